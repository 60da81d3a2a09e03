#!/bin/bash
# (needs a "wtaxcd" experiment build of a tree that still has the CART_WTA_XCD hook: kept as the record of the command)
# usage (GPU box): r03_wtaxcd.sh <tag> -- parity of the variant, then per-placement probe times (aggregate, wta) of the experiment builds
# "addr" (product kernels) and "wtaxcd" (WTA tiles of a frame on one XCD), 10 placements x 2 slot groups per process, 3 processes each
R=$GRAFT_REPO_ROOT; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/wtaxcd/libcart_engine.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random or full_size or scene or plans" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for rep in 1 2 3; do for v in addr wtaxcd; do
  CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/$v/libcart_engine.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --steps 20 --repeats 2 --placement-tries 10 > $O/x.json 2> $O/x.err
  echo "$v: $(grep '^probe' $O/x.err | awk '{printf "%s/%s ", $5, $7}')" | tee -a $O/probes.txt
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d["stages_ms_per_launch"].items() if k in ("aggregate","wta")})' $O/x.json $v | tee -a $O/probes.txt
done; done

#!/bin/bash
# usage (GPU box): r03_probe_pairs.sh <tag> <variant> [<variant> ...] -- per-placement probe times "aggregate/wta" (experiment builds print them) of
# ten placements x two slot groups per process, three processes per variant, and the bench value each process ends with
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O
for rep in 1 2 3; do for v in "$@"; do
  CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/$v/libcart_engine.so timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 20 --repeats 2 --placement-tries 10 $BENCH_ARGS > $O/x.json 2> $O/x.err || { echo "$v failed"; tail -2 $O/x.err; continue; }
  echo "$v: $(grep '^probe' $O/x.err | awk '{printf "%s/%s ", $5, $7}')" | tee -a $O/probes.txt
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d["stages_ms_per_launch"].items() if k in ("aggregate","wta")})' $O/x.json $v | tee -a $O/probes.txt
done; done

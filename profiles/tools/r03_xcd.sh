#!/bin/bash
# XCD-aware decode of the aggregation grid (frames of a launch dealt out per XCD) against the plain decode, A/B on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_xcd; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stage_by_stage or full_size_against or randomized or launch_plans" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for rep in 1 2 3; do for v in base noxcd; do
  lib=$R/cart-slam_amd/build/ab/$v/libcart_engine.so; [ "$v" = base ] && lib=$R/cart-slam_amd/build/libcart_engine.so
  for cfg in "" "--disparities 64 --paths 4" "--disparities 256 --paths 4"; do
    CART_ENGINE_LIB=$lib timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 30 $cfg > $O/x.json 2> $O/x.err || { echo "$v failed"; continue; }
    python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], "|", sys.argv[3], "|", d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d["stages_ms_per_launch"].items()})' $O/x.json $v "$cfg" | tee -a $O/summary.txt
  done
done; done

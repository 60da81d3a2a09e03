#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_xcd2; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or launch_plans or full_size_oracle_1080p" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for rep in 1 2 3; do for v in base noxcd; do
  lib=$R/cart-slam_amd/build/ab/$v/libcart_engine.so; [ "$v" = base ] && lib=$R/cart-slam_amd/build/libcart_engine.so
  for cfg in "--disparities 256 --paths 4" "--width 1920 --height 1080 --disparities 256 --paths 8 --batch 8" "--plan fused_up"; do
    CART_ENGINE_LIB=$lib timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 30 $cfg > $O/x.json 2> $O/x.err || { echo "$v failed"; continue; }
    python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], "|", sys.argv[3], "|", d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d["stages_ms_per_launch"].items()})' $O/x.json $v "$cfg" | tee -a $O/summary.txt
  done
done; done

#!/bin/bash
# row-interleaved census copy for the horizontal scans: FIFO depth 4 / 8 / 16 on top (configs[1] and the headline)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_tl_pf; mkdir -p $O
for v in base t_pf4 t_pf16; do
  lib=$R/cart-slam_amd/build/ab/$v/libcart_engine.so; [ "$v" = base ] && lib=$R/cart-slam_amd/build/libcart_engine.so
  for cfg in "--disparities 64 --paths 4" "--disparities 64 --paths 4 --no-overlap" ""; do
    CART_ENGINE_LIB=$lib timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 30 $cfg > $O/x.json 2> $O/x.err || { echo "$v failed"; continue; }
    python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "|", sys.argv[3], "|", d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d["stages_ms_per_launch"].items()})' $O/x.json $v "$cfg" | tee -a $O/summary.txt
  done
done

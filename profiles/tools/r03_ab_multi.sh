#!/bin/bash
# usage (GPU box): r03_ab_multi.sh <tag> <reps> <variant> [<variant> ...] -- the product build ("base") and experiment builds
# (cart-slam_amd/build/ab/<variant>) over the four single-GPU configurations, alternating; no parity run (ablation builds are wrong by construction)
R=$GRAFT_REPO_ROOT; T=$1; N=$2; shift 2; O=$R/gpurun_out/$T; mkdir -p $O
for rep in $(seq $N); do for v in base "$@"; do
  lib=$R/cart-slam_amd/build/ab/$v/libcart_engine.so; [ "$v" = base ] && lib=$R/cart-slam_amd/build/libcart_engine.so
  for cfg in "" "--disparities 64 --paths 4" "--disparities 256 --paths 4" "--width 1920 --height 1080 --disparities 256 --paths 8 --batch 4"; do
    CART_ENGINE_LIB=$lib timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 30 --repeats 3 $BENCH_ARGS $cfg > $O/x.json 2> $O/x.err || { echo "$v failed"; tail -2 $O/x.err; continue; }
    python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], "|", sys.argv[3], "|", d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d["stages_ms_per_launch"].items()})' $O/x.json $v "$cfg" | tee -a $O/summary.txt
  done
done; done

#!/bin/bash
# usage (GPU box): r03_ab.sh <tag> <variant dir under cart-slam_amd/build/ab> [pytest -k expr] -- parity tests on the product build, then the
# product build ("base") against the variant over the four single-GPU configurations, three alternating repetitions on one box
R=$GRAFT_REPO_ROOT; T=$1; V=$2; K=${3:-}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q ${K:+-k "$K"} > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for rep in 1 2 3; do for v in base $V; do
  lib=$R/cart-slam_amd/build/ab/$v/libcart_engine.so; [ "$v" = base ] && lib=$R/cart-slam_amd/build/libcart_engine.so
  for cfg in "" "--disparities 64 --paths 4" "--disparities 256 --paths 4" "--width 1920 --height 1080 --disparities 256 --paths 8 --batch 4"; do
    CART_ENGINE_LIB=$lib timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 30 $cfg > $O/x.json 2> $O/x.err || { echo "$v failed"; tail -2 $O/x.err; continue; }
    python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], "|", sys.argv[3], "|", d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d["stages_ms_per_launch"].items()})' $O/x.json $v "$cfg" | tee -a $O/summary.txt
  done
done; done

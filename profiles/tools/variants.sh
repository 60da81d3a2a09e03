#!/bin/bash
# usage: variants.sh <outdir> <variant...>  -- bench line (plan pairs unless PLAN is set) per experiment build under cart-slam_amd/build/ab/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  lib=$R/cart-slam_amd/build/ab/$v/libcart_engine.so; [ "$v" = base ] && lib=$R/cart-slam_amd/build/libcart_engine.so
  CART_ENGINE_LIB=$lib timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --no-overlap --steps 10 --plan ${PLAN:-pairs} ${BENCH_ARGS} > $O/$v.json 2> $O/$v.err || { echo "$v failed"; tail -3 $O/$v.err; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"])' $O/$v.json $v
done

#!/bin/bash
# usage (GPU box): r03_resident_ab.sh <tag> [bench args] -- experiment build "addr": workgroups of the aggregation launch resident per CU
# (CART_AGG_RESIDENT: unset = the engine's rule) under placement tuning, alternating, four repetitions
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O
export CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/addr/libcart_engine.so
for rep in 1 2 3 4; do for res in rule ${RES:-7 5 4 3}; do
  if [ $res = rule ]; then unset CART_AGG_RESIDENT; else export CART_AGG_RESIDENT=$res; fi
  timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 30 --repeats 3 "$@" > $O/x.json 2> $O/x.err || { echo "resident=$res failed"; tail -2 $O/x.err; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); p=d.get("placement_tuning") or {}; print("resident="+sys.argv[2], "|", d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d["stages_ms_per_launch"].items() if k in ("aggregate","wta")}, p.get("launch_pair_ms_kept"))' $O/x.json $res | tee -a $O/summary.txt
done; done

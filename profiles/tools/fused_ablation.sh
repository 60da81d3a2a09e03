#!/bin/bash
# timing experiment: bench.py against engine builds with parts of wta_fused_kernel compiled out (results are WRONG in
# those builds; only the stage times are of interest).  Libraries: cart-slam_amd/build/ab/libcart_engine_ab<mask>.so
R=$GRAFT_REPO_ROOT
for v in "" 1 2 4 7; do
  if [ -n "$v" ]; then export CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/libcart_engine_ab$v.so; else unset CART_ENGINE_LIB; fi
  timeout -k 10 200 python $R/bench.py --steps 30 --warmup 4 --no-cpu-baseline > $R/gpurun_out/ab_$v.json || exit 1
  python - <<PY
import json; j=json.load(open("$R/gpurun_out/ab_$v.json")); print("ablate=[$v]", j["value"], j["ms_per_step"], j["stages_ms_per_launch"])
PY
done

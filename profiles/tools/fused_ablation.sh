#!/bin/bash
# timing experiment: bench.py against engine builds with parts of wta_fused_kernel compiled out (results are WRONG in
# those builds; only the stage times are of interest).  Libraries: cart-slam_amd/build/ab/libcart_engine_ab<mask>.so,
# built with   hipcc $HIPFLAGS -DCART_FUSED_ABLATE=<mask> -c csrc/sgm_kernels.hip -o build/ab/sgm_<mask>.o
#              hipcc -shared -fPIC --offload-arch=gfx950 -o build/ab/libcart_engine_ab<mask>.so build/ab/sgm_<mask>.o \
#                    build/cart_engine.o build/post_kernels.o build/superpixel_kernels.o
# and selected through CART_ENGINE_LIB (cartslam/_lib.py).  NOTE: removing a store lets the compiler delete the code
# that feeds it, so these numbers bound a stage from below rather than price the store.
R=$GRAFT_REPO_ROOT
for v in "" 1 2 4 7; do
  if [ -n "$v" ]; then export CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/libcart_engine_ab$v.so; else unset CART_ENGINE_LIB; fi
  timeout -k 10 200 python $R/bench.py --steps 30 --warmup 4 --no-cpu-baseline > $R/gpurun_out/ab_$v.json || exit 1
  python - <<PY
import json; j=json.load(open("$R/gpurun_out/ab_$v.json")); print("ablate=[$v]", j["value"], j["ms_per_step"], j["stages_ms_per_launch"])
PY
done

R=$GRAFT_REPO_ROOT; export PLAN=fused_up
timeout -k 10 300 env CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/fw8/libcart_engine.so python3 -m pytest $R/tests/test_gpu_parity.py -m gpu -q -k "fused or plans or randomized" 2>&1 | tail -2
timeout -k 10 300 env CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/fw4/libcart_engine.so python3 -m pytest $R/tests/test_gpu_parity.py -m gpu -q -k "fused or plans or randomized" 2>&1 | tail -2
for cfg in "--disparities 256 --paths 4" "--disparities 256 --paths 8" "" "--width 1920 --height 1080 --disparities 256 --paths 8 --batch 4 --steps 6"; do
  echo "==== bench args: $cfg (plan fused_up)"
  BENCH_ARGS="$cfg" bash $R/profiles/tools/r02_variants.sh fw fa0 fw1 fw4 fw8 | sed "s/'census.*'aggregate/ aggregate/"
done

# One-wave workgroups for the fused sweep at D <= 128 (no cross-wave barrier drift, right-view rows per wave): timing + parity
R=$GRAFT_REPO_ROOT; export PLAN=fused_up
for cfg in "" "--disparities_128_--paths_4" "--disparities_64_--paths_4" "--width_1920_--height_1080_--batch_4"; do
  cfg=${cfg//_/ }
  echo "==== bench args: $cfg (plan fused_up)"
  BENCH_ARGS="$cfg" bash $R/profiles/tools/r02_variants.sh fw1 fa0 w1 w1rb8 | sed "s/'census.*'aggregate/ aggregate/"
done
CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/w1rb8/libcart_engine.so timeout -k 10 400 python3 -m pytest $R/tests/test_gpu_parity.py -q -m gpu -k "fused or randomized or plans_agree" 2>&1 | tail -3

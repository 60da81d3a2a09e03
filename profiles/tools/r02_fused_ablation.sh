# Fused sweep, timing experiments (results of the ablation builds are wrong by construction):
# fa<mask> = -DCART_FUSED_ABLATE=<mask> (2 no right-view atomics, 4 no slab loads, 128 no recurrence, 256 no WTA, 512 no burst,
# 1024 no path at all, 2048 burst without barriers), mw3 = 3 instead of 4 waves per SIMD for the 4-path kernels.
R=$GRAFT_REPO_ROOT; export PLAN=fused_up
for cfg in ${CFGS:-"--disparities_256_--paths_4"}; do
  cfg=${cfg//_/ }
  echo "==== bench args: $cfg (plan fused_up)"
  BENCH_ARGS="$cfg" bash $R/profiles/tools/r02_variants.sh fab base mw3 fa0 fa2 fa4 fa128 fa256 fa512 fa1024 fa2048 | sed "s/'census.*'aggregate/ aggregate/"
done

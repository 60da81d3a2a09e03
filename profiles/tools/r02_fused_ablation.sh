R=$GRAFT_REPO_ROOT; export PLAN=fused_up
for cfg in "--disparities 256 --paths 4" "--disparities 256 --paths 8" ""; do
  echo "==== bench args: $cfg (plan fused_up; ablation builds: results wrong, timing only)"
  BENCH_ARGS="$cfg" bash $R/profiles/tools/r02_variants.sh fab fa0 fa2 fa4 fa128 fa256 fa512 fa1024 | sed "s/'census.*'aggregate/ aggregate/"
done

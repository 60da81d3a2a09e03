R=$GRAFT_REPO_ROOT; export PLAN=auto
for cfg in "--disparities 64 --paths 4" "" "--disparities 256 --paths 4"; do
  echo "==== bench args: $cfg"; BENCH_ARGS="$cfg" bash $R/profiles/tools/r02_variants.sh prio prio3pf4 prio1 prio0 prio0pf4 | sed "s/'census.*'aggregate/ aggregate/"
done

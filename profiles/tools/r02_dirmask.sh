R=$GRAFT_REPO_ROOT; export PLAN=auto
for cfg in "--disparities 64 --paths 4" ""; do
  echo "==== bench args: $cfg"
  for m in 0xff 0x03 0x01 0x0c 0x04; do echo "-- dirmask $m"; CART_DEBUG_DIRMASK=$m BENCH_ARGS="$cfg" bash $R/profiles/tools/r02_variants.sh dm exp | sed 's/.*aggregate/aggregate/'; done
  echo "-- no store"; BENCH_ARGS="$cfg" bash $R/profiles/tools/r02_variants.sh dm nostore | sed 's/.*aggregate/aggregate/'
  echo "-- no store, horizontals only"; CART_DEBUG_DIRMASK=0x03 BENCH_ARGS="$cfg" bash $R/profiles/tools/r02_variants.sh dm nostore | sed 's/.*aggregate/aggregate/'
done

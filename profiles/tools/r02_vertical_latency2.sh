R=$GRAFT_REPO_ROOT; export PLAN=auto
for v in v4 nolds noldsrd noload nothing; do for b in 1 16; do echo "-- $v batch $b vertical only (0x04)"; CART_DEBUG_DIRMASK=0x04 BENCH_ARGS="--disparities 64 --paths 4 --batch $b --steps 20" bash $R/profiles/tools/r02_variants.sh vl2 $v | sed "s/.*'aggregate/aggregate/;s/, 'wta.*//"; done; done

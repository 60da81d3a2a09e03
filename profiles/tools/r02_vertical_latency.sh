R=$GRAFT_REPO_ROOT; export PLAN=auto
for v in v2 v4 v6 v4nostore; do for m in 0x04 0x0c 0x01; do echo "-- $v dirmask $m"; CART_DEBUG_DIRMASK=$m BENCH_ARGS="--disparities 64 --paths 4" bash $R/profiles/tools/r02_variants.sh vl $v | sed "s/.*'aggregate/aggregate/;s/, 'wta.*//"; done; done
echo "== batch 1 (single frame), v4"; for m in 0x04 0x01 0x0f; do CART_DEBUG_DIRMASK=$m BENCH_ARGS="--disparities 64 --paths 4 --batch 1 --steps 30" bash $R/profiles/tools/r02_variants.sh vl v4 | sed "s/.*'aggregate/aggregate/;s/, 'wta.*//"; done
echo "== batch 4, v4"; for m in 0x04 0x01 0x0f; do CART_DEBUG_DIRMASK=$m BENCH_ARGS="--disparities 64 --paths 4 --batch 4 --steps 30" bash $R/profiles/tools/r02_variants.sh vl v4 | sed "s/.*'aggregate/aggregate/;s/, 'wta.*//"; done

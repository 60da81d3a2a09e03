R=$GRAFT_REPO_ROOT; export PLAN=auto
for m in 0x0f 0x03 0x0c 0x04; do echo "-- D=64 batch 16 dirmask $m"; CART_DEBUG_DIRMASK=$m BENCH_ARGS="--disparities 64 --paths 4 --steps 20" bash $R/profiles/tools/r02_variants.sh cl v4 | sed "s/.*'aggregate/aggregate/;s/, 'post.*//"; done
for m in 0x0f 0x04; do echo "-- D=64 batch 1 dirmask $m"; CART_DEBUG_DIRMASK=$m BENCH_ARGS="--disparities 64 --paths 4 --steps 20 --batch 1" bash $R/profiles/tools/r02_variants.sh cl v4 | sed "s/.*'aggregate/aggregate/;s/, 'post.*//"; done
for m in 0xff 0x0c 0xf0; do echo "-- D=128 P=8 batch 16 dirmask $m"; CART_DEBUG_DIRMASK=$m BENCH_ARGS="--steps 20" bash $R/profiles/tools/r02_variants.sh cl v4 | sed "s/.*'aggregate/aggregate/;s/, 'post.*//"; done
echo "-- D=128 P=8 batch 1"; BENCH_ARGS="--steps 30 --batch 1" bash $R/profiles/tools/r02_variants.sh cl v4 | sed "s/.*'aggregate/aggregate/;s/, 'post.*//"

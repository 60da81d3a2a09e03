R=$GRAFT_REPO_ROOT
echo "== D=64 P=4"; PLAN=auto BENCH_ARGS="--disparities 64 --paths 4" bash $R/profiles/tools/r02_variants.sh pf_c1 base pf4_8 pf4_4
echo "== D=128 P=8"; PLAN=auto BENCH_ARGS="" bash $R/profiles/tools/r02_variants.sh pf_c2 base pf8_16
echo "== D=128 P=8 single pair"; PLAN=auto BENCH_ARGS="--batch 1 --steps 30" bash $R/profiles/tools/r02_variants.sh pf_c2s base pf8_16
echo "== D=64 P=4 single pair"; PLAN=auto BENCH_ARGS="--disparities 64 --paths 4 --batch 1 --steps 30" bash $R/profiles/tools/r02_variants.sh pf_c1s base pf4_8

R=$GRAFT_REPO_ROOT; export PLAN=auto
for rep in 1 2; do
echo "==== D=64 P=4, repetition $rep (one stream)"; BENCH_ARGS="--disparities 64 --paths 4 --steps 40" bash $R/profiles/tools/r02_variants.sh d64 exp0 pf4 pf8 aw2 aw8 pf4v6 pf8aw8 | sed "s/'census.*'aggregate/ aggregate/"
done
echo "==== D=128 P=8"; BENCH_ARGS="--steps 20" bash $R/profiles/tools/r02_variants.sh d64 exp0 aw2 aw8 | sed "s/'census.*'aggregate/ aggregate/"
echo "==== D=256 P=4"; BENCH_ARGS="--disparities 256 --paths 4 --steps 20" bash $R/profiles/tools/r02_variants.sh d64 exp0 aw2 aw8 | sed "s/'census.*'aggregate/ aggregate/"

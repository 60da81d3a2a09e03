R=$GRAFT_REPO_ROOT; export PLAN=auto
for b in 8 16 20 21 22 24 32 42; do echo "==== D=64 P=4 batch=chunk=$b"; BENCH_ARGS="--disparities 64 --paths 4 --batch $b --chunk $b" bash $R/profiles/tools/r02_variants.sh ch base | sed "s/'census.*'aggregate/ aggregate/"; done
for b in 8 16 21 32; do echo "==== D=256 P=4 batch=chunk=$b"; BENCH_ARGS="--disparities 256 --paths 4 --batch $b --chunk $b" bash $R/profiles/tools/r02_variants.sh ch base | sed "s/'census.*'aggregate/ aggregate/"; done
for b in 8 16 21 32; do echo "==== D=128 P=8 batch=chunk=$b"; BENCH_ARGS="--batch $b --chunk $b" bash $R/profiles/tools/r02_variants.sh ch base | sed "s/'census.*'aggregate/ aggregate/"; done

R=$GRAFT_REPO_ROOT
for b in 2 4 6 8; do echo "== 1920x1080 D=256 P=8 batch=chunk=$b"; PLAN=auto BENCH_ARGS="--width 1920 --height 1080 --disparities 256 --paths 8 --batch $b --chunk $b --steps 6" bash $R/profiles/tools/r02_variants.sh c3b base | sed "s/'census.*'aggregate/ aggregate/"; done

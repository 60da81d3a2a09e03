R=$GRAFT_REPO_ROOT
for b in 16 24 32 48; do for plan in fused_up slabs; do echo "== D=128 P=8 plan $plan batch=chunk=$b"; PLAN=$plan BENCH_ARGS="--batch $b --chunk $b" bash $R/profiles/tools/r02_variants.sh fc base | sed "s/'census.*'aggregate/ aggregate/"; done; done

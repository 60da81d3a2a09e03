#!/bin/bash
# GPU box, round 5, the very last commit: whole GPU suite, smoke, a second randomised campaign, the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_last; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc = 0 ] || exit $rc
timeout -k 10 200 python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }; tail -1 $O/smoke.log
BUDGET_S=400 SEED=606 timeout -k 10 800 python3 profiles/tools/parity_fuzz.py > $O/fuzz.txt 2> $O/fuzz.err || { tail -3 $O/fuzz.txt; tail -3 $O/fuzz.err; exit 1; }
tail -1 $O/fuzz.txt
bash profiles/tools/r05_default_runs.sh r05_last_default

#!/bin/bash
# GPU box: the whole GPU suite + smoke on the current tree
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_suite}; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc = 0 ] || exit $rc
timeout -k 10 200 python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }; tail -1 $O/smoke.log

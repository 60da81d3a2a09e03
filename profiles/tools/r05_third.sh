#!/bin/bash
# GPU box, round 5, third call: whole GPU suite on the collapsed CCL / fused post stage / one-communicator sharder, then the write-window A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_third; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; rc=$?; tail -14 $O/pytest.log; [ $rc = 0 ] || exit $rc
PMC_VARS="base res4 fm" bash profiles/tools/r05_ab.sh r05_window "base res3 res4 res5 fm fmnox g4nox nox" 3

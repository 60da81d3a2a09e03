#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_twelfth; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "ccl" > $O/ccl_tests.log 2>&1; rc=$?; tail -15 $O/ccl_tests.log; exit $rc

#!/bin/bash
# GPU box, round 5: four slab rows in flight in the 4-path fused sweeps (product) against two (variant ns2); whole GPU suite first
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_ns; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc = 0 ] || exit $rc
CONFIGS="ref c3p4 ref8 d128p4f" PARITY_VARS="" bash profiles/tools/r05_ab.sh r05_ns_ab "ns2 base" 3

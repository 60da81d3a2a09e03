#!/bin/bash
# GPU box: second randomised campaign (another seed), the C++ frame-loop soak and the host-loop throughput on the final code
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_campaign2; mkdir -p $O; cd $R
BUDGET_S=${BUDGET_S:-500} SEED=${SEED:-97} timeout -k 10 800 python3 profiles/tools/parity_fuzz.py > $O/fuzz.txt 2> $O/fuzz.err || { tail -3 $O/fuzz.txt; tail -3 $O/fuzz.err; exit 1; }
tail -1 $O/fuzz.txt
N=4000 timeout -k 10 400 python3 profiles/tools/host_soak.py > $O/host_soak.txt 2> $O/host_soak.err || { tail -5 $O/host_soak.txt; tail -5 $O/host_soak.err; exit 1; }; tail -2 $O/host_soak.txt
N=960 timeout -k 10 400 python3 profiles/tools/host_loop_throughput.py > $O/host_loop.txt 2> $O/host_loop.err || { tail -5 $O/host_loop.txt; tail -5 $O/host_loop.err; exit 1; }; tail -12 $O/host_loop.txt

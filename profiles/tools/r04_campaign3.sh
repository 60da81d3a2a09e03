#!/bin/bash
# GPU box: fourth randomised campaign (another seed) and soaks at configs[1] (D=64 P=4: split horizontal scans with whole-line stores), configs[2] and the reference default
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_campaign3; mkdir -p $O; cd $R
BUDGET_S=${BUDGET_S:-420} SEED=${SEED:-211} timeout -k 10 800 python3 profiles/tools/parity_fuzz.py > $O/fuzz.txt 2> $O/fuzz.err || { tail -3 $O/fuzz.txt; tail -3 $O/fuzz.err; exit 1; }
tail -1 $O/fuzz.txt
DISP=64 PATHS=4 STEPS=3000 timeout -k 10 300 python3 profiles/tools/soak.py > $O/soak_c1.txt 2> $O/soak.err || { tail -3 $O/soak_c1.txt; tail -3 $O/soak.err; exit 1; }; tail -1 $O/soak_c1.txt
STEPS=1500 timeout -k 10 300 python3 profiles/tools/soak.py > $O/soak_c2.txt 2>> $O/soak.err || { tail -3 $O/soak_c2.txt; exit 1; }; tail -1 $O/soak_c2.txt
DISP=256 PATHS=4 STEPS=1500 timeout -k 10 300 python3 profiles/tools/soak.py > $O/soak_ref.txt 2>> $O/soak.err || { tail -3 $O/soak_ref.txt; exit 1; }; tail -1 $O/soak_ref.txt
DISP=64 PATHS=8 STEPS=1500 timeout -k 10 300 python3 profiles/tools/soak.py > $O/soak_d64p8.txt 2>> $O/soak.err || { tail -3 $O/soak_d64p8.txt; exit 1; }; tail -1 $O/soak_d64p8.txt

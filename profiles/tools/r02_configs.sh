#!/bin/bash
# usage: r02_configs.sh <outdir> [pytest: 0|1] -- the -m gpu suite, then the bench line of the three single-GPU BASELINE configurations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
if [ "${2:-1}" = 1 ]; then
  timeout -k 10 900 python3 -m pytest $R/tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -20
fi
run() { name=$1; shift; timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 10 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"], d["roofline"]["frac"])' $O/$name.json $name; }
run c1_d64_p4 --disparities 64 --paths 4
run c2_d128_p8
run c3_1080p_d256_p8 --width 1920 --height 1080 --disparities 256 --paths 8 --batch 4
run ref_d256_p4 --disparities 256 --paths 4
run c1_d64_p4_fused --disparities 64 --paths 4 --plan fused_up

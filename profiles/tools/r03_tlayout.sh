#!/bin/bash
# parity subset + the four single-GPU configurations after a kernel change: gpurun_out/<tag>/{pytest.log,c1,c2,c3,ref}.json
R=$GRAFT_REPO_ROOT; T=${1:-r03_tl}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc = 0 ] || exit $rc
b() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pcie --steps 30 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["spread"]["min"], d["spread"]["max"], d["ms_per_step"], {k: round(v,3) for k,v in d["stages_ms_per_launch"].items()}, d["roofline"]["frac"])' $O/$name.json $name; }
b c1 --disparities 64 --paths 4
b c1_nooverlap --disparities 64 --paths 4 --no-overlap
b c2
b ref --disparities 256 --paths 4
b c3 --width 1920 --height 1080 --disparities 256 --paths 8 --batch 4

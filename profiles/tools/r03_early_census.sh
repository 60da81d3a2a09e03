#!/bin/bash
# usage (GPU box): r03_early_census.sh <tag> -- (experiment tree: profiles/r03_early_census.patch applied) parity of the early entry point, then
# bench.py --early 0 / 1 alternating over the four single-GPU configurations
R=$GRAFT_REPO_ROOT; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "early or side_stream" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for rep in 1 2 3 4; do for v in 0 1; do
  for cfg in "" "--disparities 64 --paths 4" "--disparities 256 --paths 4" "--width 1920 --height 1080 --disparities 256 --paths 8 --batch 4"; do
    timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 30 --repeats 3 --early $v $cfg > $O/x.json 2> $O/x.err || { echo "early=$v failed"; tail -2 $O/x.err; continue; }
    python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print("early="+sys.argv[2], "|", sys.argv[3], "|", d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d["stages_ms_per_launch"].items()})' $O/x.json $v "$cfg" | tee -a $O/summary.txt
  done
done; done

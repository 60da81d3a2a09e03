#!/bin/bash
# GPU box: the whole -m gpu suite in one process, then smoke(), then the default bench line (outputs under gpurun_out/<tag>)
R=$GRAFT_REPO_ROOT; T=${1:-r04_tests}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; rc=$?; tail -25 $O/pytest.log; [ $rc = 0 ] || exit $rc
timeout -k 10 200 python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }; tail -1 $O/smoke.log
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(d["value"], d["spread"], d["ms_per_step"], d["verified"], d["stages_ms_per_launch"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d.get("value_bgr_input"), d.get("bgr_input"), d.get("placement_tuning"))' $O/bench.json

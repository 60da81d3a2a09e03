#!/bin/bash
# GPU tests, then the default bench line, then the bench with the sequence leg (outputs under gpurun_out/<tag>)
R=$GRAFT_REPO_ROOT; T=${1:-r03_tb}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(d["value"], d["spread"], d["ms_per_step"], d["verified"], d["stages_ms_per_launch"], d["roofline"]["frac"], d["cpu_baseline"]["value"])' $O/bench.json
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pcie --sequence > $O/bench_seq.json 2> $O/bench_seq.err || { tail -5 $O/bench_seq.err; exit 1; }
python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(d["value"], d["sequence_mode"])' $O/bench_seq.json

#!/bin/bash
# GPU box, round 5, last code: whole GPU suite, smoke, the C++ frame loop (reference-style use), and the default bench line three times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_final}; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; rc=$?; tail -9 $O/pytest.log; [ $rc = 0 ] || exit $rc
timeout -k 10 200 python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }; tail -1 $O/smoke.log
for i in 1 2; do N=960 timeout -k 10 500 python3 profiles/tools/host_loop_throughput.py > $O/host_loop_$i.txt 2> $O/host_loop_$i.err || { tail -5 $O/host_loop_$i.err; exit 1; }; grep -i "steady\|frames/s\|pairs/s" $O/host_loop_$i.txt | head -12; done
for i in 1 2 3; do timeout -k 10 300 python3 bench.py > $O/bench_$i.json 2> $O/bench_$i.err || { tail -3 $O/bench_$i.err; exit 1; }
python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); p=d["placement_tuning"]; print(d["value"], d["value_without_stage_events"], d["ms_per_step"], d["verified"], d["stages_ms_per_launch"], d["roofline"]["frac"], d["roofline"]["launches_timed"], p["mode"], p["stopped_on"], p["candidates_timed"], p["launch_pair_ms_kept"], p["value_untuned"], p["seconds"], d["cpu_baseline"]["value"], d.get("value_bgr_input"), d.get("value_pcie_inclusive"))' $O/bench_$i.json; done

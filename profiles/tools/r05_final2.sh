#!/bin/bash
# GPU box, round 5, last code: whole GPU suite, smoke, rocprofv3 kernel stats of the bench command with the placement search off (every aggregation / WTA launch of
# the profile is then a bench launch: the stats' averages and the line's HIP-event stage times must agree), default bench line twice
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_final2}; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; rc=$?; tail -9 $O/pytest.log; [ $rc = 0 ] || exit $rc
timeout -k 10 200 python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }; tail -1 $O/smoke.log
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-bgr --placement-tries 1 --timing-every 1 > $O/bench_under_rocprof_tries1.json 2> $O/st.log
f=$(ls $O/st/*kernel_stats.csv | head -1); cp $f $O/kernel_stats_c2_tries1.csv; python3 $R/profiles/tools/kernel_avgs.py $f | head -3; rm -rf $O/st
python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("events of the same process:", d["value"], d["stages_ms_per_launch"], d["roofline"]["launches_timed"])' $O/bench_under_rocprof_tries1.json
cd $R
for i in 1 2; do timeout -k 10 300 python3 bench.py > $O/bench_$i.json 2> $O/bench_$i.err || { tail -3 $O/bench_$i.err; exit 1; }
python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); p=d["placement_tuning"]; print(d["value"], d["value_without_stage_events"], d["ms_per_step"], d["verified"], d["stages_ms_per_launch"], d["roofline"]["frac"], p["mode"], p["stopped_on"], p["candidates_timed"], p["launch_pair_ms_kept"], p["value_untuned"], p["seconds"], d["cpu_baseline"]["value"], d.get("value_bgr_input"), d.get("value_pcie_inclusive"))' $O/bench_$i.json; done

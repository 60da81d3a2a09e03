#!/bin/bash
# GPU box, round 5: CCL tests + plane-stage kernel times after the table kernel's batched segment loads; default bench line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_eighth; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "ccl or golden or full_size_against or plane" > $O/ccl_tests.log 2>&1; rc=$?; tail -3 $O/ccl_tests.log; [ $rc = 0 ] || exit $rc
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-bgr --no-overlap --steps 20 --repeats 2 > $O/no_overlap.json 2> $O/no_overlap.log
f=$(ls $O/st/*kernel_stats.csv | head -1); cp $f $O/kernel_stats_c2_no_overlap.csv; python3 $R/profiles/tools/kernel_avgs.py $f | head -16; rm -rf $O/st
cd $R
for i in 1 2 3; do timeout -k 10 300 python3 bench.py > $O/bench_$i.json 2> $O/bench_$i.err || { tail -3 $O/bench_$i.err; exit 1; }
python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); p=d["placement_tuning"]; print(d["value"], d["value_without_stage_events"], d["ms_per_step"], d["verified"], d["stages_ms_per_launch"], d["roofline"]["frac"], d["roofline"]["launches_timed"], p["mode"], p["stopped_on"], p["candidates_timed"], p["launch_pair_ms_kept"], p["value_untuned"], p["seconds"], d["cpu_baseline"]["value"], d.get("value_bgr_input"), d.get("value_pcie_inclusive"))' $O/bench_$i.json; done

#!/bin/bash
# GPU box, round 5: (a) is the FIRST process on a fresh box the one that finds no fast placement?  Four fresh bench processes in a row, default
# search (8 tries); (b) the whole -m gpu suite after the removal of plan PAIRS and the new placement report; (c) smoke.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=${1:-r05_second}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for i in 1 2 3 4; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 > $O/seq_$i.json 2> $O/seq_$i.err || { tail -5 $O/seq_$i.err; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); p=d["placement_tuning"]; s=d["stages_ms_per_launch"]; print("run", sys.argv[2], d["value"], "untuned", p["value_untuned"], "agg %.4f wta %.4f" % (s["aggregate"], s["wta"]), p["mode"], p["stopped_on"], p["candidates_timed"], "first %.3f kept %.3f slowest %.3f" % (p["launch_pair_ms_first"], p["launch_pair_ms_kept"], p["launch_pair_ms_slowest_seen"]), p["seconds"], "s")' $O/seq_$i.json $i | tee -a $O/summary.txt
done
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q --durations=10 > $O/pytest.log 2>&1; rc=$?; tail -18 $O/pytest.log; [ $rc = 0 ] || exit $rc
timeout -k 10 200 python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }; tail -1 $O/smoke.log

#!/bin/bash
# GPU box, round 5, fifth call: GPU suite on the re-timed placement search; what the search is worth now (1 try against the default 8, alternating
# fresh processes); frame-major order of the short directions once more at the headline, five rounds
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_fifth; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; rc=$?; tail -10 $O/pytest.log; [ $rc = 0 ] || exit $rc
for i in 1 2 3 4 5 6; do
  for t in 1 8; do
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 --placement-tries $t > $O/tries${t}_$i.json 2> $O/tries${t}_$i.err || { tail -5 $O/tries${t}_$i.err; exit 1; }
    python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); p=d.get("placement_tuning") or {}; s=d["stages_ms_per_launch"]; print("tries", sys.argv[2], d["value"], "no-events", d["value_without_stage_events"], "untuned", p.get("value_untuned"), "agg %.4f wta %.4f" % (s["aggregate"], s["wta"]), p.get("mode"), p.get("stopped_on"), p.get("candidates_timed"), "first %.3f kept %.3f slowest %.3f" % (p.get("launch_pair_ms_first", 0), p.get("launch_pair_ms_kept", 0), p.get("launch_pair_ms_slowest_seen", 0)), p.get("seconds"))' $O/tries${t}_$i.json $t | tee -a $O/summary.txt
  done
done
PARITY_VARS="" bash profiles/tools/r05_ab.sh r05_fm2 "base fm" 5

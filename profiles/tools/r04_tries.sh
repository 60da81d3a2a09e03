#!/bin/bash
# GPU box: does a longer placement search pay?  bench.py --placement-tries N, alternating, same box (headline configuration unless args are given)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_tries; mkdir -p $O; cd $R
for r in 1 2 3 4; do for n in 10 24 40; do
  timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 --placement-tries $n "$@" > $O/t.json 2> $O/t.err || { tail -3 $O/t.err; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages_ms_per_launch"]; p=d["placement_tuning"]; print("tries %2s  %8.1f pairs/s  agg %.4f wta %.4f  pair first %.4f kept %.4f  %.2f s" % (sys.argv[2], d["value"], s["aggregate"], s["wta"], p["launch_pair_ms_first"], p["launch_pair_ms_kept"], p["seconds"]))' $O/t.json $n | tee -a $O/summary.txt
done; done

#!/bin/bash
# GPU box: distribution of the default bench line's placement outcome over fresh processes (default --placement-tries)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_tries2; mkdir -p $O; cd $R
for r in $(seq ${1:-10}); do
  timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 > $O/t.json 2> $O/t.err || { tail -3 $O/t.err; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages_ms_per_launch"]; p=d["placement_tuning"]; print("tries %2d  %8.1f pairs/s  agg %.4f wta %.4f  pair first %.4f kept %.4f  %.2f s" % (p["tries"], d["value"], s["aggregate"], s["wta"], p["launch_pair_ms_first"], p["launch_pair_ms_kept"], p["seconds"]))' $O/t.json | tee -a $O/summary.txt
done

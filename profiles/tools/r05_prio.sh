#!/bin/bash
# GPU box, round 5: the disparity kernels on a high-priority stream, the plane stages beside them at normal priority -- do the main-stream kernels stretch less?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_prio; mkdir -p $O; cd $R
python3 -c 'import torch; print("priority range", torch.cuda.Stream.priority_range())' | tee $O/summary.txt
for r in 1 2 3 4; do for c in "c2 " "c1 --disparities 64 --paths 4"; do set -- $c; n=$1; shift; for m in "" "--main-priority"; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 $m "$@" > $O/${n}_${r}_${m#--}.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); s=d["stages_ms_per_launch"]; print("%-3s %-15s %8.1f pairs/s  step %.4f  census %.4f agg %.4f wta %.4f post %.4f" % (sys.argv[2], sys.argv[3] or "normal", d["value"], d["ms_per_step"], s["census"], s["aggregate"], s["wta"], s["post"]))' $O/${n}_${r}_${m#--}.json $n "$m" | tee -a $O/summary.txt
done; done; done

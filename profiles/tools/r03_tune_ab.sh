#!/bin/bash
# usage (GPU box): r03_tune_ab.sh <tag> -- bench.py --placement-tries 1 against 10 (the default), alternating, over the four single-GPU configurations
R=$GRAFT_REPO_ROOT; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
for rep in 1 2 3 4; do for tries in 1 10; do
  for cfg in "" "--disparities 64 --paths 4" "--disparities 256 --paths 4" "--width 1920 --height 1080 --disparities 256 --paths 8 --batch 4"; do
    timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 30 --repeats 3 --placement-tries $tries $cfg > $O/x.json 2> $O/x.err || { echo "tries=$tries failed"; tail -2 $O/x.err; continue; }
    python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); p=d.get("placement_tuning") or {}; print("tries="+sys.argv[2], "|", sys.argv[3], "|", d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d["stages_ms_per_launch"].items() if k in ("aggregate","wta")}, p.get("launch_pair_ms_first"), p.get("launch_pair_ms_kept"), p.get("seconds"))' $O/x.json $tries "$cfg" | tee -a $O/summary.txt
  done
done; done

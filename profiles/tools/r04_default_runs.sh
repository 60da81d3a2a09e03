#!/bin/bash
# GPU box: N fresh-process runs of the default `python bench.py` (what the driver runs): the distribution of the headline line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_default_runs; mkdir -p $O; cd $R
for i in $(seq ${N:-6}); do
  timeout -k 10 300 python3 bench.py > $O/run_$i.json 2> $O/run_$i.err || { tail -3 $O/run_$i.err; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); s=d["stages_ms_per_launch"]; p=d["placement_tuning"]; print("run %s  %.1f pairs/s (%.1f-%.1f)  step %.4f  agg %.4f  wta %.4f  frac %.4f  verified %s  bgr %.1f  pcie %.1f  cpu %.2f  tuning: %d tries %.2f s %.3f -> %.3f ms" % (sys.argv[2], d["value"], d["spread"]["min"], d["spread"]["max"], d["ms_per_step"], s["aggregate"], s["wta"], d["roofline"]["frac"], d["verified"], d["value_bgr_input"], d["value_pcie_inclusive"], d["cpu_baseline"]["value"], p["tries"], p["seconds"], p["launch_pair_ms_first"], p["launch_pair_ms_kept"]))' $O/run_$i.json $i
done

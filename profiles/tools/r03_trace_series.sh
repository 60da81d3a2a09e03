#!/bin/bash
# usage (GPU box): r03_trace_series.sh <tag> [bench args] -- rocprofv3 --kernel-trace of bench.py; prints the duration of the aggregation and WTA
# dispatches in call order (means of consecutive groups of 20) next to the bench's own HIP-event stage times of the same process
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-pcie "$@" > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - $O <<'PY'
import csv, sys, glob, json
O = sys.argv[1]
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = []
for name in ("aggregate_kernel", "wta_kernel", "wta_fused_kernel"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"] and int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]) > 100000]
    if not d: continue
    groups = [sum(d[i:i + 20]) / len(d[i:i + 20]) for i in range(0, len(d), 20)]
    out.append("%s: %d dispatches, mean %.1f us, last 250: %.1f us; means of groups of 20 in call order: %s" % (name, len(d), sum(d) / len(d), sum(d[-250:]) / len(d[-250:]), " ".join("%.0f" % g for g in groups)))
b = json.loads(open(O + "/bench.json").read().strip().splitlines()[-1])
out.append("bench (same process): %s pairs/s, %s ms/step, stages %s" % (b["value"], b["ms_per_step"], b["stages_ms_per_launch"]))
# gap between consecutive big kernels on the main queue in the timed region
open(O + "/series.txt", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
rm -rf $O/trace

#!/bin/bash
# usage (GPU box): alloc_pmc.sh <tag> -- fresh engines in one process (profiles/tools/alloc_modes.py) under rocprofv3 --kernel-trace --pmc, two passes;
# prints, per engine instance (groups of 23 dispatches of the aggregation kernel in call order), the mean duration and the mean counters
R=$GRAFT_REPO_ROOT; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "${PASS1:-TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY}" "${PASS2:-GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES}"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/p$i -o t -- python3 $R/profiles/tools/alloc_modes.py > $O/run$i.txt 2>&1 || { echo "pass $i failed"; tail -3 $O/run$i.txt; }
  python3 - $O/p$i $O/run$i.txt <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
ct = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not ct: print("no counter file in", d); sys.exit(0)
rows = list(csv.DictReader(open(ct[0])))
# per dispatch: kernel name, counters, start/end
disp = collections.OrderedDict()
for r in rows:
    k = r["Dispatch_Id"]
    e = disp.setdefault(k, {"name": r["Kernel_Name"], "c": {}, "dur": None})
    e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    if "Start_Timestamp" in r and r["Start_Timestamp"]: e["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for kern in ("aggregate_kernel", "wta_kernel"):
    ds = [e for e in disp.values() if kern in e["name"]]
    n = 23
    print(kern, len(ds), "dispatches; engine instance: mean us, counters")
    for g in range(0, len(ds), n):
        grp = ds[g:g + n]
        if len(grp) < n: break
        names = sorted(grp[0]["c"])
        dur = [e["dur"] for e in grp if e["dur"]]
        print("  inst %2d  %8.1f us  " % (g // n, sum(dur) / len(dur) if dur else -1) + "  ".join("%s %.3g" % (nm.replace("_sum", ""), sum(e["c"][nm] for e in grp) / n) for nm in names))
print(open(sys.argv[2]).read()[-900:])
PY
done
rm -rf $O/p1 $O/p2

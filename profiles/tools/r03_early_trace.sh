#!/bin/bash
# usage (GPU box): r03_early_trace.sh <tag> [bench args] -- kernel trace of bench.py --early 1; prints a window of consecutive kernels
# (name, queue, start and end in us relative to the window's first kernel) from the middle of the timed region
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 10 --repeats 1 --warmup 2 "$@" > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - $O <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
mid = len(rows) * 3 // 4
win = rows[mid:mid + 70]
t0 = int(win[0]["Start_Timestamp"])
with open(sys.argv[1] + "/window.txt", "w") as out:
    for r in win:
        line = "%-44s q%-3s %9.1f %9.1f  %7.1f" % (r["Kernel_Name"][:44], r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        out.write(line + "\n")
print(open(sys.argv[1] + "/window.txt").read())
PY

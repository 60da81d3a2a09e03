#!/bin/bash
# usage (GPU box): r03_early_trace2.sh <tag> [probe args] -- kernel trace of profiles/tools/early_probe.py (engine-only loop)
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/profiles/tools/early_probe.py "$@" > $O/probe.txt 2> $O/probe.err || { tail -5 $O/probe.err; exit 1; }
cat $O/probe.txt

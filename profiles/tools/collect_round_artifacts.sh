#!/bin/bash
# Collects the artefacts profiles/README.md lists: the bench line, the rocprofv3 kernel stats of the same command and
# the PMC passes.  usage (on the GPU box): bash profiles/tools/collect_round_artifacts.sh <tag>
set -o pipefail
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=${1:-r01}; OUT=$R/gpurun_out/$T; mkdir -p $OUT
cd $R && timeout -k 10 400 python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "bench done" | tee -a $OUT/progress.txt
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
echo "kernel stats exit=$?" | tee -a $OUT/progress.txt
cd $R && timeout -k 10 900 bash profiles/tools/pmc_passes.sh $T/pmc
python3 profiles/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt
echo "pmc done" | tee -a $OUT/progress.txt

#!/bin/bash
# GPU box: kernel durations of one step WITHOUT the two-stream overlap (every kernel alone on the GPU): what the plane stages cost by themselves
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_plane_stats; mkdir -p $O
for c in "c2 " "c1 --disparities 64 --paths 4"; do set -- $c; n=$1; shift
  cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-bgr --no-overlap --steps 20 --repeats 2 "$@" > $O/$n.json 2> $O/$n.log
  f=$(ls $O/$n/*kernel_stats.csv | head -1); cp $f $O/kernel_stats_${n}_no_overlap.csv; python3 $R/profiles/tools/kernel_avgs.py $f | head -24; rm -rf $O/$n
done

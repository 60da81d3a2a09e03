#!/bin/bash
# GPU box: CCL / plane tests, then every kernel of a step alone on the GPU (bench.py --no-overlap under rocprofv3): the plane-stage kernel times
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_ccl_check}; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "ccl or golden or full_size_against or plane or extreme or scene" > $O/ccl_tests.log 2>&1; rc=$?; tail -3 $O/ccl_tests.log; [ $rc = 0 ] || exit $rc
for c in "c2 " "c1 --disparities 64 --paths 4"; do set -- $c; n=$1; shift
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-bgr --no-overlap --steps 20 --repeats 2 "$@" > $O/no_overlap_$n.json 2> $O/no_overlap_$n.log
f=$(ls $O/st/*kernel_stats.csv | head -1); cp $f $O/kernel_stats_${n}_no_overlap.csv; python3 $R/profiles/tools/kernel_avgs.py $f | grep -i "ccl\|post_interp\|plane\|classify\|census"; rm -rf $O/st
done

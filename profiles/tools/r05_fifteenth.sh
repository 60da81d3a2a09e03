#!/bin/bash
# GPU box, round 5: whole GPU suite on the LDS-staged post stage, its kernel time, default bench lines at the headline and configs[1]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_fifteenth}; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc = 0 ] || exit $rc
for c in "c2 " "c1 --disparities 64 --paths 4" "ref --disparities 256 --paths 4"; do set -- $c; n=$1; shift
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-bgr --no-overlap --steps 20 --repeats 2 "$@" > $O/no_overlap_$n.json 2> $O/no_overlap_$n.log
f=$(ls $O/st/*kernel_stats.csv | head -1); cp $f $O/kernel_stats_${n}_no_overlap.csv; python3 $R/profiles/tools/kernel_avgs.py $f | grep -i "post\|census"; rm -rf $O/st
done
cd $R; for c in "c2 " "c1 --disparities 64 --paths 4"; do set -- $c; n=$1; shift; for i in 1 2; do timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr "$@" > $O/${n}_$i.json 2>$O/err.txt; python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"])' $O/${n}_$i.json $n; done; done

#!/bin/bash
# GPU box, round 5: the numbers DESIGN.md section 5 quotes -- bench line + rocprofv3 kernel stats + PMC passes of the four single-GPU configurations
# (collect_all.sh), kernel durations of a step without the two-stream overlap (what the plane stages cost by themselves), overlap on / off,
# and the clock / power the card holds while bench.py runs (sampled from a second process)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=${1:-r05}; cd $R
bash profiles/tools/collect_all.sh $T || exit $?
O=$R/gpurun_out/${T}_plane_stats; mkdir -p $O
for c in "c2 " "c1 --disparities 64 --paths 4"; do set -- $c; n=$1; shift
  cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-bgr --no-overlap --steps 20 --repeats 2 "$@" > $O/$n.json 2> $O/$n.log
  f=$(ls $O/$n/*kernel_stats.csv | head -1); cp $f $O/kernel_stats_${n}_no_overlap.csv; python3 $R/profiles/tools/kernel_avgs.py $f | head -24; rm -rf $O/$n
done
cd $R; O=gpurun_out/${T}_ovl; mkdir -p $O
for c in "c1 --disparities 64 --paths 4" "c2 " "ref --disparities 256 --paths 4"; do set -- $c; n=$1; shift
 for m in overlap no-overlap; do for r in 1 2; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 --$m "$@" > $O/${n}_${m}_$r.json 2>$O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages_ms_per_launch"]; print(sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], {k:round(v,4) for k,v in s.items()}, round(sum(s.values()),4))' $O/${n}_${m}_$r.json $n $m | tee -a $O/summary.txt
 done; done; done
O=$R/gpurun_out/${T}_smi; mkdir -p $O
( for k in $(seq 1 90); do echo "t=$k $(rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|Power (W)\|junction" | sed 's/.*: //' | tr '\n' ' ')" >> $O/smi.txt; sleep 0.5; done ) &
SP=$!
timeout -k 5 150 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 400 --repeats 5 > $O/long_bench.json 2> $O/long_bench.err
wait $SP
sort -t'(' -k2 $O/smi.txt | tail -3

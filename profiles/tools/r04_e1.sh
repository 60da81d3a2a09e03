#!/bin/bash
# Round 4, experiment 1 (GPU box): how the fused sweep and the D=64 aggregation respond to the number of frames per launch.
#   (a) reference default 1242x375 D=256 P=4 (plan FUSED_UP by AUTO): batch = frames per launch in {8,12,13,16,24,26,32}
#   (b) 1920x1080 D=256 P=8: batch = frames per launch in {4,6,8}
#   (c) configs[1] 1242x375 D=64 P=4: 16 / 32 / 48 frames per launch, with rocprofv3 --stats of each
# Output: gpurun_out/r04_e1/summary.txt (one line per run: tag, pairs/s, ms per step, stage times per launch, us per frame of aggregate / wta)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_e1; mkdir -p $O; cd $R
line() { python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages_ms_per_launch"]; f=d["roofline"]["frames_per_launch"]; print(sys.argv[2], d["value"], d["ms_per_step"], s, "per-frame us: agg %.1f wta %.1f" % (1e3*s.get("aggregate",0)/f, 1e3*s.get("wta",0)/f), "plan", d["config"]["launch_plan"]["plan"], "frac_moved", d["roofline"]["frac_moved"])' $1 $2 >> $O/summary.txt; tail -1 $O/summary.txt; }
run() { tag=$1; shift; timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-pcie --steps 20 --repeats 3 "$@" > $O/$tag.json 2> $O/$tag.err && line $O/$tag.json $tag || { echo "$tag FAILED"; tail -3 $O/$tag.err; }; }
for n in 16 8 12 13 24 26 32; do run ref_b$n --disparities 256 --paths 4 --batch $n --chunk $n; done
for n in 4 6 8; do run c3_b$n --width 1920 --height 1080 --disparities 256 --paths 8 --batch $n --chunk $n; done
for n in 16 32 48; do run c1_b$n --disparities 64 --paths 4 --batch $n --chunk $n; done
run c2_slabs; run c2_fused --plan fused_up
cd /tmp
for n in 16 32 48; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_c1_b$n -o s -- python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 20 --repeats 3 --disparities 64 --paths 4 --batch $n --chunk $n > $O/c1_b${n}_rocprof.json 2> $O/c1_b${n}_rocprof.err
  f=$(ls $O/st_c1_b$n/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_c1_b$n.csv && echo "c1_b$n stats:" >> $O/summary.txt && python3 $R/profiles/tools/kernel_avgs.py $f | head -4 >> $O/summary.txt; rm -rf $O/st_c1_b$n
done
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_ref -o s -- python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 20 --repeats 3 --disparities 256 --paths 4 > $O/ref_rocprof.json 2> $O/ref_rocprof.err
f=$(ls $O/st_ref/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_ref.csv && echo "ref stats:" >> $O/summary.txt && python3 $R/profiles/tools/kernel_avgs.py $f | head -5 >> $O/summary.txt; rm -rf $O/st_ref
echo done >> $O/summary.txt

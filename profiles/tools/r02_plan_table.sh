# Plan decision table: ms per step of the bench with --plan slabs / fused_up for every configuration AUTO has to decide.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/plans; mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 10 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["config"]["launch_plan"]["plan"], d["stages_ms_per_launch"])' $O/$name.json $name; }
for plan in slabs fused_up; do
 run d256p4_$plan --disparities 256 --paths 4 --plan $plan
 run d256p8_$plan --disparities 256 --paths 8 --plan $plan
 run d128p8_$plan --plan $plan
 run d128p4_$plan --disparities 128 --paths 4 --plan $plan
 run d64p4_$plan --disparities 64 --paths 4 --plan $plan
 run 1080p_d256p8_b4_$plan --width 1920 --height 1080 --disparities 256 --paths 8 --batch 4 --plan $plan
 run 1080p_d256p4_b4_$plan --width 1920 --height 1080 --disparities 256 --paths 4 --batch 4 --plan $plan
 run 1080p_d128p8_b4_$plan --width 1920 --height 1080 --disparities 128 --paths 8 --batch 4 --plan $plan
done

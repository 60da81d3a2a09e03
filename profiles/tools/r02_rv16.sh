R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rv16; mkdir -p $O
timeout -k 10 600 python3 -m pytest $R/tests -q -m gpu 2>&1 | tail -2
run() { name=$1; shift; timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 40 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["config"]["launch_plan"]["plan"], d["stages_ms_per_launch"])' $O/$name.json $name; }
for rep in 1 2; do
run ref --disparities 256 --paths 4
run c3 --width 1920 --height 1080 --disparities 256 --batch 4
run d256p8 --disparities 256 --paths 8
run c2_fused --plan fused_up
run c1_fused --disparities 64 --paths 4 --plan fused_up
done

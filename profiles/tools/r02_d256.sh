R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d256; mkdir -p $O
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_parity.py -q -m gpu -k "fused or plans or randomized or 1080p" 2>&1 | tail -2
run() { name=$1; shift; timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 20 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["config"]["launch_plan"]["plan"], d["stages_ms_per_launch"])' $O/$name.json $name; }
for i in 1 2; do
 run d256p4 --disparities 256 --paths 4
 run d256p8 --disparities 256 --paths 8
 run 1080p_d256p8_b4 --width 1920 --height 1080 --disparities 256 --paths 8 --batch 4 --steps 6
 run 1080p_d256p4_b4 --width 1920 --height 1080 --disparities 256 --paths 4 --batch 4 --steps 6
done

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/w256; mkdir -p $O
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_parity.py -q -m gpu -k "stage_by_stage or randomized or 1080p or pairs" 2>&1 | tail -2
run() { name=$1; shift; timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 20 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["config"]["launch_plan"]["plan"], d["stages_ms_per_launch"])' $O/$name.json $name; }
run d256p4_slabs --disparities 256 --paths 4 --plan slabs
run d256p4_fused --disparities 256 --paths 4 --plan fused_up
run d256p8_slabs --disparities 256 --paths 8 --plan slabs
run d256p4_slabs_b4 --disparities 256 --paths 4 --plan slabs --batch 4 --steps 50
run d256p4_auto_b4 --disparities 256 --paths 4 --batch 4 --steps 50
run d256p4_fused_b4 --disparities 256 --paths 4 --plan fused_up --batch 4 --steps 50

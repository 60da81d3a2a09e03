#!/bin/bash
# headline configuration: plan slabs against plan fused_up, alternating, on one box (bench defaults otherwise)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_plans_ab; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for rep in 1 2 3 4 5; do for plan in slabs fused_up; do
  timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 50 --plan $plan ${BENCH_ARGS} > $O/x.json 2> $O/x.err || { echo failed; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], "|", d["value"], d["spread"]["min"], d["spread"]["max"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"], d["stages_ms_per_launch"]["wta"])' $O/x.json $plan | tee -a $O/summary.txt
done; done

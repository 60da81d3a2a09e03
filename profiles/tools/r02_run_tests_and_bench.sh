#!/bin/bash
# one GPU call: the -m gpu suite, then the bench line per launch plan (no CPU baseline / PCIe legs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-run}; mkdir -p $O
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -30
grep -q "rc=0" $O/pytest.rc || echo "TESTS FAILED (continuing to the bench)"
for plan in slabs fused_up pairs; do
  timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 10 --plan $plan ${BENCH_ARGS} > $O/bench_$plan.json 2> $O/bench_$plan.err || { echo "bench $plan failed"; tail -5 $O/bench_$plan.err; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"], d["roofline"]["frac"])' $O/bench_$plan.json $plan
done

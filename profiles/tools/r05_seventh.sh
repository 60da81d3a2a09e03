#!/bin/bash
# GPU box, round 5, seventh call: randomised differential campaign on the round's code (component table in the CCL pass, fused post stage, new placement
# search), soaks, the two launch plans at the headline under the new search, and the clock / power the card holds while bench.py runs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_seventh; mkdir -p $O; cd $R
BUDGET_S=${BUDGET_S:-300} SEED=${SEED:-505} timeout -k 10 700 python3 profiles/tools/parity_fuzz.py > $O/fuzz.txt 2> $O/fuzz.err || { tail -3 $O/fuzz.txt; tail -3 $O/fuzz.err; exit 1; }
tail -1 $O/fuzz.txt
STEPS=1500 timeout -k 10 300 python3 profiles/tools/soak.py > $O/soak_c2.txt 2> $O/soak.err || { tail -3 $O/soak_c2.txt; tail -3 $O/soak.err; exit 1; }; tail -1 $O/soak_c2.txt
DISP=64 PATHS=4 STEPS=3000 timeout -k 10 300 python3 profiles/tools/soak.py > $O/soak_c1.txt 2>> $O/soak.err || { tail -3 $O/soak_c1.txt; exit 1; }; tail -1 $O/soak_c1.txt
for i in 1 2 3 4; do
  for plan in slabs fused_up; do
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 --plan $plan > $O/plan_${plan}_$i.json 2> $O/plan_${plan}_$i.err || { tail -5 $O/plan_${plan}_$i.err; exit 1; }
    python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); p=d.get("placement_tuning") or {}; s=d["stages_ms_per_launch"]; print("plan", sys.argv[2], d["value"], "agg %.4f wta %.4f" % (s["aggregate"], s["wta"]), p.get("mode"), "kept %.3f" % p.get("launch_pair_ms_kept", 0))' $O/plan_${plan}_$i.json $plan | tee -a $O/summary.txt
  done
done
# clock and power while the default bench runs (sampled from a second process; reading only)
( timeout -k 5 120 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 400 --repeats 5 > $O/long_bench.json 2> $O/long_bench.err ) &
BP=$!
sleep 25
for k in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|mclk\|power\|junction\|edge" | tr '\n' ';' >> $O/smi.txt; echo >> $O/smi.txt; sleep 1.5; done
wait $BP
tail -3 $O/smi.txt

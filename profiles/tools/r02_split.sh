# census of batch i+1 on a third stream + post stages on the side stream (--split) against everything but the plane stages on the main stream (default)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/split; mkdir -p $O
timeout -k 10 600 python3 -m pytest $R/tests -q -m gpu 2>&1 | tail -3
run() { name=$1; shift; timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 50 "$@" > $O/x.json 2> $O/x.err || { echo "$name failed"; tail -3 $O/x.err; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"])' $O/x.json "$name"; }
for rep in 1 2 3; do
 for cfg in "c2:" "c1:--disparities_64_--paths_4" "ref:--disparities_256_--paths_4" "c3:--width_1920_--height_1080_--disparities_256_--batch_4"; do
  name=${cfg%%:*}; a=${cfg#*:}; a=${a//_/ }
  run "$name split" $a --split; run "$name no-split" $a
 done
done

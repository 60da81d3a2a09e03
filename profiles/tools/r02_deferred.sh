# Where do the plane stages of a batch go?  one stream / side stream right away / side stream gated behind the next aggregation
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/deferred; mkdir -p $O
timeout -k 10 300 python3 -m pytest $R/tests/test_gpu_parity.py -q -m gpu -k "deferred or gated" 2>&1 | tail -2
run() { name=$1; shift; timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 40 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["config"]["two_stream_pipelining"], d["stages_ms_per_launch"])' $O/$name.json $name; }
for cfg in "c2:" "c1:--disparities_64_--paths_4" "ref:--disparities_256_--paths_4" "d128p4:--disparities_128_--paths_4" "c3:--width_1920_--height_1080_--disparities_256_--batch_4"; do
  name=${cfg%%:*}; a=${cfg#*:}; a=${a//_/ }
  for mode in no-overlap overlap deferred; do run ${name}_$mode $a --$mode; done
done

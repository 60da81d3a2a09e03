# A/B of the product's residency cap against "everything resident" (experiment build, CART_AGG_DYNLDS=0), bench defaults otherwise
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/resid; mkdir -p $O
timeout -k 10 600 python3 -m pytest $R/tests -q -m gpu 2>&1 | tail -2
export CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/exp/libcart_engine.so
one() { a=${1//_/ }; label=$2
  timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 40 $a > $O/x.json 2> $O/x.err || { echo "failed $a $label"; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"])' $O/x.json "$a" $label; }
for rep in 1 2 3; do
 for cfg in "" "--disparities_64_--paths_4" "--disparities_256_--paths_4" "--width_1920_--height_1080_--disparities_256_--batch_4" "--disparities_128_--paths_4"; do
  unset CART_AGG_DYNLDS; one "$cfg" capped
  export CART_AGG_DYNLDS=0; one "$cfg" all_resident
 done
done
unset CART_AGG_DYNLDS CART_ENGINE_LIB
N=960 ONLY=0,1 timeout -k 10 300 python3 $R/profiles/tools/host_loop_throughput.py 2>&1 | grep "steady\|frames_per_launch" | sed 's/| 960 frames.*//'

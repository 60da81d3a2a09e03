R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/resid; mkdir -p $O
export CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/exp/libcart_engine.so
one() { a=${1//_/ }; pad=$2
  CART_AGG_DYNLDS=$pad timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --no-overlap --steps 30 $a > $O/x.json 2> $O/x.err || { echo "failed $a $pad"; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], "pad", sys.argv[3], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"])' $O/x.json "$a" $pad; }
for rep in 1 2 3 4; do for pad in 0 23000 30000; do one "" $pad; done; done
for rep in 1 2 3; do for pad in 0 23000 44000 70000; do one "--disparities_64_--paths_4" $pad; done; done
for rep in 1 2; do for pad in 0 23000 44000; do one "--disparities_256_--paths_4" $pad; done; done
for rep in 1 2; do for pad in 0 23000 44000; do one "--width_1920_--height_1080_--disparities_256_--batch_4" $pad; done; done
for rep in 1 2; do for pad in 0 30000 44000; do one "--disparities_128_--paths_4" $pad; done; done

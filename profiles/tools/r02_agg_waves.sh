# waves per workgroup of the aggregation launch (nothing is shared between its waves): 4 (product) / 2 / 1, same waves resident per CU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/resid; mkdir -p $O
for rep in 1 2; do for cfg in "" "--disparities_64_--paths_4" "--disparities_256_--paths_4" "--width_1920_--height_1080_--disparities_256_--batch_4"; do a=${cfg//_/ }
 for v in exp aw8; do
  CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/$v/libcart_engine.so timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 40 $a > $O/x.json 2> $O/x.err || { echo "failed $v"; tail -2 $O/x.err; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"])' $O/x.json "$a" $v
 done; done; done
CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/aw8/libcart_engine.so timeout -k 10 300 python3 -m pytest $R/tests/test_gpu_parity.py -q -m gpu -k "stage_by_stage or randomized" 2>&1 | tail -2

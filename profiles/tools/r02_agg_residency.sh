# Does the aggregation launch of the 4-path configurations balance better when fewer workgroups are resident per CU
# (unused dynamic LDS as the cap; the hardware dispatcher then hands the remaining blocks to whichever CU frees a slot)?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/resid; mkdir -p $O
export CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/exp/libcart_engine.so
for cfg in "--disparities_64_--paths_4" "--disparities_128_--paths_4" "--disparities_256_--paths_4_--plan_slabs" ""; do
  a=${cfg//_/ }
  for pad in 0 8000 17000 23000 30000 44000 70000; do
    CART_AGG_DYNLDS=$pad timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --no-overlap --steps 30 $a > $O/x.json 2> $O/x.err || { echo "failed $a $pad"; tail -2 $O/x.err; continue; }
    python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], "pad", sys.argv[3], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"])' $O/x.json "$a" $pad
  done
done

# frame loop (D=128, 8 paths, ~6 frames per launch) with and without a cap of 4 workgroups per CU on the aggregation launch
R=$GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$R/cart-slam_amd/build/ab/exp:$LD_LIBRARY_PATH   # RUNPATH of the executable yields to it: the experiment build of the engine
for rep in 1 2 3; do for pad in 0 30464; do
  echo "== pad $pad"; CART_AGG_DYNLDS=$pad N=960 ONLY=0 timeout -k 10 300 python3 $R/profiles/tools/host_loop_throughput.py 2>&1 | grep "steady"
done; done

R=$GRAFT_REPO_ROOT; export PLAN=auto; export TMPDIR=/tmp
for v in v4 allprio3 allprio1 nothing; do for m in 0x04 0x0f; do echo "-- $v batch 16 dirmask $m"; CART_DEBUG_DIRMASK=$m BENCH_ARGS="--disparities 64 --paths 4 --steps 20" bash $R/profiles/tools/r02_variants.sh ap $v | sed "s/.*'aggregate/aggregate/;s/, 'wta.*//"; done; done
cd /tmp
for m in 0x04 0x01; do
CART_DEBUG_DIRMASK=$m CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/v4/libcart_engine.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ap/trace_$m -o s -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-overlap --steps 20 --disparities 64 --paths 4 --batch 1 > /dev/null 2>&1
echo "batch 1 dirmask $m kernel durations:"; python3 $R/profiles/tools/kernel_avgs.py $(ls $R/gpurun_out/ap/trace_$m/*kernel_stats.csv | head -1) | head -4; rm -rf $R/gpurun_out/ap/trace_$m
done

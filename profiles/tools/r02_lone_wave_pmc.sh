R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/lone; mkdir -p $O; cd /tmp
for m in 0x04 0x01; do
 for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  CART_DEBUG_DIRMASK=$m CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/v4/libcart_engine.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$m/p -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-overlap --steps 3 --warmup 1 --disparities 64 --paths 4 --batch 1 > /dev/null 2>&1
 done
 echo "== dirmask $m"; python3 $R/profiles/pmc_summary.py $O/pmc_$m aggregate; rm -rf $O/pmc_$m
done

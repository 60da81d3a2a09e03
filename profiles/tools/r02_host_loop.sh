# profiles/r02_host_loop.txt: the C++ frame loop at the reference's 12 frames in flight (four module lists), then 32 in flight
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/host_loop_r02.txt
timeout -k 10 300 python3 -m pytest $R/tests/test_host.py -q -m gpu 2>&1 | tail -2
( echo "# 12 frames in flight (CARTSLAM_CONCURRENT_RUN_LIMIT of the reference), CARTSLAM_COALESCE default (1)"
  N=960 ONLY=0,1,2,3 timeout -k 10 500 python3 $R/profiles/tools/host_loop_throughput.py 2>&1 | grep -v amdgpu.ids
  echo "# --inflight 32"
  EXTRA="--inflight 32" N=960 ONLY=0,1,4 timeout -k 10 400 python3 $R/profiles/tools/host_loop_throughput.py 2>&1 | grep -v amdgpu.ids ) > $O
grep "^#\|steady\|frames_per_launch" $O | sed 's/| 960 frames.*//'

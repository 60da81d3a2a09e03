R=$GRAFT_REPO_ROOT
for c in 1 2 3; do for inflight in 12 32; do
  echo "== CARTSLAM_COALESCE=$c inflight=$inflight"
  CARTSLAM_COALESCE=$c EXTRA="--inflight $inflight" N=960 ONLY=0,1 timeout -k 10 300 python3 $R/profiles/tools/host_loop_throughput.py 2>&1 | grep "steady\|frames_per_launch" | sed 's/| 960 frames.*//'
done; done

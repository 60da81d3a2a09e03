# one group per module on the GPU (CARTSLAM_COALESCE=1) against two with the second one queued only when N requests have gathered
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "1 1" "2 4" "2 6" "2 8" "3 4"; do set -- $cfg
  echo "== CARTSLAM_COALESCE=$1 AHEAD=$2"
  CARTSLAM_COALESCE=$1 CARTSLAM_COALESCE_AHEAD=$2 N=960 ONLY=0,1 timeout -k 10 300 python3 $R/profiles/tools/host_loop_throughput.py 2>&1 | grep "steady\|frames_per_launch" | sed 's/| 960 frames.*//; s/:.*frames_per_launch/ fpl/'
done; done

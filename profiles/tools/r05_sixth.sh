#!/bin/bash
# GPU box, round 5, sixth call: do the bytes between the slots of a slab group decide the placement mode?  (profiles/tools/r05_place_log.py, four fresh processes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_sixth; mkdir -p $O; cd $R
for i in 1 2 3 4; do
  timeout -k 10 250 python3 profiles/tools/r05_place_log.py 3 17 > $O/log_$i.txt 2> $O/log_$i.err || { tail -5 $O/log_$i.err; exit 1; }
  tail -1 $O/log_$i.txt
done

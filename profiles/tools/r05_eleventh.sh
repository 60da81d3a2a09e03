#!/bin/bash
# GPU box, round 5: the aggregation launch's knobs in the fast placement mode (prefetch depth, waves per workgroup, store kind, priority, launch bound, FIFO depth)
R=$GRAFT_REPO_ROOT; cd $R
PARITY_VARS="vd4 ntoff prio0 lb7 lb5 hspf4 hspf16" PARITY_K="launch_plans_agree or xcd_placed or randomized_configurations or full_size_against_oracle" bash profiles/tools/r05_ab.sh r05_knobs "base vd4 w8 w2 ntoff prio0 lb7 lb5 hspf4 hspf16" 3

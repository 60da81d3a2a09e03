#!/bin/bash
# GPU box, round 5, fourth call: (a) VERDICT r4 item 3 -- 1080p aggregation launch with per-XCD private census copies (can the Infinity Cache be what
# serves the 8-fold re-fetch?), (b) the frame-major order of the vertical / diagonal directions on the other launch shapes
R=$GRAFT_REPO_ROOT; cd $R
CONFIGS="c3" PMC_VARS="base privcen0 privcen" PMC_PASSES="FETCH_SIZE;TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" PARITY_K="full_size_oracle_1080p or launch_plans_agree or randomized_configurations" \
  bash profiles/tools/r05_ab.sh r05_privcen "base privcen0 privcen" 3 || exit $?
CONFIGS="c2b32 c1 c2b8 c3" PARITY_VARS="" bash profiles/tools/r05_ab.sh r05_fm "base fm" 3

#!/bin/bash
# usage (GPU box): collect_all.sh <round tag, e.g. r03>: bench line + kernel stats + PMC passes of the four single-GPU configurations
# DESIGN.md section 5 quotes, under gpurun_out/<round>_{c2,c1,c3,ref}/ (one GPU call, ~4 minutes)
R=$GRAFT_REPO_ROOT; T=${1:-r03}
FULL=1 bash $R/profiles/tools/collect.sh ${T}_c2 &&
bash $R/profiles/tools/collect.sh ${T}_c1 --disparities 64 --paths 4 &&
bash $R/profiles/tools/collect.sh ${T}_c3 --width 1920 --height 1080 --disparities 256 --paths 8 --batch 4 &&
bash $R/profiles/tools/collect.sh ${T}_ref --disparities 256 --paths 4

# final round-2 artefacts of the four configurations DESIGN.md section 5 quotes (one GPU call)
R=$GRAFT_REPO_ROOT
FULL=1 bash $R/profiles/tools/r02_collect.sh r02_c2 &&
bash $R/profiles/tools/r02_collect.sh r02_c1 --disparities 64 --paths 4 &&
bash $R/profiles/tools/r02_collect.sh r02_c3 --width 1920 --height 1080 --disparities 256 --paths 8 --batch 4 &&
bash $R/profiles/tools/r02_collect.sh r02_ref --disparities 256 --paths 4

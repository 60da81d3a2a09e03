R=$GRAFT_REPO_ROOT
for cfg in "64 4" "256 4" "128 8"; do set -- $cfg; D=$1 P=$2 timeout -k 10 300 python3 $R/profiles/tools/cu_mask_overlap.py 2>&1 | grep -v amdgpu.ids; done

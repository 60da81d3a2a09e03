#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_rvdiag; mkdir -p $O; cd $R
timeout -k 10 120 python3 profiles/tools/r04_rvdiag.py $O/base.npz 2> $O/base.err || exit 1
CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/spec/libcart_engine.so timeout -k 10 120 python3 profiles/tools/r04_rvdiag.py $O/spec.npz 2> $O/spec.err || { grep -qs "fault" $O/spec.err && exit 9; exit 1; }
python3 - <<'PY'
import numpy as np, os
O=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r04_rvdiag")
a=np.load(O+"/base.npz"); b=np.load(O+"/spec.npz")
for k in ("wl","wr","disp"):
    df=np.argwhere(a[k]!=b[k]); print(k, len(df), "differences")
    for y,x in df[:40]: print("  y",y,"x",x,"x%16",x%16,"base",hex(int(a[k][y,x])),"spec",hex(int(b[k][y,x])))
PY

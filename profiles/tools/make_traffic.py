#!/usr/bin/env python3
"""Adds the HBM traffic of one configuration's aggregate / WTA launches to profiles/traffic.json.
usage: make_traffic.py <pmc_summary.txt> <W> <H> <D> <P> <frames per launch> <plan> [round]
Input = profiles/pmc_summary.py output holding FETCH_SIZE and WRITE_SIZE means (KB per dispatch, separate --pmc passes).
gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE counts 128-byte requests at 64 bytes -> doubled;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import json
import os
import re
import sys

summary, w, h, D, P, fpl, plan = sys.argv[1], *(int(v) for v in sys.argv[2:7]), sys.argv[7]
rnd = int(sys.argv[8]) if len(sys.argv) > 8 else 3
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(root, "profiles", "traffic.json")
tj = json.load(open(path)) if os.path.exists(path) else {}
kernel, vals = None, {}
for line in open(summary):
    if not line.startswith(" "):
        kernel = line.strip()
        continue
    m = re.match(r"\s+(\S+)\s+mean\s+([0-9.]+)", line)
    if m and kernel:
        vals.setdefault(kernel, {})[m.group(1)] = float(m.group(2))
suffix = "" if plan == "slabs" else "_" + plan
slabs = {"slabs": P, "fused_up": P - 1}[plan]
for k, v in vals.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    if "aggregate_kernel" in k:
        name, alg = "aggregate", w * h * (8 * slabs + slabs * D) * fpl
    elif "wta_kernel" in k or "wta_fused_kernel" in k:
        name, alg = "wta", w * h * (slabs * D + 4) * fpl
    else:
        continue
    tj[f"{name}_{w}x{h}_D{D}_P{P}_B{fpl}{suffix}"] = {
        "kernel": k.replace("void cart_amd::", ""), "fetch_size_kb_raw": round(v["FETCH_SIZE"], 1), "write_size_kb": round(v["WRITE_SIZE"], 1),
        "hbm_bytes_per_launch": int(round(v["WRITE_SIZE"] * 1024 + 2 * v["FETCH_SIZE"] * 1024)), "alg_bytes_per_launch": alg, "round": rnd}
    if "SQ_INSTS_VALU" in v:   # wave-level VALU instructions per launch (bench.py prints the aggregation launch's issue fraction beside its HBM fraction)
        tj[f"{name}_{w}x{h}_D{D}_P{P}_B{fpl}{suffix}"]["valu_insts_per_launch"] = int(round(v["SQ_INSTS_VALU"]))
json.dump(tj, open(path, "w"), indent=1)
print("updated", path, [k for k in tj if k != "_note"])

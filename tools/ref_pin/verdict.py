#!/usr/bin/env python3
"""What the reference outputs say: compares every tests/golden/ref/ref_disparity_<case>.bin (written by ref_pin.cpp) with the CPU oracle
under each of the eight settings of the three choices that are open upstream (S8, S7, S5: oracle/cart_oracle.h) and prints the ONE thing a
maintainer has to do with the answer -- which `cart_engine_set_option` defaults to flip, or that nothing has to change, or that the difference
is none of the three.  CPU only (oracle); exit 0 = the oracle's current defaults reproduce the reference, 3 = a flip of defaults does,
4 = no setting does.
    python tools/ref_pin/verdict.py [ref dir = tests/golden/ref]"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "cart-slam_amd")]

OPTIONS = {1: ("CART_OPT_SPEC_S8_ZERO_INVALID", "CART_ORACLE_VARIANT_S8_ZERO_INVALID", "the LR check also invalidates integer disparity 0"),
           2: ("CART_OPT_SPEC_S7_REPLICATE_BORDER", "CART_ORACLE_VARIANT_S7_REPLICATE_BORDER", "3x3 medians over a replicated border"),
           4: ("CART_OPT_SPEC_S5_TOP2", "CART_ORACLE_VARIANT_S5_TOP2", "uniqueness from the second-best cost only")}


def verdict(ref_dir):
    """-> dict(files, version, diffs {variant set: differing pixels summed over the cases}, best, exact)."""
    import oracle_lib as O   # the checker
    from test_ref_pin import case_name, load_ref, ref_files
    files = ref_files(ref_dir)
    diffs = {v: 0 for v in range(8)}
    per_case = {}
    for p in files:
        ref, (l, r, md, D, P) = load_ref(p)
        per_case[case_name(p)] = {}
        for v in range(8):
            n = int((O.disparity_module(l, r, D, P, md, radius=-1, variants=v) != ref).sum())
            diffs[v] += n
            per_case[case_name(p)][v] = n
    best = min(diffs, key=lambda v: (diffs[v], v))
    vf = os.path.join(ref_dir, "OPENCV_VERSION.txt")
    return {"files": len(files), "version": open(vf).read().strip() if os.path.exists(vf) else "unknown", "diffs": diffs, "per_case": per_case,
            "best": best, "exact": bool(files) and diffs[best] == 0}


def report(v):
    lines = [f"reference outputs: {v['files']} cases, OpenCV {v['version']}",
             "differing pixels by variant set (bit 1 = S8, 2 = S7, 4 = S5): " + ", ".join(f"{k}: {n}" for k, n in v["diffs"].items())]
    if not v["files"]:
        return lines + ["no ref_disparity_*.bin files found: run tools/ref_pin/run.sh first"], 1
    if v["exact"] and v["best"] == 0:
        return lines + ["VERDICT: the oracle's defaults reproduce the reference bit for bit.  Nothing to flip: commit tests/golden/ref/ and parity is pinned."], 0
    if v["exact"]:
        lines.append(f"VERDICT: variant set {v['best']} reproduces the reference bit for bit.  Flip these defaults (engine, per engine before the first call / oracle):")
        for bit, (opt, var, what) in OPTIONS.items():
            if v["best"] & bit:
                lines.append(f"  cart_engine_set_option(engine, {opt}, 1);   oracle: variants |= {var}   ({what})")
        lines.append("  to make them the defaults: `opt_spec`'s initial value in cart-slam_amd/csrc/cart_engine.hip (struct cart_engine), the default `variants` in "
                     "oracle/cart_oracle.c::cart_oracle_sgm and in tests/oracle_lib.py; then python tests/golden/make_golden.py and both test suites")
        return lines, 3
    worst_case = max(v["per_case"], key=lambda c: v["per_case"][c][v["best"]])
    lines.append(f"VERDICT: no setting of S8 / S7 / S5 reproduces the reference (closest: set {v['best']}, {v['diffs'][v['best']]} pixels; worst case {worst_case}).  "
                 "The difference is another spec item: tools/ref_pin/README.md, 'With n > 0 for every v'.")
    return lines, 4


if __name__ == "__main__":
    lines, rc = report(verdict(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ref")))
    print("\n".join(lines))
    sys.exit(rc)

"""Randomised differential campaign, GPU engine against the CPU oracle (test infrastructure, not part of the suite): random image
sizes (ragged, 16..2048 wide), D, paths, min_disparity, P1 / P2, uniqueness ratio, smoothing radius / iterations, spec variants,
launch plan, chunking, batch size, gray / BGR, scene family -- whole disparity module of every frame of the batch, then plane
derivative + histogram + classification + CCL (every other case with the component table) of frame 0.  Time-boxed.   BUDGET_S=600 SEED=1 python profiles/tools/parity_fuzz.py
Prints one line per case; exits non-zero on the first differing value (after printing the parameters that reproduce it)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib as O      # checker
from cartslam import Engine, synth

budget = float(os.environ.get("BUDGET_S", 300)); seed = int(os.environ.get("SEED", 1))
rng = random.Random(seed)
t0 = time.time(); cases = frames = 0
while time.time() - t0 < budget:
    D = rng.choice((64, 128, 256)); P = rng.choice((4, 8))
    big = rng.random() < 0.25
    w = rng.randint(600, 2048) if big else rng.randint(16, 600)
    h = rng.randint(200, 800) if big else rng.randint(8, 200)
    if w * h * D * P > 3.0e9:      # keep one oracle frame under ~10 s
        continue
    md = rng.choice((0, 1, 4, 4, 4, 7, 16, 33, 64))
    p1 = rng.randint(0, 40); p2 = rng.randint(p1, 224)
    uniq = rng.choice((0, 1, 5, 12, 12, 12, 15, 30, 50, 99, 100))
    radius = rng.choice((-1, -1, 1, 2, 2, 3, 5, 8)); iters = rng.randint(1, 5)
    variants = rng.choice((0, 0, 0, 1, 2, 4, 3, 5, 6, 7))
    plan = rng.choice(("auto", "auto", "slabs", "fused_up"))
    B = rng.choice((1, 1, 2, 3, 5, 8)) if not big else rng.choice((1, 2))
    chunk = rng.choice((0, 0, 0, 1, 2, 3))
    ch = rng.choice((1, 1, 3)); scene = rng.choice(synth.SCENES); s = rng.randint(0, 1 << 30)
    desc = f"w={w} h={h} D={D} P={P} md={md} p1={p1} p2={p2} uniq={uniq} r={radius} it={iters} var={variants} plan={plan} B={B} chunk={chunk} ch={ch} scene={scene} seed={s}"
    # a third of the cases tune the slab placement first (cart_engine_tune_placement: the workspace is swapped for another one); some give the
    # engine more slots than the call needs, large cases up to a workspace above 8 GiB (several slot groups, each its own hipMalloc)
    inflight = B
    if rng.random() < 0.3:
        inflight = B + rng.randint(1, 24)
        while inflight > B and inflight * w * h * D * P > 24e9:
            inflight -= 1
    tune = rng.random() < 0.33
    desc += f" inflight={inflight} tune={int(tune)}"
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=md, p1=p1, p2=p2, uniqueness_ratio=uniq, smoothing_radius=radius,
                 smoothing_iterations=iters, max_inflight=inflight)
    eng.set_plan(plan)
    if chunk:
        eng.set_chunk_frames(chunk)
    eng.set_spec_variants(s8_zero_invalid=bool(variants & 1), s7_replicate_border=bool(variants & 2), s5_top2=bool(variants & 4))
    if tune:
        eng.tune_placement(B, 3)
    pairs = [synth.make_pair(w, h, D, md, seed=s + k, frame=k, channels=ch, scene=scene)[:2] for k in range(B)]
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda(); R = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    got = eng.compute_disparity(L, R).cpu().numpy()
    diff = 0
    for k in range(B):
        exp = O.disparity_module(pairs[k][0], pairs[k][1], D, P, md, p1=p1, p2=p2, uniq=uniq, radius=radius, iterations=iters, variants=variants)
        diff += int((got[k] != exp).sum())
        if k == 0:
            exp0 = exp
    d0 = torch.from_numpy(got[0]).cuda()
    hist = torch.zeros(256, dtype=torch.int32, device="cuda")
    pd = eng.plane_derivative_hist(d0, hist)
    eb, eh = O.plane_derivative(exp0)
    ok, pp = O.histogram_peak_params(eh)
    if not ok:
        pp = (6, 18, -5, 6, 11, 0)
    planes = eng.plane_classify(pd, pp)
    ep = O.classify(eb, pp); eids, en = O.ccl(ep)
    if cases % 2:   # ids + count alone (three launches) ...
        ids, n = eng.plane_ccl(planes)
    else:           # ... or with the component table from the same pass (cart_plane_ccl_table), whose scratch must come back all zeros
        cap = rng.choice((en + 1, max(1, en // 2), 4096))
        ids, table, n = eng.plane_ccl_table(planes, max_components=cap)
        et, _ = O.ccl_stats(ep, eids, max_components=cap)
        diff += int((table.cpu().numpy().reshape(-1, 7)[:len(et)] != et).sum()) + int(eng.debug_ccl_scratch_nonzero())
    diff += int((pd.cpu().numpy() != eb).sum()) + int((hist.cpu().numpy() != eh).sum()) + int((planes.cpu().numpy() != ep).sum()) + \
            int((ids.cpu().numpy() != eids).sum()) + int(int(n.item()) != en)
    eng.close()
    cases += 1; frames += B
    print(f"{cases:4d} {desc}: valid {float((exp0 != -32768).mean()):.3f} components {en} differing values {diff}", flush=True)
    if diff:
        print("FAILED: " + desc); sys.exit(1)
print(f"parity fuzz: seed {seed}, {cases} cases, {frames} frames, 0 differing values, {time.time() - t0:.0f} s")

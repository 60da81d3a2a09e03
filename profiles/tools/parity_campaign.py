"""One-off parity campaign at full size (not part of the test suite: ~2 minutes of oracle time): every scene variant x several seeds
and frames x the three 1242x375 configurations + BGR inputs, whole disparity module + plane labelling + CCL against the oracle.
Prints one line per case and a total; exits non-zero on any differing pixel.   N_SEEDS=4 python profiles/tools/parity_campaign.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib as O      # checker (this tool is test infrastructure)
from cartslam import Engine, synth
w, h = 1242, 375
n_seeds = int(os.environ.get("N_SEEDS", 3))
cases, bad, t0 = 0, 0, time.time()
for (D, P) in ((128, 8), (64, 4), (256, 4)):
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2)
    for scene in synth.SCENES:
        for s in range(n_seeds):
            ch = 3 if (s % 3 == 2) else 1
            l, r, _ = synth.make_pair(w, h, D, 4, seed=9000 + 17 * s + D, frame=5 * s, channels=ch, scene=scene)
            exp = O.disparity_module(l, r, D, P, 4, radius=2, iterations=1)
            d = eng.compute_disparity(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda())
            hist = torch.zeros(256, dtype=torch.int32, device="cuda")
            pd = eng.plane_derivative_hist(d, hist)
            eb, eh = O.plane_derivative(exp)
            ok, pp = O.histogram_peak_params(eh)
            if not ok:
                pp = (6, 18, -5, 6, 11, 0)
            planes = eng.plane_classify(pd, pp)
            ids, n = eng.plane_ccl(planes)
            ep = O.classify(eb, pp)
            eids, en = O.ccl(ep)
            diff = int((d.cpu().numpy() != exp).sum()) + int((pd.cpu().numpy() != eb).sum()) + int((hist.cpu().numpy() != eh).sum()) + \
                   int((planes.cpu().numpy() != ep).sum()) + int((ids.cpu().numpy() != eids).sum()) + int(int(n.item()) != en)
            cases += 1; bad += diff != 0
            print(f"D={D} P={P} {scene:11s} seed {s} ch {ch}: valid {float((exp != -32768).mean()):.3f} components {en:5d} differing values {diff}")
    eng.close()
print(f"parity campaign: {cases} full-size frames, {bad} with differences, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)

"""Diagnostic: the product path against FROG_REFERENCE_ORDER=1 (both on the device), free-running over a schedule; prints the
relative deviation of E and of the coordinates after every iteration so that a discrete event (one half-link deciding
`w < threshold` differently, imageGroup.cxx:274) shows as a jump.  Usage: diag_ref_trajectory.py [n_images n_points pairs li dl di]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from frog_amd.pairs import Pairs
import test_gpu_reference_order as T

a = [int(x) for x in sys.argv[1:]] or [6, 3000, 1500, 50, 3, 200]
# optional 7th argument: partner images per image (0 = every image pair), 8th: seed
pairs = Pairs.synthetic(a[0], a[1], a[2], seed=(a[7] if len(a) > 7 else 7), partners_per_image=(a[6] if len(a) > 6 else 0))
os.environ["FROG_REFERENCE_ORDER"] = "1"
ref = T.Side(pairs)
del os.environ["FROG_REFERENCE_ORDER"]
fast = T.Side(pairs)
prev = [0.0]
def check(tag, sides, e=None, infos=None):
    if e is None or tag[0] == "step" or e[0] < 0:
        if e is not None and e[0] < 0: print(tag, "rejected")
        return
    de = abs(e[0] - e[1]) / abs(e[1])
    x0, x1 = sides[0].xyz2().astype(np.float64), sides[1].xyz2().astype(np.float64)
    dx = float(np.max(np.abs(x0 - x1)))
    ca, cb = sides[0].g.countInliers(), sides[1].g.countInliers()
    census = [ca[i].inliers - cb[i].inliers for i in range(pairs.n_images)]
    flag = "  <-- jump" if de > 10 * max(prev[0], 1e-9) else ""
    if pairs.n_images > 12:          # large groups: the images whose census differs, not the whole list
        census = {i: c for i, c in enumerate(census) if c}
    print(tag, f"dE {de:.2e} max|dxyz2| {dx:.2e} mm census diff {census}{flag}", flush=True)
    prev[0] = de
T.lockstep([fast, ref], a[3], a[4], a[5], check)

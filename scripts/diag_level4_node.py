"""Diagnostic: which image's proposal makes the group mean differ at a level-4 node (product path vs reference order)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from frog_amd.pairs import Pairs
import test_gpu_reference_order as T
from lattice_util import lattice_taps

pairs = Pairs.synthetic(40, 20000, 16667, seed=2, partners_per_image=20)
os.environ["FROG_REFERENCE_ORDER"] = "1"
ref = T.Side(pairs)
del os.environ["FROG_REFERENCE_ORDER"]
fast = T.Side(pairs)
po = np.asarray(pairs.point_offset)
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 4


class Stop(Exception):
    pass


def check(tag, sides, e=None, infos=None):
    if isinstance(tag, tuple) and tag[0] in ("deformable", "step") and tag[1] >= 3:
        print(tag, "stray points so far (product path):", sides[0].g.stray_points(), flush=True)
    if not (isinstance(tag, tuple) and tag[0] == "deformable" and tag[1] == 4 and tag[2] == n_it - 1):
        return
    k = sides[0].num_grids() - 1
    c0 = [sides[0].grid(i, k)[1] for i in range(40)]
    info = sides[0].grid(0, k)[0]
    c1 = [sides[1].grid(i, k)[1] for i in range(40)]
    dev = np.stack([np.abs(a - b).max(axis=1) for a, b in zip(c0, c1)])          # [image, node]
    img, node = np.unravel_index(np.argmax(dev), dev.shape)
    print("worst deviation", dev[img, node], "image", img, "node", node, "of", dev.shape, "max|c|", max(np.abs(b).max() for b in c1), flush=True)
    gw = np.array([sides[1].g.gradient(i, len(c1[0]))[node] for i in range(40)])  # reference-order gradient at that node, last step
    for i in range(40):
        print(f"  image {i:2d} gw {gw[i][3]:.6e} g {gw[i][:3]} c_fast {c0[i][node]} c_ref {c1[i][node]} diff {c0[i][node] - c1[i][node]}", flush=True)
    # the points of the touched images that support the node
    xyz = sides[1].xyz()
    for i in np.nonzero(gw[:, 3] > 0)[0][:6]:
        idx, wt = lattice_taps(xyz[po[i]:po[i + 1]], info)
        hit = np.nonzero(idx == node)
        ps = sides[1].point_sums()[po[i]:po[i + 1]]
        psf = sides[0].point_sums()[po[i]:po[i + 1]]
        print("  image", i, "supporting points:", [(int(p), float(wt[p, t]), ps[p].tolist(), psf[p].tolist()) for p, t in zip(*hit) if ps[p][3] != 0 or psf[p][3] != 0][:6], flush=True)
    # image `img`: the points around the node
    i = int(img)
    dims = list(info.dims)
    nx, ny, nz = node % dims[0], (node // dims[0]) % dims[1], node // (dims[0] * dims[1])
    print("  node", (nx, ny, nz), "of", dims, "origin", list(info.origin), "spacing", list(info.spacing), flush=True)
    pts = xyz[po[i]:po[i + 1]].astype(np.float64)
    q = (pts - np.array(list(info.origin))) / np.array(list(info.spacing))
    near = np.nonzero((np.abs(q[:, 0] - nx) < 3.0) & (np.abs(q[:, 1] - ny) < 3.0) & (np.abs(q[:, 2] - nz) < 3.0))[0]
    ps, psf = sides[1].point_sums()[po[i]:po[i + 1]], sides[0].point_sums()[po[i]:po[i + 1]]
    for p in near:
        print("   point", int(p), "lattice coords", q[p].tolist(), "sums ref", ps[p].tolist(), "fast", psf[p].tolist(), flush=True)
    raise Stop()


try:
    T.lockstep([fast, ref], 20, 5, 12, check)
except Stop:
    pass

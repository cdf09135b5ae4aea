"""Diagnostic: are the level-4 deviations between the product path and reference order the control points that one run's
scatter reaches with a vanishing weight (a point within an ulp of a cell face) and the other's does not reach at all?

A control point reached only by such a point gets gw = 1e-17 > 0 and the full proposal g / gw (imageGroup.cxx:346-375 divides
whatever the weight is); reached by nothing it keeps its value.  So one ulp in a point's position, across a cell face, is one
whole step at up to 16 control points of that image -- and, through the group mean, step / nImages at the same control points
of every other image."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from frog_amd.pairs import Pairs
import test_gpu_reference_order as T

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 40
li, dl, di = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (20, 5, 12)
pairs = Pairs.synthetic(n_images, 20000, 16667, seed=2, partners_per_image=20)
os.environ["FROG_REFERENCE_ORDER"] = "1"
ref = T.Side(pairs)
del os.environ["FROG_REFERENCE_ORDER"]
fast = T.Side(pairs)
po = np.asarray(pairs.point_offset)


def cells(x, info):
    """scatter's cell and fraction (imageGroup.cxx:303-310): the lattice coordinate rounded to f32, floor."""
    q = ((x.astype(np.float64) - np.array(list(info.origin))) / np.array(list(info.spacing))).astype(np.float32)
    c = np.floor(q)
    return c.astype(np.int64), q - c


def stencil_nodes(c, dims):
    """node indices of the 4^3 stencils of cells c [n, 3] -> [n, 64] (-1 outside the lattice)"""
    o = np.arange(-1, 3)
    gx = c[:, 0, None] + o[None, :]; gy = c[:, 1, None] + o[None, :]; gz = c[:, 2, None] + o[None, :]
    ok = ((gx >= 0) & (gx < dims[0]))[:, None, None, :] & ((gy >= 0) & (gy < dims[1]))[:, None, :, None] & ((gz >= 0) & (gz < dims[2]))[:, :, None, None]
    idx = gx[:, None, None, :] + dims[0] * (gy[:, None, :, None] + dims[1] * gz[:, :, None, None])
    return np.where(ok, idx, -1).reshape(len(c), 64)


state = {}


def check(tag, sides, e=None, infos=None):
    if not isinstance(tag, tuple):
        return
    if tag[0] == "setup":
        state["xyz"] = [s.xyz().copy() for s in sides]          # a lattice's input positions: fixed until the next set-up
        return
    if not ((tag[0] == "deformable" and (tag[2] == di - 1 or (tag[1] == 4 and tag[2] < 6))) or (tag[0] == "step" and e[0] < 0)):
        return
    level = tag[1]
    k = sides[0].num_grids() - 1
    info = sides[0].grid(0, k)[0]
    dims = list(info.dims)
    n_cp = dims[0] * dims[1] * dims[2]
    fragile = np.zeros(n_cp, bool)          # control points some image reaches in one run only (or with weight 0 in one)
    n_pts = 0
    sums = [s.point_sums() for s in sides]
    for i in range(n_images):
        a, b = state["xyz"][0][po[i]:po[i + 1]], state["xyz"][1][po[i]:po[i + 1]]
        ca, fa = cells(a, info)
        cb, fb = cells(b, info)
        live = (sums[0][po[i]:po[i + 1], 3] != 0) | (sums[1][po[i]:po[i + 1], 3] != 0)
        m = live & (np.any(ca != cb, axis=1) | np.any((fa == 0) != (fb == 0), axis=1))
        n_pts += int(m.sum())
        if m.any():
            na, nb = stencil_nodes(ca[m], dims), stencil_nodes(cb[m], dims)
            for r in range(len(na)):
                # every control point of both stencils: the ones in one stencil only, and the face planes whose weight is
                # 0 on one side and 1e-17 on the other
                for n in np.concatenate([na[r], nb[r]]):
                    if n >= 0: fragile[n] = True
    worst_in, worst_out, scale = 0.0, 0.0, 0.0
    n_big_out = 0
    for i in range(n_images):
        c0, c1 = sides[0].grid(i, k)[1], sides[1].grid(i, k)[1]
        d = np.abs(c0.astype(np.float64) - c1).max(axis=1)
        scale = max(scale, float(np.abs(c1).max()))
        if fragile.any(): worst_in = max(worst_in, float(d[fragile].max()))
        worst_out = max(worst_out, float(d[~fragile].max()))
        n_big_out += int((d[~fragile] > 1e-3).sum())
    print(tag, f"level {level} lattice {k} dims {dims}: points across a face {n_pts}, fragile control points {int(fragile.sum())} of {n_cp}; "
          f"max |dc| on them {worst_in:.3e}, elsewhere {worst_out:.3e} ({n_big_out} beyond 1e-3), max |c| {scale:.3e}", flush=True)


T.lockstep([fast, ref], li, dl, di, check)

"""Diagnostic: where the product path and FROG_REFERENCE_ORDER=1 part at level 4 of cfg 5 (no census difference)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from frog_amd.pairs import Pairs
import test_gpu_reference_order as T

pairs = Pairs.synthetic(500, 20000, 16667, seed=1, partners_per_image=60)
os.environ["FROG_REFERENCE_ORDER"] = "1"
ref = T.Side(pairs)
del os.environ["FROG_REFERENCE_ORDER"]
fast = T.Side(pairs)
po = np.asarray(pairs.point_offset)


class Stop(Exception):
    pass


def check(tag, sides, e=None, infos=None):
    if isinstance(tag, tuple) and tag[0] == "step" and tag[1] == 4 and tag[2] in (0, 3, 6):
        ps0, ps1 = sides[0].point_sums(), sides[1].point_sums()
        d = np.abs(ps0 - ps1)
        print(tag, "point sums: max abs dev", d.max(axis=0), "rel of max", d.max() / np.abs(ps1).max(), flush=True)
    if isinstance(tag, tuple) and tag[0] == "deformable" and tag[1] == 4 and tag[2] in (0, 3, 6):
        x0, x1 = sides[0].xyz2().astype(np.float64), sides[1].xyz2().astype(np.float64)
        d = np.linalg.norm(x0 - x1, axis=1)
        top = np.argsort(-d)[:8]
        print(tag, "max |dxyz2|", d.max(), "points over 1e-3 mm:", int((d > 1e-3).sum()), "over 1e-2:", int((d > 1e-2).sum()), flush=True)
        ps0, ps1 = sides[0].point_sums(), sides[1].point_sums()
        for p in top:
            img = int(np.searchsorted(po, p, side="right") - 1)
            print("  point", int(p), "image", img, "dev mm", d[p], "xyz", x1[p], "sums fast", ps0[p], "ref", ps1[p], "links", int(pairs.row_ptr[p + 1] - pairs.row_ptr[p]), flush=True)
        if tag[2] == 6:
            k = sides[0].num_grids() - 1
            img = int(np.searchsorted(po, top[0], side="right") - 1)
            info, c0 = sides[0].grid(img, k)
            _, c1 = sides[1].grid(img, k)
            dc = np.abs(c0 - c1).max(axis=1)
            worst = np.argsort(-dc)[:6]
            print("  lattice", k, "image", img, "dims", list(info.dims), "max|c|", np.abs(c1).max(), "worst nodes", [(int(n), float(dc[n]), c0[n].tolist(), c1[n].tolist()) for n in worst], flush=True)
            gr = sides[1].g.gradient(img, len(c0))          # reference-order mode: the gradient image as the scatter left it
            print("  reference-order gradient at the worst nodes (sum w sDisp xyz, sum w sWeight):", [(int(n), gr[n].tolist()) for n in worst], flush=True)
            print("  nodes of this image with 0 < gw < 1e-30:", int(((gr[:, 3] > 0) & (gr[:, 3] < 1e-30)).sum()), " < 1e-20:", int(((gr[:, 3] > 0) & (gr[:, 3] < 1e-20)).sum()),
                  " < 1e-10:", int(((gr[:, 3] > 0) & (gr[:, 3] < 1e-10)).sum()), " touched:", int((gr[:, 3] > 0).sum()), "of", len(gr), flush=True)
            small = (gr[:, 3] > 0) & (gr[:, 3] < 1e-20)
            print("  max coefficient deviation on nodes with gw < 1e-20:", float(dc[small].max()) if small.any() else None,
                  " on nodes with gw >= 1e-10:", float(dc[gr[:, 3] >= 1e-10].max()), " on untouched nodes:", float(dc[gr[:, 3] == 0].max()), flush=True)
            raise Stop()


try:
    T.lockstep([fast, ref], 20, 5, 40, check)
except Stop:
    pass

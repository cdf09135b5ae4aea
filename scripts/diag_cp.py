#!/usr/bin/env python3
"""Free-running comparison, seed 3: when does control point (12,2,7) of image 2, level 1, leave the oracle?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frog_amd import _abi                      # noqa: E402
from frog_amd.image_group import ImageGroup    # noqa: E402
from frog_amd.pairs import Pairs               # noqa: E402
from oracle.oracle_api import OracleGroup      # noqa: E402

pairs = Pairs.synthetic(6, 3000, 1500, seed=3)
g = ImageGroup(pairs)
ref = OracleGroup(pairs.model, _abi.FrogOptions.default())
ref.setup_stats()
g.setupLinearTransforms(); ref.linear_init()
g.transformPoints(); ref.transform_points()
for it in range(50):
    if it % 10 == 0:
        g.updateStats(); ref.update_stats()
    g.updateLinearTransforms(); ref.linear_step()
    g.transformPoints(); ref.transform_points()
g.transformPoints(True); ref.transform_points(True)
for level in range(2):
    info = g.setupDeformableTransforms(level); ref.deformable_setup(level, _abi.FrogGridInfo())
    g.transformPoints(); ref.transform_points()
    dims = list(info.dims)
    j = 12 + dims[0] * (2 + dims[1] * 7) if level == 1 else 0
    for it in range(40 if level == 0 else 3):
        if it % 10 == 0:
            g.updateStats(); ref.update_stats()
        e = g.updateDeformableTransforms(0.02); er = ref.deformable_step(0.02)
        g.transformPoints(); ref.transform_points()
        if level == 1:
            c = g.grid(2, 1)[1][j]; rc = ref.grid(2, 1, _abi.FrogGridInfo())[1][j]
            ps, rps = g.point_sums(), ref.point_sums()
            dw = np.abs(ps[:, 3] - rps[:, 3])
            k = int(np.argmax(dw / np.maximum(rps[:, 3], 1e-3)))
            gr, rgr = g.gradient(2, dims[0] * dims[1] * dims[2])[j], ref.gradient(2, dims[0] * dims[1] * dims[2])[j]
            if it < 3:
                print("   grad", gr, "vs", rgr)
            print(f"it {it}: c {c} vs {rc}  | dev {np.abs(c - rc).max():.2e} | worst sWeight point {k}: {ps[k, 3]:.6f} vs {rps[k, 3]:.6f} | em dev {max(np.abs(np.array(g.em(i)) - np.array(ref.em(i))).max() for i in range(6)):.1e}")
    g.transformPoints(True); ref.transform_points(True)

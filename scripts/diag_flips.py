#!/usr/bin/env python3
"""How often does a half-link fall on the other side of the inlier threshold on the device than in the
oracle?  Lock-step deformable iterations (identical xyz2 and EM parameters fed to both each iteration),
counting points whose sWeight differs by more than rounding."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frog_amd import _abi                      # noqa: E402
from frog_amd.image_group import ImageGroup    # noqa: E402
from frog_amd.pairs import Pairs               # noqa: E402
from oracle.oracle_api import OracleGroup      # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pairs = Pairs.synthetic(6, 3000, 1500, seed=seed)
g = ImageGroup(pairs)
ref = OracleGroup(pairs.model, _abi.FrogOptions.default())
ref.setup_stats()
g.setupLinearTransforms(); ref.linear_init()
g.transformPoints(); ref.transform_points()
for it in range(50):
    if it % 10 == 0:
        ref.update_stats()
        for i in range(pairs.n_images):
            g.set_em(i, ref.em(i))
    g.updateLinearTransforms(); ref.linear_step()
    g.transformPoints(); ref.transform_points()
g.transformPoints(True); ref.transform_points(True)
flips_total = 0
for level in range(3):
    g.setupDeformableTransforms(level); ref.deformable_setup(level, _abi.FrogGridInfo())
    g.transformPoints(); ref.transform_points()
    for it in range(40):
        g.set_points2(ref.xyz2())
        if it % 10 == 0:
            ref.update_stats()
            for i in range(pairs.n_images):
                g.set_em(i, ref.em(i))
        e = g.updateDeformableTransforms(0.02); er = ref.deformable_step(0.02)
        ps, rps = g.point_sums(), ref.point_sums()
        dw = np.abs(ps[:, 3] - rps[:, 3])
        flips = int(np.sum(dw > 1e-3 * np.maximum(rps[:, 3], 1e-6) + 1e-6))
        flips_total += flips
        if flips:
            k = int(np.argmax(dw))
            print(f"level {level} it {it}: {flips} points differ; worst point {k}: sWeight {ps[k, 3]:.6f} vs {rps[k, 3]:.6f}")
        g.transformPoints(); ref.transform_points()
    g.transformPoints(True); ref.transform_points(True)
print("total flipped points over 120 deformable iterations:", flips_total, "of", pairs.n_half_links, "half-links per iteration")

#!/usr/bin/env python3
"""Parity margins over a matrix of seeds and options: full run() of the HIP path vs the oracle
(small groups), reporting the largest deviations of E, matrices and lattice coefficients."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frog_amd import _abi                      # noqa: E402
from frog_amd.image_group import ImageGroup    # noqa: E402
from frog_amd.pairs import Pairs               # noqa: E402
from oracle.oracle_api import OracleGroup      # noqa: E402


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


worst = {}
cases = []
for seed in (1, 2, 3):
    cases.append((dict(n=6, pts=3000, ppb=1500, seed=seed), {}))
cases += [
    (dict(n=6, pts=3000, ppb=1500, seed=4), dict(use_scale=0)),
    (dict(n=6, pts=3000, ppb=1500, seed=5), dict(inlier_threshold=0.3)),
    (dict(n=6, pts=3000, ppb=1500, seed=6), dict(guarantee_diffeomorphism=0)),
    (dict(n=6, pts=3000, ppb=1500, seed=7), dict(initial_grid_size=60.0)),
    (dict(n=6, pts=3000, ppb=1500, seed=8), dict(max_displacement_ratio=0.2)),
    (dict(n=10, pts=1500, ppb=600, seed=9), dict(stats_max_size=3000)),
    (dict(n=3, pts=6000, ppb=4000, seed=10), dict(linear_alpha=0.3)),
]
for cfg, opt in cases:
    pairs = Pairs.synthetic(cfg["n"], cfg["pts"], cfg["ppb"], seed=cfg["seed"])
    g = ImageGroup(pairs, **opt)
    g.linearIterations, g.deformableLevels, g.deformableIterations = 50, 3, 40
    ref = OracleGroup(pairs.model, _abi.FrogOptions.default(**opt))
    E = g.run()
    rE, rgrids = ref.run(li=50, dl=3, di=40)
    dev = {"E": max(abs(a - b) / abs(b) for a, b in zip(E, rE)) if len(E) == len(rE) else float("inf"),
           "grids": (list(g.gridsPerLevel), list(rgrids))}
    dev["matrix"] = max(rel(np.diag(g.matrix(i))[:3], np.diag(ref.matrix(i))[:3]) for i in range(pairs.n_images))
    dev["translation"] = max(rel(g.matrix(i)[:3, 3], ref.matrix(i)[:3, 3]) for i in range(pairs.n_images))
    co = 0.0
    if g.gridsPerLevel == list(rgrids):
        for k in range(ref.num_grids()):
            for i in range(pairs.n_images):
                co = max(co, rel(g.grid(i, k)[1], ref.grid(i, k, _abi.FrogGridInfo())[1]))
    else:
        co = float("inf")
    dev["coeff"] = co
    dev["xyz2"] = rel(g.points()[1], ref.xyz2())
    print(cfg, opt, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in dev.items()}, flush=True)

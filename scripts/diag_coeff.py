#!/usr/bin/env python3
"""Where does the largest coefficient deviation of a free-running comparison sit?"""
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
g.linearIterations, g.deformableLevels, g.deformableIterations = 50, 3, 40
ref = OracleGroup(pairs.model, _abi.FrogOptions.default())
g.run(); ref.run(li=50, dl=3, di=40)
for k in range(ref.num_grids()):
    info = _abi.FrogGridInfo()
    for i in range(pairs.n_images):
        gi, c = g.grid(i, k)
        _, rc = ref.grid(i, k, info)
        d = np.abs(c - rc).max(1)
        j = int(np.argmax(d))
        mx = np.abs(rc).max()
        dims = list(info.dims)
        x, y, z = j % dims[0], (j // dims[0]) % dims[1], j // (dims[0] * dims[1])
        print(f"grid {k} image {i}: max dev {d[j] / mx:.2e} of max |c| {mx:.3f} at cp ({x},{y},{z}) of {dims}: c {c[j]} vs {rc[j]}; "
              f"rms dev {np.sqrt((d * d).mean()) / mx:.2e}; cps with dev > 1e-4 max: {int((d > 1e-4 * mx).sum())} of {len(d)}")

"""Child process of tests/test_gpu_round6.py: a short free-running schedule on a small group through whichever device library
FROG_HIP_LIB names, results into an .npz (the library is chosen once per process, so the two builds need two processes).
usage: run_lattices.py OUT.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from frog_amd.pairs import Pairs            # noqa: E402
import test_gpu_reference_order as T        # noqa: E402

pairs = Pairs.synthetic(6, 3000, 1500, seed=7)
s = T.Side(pairs)
energies = []


def check(tag, sides, e=None, infos=None):
    if e is not None:
        energies.append(float(e[0]))


grids = T.lockstep([s], 8, 3, 12, check)
out = {"xyz2": s.xyz2(), "E": np.array(energies), "grids": np.array(grids)}
for k in range(s.num_grids()):
    out[f"lattice{k}"] = np.stack([s.grid(i, k)[1] for i in range(pairs.n_images)])
np.savez(sys.argv[1], **out)

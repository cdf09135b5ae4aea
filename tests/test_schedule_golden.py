"""tests/golden/schedule_golden.npz (the CPU oracle's run of BASELINE.json configs[2] over the reference's whole default schedule,
made by tests/golden/make_schedule_golden.py; compared with the device by tests/test_gpu_schedule_golden.py): the file is what
its generator says -- its shapes hang together, and the oracle as it is built HERE, on the same synthetic group, prints the
file's first energies to the bit (the first refresh and three linear iterations: ten seconds; the whole run took twenty minutes on eight CPUs)."""
import os

import numpy as np
import pytest

from frog_amd import _abi
from frog_amd.pairs import Pairs
from oracle.oracle_api import OracleGroup

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "schedule_golden.npz")
GOLDEN_CFG5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "schedule_golden_cfg5.npz")


@pytest.mark.parametrize("path, n_img, schedule", [(GOLDEN, 100, (50, 3, 200)),         # imageGroup.h:52-82, the reference's defaults
                                                   (GOLDEN_CFG5, 500, (20, 5, 40)),
                                                   (os.path.join(os.path.dirname(GOLDEN), "schedule_golden_cfg2.npz"), 20, (50, 0, 0))])
def test_fixture_is_self_consistent(path, n_img, schedule):
    g = np.load(path)
    li, dl, di = (int(v) for v in g["schedule"])
    assert (li, dl, di) == schedule
    assert len(g["E"]) == li + dl * di and np.all(np.isfinite(g["E"])) and np.all(g["E"] > 0)
    assert np.array_equal(g["E"], g["E"].astype(np.float32).astype(np.float64))      # measures are printed from a float
    n_lat = int(g["grids"].sum())
    assert len(g["grids"]) == dl and g["dims"].shape == (n_lat, 3) and g["origin"].shape == (n_lat, 3) and g["spacing"].shape == (n_lat, 3)
    assert g["sha_grid"].shape == (n_lat, n_img, 32) and g["matrices"].shape == (n_img, 4, 4) and g["em"].shape == (n_img, 3)
    for k in range(n_lat):
        nodes, stride = int(np.prod(g["dims"][k])), int(g["node_stride"][k])
        c = g[f"coeff_{k}"]
        assert c.shape == (len(g["images"]), (nodes + stride - 1) // stride, 3) and c.dtype == np.float32
        assert float(np.abs(c).max()) <= float(g["max_coeff"][k])
    # lattices of one level share their spacing's level: g / 2^level, rounded to whole cells of the group's box (imageGroup.cxx:159-218)
    level = np.repeat(np.arange(dl), g["grids"])
    assert np.all(g["spacing"].max(axis=1) <= 100.0 / 2.0 ** level * 1.5) and np.all(g["spacing"].min(axis=1) >= 100.0 / 2.0 ** level / 1.5)
    # the linear matrices are axis-aligned scale + translation (imageGroup.cxx:1063-1149)
    m = g["matrices"]
    off = m[:, :3, :3] * (1 - np.eye(3))
    assert np.all(off == 0) and np.all(m[:, 3] == [0, 0, 0, 1])
    assert g["inliers"].shape == (n_img,) and int(g["inliers"].sum() + g["outliers"].sum()) == int(g["n_half_links"])


def test_the_oracle_built_here_prints_the_fixtures_first_energies():
    g = np.load(GOLDEN)
    pairs = Pairs.synthetic(100, 20000, 10101, seed=1)
    assert pairs.n_half_links == int(g["n_half_links"])
    ref = OracleGroup(pairs.model, _abi.FrogOptions.default())
    ref.setup_stats()
    ref.linear_init()
    ref.transform_points()
    ref.update_stats()
    for it in range(3):
        e = float(np.float32(ref.linear_step()))
        ref.transform_points()
        assert e == float(g["E"][it]), (it, e, float(g["E"][it]))

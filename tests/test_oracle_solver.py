"""Self-checks of the oracle's solver loops (parity unpinned by the reference:
imageGroup.cxx cannot be built here and the reference ships no vectors), through
analytic properties of the algorithm."""
import numpy as np
import pytest

from frog_amd import _abi
from frog_amd.pairs import Pairs
from oracle.oracle_api import OracleGroup


def exact_copies(n_images=4, n_points=400, seed=0, scales=None, shifts=None):
    """Images that are exact scaled/translated copies of one cloud, every point linked."""
    rng = np.random.default_rng(seed)
    cloud = rng.uniform(0, 200, size=(n_points, 3))
    xyz, po = [], [0]
    for i in range(n_images):
        s = scales[i] if scales is not None else np.ones(3)
        t = shifts[i] if shifts is not None else np.zeros(3)
        xyz.append((cloud * s + t).astype(np.float32))
        po.append(po[-1] + n_points)
    idx = np.arange(n_points, dtype=np.uint32)
    blocks = [(i, j, idx, idx) for i in range(n_images) for j in range(i + 1, n_images)]
    return Pairs.from_arrays(po, np.concatenate(xyz), blocks)


def group(pairs, **opt):
    g = OracleGroup(pairs.model, _abi.FrogOptions.default(**opt))
    g.setup_stats()
    return g


def test_identical_images_are_a_fixed_point():
    p = exact_copies()
    g = group(p)
    g.linear_init(); g.transform_points()
    before = g.xyz2().copy()
    g.update_stats()
    e = g.linear_step()
    g.transform_points()
    assert e == 0.0
    assert np.allclose(g.xyz2(), before, atol=1e-4)
    for i in range(p.n_images):
        assert np.allclose(np.diag(g.matrix(i))[:3], 1.0, atol=1e-6)


def test_linear_stage_recovers_translation_and_scale():
    rng = np.random.default_rng(5)
    scales = rng.uniform(0.85, 1.2, size=(4, 3))
    shifts = rng.uniform(-40, 40, size=(4, 3))
    p = exact_copies(scales=scales, shifts=shifts)
    g = group(p)
    E, _ = g.run(li=60, dl=0, di=0)
    assert E[-1] < 1e-2 * E[0]
    xyz2 = g.xyz2().reshape(4, -1, 3)
    # all images land on the same common-space cloud
    assert np.max(np.abs(xyz2 - xyz2.mean(axis=0))) < 0.05
    # the recovered per-axis scales have the planted ratios
    s = np.array([np.diag(g.matrix(i))[:3] for i in range(4)])
    assert np.allclose(s * scales, (s * scales)[0], rtol=1e-3)


def test_deformable_updates_have_zero_mean_across_images(tiny_pairs):
    g = group(tiny_pairs)
    E, grids = g.run(li=15, dl=2, di=12)
    assert len(E) == 15 + 24 and all(np.isfinite(E))
    for k in range(g.num_grids()):
        tot = sum(g.grid(i, k, _abi.FrogGridInfo())[1].astype(np.float64) for i in range(tiny_pairs.n_images))
        scale = max(np.max(np.abs(g.grid(i, k, _abi.FrogGridInfo())[1])) for i in range(tiny_pairs.n_images))
        # imageGroup.cxx:417-423: the cross-image mean is subtracted every step
        assert np.max(np.abs(tot)) <= 1e-5 * max(scale, 1e-3) * tiny_pairs.n_images


def test_lattice_geometry(tiny_pairs):
    g = group(tiny_pairs)
    g.linear_init(); g.transform_points(); g.transform_points(True)
    xyz = g.xyz()
    for level in (0, 1, 2):
        info = g.deformable_setup(level, _abi.FrogGridInfo())
        size = 100.0 / 2 ** level
        for k in range(3):
            lo, hi = float(xyz[:, k].min()), float(xyz[:, k].max())
            s = float(np.float32(1) + np.float32(2) * np.float32(0.1))
            cen = 0.5 * (lo + hi)
            length = (cen + s * (hi - cen)) - (cen + s * (lo - cen))
            n = max(1, int(round(length / size)))
            assert info.dims[k] == n + 3
            assert info.spacing[k] == pytest.approx(length / n, rel=1e-12)
            assert info.origin[k] == pytest.approx(cen + s * (lo - cen) - length / n, rel=1e-9, abs=1e-9)
            # every point's 4^3 stencil stays inside the lattice
            c = (xyz[:, k] - info.origin[k]) / info.spacing[k]
            assert c.min() >= 1 and np.floor(c).max() <= info.dims[k] - 3


def test_zero_lattice_is_identity_and_guard_rejects(tiny_pairs):
    g = group(tiny_pairs, max_displacement_ratio=1e-5)
    g.linear_init(); g.transform_points()
    for it in range(5):
        if it == 0:
            g.update_stats()
        g.linear_step(); g.transform_points()
    g.transform_points(True)
    g.deformable_setup(1, _abi.FrogGridInfo())
    g.transform_points()
    assert np.array_equal(g.xyz2(), g.xyz())          # all-zero coefficients displace nothing
    assert g.deformable_step(0.02) == -1.0             # imageGroup.cxx:434-439
    assert not g.grid(0, 0, _abi.FrogGridInfo())[1].any()


def test_thread_count_does_not_change_the_transforms(tiny_pairs):
    from oracle.oracle_api import lib
    res = []
    n0 = lib().frogo_get_max_threads()
    for nt in (1, 3):
        lib().frogo_set_threads(nt)
        g = group(tiny_pairs)
        g.run(li=12, dl=1, di=8)
        res.append((g.xyz(), [g.matrix(i) for i in range(tiny_pairs.n_images)]))
    lib().frogo_set_threads(n0)
    assert np.array_equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)

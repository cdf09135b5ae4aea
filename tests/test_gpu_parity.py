"""HIP path vs CPU oracle, through the C ABI (include/frog_hip.h).

Bars (BASELINE.json north_star): sampled-link ordinals, sample values and
histogram bins bit-exact on identical inputs; transform parameters (matrix
entries, B-spline coefficients) within 1e-4 relative.
"""
import ctypes as C

import numpy as np
import pytest

from frog_amd import _abi
from frog_amd.image_group import ImageGroup
from oracle.oracle_api import OracleGroup
from lattice_util import compare_lattice, node_weights

pytestmark = pytest.mark.gpu

REL = 1e-4          # north-star tolerance for transform parameters


def make(pairs, setup_stats=True, **opt):
    o = _abi.FrogOptions.default(**opt)
    g = ImageGroup(pairs, **opt)
    ref = OracleGroup(pairs.model, o)
    if setup_stats:                     # OracleGroup.run() does it itself, like ImageGroup::run
        ref.setup_stats()
    return g, ref


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)


def same_inputs(g, ref):
    """Identical xyz2 and (c1, c2, ratio) on both sides: integer outputs are then compared with ==."""
    g.set_points2(ref.xyz2())
    for i in range(ref.n_images):
        g.set_em(i, ref.em(i))


def lattices_agree(g, ref, pairs, k=0):
    """Lattice k of every image: coefficients and displacement field by the criterion of tests/lattice_util.py (a raw
    max-norm over ALL control points also measures the conditioning of the rim of the box: 1e-4 per ulp of coordinate)."""
    x, po = ref.xyz(), np.asarray(pairs.point_offset)
    w = node_weights(ref, k, po, x)
    for i in range(ref.n_images):
        dev_c, dev_d, _, _ = compare_lattice(g, ref, k, i, x[po[i]:po[i + 1]], w)
        assert dev_c <= REL and dev_d <= REL, f"lattice {k} image {i}: coefficients {dev_c:.2e}, field {dev_d:.2e}"


def start(g, ref):
    g.setupLinearTransforms(); ref.linear_init()
    g.transformPoints(); ref.transform_points()


def test_linear_init_and_transform_bit_exact(small_pairs):
    g, ref = make(small_pairs)
    start(g, ref)
    for i in range(small_pairs.n_images):
        assert np.array_equal(g.matrix(i), ref.matrix(i))
    xyz, xyz2 = g.points()
    assert np.array_equal(xyz, ref.xyz())
    assert np.array_equal(xyz2, ref.xyz2())


@pytest.mark.parametrize("max_size", [10000, 2000])
def test_update_stats_samples_bit_exact(small_pairs, max_size):
    # 3000 points x ~5 partners x 1500 pairs -> 15000 half-links per image:
    # max_size 10000 and 2000 both exercise the mt19937 reservoir
    g, ref = make(small_pairs, stats_max_size=max_size)
    start(g, ref)
    for refresh in range(3):            # generator state carries over between refreshes
        g.updateStats(); ref.update_stats()
        for i in range(small_pairs.n_images):
            s, o = g.samples(i)
            rs, ro = ref.samples(i)
            assert np.array_equal(o, ro), f"ordinals differ, image {i}, refresh {refresh}"
            assert np.array_equal(s, rs), f"sample distances differ, image {i}, refresh {refresh}"
            assert np.array_equal(g.histogram(i), ref.histogram(i))
            assert np.array_equal(g.em(i), ref.em(i))       # same samples, sums in sample order: same bits


def test_update_stats_without_reservoir(tiny_pairs):
    # 600 points x 3 partners x 300 pairs = 1800 half-links < maxSize: every link is a sample
    g, ref = make(tiny_pairs)
    start(g, ref)
    g.updateStats(); ref.update_stats()
    for i in range(tiny_pairs.n_images):
        s, o = g.samples(i)
        rs, ro = ref.samples(i)
        assert len(s) == len(rs) and np.array_equal(o, ro) and np.array_equal(s, rs)


def test_linear_step_matches(small_pairs):
    g, ref = make(small_pairs)
    start(g, ref)
    ref.update_stats()
    for i in range(small_pairs.n_images):
        g.set_em(i, ref.em(i))          # identical EM parameters on both sides
    e = g.updateLinearTransforms(); er = ref.linear_step()
    assert abs(e - er) / er < 1e-6
    for i in range(small_pairs.n_images):
        m, mr = g.matrix(i), ref.matrix(i)
        assert relerr(np.diag(m)[:3], np.diag(mr)[:3]) < 1e-6
        assert relerr(m[:3, 3], mr[:3, 3]) < 1e-6


def test_linear_stage_50_iterations(small_pairs):
    g, ref = make(small_pairs)
    start(g, ref)
    for it in range(50):
        if it % 10 == 0:
            g.updateStats(); ref.update_stats()
        e = g.updateLinearTransforms(); er = ref.linear_step()
        g.transformPoints(); ref.transform_points()
        assert abs(e - er) / er < REL
    for i in range(small_pairs.n_images):
        m, mr = g.matrix(i), ref.matrix(i)
        assert relerr(np.diag(m)[:3], np.diag(mr)[:3]) < REL
        assert relerr(m[:3, 3], mr[:3, 3]) < REL
    assert relerr(g.points()[1], ref.xyz2()) < REL


def _to_deformable(g, ref, iters=20):
    start(g, ref)
    for it in range(iters):
        if it % 10 == 0:
            ref.update_stats()
            for i in range(ref.n_images):
                g.set_em(i, ref.em(i))
        g.updateLinearTransforms(); ref.linear_step()
        g.transformPoints(); ref.transform_points()
    g.transformPoints(True); ref.transform_points(True)
    # continue from identical coordinates so that the deformable kernels are compared in isolation
    g_xyz, _ = g.points()
    return g_xyz


@pytest.mark.parametrize("level", [0, 2])
def test_deformable_step_pieces(small_pairs, level):
    g, ref = make(small_pairs)
    _to_deformable(g, ref)
    info = g.setupDeformableTransforms(level)
    rinfo = ref.deformable_setup(level, _abi.FrogGridInfo())
    assert list(info.dims) == list(rinfo.dims)
    np.testing.assert_allclose(list(info.origin), list(rinfo.origin), rtol=1e-6)
    np.testing.assert_allclose(list(info.spacing), list(rinfo.spacing), rtol=1e-6)
    g.transformPoints(); ref.transform_points()
    assert relerr(g.points()[1], ref.xyz2()) < 1e-6
    ref.update_stats()
    for i in range(ref.n_images):
        g.set_em(i, ref.em(i))
    n_cp = info.dims[0] * info.dims[1] * info.dims[2]
    e = g.updateDeformableTransforms(0.02); er = ref.deformable_step(0.02)
    assert er > 0 and abs(e - er) / er < 1e-5
    ps, rps = g.point_sums(), ref.point_sums()
    assert relerr(ps, rps) < 1e-5
    lattices_agree(g, ref, small_pairs)
    g.transformPoints(); ref.transform_points()
    assert relerr(g.points()[1], ref.xyz2()) < 1e-6


def test_diffeomorphism_guard_rejects_without_state_change(small_pairs):
    g, ref = make(small_pairs, max_displacement_ratio=1e-4)
    _to_deformable(g, ref)
    g.setupDeformableTransforms(1); ref.deformable_setup(1, _abi.FrogGridInfo())
    g.transformPoints(); ref.transform_points()
    before = g.points()[1].copy()
    e = g.updateDeformableTransforms(0.02); er = ref.deformable_step(0.02)
    assert e == -1.0 and er == -1.0
    _, c = g.grid(0, 0)
    assert not c.any()
    g.transformPoints()
    assert np.array_equal(g.points()[1], before)


def test_count_inliers(small_pairs):
    g, ref = make(small_pairs)
    _to_deformable(g, ref)
    same_inputs(g, ref)
    c = g.countInliers()
    rc = ref.count_inliers((_abi.FrogCounts * ref.n_images)())
    for i in range(ref.n_images):
        assert c[i].points == rc[i].points and c[i].pairs == rc[i].pairs
        # an integer census on identical inputs: weights near the threshold are decided by the reference's own
        # arithmetic on the device (k_links.hip.h THRESHOLD_BAND), so the counts are equal, not close
        assert c[i].inliers == rc[i].inliers and c[i].outliers == rc[i].outliers


def test_full_run_parity(small_pairs):
    """Whole schedule of run() (imageGroup.cxx:31-157) at reduced iteration counts."""
    g, ref = make(small_pairs, setup_stats=False)
    g.linearIterations, g.deformableLevels, g.deformableIterations = 30, 3, 40
    E = np.array(g.run())
    Er, grids_r = ref.run(li=30, dl=3, di=40)
    assert g.gridsPerLevel == grids_r
    assert len(E) == len(Er)
    assert np.max(np.abs(E - Er) / Er) < REL
    for i in range(ref.n_images):
        m, mr = g.matrix(i), ref.matrix(i)
        assert relerr(np.diag(m)[:3], np.diag(mr)[:3]) < REL
        assert relerr(m[:3, 3], mr[:3, 3]) < REL
    assert g.num_grids() == ref.num_grids()
    for k in range(ref.num_grids()):
        for i in range(ref.n_images):
            info, c = g.grid(i, k)
            rinfo, rc = ref.grid(i, k, _abi.FrogGridInfo())
            assert list(info.dims) == list(rinfo.dims)
            assert relerr(c, rc) < REL, f"lattice {k} image {i}"
    assert relerr(g.points()[0], ref.xyz()) < REL


def ragged_pairs(seed=5):
    """Images of 25 .. 1500 points observing subsets of one landmark cloud: true matches between
    co-observed landmarks, 25 % false matches, points without any link, one image pair with heavy
    duplication (400 links of ONE point into the same partner image) and one image pair whose
    block appears twice in the file."""
    rng = np.random.default_rng(seed)
    sizes = [1500, 25, 700, 40, 1100, 260]
    po = np.concatenate([[0], np.cumsum(sizes)])
    cloud = rng.uniform(0, 300, size=(1500, 3))
    seen = [rng.permutation(1500)[:n] for n in sizes]            # landmark of every point
    xyz = np.concatenate([(cloud[seen[i]] * rng.uniform(0.9, 1.1, 3) + rng.uniform(-30, 30, 3)
                           + rng.normal(0, 1.5, (sizes[i], 3))).astype(np.float32) for i in range(len(sizes))])
    where = []
    for i in range(len(sizes)):
        w = -np.ones(1500, np.int64); w[seen[i]] = np.arange(sizes[i]); where.append(w)
    blocks = []
    for i in range(len(sizes)):
        for j in range(i + 1, len(sizes)):
            both = np.nonzero((where[i] >= 0) & (where[j] >= 0))[0]
            both = both[rng.random(len(both)) < 0.8]             # some co-observed landmarks stay unlinked
            p1, p2 = where[i][both], where[j][both]
            nf = max(2, len(both) // 3)
            p1 = np.concatenate([p1, rng.integers(0, sizes[i], nf)])
            p2 = np.concatenate([p2, rng.integers(0, sizes[j], nf)])
            if (i, j) == (0, 2):
                p1 = np.concatenate([p1, np.full(400, 7)])       # 400 links of point 7 of image 0 into image 2
                p2 = np.concatenate([p2, rng.integers(0, 20, 400)])
            order = np.argsort(p1, kind="stable")
            blocks.append((i, j, p1[order].astype(np.uint32), p2[order].astype(np.uint32)))
    blocks.append(blocks[1])                                     # the same image pair appears twice in the file
    from frog_amd.pairs import Pairs
    return Pairs.from_arrays(po, xyz, blocks)


def test_ragged_group_with_duplicate_links():
    pairs = ragged_pairs()
    g, ref = make(pairs, stats_max_size=500)
    start(g, ref)
    assert np.array_equal(g.points()[1], ref.xyz2())
    for it in range(12):
        if it % 5 == 0:
            g.updateStats(); ref.update_stats()
            for i in range(pairs.n_images):
                s, o = g.samples(i)
                rs, ro = ref.samples(i)
                if it == 0:
                    assert np.array_equal(o, ro) and np.array_equal(s, rs)
                else:
                    assert np.array_equal(o, ro) and np.allclose(s, rs, rtol=1e-3, atol=1e-2)
        e = g.updateLinearTransforms(); er = ref.linear_step()
        g.transformPoints(); ref.transform_points()
        assert abs(e - er) / er < REL
    g.transformPoints(True); ref.transform_points(True)
    g.setupDeformableTransforms(1); ref.deformable_setup(1, _abi.FrogGridInfo())
    g.transformPoints(); ref.transform_points()
    g.updateStats(); ref.update_stats()
    for i in range(pairs.n_images):
        g.set_em(i, ref.em(i))
    e = g.updateDeformableTransforms(0.02); er = ref.deformable_step(0.02)
    assert (e < 0) == (er < 0)
    # per-point sums: point 7 of image 0 receives 400 duplicate links in one partner group
    ps, rps = g.point_sums(), ref.point_sums()
    assert relerr(ps, rps) < 1e-5
    if e >= 0:
        assert abs(e - er) / er < 1e-5
        lattices_agree(g, ref, pairs)
    same_inputs(g, ref)
    c = g.countInliers()
    rc = ref.count_inliers((_abi.FrogCounts * pairs.n_images)())
    for i in range(pairs.n_images):
        assert c[i].pairs == rc[i].pairs and c[i].inliers == rc[i].inliers


def test_abi_call_order_and_option_validation(tiny_pairs):
    import ctypes as C
    lib = _abi.hip_lib()
    g = ImageGroup(tiny_pairs)
    e = C.c_double()
    # deformable entry points before a lattice exists
    assert lib.frog_deformable_step(g._ctx, 0.02, C.byref(e)) == _abi.FROG_E_STATE
    assert lib.frog_deformable_phase_b(g._ctx) == _abi.FROG_E_STATE
    g.setupLinearTransforms(); g.transformPoints(); g.updateStats()
    g.updateLinearTransforms(); g.transformPoints(True)
    g.setupDeformableTransforms(0); g.transformPoints()
    # phase C without B, linear step after the lattice exists
    assert lib.frog_deformable_phase_a(g._ctx, 0.02) == _abi.FROG_OK
    assert lib.frog_deformable_phase_c(g._ctx, C.byref(e)) == _abi.FROG_E_STATE
    assert lib.frog_linear_step(g._ctx, C.byref(e)) == _abi.FROG_E_STATE
    assert lib.frog_deformable_phase_b(g._ctx) == _abi.FROG_OK
    assert lib.frog_deformable_phase_c(g._ctx, C.byref(e)) == _abi.FROG_OK and (e.value > 0 or e.value == -1.0)
    # bad arguments
    buf = (C.c_float * 3)()
    assert lib.frog_get_em(g._ctx, 99, buf) == _abi.FROG_E_INVALID
    assert lib.frog_get_grid(g._ctx, 0, 7, None, None, 0) == _abi.FROG_E_INVALID
    ctx = C.c_void_p()
    o = _abi.FrogOptions.default(stats_max_size=0)
    assert lib.frog_create(C.byref(tiny_pairs.model), C.byref(o), 0, 0, 4, C.byref(ctx)) == _abi.FROG_E_INVALID
    assert lib.frog_create(C.byref(tiny_pairs.model), C.byref(_abi.FrogOptions.default()), 99, 0, 4, C.byref(ctx)) == _abi.FROG_E_INVALID


def test_external_stream_and_sub_range_context(tiny_pairs):
    """frog_set_stream with a stream owned by the caller; a context that owns only images [1,3)
    refuses the whole-group entry points (the collectives are the caller's job)."""
    import ctypes as C
    import torch
    lib = _abi.hip_lib()
    g = ImageGroup(tiny_pairs)
    s = torch.cuda.Stream()
    assert lib.frog_set_stream(g._ctx, C.c_void_p(s.cuda_stream)) == _abi.FROG_OK
    g.setupLinearTransforms(); g.transformPoints(); g.updateStats()
    e1 = g.updateLinearTransforms()
    ref = ImageGroup(tiny_pairs)
    ref.setupLinearTransforms(); ref.transformPoints(); ref.updateStats()
    assert ref.updateLinearTransforms() == e1
    part = ImageGroup(tiny_pairs, image_range=(1, 3))
    e = C.c_double()
    assert lib.frog_update_stats(part._ctx) == _abi.FROG_E_STATE
    assert lib.frog_deformable_setup(part._ctx, 0, None) == _abi.FROG_E_STATE
    assert lib.frog_update_stats_local(part._ctx) == _abi.FROG_OK
    em = np.array([part.em(i) for i in range(4)])
    assert not em[0].any() and not em[3].any() and em[1].all() and em[2].all()   # other ranks' rows are zero


@pytest.mark.parametrize("which", ["small", "ragged"])
def test_wide_and_narrow_link_records_agree_bit_for_bit(small_pairs, which, monkeypatch):
    # The sweep reads 4-byte link records when the field widths allow it and 8-byte ones
    # otherwise (ctx.h RecFormat); FROG_WIDE_RECORDS=1 forces the 8-byte form at frog_create.
    # Same links in the same order, so every observable must be identical.
    pairs = small_pairs if which == "small" else ragged_pairs()

    def run(wide):
        if wide:
            monkeypatch.setenv("FROG_WIDE_RECORDS", "1")
        else:
            monkeypatch.delenv("FROG_WIDE_RECORDS", raising=False)
        g = ImageGroup(pairs, stats_max_size=2000)
        g.setupLinearTransforms(); g.transformPoints()
        out = []
        for it in range(6):
            if it % 5 == 0:
                g.updateStats()
            out.append(g.updateLinearTransforms())
            g.transformPoints()
        g.transformPoints(True)
        g.setupDeformableTransforms(1); g.transformPoints()
        g.updateStats()
        out.append(g.updateDeformableTransforms(0.02))
        sums = g.point_sums().copy()
        counts = [(c.inliers, c.outliers) for c in g.countInliers()]
        return out, sums, counts

    e_n, s_n, c_n = run(False)
    e_w, s_w, c_w = run(True)
    assert e_n == e_w
    assert np.array_equal(s_n, s_w)
    assert c_n == c_w


def test_error_map_matches_oracle(small_pairs):
    # saveErrorMaps (imageGroup.cxx:475-567) on identical coordinates and EM parameters: the
    # per-point sums come from the device sweep, the nearest-node binning keeps the reference's order
    g, ref = make(small_pairs)
    _to_deformable(g, ref)
    info = g.setupDeformableTransforms(1)
    ref.deformable_setup(1, _abi.FrogGridInfo())
    g.transformPoints(); ref.transform_points()
    ref.update_stats()
    for i in range(ref.n_images):
        g.set_em(i, ref.em(i))
    for _ in range(3):
        assert g.updateDeformableTransforms(0.02) >= 0 and ref.deformable_step(0.02) >= 0
        g.transformPoints(); ref.transform_points()
    g.transformPoints(True); ref.transform_points(True)            # run() :126, then :141
    assert relerr(g.points()[1], ref.xyz2()) < 1e-6
    g.set_points2(ref.xyz2())                                       # identical inputs for the comparison
    n_cp = info.dims[0] * info.dims[1] * info.dims[2]
    with pytest.raises(RuntimeError):
        g.errorMap(0)                                               # needs residualSums() first
    g.residualSums()
    for i in range(ref.n_images):
        ginfo, m = g.errorMap(i)
        r = ref.error_map(i, n_cp)
        assert list(ginfo.dims) == list(info.dims)
        # same nodes hit, same weights, mean residuals to f32 rounding of the per-point sums
        assert np.array_equal(m[:, 3] > 0, r[:, 3] > 0)
        assert relerr(m[:, 3], r[:, 3]) < 1e-5
        assert np.max(np.abs(m[:, :3] - r[:, :3])) < 1e-4 * max(1.0, np.max(np.abs(r[:, :3])))
    g.transformPoints()
    with pytest.raises(RuntimeError):
        g.errorMap(0)                                               # any later step invalidates the sums


@pytest.mark.parametrize("n_sub", [2, 8])
def test_sweep_sub_passes_match_the_oracle(small_pairs, n_sub, monkeypatch):
    # Large models split the partner images into 8 * n_sub groups and launch the sweep n_sub times,
    # each launch continuing the per-XCD partial sums (ctx.h); FROG_SUBPASSES forces it on a small one.
    monkeypatch.setenv("FROG_SUBPASSES", str(n_sub))
    g, ref = make(small_pairs)
    _to_deformable(g, ref, iters=12)
    info = g.setupDeformableTransforms(1)
    ref.deformable_setup(1, _abi.FrogGridInfo())
    g.transformPoints(); ref.transform_points()
    ref.update_stats()
    for i in range(ref.n_images):
        g.set_em(i, ref.em(i))
    e = g.updateDeformableTransforms(0.02); er = ref.deformable_step(0.02)
    assert er > 0 and abs(e - er) / er < 1e-5
    assert relerr(g.point_sums(), ref.point_sums()) < 1e-5
    lattices_agree(g, ref, small_pairs)
    same_inputs(g, ref)
    c = g.countInliers()
    rc = ref.count_inliers((_abi.FrogCounts * small_pairs.n_images)())
    for i in range(small_pairs.n_images):
        assert c[i].pairs == rc[i].pairs and c[i].inliers == rc[i].inliers


def test_hard_links_of_landmark_constraints(small_pairs):
    # -lc: every landmark is hard-linked to the other landmarks of its name; in the deformable step and the
    # error maps a hard link adds weight2 * (pB - pA) and weight2 after the regular links, and
    # weight2 * dist2 / weight2 to the energy sums (imageGroup.cxx:280-295, :520-533)
    from frog_amd.pairs import Pairs
    pairs = Pairs.synthetic(6, 3000, 1500, seed=7)
    po0 = np.asarray(pairs.point_offset).copy()
    rng = np.random.default_rng(8)
    for i in range(6):                                   # 3 landmarks per image, as link-less points
        pts = np.asarray(pairs.xyz)[po0[i]:po0[i + 1]]
        pairs.append_points(i, pts[rng.choice(3000, 3, replace=False)])
    po = np.asarray(pairs.point_offset)
    point, partner = [], []
    for k in range(3):                                   # name k: the k-th landmark of every image
        ids = [po[i] + 3000 + k for i in range(6)]
        for a in ids:
            for b in ids:
                if a != b:
                    point.append(a); partner.append(b)
    w2 = np.float32(6 * 50.0) ** 2
    g, ref = make(pairs)
    _to_deformable(g, ref, iters=12)
    info = g.setupDeformableTransforms(1)
    ref.deformable_setup(1, _abi.FrogGridInfo())
    g.transformPoints(); ref.transform_points()
    ref.update_stats()
    for i in range(ref.n_images):
        g.set_em(i, ref.em(i))
    e0 = g.updateDeformableTransforms(0.0)               # alpha 0: state unchanged, E without constraints
    g.transformPoints()
    g.set_hard_links(point, partner, w2); ref.set_hard_links(point, partner, w2)
    e = g.updateDeformableTransforms(0.02); er = ref.deformable_step(0.02)
    assert er > 0 and abs(e - er) / er < 1e-5 and e > 2 * e0      # the constraint terms dominate the energy sums
    ps, rps = g.point_sums(), ref.point_sums()
    assert relerr(ps, rps) < 1e-5
    assert all(rps[a, 3] >= 5 * w2 * 0.999 for a in set(point))  # 5 hard links each
    lattices_agree(g, ref, pairs)
    g.transformPoints(); ref.transform_points()
    g.set_points2(ref.xyz2())
    g.residualSums()
    n_cp = info.dims[0] * info.dims[1] * info.dims[2]
    for i in range(ref.n_images):
        m, r = g.errorMap(i)[1], ref.error_map(i, n_cp)
        assert relerr(m[:, 3], r[:, 3]) < 1e-5 and np.max(np.abs(m[:, :3] - r[:, :3])) < 1e-4 * max(1.0, np.max(np.abs(r[:, :3])))
    g.set_hard_links([], [], 0.0)                        # n = 0 removes them
    e2 = g.updateDeformableTransforms(0.0)
    assert e2 < 0.5 * e


@pytest.mark.parametrize("n_fixed", [1, 4])
def test_fixed_images_match_oracle(small_pairs, n_fixed):
    """-fi n (imageGroup.cxx:34,195,240,385,398,442,823,913,1003,1068): the first n images keep their
    position, every loop but updateStats starts at image n and the group mean is not removed."""
    n = small_pairs.n_images
    g = ImageGroup(small_pairs, n_fixed_images=n_fixed)
    ref = OracleGroup(small_pairs.model, _abi.FrogOptions.default(n_fixed_images=n_fixed))
    g.linearIterations, g.deformableLevels, g.deformableIterations = 20, 2, 20
    E = np.array(g.run())
    Er, grids_r = ref.run(li=20, dl=2, di=20)
    assert g.gridsPerLevel == grids_r and len(E) == len(Er)
    assert np.max(np.abs(E - Er) / Er) < REL
    xyz, xyz2 = g.points()
    po = small_pairs.point_offset
    assert np.array_equal(xyz[:po[n_fixed]], small_pairs.xyz[:po[n_fixed]])          # fixed images never move
    assert np.array_equal(xyz2[:po[n_fixed]], small_pairs.xyz[:po[n_fixed]])
    assert relerr(xyz, ref.xyz()) < REL and relerr(xyz2, ref.xyz2()) < REL
    for i in range(n):
        assert np.allclose(g.em(i), ref.em(i), rtol=1e-5), f"mixture of image {i}"
        s, o = g.samples(i)
        rs, ro = ref.samples(i)
        assert np.array_equal(o, ro), f"sample ordinals of image {i}"
    for i in range(n_fixed, n):
        m, mr = g.matrix(i), ref.matrix(i)
        assert relerr(np.diag(m)[:3], np.diag(mr)[:3]) < REL and relerr(m[:3, 3], mr[:3, 3]) < REL
        for k in range(ref.num_grids()):
            assert relerr(g.grid(i, k)[1], ref.grid(i, k, _abi.FrogGridInfo())[1]) < REL, f"lattice {k} image {i}"
    # the lattices no longer sum to zero over the images (no mean removal)
    tot = sum(g.grid(i, 0)[1].astype(np.float64) for i in range(n_fixed, n))
    assert np.max(np.abs(tot)) > 1e-3
    same_inputs(g, ref)
    c = g.countInliers()
    rc = ref.count_inliers((_abi.FrogCounts * n)())
    for i in range(n_fixed, n):
        assert c[i].pairs == rc[i].pairs and c[i].inliers == rc[i].inliers
    # a context cannot be given a sub-range together with fixed images
    ctx = C.c_void_p()
    o = _abi.FrogOptions.default(n_fixed_images=1)
    assert _abi.hip_lib().frog_create(C.byref(small_pairs.model), C.byref(o), 0, 1, n, C.byref(ctx)) == _abi.FROG_E_INVALID
    o = _abi.FrogOptions.default(n_fixed_images=n)
    assert _abi.hip_lib().frog_create(C.byref(small_pairs.model), C.byref(o), 0, 0, n, C.byref(ctx)) == _abi.FROG_E_INVALID


def test_ransac_stage_matches_oracle(small_pairs):
    """run() with fixed images and -r 1 (imageGroup.cxx:40-49): RANSAC per moving image instead of the linear
    iterations, then the deformable levels.  Candidates and the refit are f64 host arithmetic on both sides
    (two independent eigen-solvers: agreement to rounding), the census is integer."""
    n, nf = small_pairs.n_images, 2
    g = ImageGroup(small_pairs, n_fixed_images=nf)
    ref = OracleGroup(small_pairs.model, _abi.FrogOptions.default(n_fixed_images=nf))
    ref.setup_stats()
    start(g, ref)
    for i in range(nf, n):
        a = g.RANSAC(i, iterations=600, batches=4, inlier_distance=50.0, max_scale=10.0)
        b = ref.ransac(i, iterations=600, batches=4, inlier_distance=50.0, max_scale=10.0)
        assert a == b and a > 1000, f"best census of image {i}"
        assert np.allclose(g.matrix(i), ref.matrix(i), rtol=1e-9, atol=1e-9), f"refitted matrix of image {i}"
        m = g.matrix(i)[:3, :3]
        s = np.cbrt(np.linalg.det(m))
        assert np.allclose(m @ m.T, s * s * np.eye(3), atol=1e-9)        # a similarity: rotation times one scale
    g.transformPoints(); ref.transform_points()
    g.updateStats(); ref.update_stats()
    assert relerr(g.points()[1], ref.xyz2()) < 1e-6
    g.transformPoints(True); ref.transform_points(True)
    info = g.setupDeformableTransforms(0); ref.deformable_setup(0, _abi.FrogGridInfo())
    g.transformPoints(); ref.transform_points()
    for it in range(10):
        if it % 10 == 0:
            g.updateStats(); ref.update_stats()
        e, er = g.updateDeformableTransforms(0.02), ref.deformable_step(0.02)
        assert abs(e - er) / er < REL
        g.transformPoints(); ref.transform_points()
    # no candidate at all (iterations / batches rounds down to 0, :636): the census is 0 and the matrix is the refit of
    # what the CURRENT matrix brings within the distance -- on both sides
    g3 = ImageGroup(small_pairs, n_fixed_images=nf)
    ref3 = OracleGroup(small_pairs.model, _abi.FrogOptions.default(n_fixed_images=nf))
    ref3.setup_stats()
    start(g3, ref3)
    assert g3.RANSAC(nf, iterations=3, batches=4, inlier_distance=80.0) == ref3.ransac(nf, iterations=3, batches=4, inlier_distance=80.0) == 0
    assert np.allclose(g3.matrix(nf), ref3.matrix(nf), rtol=1e-9, atol=1e-9)
    # degenerate requests are refused before any launch
    with pytest.raises(Exception):
        g2 = ImageGroup(small_pairs, n_fixed_images=nf)
        g2.setupLinearTransforms(); g2.transformPoints()
        g2.RANSAC(0)                                                     # a fixed image is not owned

"""The raw-coefficient story on an equality test instead of an argument.

`FROG_REFERENCE_ORDER=1` (frog_amd/csrc/device/k_reforder.hip.h) runs the solver loops on the device with nothing
re-associated: the reference's own weight arithmetic for every half-link, a point's f32 sums as one chain in readPairs order,
the linear step's f64 sums as one chain per image, the B-spline scatter point by point with the f64 tap product and f32
read-modify-write, the transform without fused multiply-adds.

  * reference-order mode == CPU oracle, `np.array_equal`, free-running over ImageGroup::run's schedule: per-point sums,
    gradient images, proposals, every lattice, matrices, mixtures, coordinates (energies: the reference's own sum order
    depends on its thread count, imageGroup.cxx:239 -- compared to 1e-12);
  * the product path (fast weight, culling lists, fused sweep, tiled scatter) against reference-order mode ON THE DEVICE, at
    sizes and schedule lengths the oracle cannot follow inside a test -- by construction every deviation found there is
    re-association, and its size is recorded per lattice: raw coefficients, support-weighted coefficients, and the
    displacement field on a DENSE lattice over the group's bounding box (what tools/VolumeTransform.cxx:119-136 samples),
    not only at the keypoints.
"""
import os

import numpy as np
import pytest

from frog_amd import _abi
from frog_amd.image_group import ImageGroup
from frog_amd.pairs import Pairs
from oracle.oracle_api import OracleGroup
from lattice_util import lattice_deviation, node_weights, lattice_taps, face_crossing_nodes

pytestmark = pytest.mark.gpu
REL = 1e-4
RAW_REL = 1e-3      # raw coefficients, product path vs reference-order mode, levels 0-3 at benchmark sizes: measured <= 2.7e-4 (a ten-fold regression fails)


def note(name, value):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "test_numbers.txt"), "a") as fh:
            fh.write(f"{name} {value}\n")


class Side:
    """One driver interface over the HIP path (ImageGroup) and the oracle (OracleGroup)."""

    def __init__(self, pairs, oracle=False, **opt):
        self.oracle = oracle
        self.pairs = pairs
        self.first = int(opt.get("n_fixed_images", 0))       # -fi: the first images are fixed, the context owns the others
        if oracle:
            self.g = OracleGroup(pairs.model, _abi.FrogOptions.default(**opt))
            self.g.setup_stats()
            self.g.keep_raw_gradient(True)
        else:
            self.g = ImageGroup(pairs, **opt)

    def init(self):
        (self.g.linear_init if self.oracle else self.g.setupLinearTransforms)()

    def transform(self, apply=False):
        (self.g.transform_points if self.oracle else self.g.transformPoints)(apply)

    def stats(self):
        (self.g.update_stats if self.oracle else self.g.updateStats)()

    def linear(self):
        return self.g.linear_step() if self.oracle else self.g.updateLinearTransforms()

    def setup(self, level):
        return self.g.deformable_setup(level, _abi.FrogGridInfo()) if self.oracle else self.g.setupDeformableTransforms(level)

    def deformable(self, alpha):
        return self.g.deformable_step(alpha) if self.oracle else self.g.updateDeformableTransforms(alpha)

    def xyz(self):
        return self.g.xyz() if self.oracle else self.g.points()[0]

    def xyz2(self):
        return self.g.xyz2() if self.oracle else self.g.points()[1]

    def matrices(self):
        return np.stack([self.g.matrix(i) for i in range(self.first, self.pairs.n_images)])

    def ems(self):
        return np.stack([self.g.em(i) for i in range(self.pairs.n_images)])

    def point_sums(self):
        return self.g.point_sums()

    def gradient_raw(self, image, n_cp):
        return self.g.gradient_raw(image, n_cp) if self.oracle else self.g.gradient(image, n_cp)

    def grid(self, image, k):
        return self.g.grid(image, k, _abi.FrogGridInfo()) if self.oracle else self.g.grid(image, k)

    def num_grids(self):
        return self.g.num_grids()


def lockstep(sides, li, dl, di, check, alpha0=0.02):
    """ImageGroup::run's schedule (imageGroup.cxx:54-128) on every side, each on its own state; `check(tag, ...)` is called
    with the sides after every step.  Returns the lattices created per level."""
    for s in sides: s.init()
    for s in sides: s.transform()
    check("init", sides)
    for it in range(li):
        if it % 10 == 0:
            for s in sides: s.stats()
        e = [s.linear() for s in sides]
        for s in sides: s.transform()
        check(("linear", it), sides, e)
    for s in sides: s.transform(True)
    grids = []
    for level in range(dl):
        def setup():
            infos = [s.setup(level) for s in sides]
            for s in sides: s.transform()
            check(("setup", level), sides, infos=infos)
        setup()
        alpha, nd, it, n_g = np.float32(alpha0), 0, 0, 1
        while it < di:
            if it % 10 == 0:
                for s in sides: s.stats()
            e = [s.deformable(float(alpha)) for s in sides]
            assert len({x < 0 for x in e}) == 1, f"guard decisions differ at level {level}, iteration {it}: {e}"
            check(("step", level, it), sides, e)
            if e[0] < 0:
                if nd == 0:
                    alpha = np.float32(alpha / np.float32(2))
                n_g += 1
                for s in sides: s.transform(True)
                setup()
                nd = 0
                continue
            nd += 1
            for s in sides: s.transform()
            check(("deformable", level, it), sides, e)
            it += 1
        grids.append(n_g)
        for s in sides: s.transform(True)
    return grids


def equality_checker(images, counters):
    """Every comparable state of two sides, with np.array_equal."""
    def check(tag, sides, e=None, infos=None):
        a, b = sides
        kind = tag if isinstance(tag, str) else tag[0]
        if infos is not None:
            for f in ("dims", "origin", "spacing"):
                assert list(getattr(infos[0], f)) == list(getattr(infos[1], f)), (tag, f)
        if e is not None and e[0] >= 0:
            assert abs(e[0] - e[1]) <= 1e-12 * abs(e[1]), (tag, e)
        if kind == "step":
            assert np.array_equal(a.point_sums(), b.point_sums()), f"{tag}: per-point sums differ"
            info = a.grid(images[0], a.num_grids() - 1)[0]
            n_cp = info.dims[0] * info.dims[1] * info.dims[2]
            for i in images:
                assert np.array_equal(a.gradient_raw(i, n_cp), b.gradient_raw(i, n_cp)), f"{tag}: gradient image {i} differs"
            counters["steps"] += 1
            return
        assert np.array_equal(a.xyz2(), b.xyz2()), f"{tag}: coordinates differ"
        if kind in ("linear", "init"):
            assert np.array_equal(a.matrices(), b.matrices()), f"{tag}: matrices differ"
        assert np.array_equal(a.ems(), b.ems()), f"{tag}: mixtures differ"
        if kind == "deformable":
            k = a.num_grids() - 1
            for i in images:
                assert np.array_equal(a.grid(i, k)[1], b.grid(i, k)[1]), f"{tag}: lattice {k} of image {i} differs"
        counters[kind] = counters.get(kind, 0) + 1
    return check


def run_equal(pairs, li, dl, di, monkeypatch, images=None, **opt):
    monkeypatch.setenv("FROG_REFERENCE_ORDER", "1")
    dev = Side(pairs, **opt)
    monkeypatch.delenv("FROG_REFERENCE_ORDER")
    ref = Side(pairs, oracle=True, **opt)
    images = list(images if images is not None else range(dev.first, pairs.n_images))
    counters = {"steps": 0}
    grids = lockstep([dev, ref], li, dl, di, equality_checker(images, counters))
    assert dev.num_grids() == ref.num_grids() == sum(grids)
    for k in range(ref.num_grids()):                       # every lattice of the chain, finished ones included
        for i in range(dev.first, pairs.n_images):
            assert np.array_equal(dev.grid(i, k)[1], ref.grid(i, k)[1]), f"lattice {k} of image {i} differs"
    assert np.array_equal(dev.xyz(), ref.xyz()) and np.array_equal(dev.matrices(), ref.matrices())
    return grids, counters


def test_reference_order_mode_equals_the_oracle_small_group(monkeypatch):
    """6 images x 3 000 keypoints over the reference's WHOLE default schedule (-li 50 -dl 3 -di 200: 650 accepted iterations,
    65 statistics refreshes, the guard's rejections with their re-basing, regrids and alpha-halving as they come): every
    per-point sum, gradient image, lattice, matrix, mixture and coordinate has the oracle's bits after every step."""
    pairs = Pairs.synthetic(6, 3000, 1500, seed=7)
    grids, counters = run_equal(pairs, 50, 3, 200, monkeypatch)
    note("reference_order_small", f"equal bits over 50 + 3 x 200 iterations, lattices per level {grids}, compared steps {counters}")
    assert counters["linear"] == 50 and counters["deformable"] == 600
    assert sum(grids) > 3, "the schedule was meant to include rejected steps and regrids"


def test_reference_order_mode_equals_the_oracle_config5_shaped(monkeypatch):
    """40 images x 20 000 keypoints, ~20 partner images each (1.2e7 half-links), 10 linear + 5 levels x 3 iterations: the
    shape of BASELINE.json configs[4] at a size the oracle walks in seconds; level 4 has 870 000 nodes per image."""
    pairs = Pairs.synthetic(40, 20000, 16667, seed=2, partners_per_image=20)
    grids, counters = run_equal(pairs, 10, 5, 3, monkeypatch, images=range(0, 40, 7))
    note("reference_order_cfg5_shaped", f"equal bits over 10 + 5 x 3 iterations, lattices per level {grids}, compared steps {counters}")
    assert len(grids) == 5


# ---- the product path against reference-order mode, both on the device --------------------------------------------------

def dense_field_deviation(a, b, k, images, xyz, n_per_axis=24, skip=None):
    """Displacement of lattice k on both sides on a dense lattice of points over the bounding box of the coordinates the
    lattice acts on (what a resampler evaluates: tools/VolumeTransform.cxx:119-136), relative to the largest displacement.
    skip: mask of control points whose difference is left out (tests/lattice_util.py face_crossing_nodes)."""
    lo, hi = xyz.min(axis=0).astype(np.float64), xyz.max(axis=0).astype(np.float64)
    axes = [np.linspace(lo[d], hi[d], n_per_axis) for d in range(3)]
    pts = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, 3)
    worst, scale = 0.0, 0.0
    for i in images:
        info, ca = a.grid(i, k)
        _, cb = b.grid(i, k)
        idx, wt = lattice_taps(pts, info)
        diff = ca.astype(np.float64) - cb
        if skip is not None:
            diff[skip] = 0.0
        db = np.einsum("nt,ntk->nk", wt, cb.astype(np.float64)[idx])
        worst = max(worst, float(np.max(np.abs(np.einsum("nt,ntk->nk", wt, diff[idx])))))
        scale = max(scale, float(np.max(np.abs(db))))
    return worst / max(scale, 1e-30), scale


class RefAsOracle:
    """Adapter: a reference-order device side seen through the oracle's getter names (for tests/lattice_util.py)."""

    def __init__(self, side):
        self.s = side
        self.n_images = side.pairs.n_images

    def grid(self, image, k, info=None):
        return self.s.g.grid(image, k)


def fast_against_reference_order(pairs, li, dl, di, monkeypatch, images, **opt):
    monkeypatch.setenv("FROG_REFERENCE_ORDER", "1")
    ref = Side(pairs, **opt)
    monkeypatch.delenv("FROG_REFERENCE_ORDER")
    fast = Side(pairs, **opt)
    po = np.asarray(pairs.point_offset)
    snaps, snaps_fast, worst = [], [], {"E": 0.0}

    def check(tag, sides, e=None, infos=None):
        if infos is not None:
            assert list(infos[0].dims) == list(infos[1].dims), tag
            snaps.append(sides[1].xyz().copy())
            snaps_fast.append(sides[0].xyz().copy())
        if e is not None and e[0] >= 0 and tag[0] != "step":
            worst["E"] = max(worst["E"], abs(e[0] - e[1]) / abs(e[1]))
    grids = lockstep([fast, ref], li, dl, di, check)
    out = []
    adapter = RefAsOracle(ref)
    for k in range(ref.num_grids()):
        w = node_weights(adapter, k, po, snaps[k])
        d = {"raw": 0.0, "weighted": 0.0, "field": 0.0}
        for i in images:
            r = lattice_deviation(fast.g, adapter, k, i, snaps[k][po[i]:po[i + 1]], w)
            if r["raw"] >= d["raw"]:
                d["raw_image"] = int(i)
                for key in ("raw_node", "raw_node_weight", "raw_node_support", "raw_node_points"):
                    d[key] = r[key]
            for key in ("raw", "weighted", "field"):
                d[key] = max(d[key], r[key])
            d["weak"], d["nodes"] = r["weak"], r["nodes"]
        d["dense_field"], d["max_disp"] = dense_field_deviation(fast.g, ref.g, k, images, snaps[k])
        # control points one of the two runs reaches across a cell face (tests/lattice_util.py): counted, and the same two
        # numbers without them
        info = ref.g.grid(0 + ref.first, k)[0]
        crossing, skip = face_crossing_nodes(snaps_fast[k], snaps[k], info)
        d["face_crossings"], d["crossing_nodes"] = int(len(crossing)), int(skip.sum())
        d["raw_elsewhere"], d["dense_field_elsewhere"] = d["raw"], d["dense_field"]
        if len(crossing):
            d["raw_elsewhere"], scale = 0.0, 0.0
            for i in images:
                ca, cb = fast.g.grid(i, k)[1], ref.g.grid(i, k)[1]
                d["raw_elsewhere"] = max(d["raw_elsewhere"], float(np.abs(ca.astype(np.float64) - cb)[~skip].max()))
                scale = max(scale, float(np.abs(cb).max()))
            d["raw_elsewhere"] /= max(scale, 1e-30)
            d["dense_field_elsewhere"] = dense_field_deviation(fast.g, ref.g, k, images, snaps[k], skip=skip)[0]
        out.append(d)
    # the WHOLE chain of an image (matrix, then every lattice in creation order: what transforms/<i>.json holds and
    # tools/VolumeTransform.cxx / PointsTransform.cxx evaluate) on a dense lattice of points over the image's own keypoint
    # box, through the device's chain evaluation (include/frog_chain.h): deviation of the displacement T(x) - x
    from frog_amd.chain import Chain, Link
    chain = {"rel": 0.0, "mm": 0.0, "max_disp_mm": 0.0}
    x0 = np.asarray(pairs.xyz, np.float64).reshape(-1, 3)
    for i in images:
        pts_i = x0[po[i]:po[i + 1]]
        lo, hi = pts_i.min(axis=0), pts_i.max(axis=0)
        grid = np.stack(np.meshgrid(*[np.linspace(lo[d], hi[d], 20) for d in range(3)], indexing="ij"), axis=-1).reshape(-1, 3)
        disp = []
        for side in (fast, ref):
            links = [Link.linear(side.g.matrix(i))]
            for k in range(side.num_grids()):
                info, c = side.g.grid(i, k)
                links.append(Link.bspline(list(info.dims), list(info.origin), list(info.spacing), c))
            ch = Chain(links)
            disp.append(ch.apply(grid) - grid)
            ch.close()
        dev = float(np.max(np.abs(disp[0] - disp[1])))
        scale = float(np.max(np.abs(disp[1])))
        chain["mm"] = max(chain["mm"], dev); chain["max_disp_mm"] = max(chain["max_disp_mm"], scale)
        chain["rel"] = max(chain["rel"], dev / max(scale, 1e-30))
    mf, mr = fast.matrices(), ref.matrices()
    diag = lambda a: np.stack([a[:, 0, 0], a[:, 1, 1], a[:, 2, 2]])
    m = max(float(np.max(np.abs(diag(mf) - diag(mr))) / np.max(np.abs(diag(mr)))),
            float(np.max(np.abs(mf[:, :3, 3] - mr[:, :3, 3])) / np.max(np.abs(mr[:, :3, 3]))))
    x = float(np.max(np.abs(fast.xyz().astype(np.float64) - ref.xyz())) / np.max(np.abs(ref.xyz())))
    ca, cb = fast.g.countInliers(), ref.g.countInliers()
    census = sum(abs(ca[i].inliers - cb[i].inliers) for i in range(pairs.n_images))
    return {"lattices": out, "grids": grids, "E": worst["E"], "matrices": m, "xyz": x, "census": census, "chain": chain}


def report(name, r):
    note(name, f"E {r['E']:.2e} matrices {r['matrices']:.2e} xyz {r['xyz']:.2e} grids {r['grids']} census_differs_by {r['census']} "
               f"whole_chain_dense rel {r['chain']['rel']:.2e} abs {r['chain']['mm']:.2e} mm of {r['chain']['max_disp_mm']:.1f} mm")
    for k, d in enumerate(r["lattices"]):
        note(f"{name}_lattice_{k}", " ".join(f"{a} {b:.2e}" if isinstance(b, float) else f"{a} {b}" for a, b in d.items()))


def test_fast_path_against_reference_order_small_group(monkeypatch):
    """The product path against the mode the tests above hold equal to the CPU oracle, on the same group and schedule
    (6 images, 50 + 3 x 40 iterations): same guard decisions; energies 1e-6; per lattice the displacement field at the
    keypoints AND on a dense lattice over the bounding box within 1e-4 of the largest displacement, support-weighted
    coefficients 1e-4, raw coefficients 1e-3 (measured 4.4e-5), the whole chain on a dense lattice 1e-5 (measured 5.5e-7)."""
    pairs = Pairs.synthetic(6, 3000, 1500, seed=7)
    r = fast_against_reference_order(pairs, 50, 3, 40, monkeypatch, range(6))
    report("fast_vs_reference_order_small", r)
    for k, d in enumerate(r["lattices"]):
        assert d["field"] <= REL and d["dense_field"] <= REL and d["weighted"] <= REL and d["raw"] <= RAW_REL, (k, d)
    assert r["E"] < 1e-6 and r["xyz"] < 1e-6 and r["chain"]["rel"] <= 1e-5


def test_fast_path_against_reference_order_small_group_whole_schedule(monkeypatch):
    """The same over the reference's whole default schedule (-li 50 -dl 3 -di 200).  updateDeformableTransforms drops a
    half-link when its weight is below the threshold (imageGroup.cxx:274): a discontinuity, so two runs whose coordinates
    differ in the last bits sooner or later decide ONE link differently, and on a group of 9e4 half-links one link is 1e-5 of
    the energy -- from then on the two runs are two different (equally valid) trajectories.  Reported: deviations and by how
    many half-links the two runs' final inlier census differs.  Asserted: the loose 1e-3 bars (same decisions of the guard,
    same lattices)."""
    pairs = Pairs.synthetic(6, 3000, 1500, seed=7)
    r = fast_against_reference_order(pairs, 50, 3, 200, monkeypatch, range(6))
    report("fast_vs_reference_order_small_whole_schedule", r)
    for k, d in enumerate(r["lattices"]):
        assert d["field"] <= 1e-3 and d["dense_field"] <= 1e-3 and d["raw"] <= 2e-3, (k, d)          # measured 3.8e-4 / 1.6e-4 / 2.6e-4
    assert r["E"] < 1e-3 and r["chain"]["rel"] <= 1e-4                                                   # measured 6.4e-5 / 6.2e-6


def test_fast_path_against_reference_order_config3(monkeypatch):
    """BASELINE.json configs[2] at its size (100 images x 20 000 keypoints, 1e8 half-links): 20 linear + 3 x 20 deformable
    iterations, product path against reference-order mode -- both on the device, so the comparison costs seconds where the
    oracle takes minutes.  The full default schedule is run once per round by scripts/parity_reference_order.py and its
    numbers kept in profiles/."""
    pairs = Pairs.synthetic(100, 20000, 10101, seed=1)
    r = fast_against_reference_order(pairs, 20, 3, 20, monkeypatch, range(0, 100, 9))
    report("fast_vs_reference_order_cfg3", r)
    for k, d in enumerate(r["lattices"]):
        # measured: field 5.3e-6, dense 5.4e-6, weighted 3.7e-6, raw 2.7e-4 (level 2: 7 100 of 18 216 nodes on the rim)
        assert d["field"] <= REL and d["dense_field"] <= REL and d["weighted"] <= REL and d["raw"] <= RAW_REL, (k, d)
    assert r["E"] < 1e-6 and r["xyz"] < 1e-6 and r["chain"]["rel"] <= 1e-5      # measured 6.2e-8, 2.0e-7, 1.3e-7


def test_fast_path_against_reference_order_config5_full_size(monkeypatch):
    """BASELINE.json configs[4] at its size -- 500 images x 20 000 keypoints, ~60 partner images each, 4.6e8 half-links, five
    levels, -gd 1 (level 4: 9e5 nodes per image, bricks of 8^3 cells) -- which no CPU run can follow inside a test: the
    product path against reference-order mode (bit-equal to the oracle on the same kind of group at 40 images: above), both
    on the device, 4 linear + 5 x 2 deformable iterations.  Same guard decisions and lattices; energies 1e-6; per lattice
    the displacement field at the keypoints and on a dense lattice 1e-4, support-weighted coefficients 1e-4, raw 1e-3 (measured
    4.8e-5 on level 4), the whole chain on a dense lattice 1e-5 (measured 3.3e-8)."""
    pairs = Pairs.synthetic(500, 20000, 16667, seed=1, partners_per_image=60)
    r = fast_against_reference_order(pairs, 4, 5, 2, monkeypatch, range(0, 500, 83))
    report("fast_vs_reference_order_cfg5", r)
    assert len(r["grids"]) == 5
    for k, d in enumerate(r["lattices"]):
        assert d["field"] <= REL and d["dense_field"] <= REL and d["weighted"] <= REL and d["raw"] <= RAW_REL, (k, d)
    assert r["E"] < 1e-6 and r["xyz"] < 1e-6 and r["chain"]["rel"] <= 1e-5


def test_fast_path_against_reference_order_config5_shaped_long_level4(monkeypatch):
    """cfg-5-shaped group of 40 images (20 000 keypoints, 20 partner images, five levels: level 4 has 8.5e5 control points per
    image, about one keypoint per cell) over 20 + 5 x 12 iterations -- long enough for the level-4 guard to reject a step and
    for a lattice to live ten iterations.  That is where the reference's update is discontinuous in a way the coarser levels
    hide (tests/lattice_util.py, face_crossing_nodes): a control point whose only support is a point within an ulp of a cell
    face moves a full step per iteration in the run that sees the point on one side, and not at all in the other.  Reported
    per lattice: how many points the two runs place across a face, the deviations with and without the control points of
    those points' stencils.  Asserted: same guard decisions; energies; the field at the keypoints (what the energy sees);
    coefficients and dense field away from the crossings."""
    pairs = Pairs.synthetic(40, 20000, 16667, seed=2, partners_per_image=20)
    r = fast_against_reference_order(pairs, 20, 5, 12, monkeypatch, range(40))
    report("fast_vs_reference_order_cfg5_shaped_long", r)
    assert len(r["grids"]) == 5 and sum(r["grids"]) > 5           # the guard rejected at least once
    n_coarse = sum(r["grids"][:4])             # lattices of levels 0-3
    for k, d in enumerate(r["lattices"]):
        assert d["face_crossings"] <= 8, (k, d)
        # levels 0-3: measured raw <= 8.0e-4 (away from crossings 2.8e-4 .. 8.0e-4), dense field <= 3.5e-5; level 4: raw away
        # from the crossings 2.6e-3, dense field 2.3e-4 away from them (7.7e-4 with them), field at the keypoints 5.7e-4
        if k < n_coarse:
            assert d["raw_elsewhere"] <= 2e-3 and d["field"] <= REL and d["dense_field_elsewhere"] <= REL, (k, d)
        else:
            assert d["field"] <= 1e-3 and d["raw_elsewhere"] <= 1e-2 and d["dense_field_elsewhere"] <= 1e-3, (k, d)
    assert r["E"] < 1e-6 and r["chain"]["rel"] <= 1e-5          # measured 3.5e-8, 2.0e-6


# ---- reference-order mode on the inputs the reference's loops have special cases for ------------------------------------

def test_reference_order_mode_equals_the_oracle_ragged_group_with_duplicate_links(monkeypatch):
    """Images of 25 .. 1 500 points, points without links, 400 duplicate links of ONE point into one partner image, an image
    pair whose block appears twice in the file (tests/test_gpu_parity.py ragged_pairs), a reservoir smaller than the link
    count: 12 linear + 2 x 8 deformable iterations, every quantity equal to the oracle's after every step."""
    from test_gpu_parity import ragged_pairs
    pairs = ragged_pairs()
    grids, counters = run_equal(pairs, 12, 2, 8, monkeypatch, stats_max_size=500)
    assert counters["linear"] == 12 and counters["deformable"] == 16


def test_reference_order_mode_equals_the_oracle_with_landmark_constraints_and_error_maps(monkeypatch):
    """-lc: hard links between landmarks (link-less extra points) enter a point's f32 sums after its regular links
    (imageGroup.cxx:280-295) and the error maps (:520-533): per-point sums, lattices and error maps equal to the oracle's."""
    pairs = Pairs.synthetic(6, 3000, 1500, seed=7)
    po0 = np.asarray(pairs.point_offset).copy()
    rng = np.random.default_rng(8)
    for i in range(6):
        pts = np.asarray(pairs.xyz)[po0[i]:po0[i + 1]]
        pairs.append_points(i, pts[rng.choice(3000, 3, replace=False)])
    po = np.asarray(pairs.point_offset)
    point, partner = [], []
    for k in range(3):
        ids = [po[i] + 3000 + k for i in range(6)]
        for a in ids:
            for b in ids:
                if a != b:
                    point.append(a); partner.append(b)
    w2 = np.float32(6 * 50.0) ** 2
    monkeypatch.setenv("FROG_REFERENCE_ORDER", "1")
    dev = Side(pairs)
    monkeypatch.delenv("FROG_REFERENCE_ORDER")
    ref = Side(pairs, oracle=True)
    dev.g.set_hard_links(point, partner, w2); ref.g.set_hard_links(point, partner, w2)
    counters = {"steps": 0}
    lockstep([dev, ref], 12, 2, 6, equality_checker(range(6), counters))
    assert counters["deformable"] == 12
    dev.g.residualSums()
    info = dev.grid(0, dev.num_grids() - 1)[0]
    n_cp = info.dims[0] * info.dims[1] * info.dims[2]
    for i in range(6):
        assert np.array_equal(dev.g.errorMap(i)[1], ref.g.error_map(i, n_cp)), f"error map of image {i}"


@pytest.mark.parametrize("n_fixed", [1, 4])
def test_reference_order_mode_equals_the_oracle_with_fixed_images(monkeypatch, n_fixed):
    """-fi n: the first n images keep their position, every loop but updateStats starts at image n, the group mean is not
    removed (imageGroup.cxx:398): a 15 + 2 x 12 schedule, equal to the oracle's after every step."""
    pairs = Pairs.synthetic(6, 3000, 1500, seed=7)
    grids, counters = run_equal(pairs, 15, 2, 12, monkeypatch, n_fixed_images=n_fixed)
    assert counters["deformable"] == 24

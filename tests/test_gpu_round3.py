"""Round-3 parity tests on the GPU (all through the C ABI):
  * the headline configuration (cfg 3: 100 images x 20 000 keypoints, 1e8 half-links) FREE-RUNNING against the oracle over
    a schedule of 10 linear + 3 levels x 10 deformable iterations, regrids as they come -- with the raw max-norm of the
    coefficient deviation reported per lattice next to the support-weighted one (tests/lattice_util.py);
  * a well-supported small case whose raw coefficients are held to the plain 1e-4 bar;
  * the same kind of case with every inlier weight forced through the reference's own arithmetic (FROG_WEIGHT_EXACT=1):
    the deviations on weakly supported nodes do not come from the fast f32 weight.
"""
import os

import numpy as np
import pytest

from frog_amd import _abi
from frog_amd.image_group import ImageGroup
from frog_amd.pairs import Pairs
from oracle.oracle_api import OracleGroup
from lattice_util import lattice_deviation, node_weights

pytestmark = pytest.mark.gpu
REL = 1e-4


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def note(name, value):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "test_numbers.txt"), "a") as fh:
            fh.write(f"{name} {value}\n")


def free_run(pairs, li, dl, di, images=None, **opt):
    """ImageGroup::run's schedule (imageGroup.cxx:54-128) on the HIP path and on the oracle, both free-running from the
    same pairs.  Returns per-lattice deviations and the worst energy / matrix deviation."""
    g = ImageGroup(pairs, **opt)
    ref = OracleGroup(pairs.model, _abi.FrogOptions.default(**opt))
    ref.setup_stats()
    po = np.asarray(pairs.point_offset)
    g.setupLinearTransforms(); ref.linear_init()
    g.transformPoints(); ref.transform_points()
    worst_e = 0.0
    for it in range(li):
        if it % 10 == 0:
            g.updateStats(); ref.update_stats()
        e, er = g.updateLinearTransforms(), ref.linear_step()
        g.transformPoints(); ref.transform_points()
        worst_e = max(worst_e, abs(e - er) / er)
    worst_m = 0.0
    for i in range(pairs.n_images):
        m, mr = g.matrix(i), ref.matrix(i)
        worst_m = max(worst_m, relerr(np.diag(m)[:3], np.diag(mr)[:3]), relerr(m[:3, 3], mr[:3, 3]))
    g.transformPoints(True); ref.transform_points(True)
    snapshots, levels, grids = [], [], []
    for level in range(dl):
        def setup():
            info = g.setupDeformableTransforms(level)
            rinfo = ref.deformable_setup(level, _abi.FrogGridInfo())
            assert list(info.dims) == list(rinfo.dims), f"lattice dimensions differ at level {level}"
            snapshots.append(ref.xyz().copy()); levels.append(level)
            g.transformPoints(); ref.transform_points()
        setup()
        alpha, nd, it, n_g = np.float32(0.02), 0, 0, 1
        while it < di:
            if it % 10 == 0:
                g.updateStats(); ref.update_stats()
            e, er = g.updateDeformableTransforms(float(alpha)), ref.deformable_step(float(alpha))
            assert (e < 0) == (er < 0), f"guard decisions differ at level {level}, iteration {it}"
            if e < 0:
                if nd == 0:
                    alpha = np.float32(alpha / np.float32(2))
                n_g += 1
                g.transformPoints(True); ref.transform_points(True)
                setup()
                nd = 0
                continue
            nd += 1
            g.transformPoints(); ref.transform_points()
            worst_e = max(worst_e, abs(e - er) / er)
            it += 1
        grids.append(n_g)
        g.transformPoints(True); ref.transform_points(True)
    assert g.num_grids() == ref.num_grids() == len(snapshots)
    per_lattice = []
    for k in range(ref.num_grids()):
        w = node_weights(ref, k, po, snapshots[k])
        worst = {"raw": 0.0, "weighted": 0.0, "field": 0.0, "dense": 0.0, "weak": 0, "nodes": 0, "level": levels[k]}
        for i in (images if images is not None else range(pairs.n_images)):
            d = lattice_deviation(g, ref, k, i, snapshots[k][po[i]:po[i + 1]], w)
            for key in ("raw", "weighted", "field", "dense"):
                worst[key] = max(worst[key], d[key])
            worst["weak"], worst["nodes"] = d["weak"], d["nodes"]
        per_lattice.append(worst)
    final = relerr(g.points()[0], ref.xyz())
    # the whole chain of an image (matrix + every lattice) on a dense lattice over its keypoint box, through the device's
    # chain evaluation (include/frog_chain.h; what tools/PointsTransform.cxx / VolumeTransform.cxx do with transforms/<i>.json)
    from frog_amd.chain import Chain, Link
    x0 = np.asarray(pairs.xyz, np.float64).reshape(-1, 3)
    chain_rel = chain_mm = 0.0
    for i in list(images if images is not None else range(pairs.n_images))[:12]:
        lo, hi = x0[po[i]:po[i + 1]].min(axis=0), x0[po[i]:po[i + 1]].max(axis=0)
        pts = np.stack(np.meshgrid(*[np.linspace(lo[d], hi[d], 16) for d in range(3)], indexing="ij"), axis=-1).reshape(-1, 3)
        disp = []
        for m, grid_of in ((g.matrix(i), lambda k: g.grid(i, k)), (ref.matrix(i), lambda k: ref.grid(i, k, _abi.FrogGridInfo()))):
            links = [Link.linear(m)]
            for k in range(ref.num_grids()):
                info, c = grid_of(k)
                links.append(Link.bspline(list(info.dims), list(info.origin), list(info.spacing), c))
            ch = Chain(links); disp.append(ch.apply(pts) - pts); ch.close()
        dev = float(np.max(np.abs(disp[0] - disp[1])))
        chain_mm = max(chain_mm, dev); chain_rel = max(chain_rel, dev / max(float(np.max(np.abs(disp[1]))), 1e-30))
    return {"E": worst_e, "matrices": worst_m, "lattices": per_lattice, "grids_per_level": grids, "final_xyz": final,
            "chain_rel": chain_rel, "chain_mm": chain_mm}


def test_config3_free_running_schedule_against_the_oracle():
    """BASELINE.json configs[2] at its size: 10 linear iterations + 3 levels x 10 deformable iterations (statistics
    refreshes, lattice set-ups, re-basing, any regrid the guard asks for), device and oracle each on their own state
    from the first step on.  E series 1e-4, matrices 1e-6, the same lattices per level, displacement field of every
    lattice 1e-4 at every point of every image, support-weighted coefficients 1e-4 -- and the RAW max-norm of the
    coefficient deviation per lattice is printed, not hidden behind the weighted bar."""
    pairs = Pairs.synthetic(100, 20000, 10101, seed=1)
    r = free_run(pairs, 10, 3, 10)
    note("cfg3_free_run", f"E {r['E']:.2e} matrices {r['matrices']:.2e} final_xyz {r['final_xyz']:.2e} grids {r['grids_per_level']} "
                          f"whole_chain_dense rel {r['chain_rel']:.2e} abs {r['chain_mm']:.2e} mm")
    assert r["chain_rel"] <= REL
    for k, d in enumerate(r["lattices"]):
        note(f"cfg3_free_run_lattice_{k}", f"level {d['level']} raw {d['raw']:.2e} weighted {d['weighted']:.2e} field {d['field']:.2e} dense_field {d['dense']:.2e} "
                                             f"weak_nodes {d['weak']}/{d['nodes']}")
    assert r["E"] < REL
    assert r["matrices"] < 1e-6
    for k, d in enumerate(r["lattices"]):
        assert d["field"] <= REL, f"lattice {k}: displacement field off by {d['field']:.2e}"
        assert d["dense"] <= REL, f"lattice {k}: displacement field on the dense lattice off by {d['dense']:.2e}"
        assert d["weighted"] <= REL, f"lattice {k}: supported coefficients off by {d['weighted']:.2e}"
        assert d["raw"] <= 1e-3, f"lattice {k}: raw coefficients off by {d['raw']:.2e}"       # measured 1.05e-4
    assert r["final_xyz"] < 1e-6


# a lattice whose every node is determined by thousands of points: -g 700 gives 1 x 1 x 1 cells (4^3 nodes) at level 0 and
# 1 x 1 x 2 at level 1 on the 480 x 480 x 720 mm box of the synthetic groups
WELL_SUPPORTED = dict(initial_grid_size=700.0)


def test_raw_coefficients_of_a_well_supported_lattice():
    """The plain bar, no weighting: every raw coefficient within 1e-4 of the lattice's largest on a case where no node is
    weakly supported -- a 1e-4..1e-2 regression of the arithmetic shows here whatever the rim criterion forgives."""
    pairs = Pairs.synthetic(6, 3000, 1500, seed=7)
    r = free_run(pairs, 50, 2, 40, **WELL_SUPPORTED)
    for k, d in enumerate(r["lattices"]):
        note(f"well_supported_lattice_{k}", f"raw {d['raw']:.2e} field {d['field']:.2e} weak_nodes {d['weak']}/{d['nodes']}")
        assert d["raw"] <= REL, f"lattice {k}: raw coefficients off by {d['raw']:.2e}"
        assert d["field"] <= REL
    assert r["E"] < REL and r["matrices"] < REL


def test_rim_deviations_with_the_fast_and_with_the_exact_weight(monkeypatch):
    """The same free-running case twice: inlier weights in the fast f32 form (k_links.hip.h inlier_probability, within
    2^-16 of the reference's) and forced through the reference's own promotions for EVERY half-link
    (FROG_WEIGHT_EXACT=1).  Both runs have to meet the bars; the raw deviations on weakly supported nodes are reported for
    both.  What they show has changed with the kernels: in round 4 (f64 FMA transform) 3.35e-4 fast / 3.81e-4 exact -- "not
    the weight"; in round 5 (f32 transform) 3.35e-4 fast / 5.6e-6 exact -- on this group the product path with exact weights
    now lands within 6e-6 of the oracle on EVERY node, and what the rim amplifies in the default path is the fast weight's
    1.25e-6.  Either way an ulp-sized input difference, amplified by nodes the data do not determine (tests/lattice_util.py);
    the displacement field agrees to 1e-5 in both."""
    pairs = Pairs.synthetic(6, 3000, 1500, seed=3)
    fast = free_run(pairs, 50, 3, 40)
    monkeypatch.setenv("FROG_WEIGHT_EXACT", "1")
    exact = free_run(pairs, 50, 3, 40)
    monkeypatch.delenv("FROG_WEIGHT_EXACT")
    # ... and a third time with the deformable sweeps' one-exponential form switched off (FROG_WEIGHT_GENERAL=1: no image gets
    # a range, every weight is min of two inlier_probability values as before round 5's second session): the same bars
    monkeypatch.setenv("FROG_WEIGHT_GENERAL", "1")
    general = free_run(pairs, 50, 3, 40)
    monkeypatch.delenv("FROG_WEIGHT_GENERAL")
    raw_f = max(d["raw"] for d in fast["lattices"]); raw_x = max(d["raw"] for d in exact["lattices"])
    note("rim_deviation_fast_vs_exact_weights", f"raw fast {raw_f:.2e} exact {raw_x:.2e} general {max(d['raw'] for d in general['lattices']):.2e} field fast "
         f"{max(d['field'] for d in fast['lattices']):.2e} exact {max(d['field'] for d in exact['lattices']):.2e} "
         f"general {max(d['field'] for d in general['lattices']):.2e}; E fast {fast['E']:.2e} general {general['E']:.2e}")
    for r in (fast, exact, general):
        for d in r["lattices"]:
            assert d["weighted"] <= REL and d["field"] <= REL and d["raw"] <= 1e-3       # measured 3.4e-4 / 5.6e-6


def test_config4_eight_contexts_on_the_config3_group(tmp_path):
    """BASELINE.json configs[3] = configs[2] sharded 8 ways, AT ITS WORKLOAD: `bin/frog -ngl 8` (the C++ multi-GPU host:
    eight contexts, ragged shards of 12-13 images, eight replicas of the 24 MB coordinate table, every collective of
    include/frog_comm.h incl. the all-reduce of the 3 G proposal sums, host-staged because the box has one GPU) on the
    100-image / 1e8-half-link group for 2 linear + 3 levels x 3 deformable iterations, against the one-context run of
    the same binary: energies to the six digits measures.csv prints, matrices 1e-6, lattices 1e-5, identical census,
    and the eight replicas of xyz2 and of the mixture table bit-equal at the end (FROG_CHECK_REPLICAS)."""
    import csv
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = Pairs.synthetic(100, 20000, 10101, seed=1)
    assert 9.9e7 < pairs.n_half_links < 1.01e8
    one, eight = tmp_path / "one", tmp_path / "eight"
    one.mkdir(); eight.mkdir()
    pairs.write(one / "pairs.bin")
    os.symlink(one / "pairs.bin", eight / "pairs.bin")
    del pairs

    def run(cwd, *flags):
        r = subprocess.run([os.path.join(root, "bin", "frog"), "pairs.bin", "-li", "2", "-dl", "3", "-di", "3", "-j", "-q", "1", *flags],
                           cwd=cwd, capture_output=True, text=True, timeout=900, env=dict(os.environ, FROG_CHECK_REPLICAS="1"))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return r.stdout
    run(one)
    out = run(eight, "-ngl", "8")
    assert "Images sharded over 8 contexts" in out
    assert "Replicas identical : yes (8 contexts" in out
    ea = np.array([float(x[1]) for x in list(csv.reader(open(one / "measures.csv")))[1:]])
    eb = np.array([float(x[1]) for x in list(csv.reader(open(eight / "measures.csv")))[1:]])
    assert len(ea) == len(eb) == 11 and np.max(np.abs(ea - eb) / eb) < 1e-5          # six printed digits
    worst_m = worst_c = 0.0
    for i in range(100):
        ta = json.load(open(one / "transforms" / f"{i}.json"))["transforms"]
        tb = json.load(open(eight / "transforms" / f"{i}.json"))["transforms"]
        assert len(ta) == len(tb) >= 4
        worst_m = max(worst_m, relerr(ta[0]["matrix"], tb[0]["matrix"]))
        for x, y in zip(ta[1:], tb[1:]):
            assert x["dimensions"] == y["dimensions"]
            worst_c = max(worst_c, relerr(x["coeffs"], y["coeffs"]))
    note("cfg4_eight_contexts_vs_one", f"E {np.max(np.abs(ea - eb) / eb):.2e} matrices {worst_m:.2e} lattices {worst_c:.2e}")
    assert worst_m < 1e-6 and worst_c < 1e-5
    ba, bb = json.load(open(one / "bbox.json")), json.load(open(eight / "bbox.json"))
    assert ba["halfPairs"] == bb["halfPairs"] and abs(ba["inliers"] - bb["inliers"]) <= 2


def _short_run(pairs, **opt):
    g = ImageGroup(pairs, **opt)
    g.setupLinearTransforms(); g.transformPoints()
    es = []
    for it in range(6):
        if it % 10 == 0:
            g.updateStats()
        es.append(g.updateLinearTransforms()); g.transformPoints()
    g.transformPoints(True)
    sums = []
    for level in range(2):
        g.setupDeformableTransforms(level); g.transformPoints()
        for it in range(12):
            if it % 10 == 0:
                g.updateStats()
            e = g.updateDeformableTransforms(0.02)
            assert e >= 0
            es.append(e); g.transformPoints()
            if it in (0, 11):
                sums.append(g.point_sums().copy())
        g.transformPoints(True)
    grids = [g.grid(i, k)[1].copy() for k in range(g.num_grids()) for i in range(pairs.n_images)]
    return np.array(es), sums, grids, g.points()[0].copy()


@pytest.mark.parametrize("wide", ["0", "1"])
def test_fused_sweep_equals_the_per_group_sweep_bit_for_bit(monkeypatch, wide):
    """The deformable sweep as one block per tile with the eight group sums added in LDS (k_links.hip.h FUSED: one float4
    per point to memory) against one block per (4 tiles, partner group) + the scatter adding the eight partial sums: the
    same additions in the same order -- energies, per-point sums, lattices and coordinates identical bits, with narrow
    and wide records, with the culling list (built in the sweep) and without it."""
    pairs = Pairs.synthetic(6, 3000, 1500, seed=7)
    monkeypatch.setenv("FROG_WIDE_RECORDS", wide)
    for cull in ("1", "0"):
        monkeypatch.setenv("FROG_CULL", cull)
        monkeypatch.setenv("FROG_SWEEP_FUSED", "0")
        e0, s0, g0, x0 = _short_run(pairs)
        monkeypatch.setenv("FROG_SWEEP_FUSED", "1")
        e1, s1, g1, x1 = _short_run(pairs)
        assert np.array_equal(e0, e1)
        for a, b in zip(s0, s1):
            assert np.array_equal(a, b)
        for a, b in zip(g0, g1):
            assert np.array_equal(a, b)
        assert np.array_equal(x0, x1)


def test_block_order_of_the_fused_sweep_does_not_change_a_bit(monkeypatch):
    """FROG_TILE_SLICES only decides WHEN a tile is walked (frog_hip.hip: the fused launch's blocks go slice by slice of
    each image's eighth): image by image, two slices and eight slices give identical energies, sums, lattices, coordinates."""
    pairs = Pairs.synthetic(6, 3000, 1500, seed=7)
    monkeypatch.setenv("FROG_SWEEP_FUSED", "1")
    runs = []
    for slices in ("1", "2", "8"):
        monkeypatch.setenv("FROG_TILE_SLICES", slices)
        runs.append(_short_run(pairs))
    for e, s_, g_, x in runs[1:]:
        assert np.array_equal(runs[0][0], e) and np.array_equal(runs[0][3], x)
        for a, b in zip(runs[0][1], s_):
            assert np.array_equal(a, b)
        for a, b in zip(runs[0][2], g_):
            assert np.array_equal(a, b)


def test_apply_in_the_middle_of_a_lattice_against_the_oracle(small_pairs):
    """transformPoints(apply) rewrites the positions a lattice acts on; the C ABI allows going on with the SAME lattice
    afterwards (the reference's run() never does).  The scatter and the B-spline transforms read the positions from a copy
    in the lattice's brick order (ctx.h pos_b) that is gathered at set-up: an apply must mark it stale.  Same sequence on
    both sides, coordinates compared."""
    g = ImageGroup(small_pairs)
    ref = OracleGroup(small_pairs.model, _abi.FrogOptions.default())
    ref.setup_stats()
    g.setupLinearTransforms(); ref.linear_init()
    g.transformPoints(); ref.transform_points()
    for it in range(10):
        if it % 10 == 0:
            g.updateStats(); ref.update_stats()
        g.updateLinearTransforms(); ref.linear_step()
        g.transformPoints(); ref.transform_points()
    g.transformPoints(True); ref.transform_points(True)
    g.setupDeformableTransforms(1); ref.deformable_setup(1, _abi.FrogGridInfo())
    g.transformPoints(); ref.transform_points()
    g.updateStats(); ref.update_stats()
    for phase in range(2):
        for _ in range(3):
            assert g.updateDeformableTransforms(0.02) >= 0
            ref.deformable_step(0.02)
            g.transformPoints(); ref.transform_points()
        if phase == 0:
            g.transformPoints(True); ref.transform_points(True)       # the lattice stands, the positions under it moved
            g.transformPoints(); ref.transform_points()
    x, rx = g.points()[1], ref.xyz2()
    scale = float(np.max(np.abs(rx)))
    assert float(np.max(np.abs(x.astype(np.float64) - rx))) <= 1e-5 * scale


def _linear_run(pairs, n_it, **opt):
    g = ImageGroup(pairs, **opt)
    g.setupLinearTransforms(); g.transformPoints()
    es = []
    for it in range(n_it):
        if it % 10 == 0:
            g.updateStats()
        es.append(g.updateLinearTransforms()); g.transformPoints()
    mats = np.stack([g.matrix(i) for i in range(pairs.n_images)])
    stats = g.cull_stats_linear()
    return np.array(es), mats, g.points()[1].copy(), stats


@pytest.mark.parametrize("skin", ["1.25,10", "1.0,0"])
def test_linear_stage_with_the_zero_weight_list_equals_the_full_sweep(monkeypatch, skin):
    """updateLinearTransforms has no threshold (imageGroup.cxx:1100-1117), but a half-link whose weight is exactly zero
    adds nothing to the 18 sums: the linear sweep walks a list that leaves out the half-links whose distance puts the
    sweep's weight at exactly +0 (k_cull.hip.h cull_cutoff_linear_of).  40 linear iterations with the list (default skin,
    and a zero skin that invalidates the list at every step so that every sweep walks all records and rewrites it) against
    FROG_CULL_LINEAR=0.  Every left-out link would have added +-0.0, so the sums are the same real numbers; they are per-lane f64
    accumulators and compaction moves records between lanes, so equality holds up to f64 re-association, not by construction
    to the last bit: energies 1e-13, matrices 1e-12, coordinates one f32 ulp (whether they were in fact equal is reported);
    and the list really leaves links out once the images have come together."""
    pairs = Pairs.synthetic(8, 3000, 1200, seed=5)
    monkeypatch.setenv("FROG_CULL_LINEAR", "0")
    e0, m0, x0, st0 = _linear_run(pairs, 40)
    assert st0[0] == 0
    monkeypatch.setenv("FROG_CULL_LINEAR", "1")
    monkeypatch.setenv("FROG_CULL_SKIN_LINEAR", skin)
    e1, m1, x1, st1 = _linear_run(pairs, 40)
    assert np.max(np.abs(e0 - e1) / e0) <= 1e-13
    assert np.max(np.abs(m0 - m1)) <= 1e-12 * np.max(np.abs(m0))
    assert np.max(np.abs(x0.astype(np.float64) - x1)) <= 1.2e-7 * np.max(np.abs(x0))
    lists, listed, owned = st1
    same = bool(np.array_equal(e0, e1) and np.array_equal(m0, m1) and np.array_equal(x0, x1))
    note(f"linear_list_skin_{skin}", f"lists {lists} listed {listed} of {owned} half-links; identical bits: {same}")
    assert lists >= 4 and 0 < listed < owned


def test_config5_shaped_group_five_levels_against_the_oracle():
    """BASELINE.json configs[4] has 500 images; the oracle cannot follow that, so its deformable levels are only checked by
    properties there (tests/test_gpu_config5.py).  This is the same KIND of group at a size the oracle walks in seconds --
    40 images x 20 000 keypoints, ~20 partner images each, five levels, -gd 1 -- free-running on both sides: levels 3 and 4
    have under 24 points per brick of 4^3 cells and take the bricks of 8^3 cells (11^3-node tiles, the point-by-point
    B-spline transform), which the small parity cases never reach by themselves.  E series, guard decisions and lattices per
    level equal; displacement field and support-weighted coefficients 1e-4; raw coefficients reported."""
    pairs = Pairs.synthetic(40, 20000, 16667, seed=2, partners_per_image=20)
    assert pairs.n_half_links > 1.0e7
    r = free_run(pairs, 10, 5, 3, images=range(0, 40, 7))
    note("cfg5_shaped_free_run", f"E {r['E']:.2e} matrices {r['matrices']:.2e} grids {r['grids_per_level']} half_links {pairs.n_half_links}")
    for k, d in enumerate(r["lattices"]):
        note(f"cfg5_shaped_lattice_{k}", f"level {d['level']} raw {d['raw']:.2e} weighted {d['weighted']:.2e} field {d['field']:.2e} "
                                          f"weak_nodes {d['weak']}/{d['nodes']}")
    assert len(r["grids_per_level"]) == 5 and r["E"] < REL and r["matrices"] < 1e-6
    assert max(d["nodes"] for d in r["lattices"]) > 500000            # the fine lattice really is fine
    for k, d in enumerate(r["lattices"]):
        assert d["field"] <= REL and d["weighted"] <= REL and d["raw"] <= (1e-3 if d["level"] < 4 else 2e-3), (k, d)       # measured 3.3e-5 / 4.8e-4 (level 4)

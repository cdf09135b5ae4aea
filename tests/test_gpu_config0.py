"""BASELINE.json configs[0]: 4 keypoint files -> bin/match -> bin/frog at the reference's DEFAULT schedule (no schedule flag:
-li 50 -dl 3 -di 200, imageGroup.h:52-82), files compared with the oracle's run of the same pairs.bin.

The volumes -> surf3d half of the configuration cannot exist here (DummyVolumeGenerator writes an empty
lattice, the surf3d submodule is absent); the chain starts at what surf3d would have written: pointsK.csv.gz.

Two runs of bin/frog on the pairs.bin bin/match wrote:
  * `bin/frog -exact 1` (frog_options::reference_order; frog_amd/csrc/device/k_reforder.hip.h): every number in the files equals the oracle's --
    matrices, every coefficient of every lattice (compact .nii.gz sidecars, the default output form), energies as printed,
    histograms, census;
  * the product path: same lattices and guard decisions; deviations reported.  Four images of 900 keypoints are 1e4
    half-links: one link that crosses the inlier threshold an iteration earlier in one run than in the other is 1e-4 of the
    energy (profiles/r04_threshold_flip_small_group.txt), so the bars on this run are the loose ones (1e-3).
"""
import csv
import json
import os
import subprocess

import numpy as np
import pytest

from frog_amd import _abi
from frog_amd.match import synthetic_keypoints, write_keypoints, Keypoints
from frog_amd.pairs import Pairs
from oracle.oracle_api import OracleGroup
from nifti_util import read_nifti
from lattice_util import lattice_taps

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def note(name, value):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "test_numbers.txt"), "a") as fh:
            fh.write(f"{name} {value}\n")


def moved_keypoints(n_images, n_points, seed):
    """synthetic_keypoints, every image seen through its own anisotropic scale, translation and a smooth bump of a few mm: what
    the linear stage and the lattices are there to undo."""
    rng = np.random.default_rng(seed + 1000)
    out = []
    for kp in synthetic_keypoints(n_images, n_points, seed=seed):
        x = kp.xyz.astype(np.float64)
        bump = np.zeros_like(x)
        for _ in range(3):
            c, a = rng.uniform(50, 350, 3), rng.uniform(-6, 6, 3)
            bump += a * np.exp(-np.sum((x - c) ** 2, axis=1, keepdims=True) / (2 * 90.0 ** 2))
        y = (x + bump) * rng.uniform(0.9, 1.1, 3) + rng.uniform(-40, 40, 3)
        out.append(Keypoints(y.astype(np.float32), kp.scale, kp.laplacian, kp.response, kp.desc))
    return out


def read_transforms(d, n_images):
    out = []
    for i in range(n_images):
        t = json.load(open(d / "transforms" / f"{i}.json"))["transforms"]
        assert t[0]["type"] == "vtkMatrixToLinearTransform"
        lattices = []
        for k, entry in enumerate(t[1:]):
            assert entry["type"] == "vtkBSplineTransform" and set(entry) == {"type", "file"}      # compact form: the default
            h, vox = read_nifti(d / "transforms" / entry["file"])
            lattices.append((list(h["dim"][1:4]), np.array(h["qoffset"], np.float64), np.array(h["pixdim"][1:4], np.float64), vox))
        out.append((np.array(t[0]["matrix"]).reshape(4, 4), lattices))
    return out


def run_frog(d, flags=()):
    env = dict(os.environ)
    env.pop("FROG_REFERENCE_ORDER", None)
    r = subprocess.run([os.path.join(ROOT, "bin", "frog"), "pairs.bin", *flags], cwd=d, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_config0_keypoint_files_through_match_and_frog_at_the_default_schedule(tmp_path):
    n_images = 4
    imgs = moved_keypoints(n_images, 900, seed=21)
    names = []
    for i, kp in enumerate(imgs):
        p = tmp_path / f"points{i}.csv.gz"
        write_keypoints(p, kp)
        names.append(str(p))
    (tmp_path / "list.txt").write_text("".join(f"{n}\n" for n in names))
    r = subprocess.run([os.path.join(ROOT, "bin", "match"), "list.txt", "-o", "pairs.bin", "-d", "1"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Nb Match :" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    pairs = Pairs.read(str(tmp_path / "pairs.bin"))
    assert pairs.n_images == n_images and pairs.n_half_links > 2000

    # the oracle, the reference's defaults
    ref = OracleGroup(pairs.model, _abi.FrogOptions.default())
    E, grids_per_level = ref.run()
    assert len(E) == 650 and len(grids_per_level) == 3
    counts = ref.count_inliers((_abi.FrogCounts * n_images)())

    runs = {}
    # `-exact 1` = frog_options::reference_order: the mode a user asks bin/frog for, not a test hook
    for tag, flags in (("reference_order", ("-exact", "1")), ("product", ())):
        d = tmp_path / tag
        d.mkdir()
        os.link(tmp_path / "pairs.bin", d / "pairs.bin")
        out = run_frog(d, flags)
        assert "Linear registration" in out and "Total time" in out
        runs[tag] = d

    # ---- reference-order run: the oracle's numbers ----
    d = runs["reference_order"]
    got = read_transforms(d, n_images)
    for i, (m, lattices) in enumerate(got):
        assert np.array_equal(m, ref.matrix(i)), f"matrix of image {i}"
        assert len(lattices) == ref.num_grids() == sum(grids_per_level)
        for k, (dims, origin, spacing, vox) in enumerate(lattices):
            info, c = ref.grid(i, k, _abi.FrogGridInfo())
            assert dims == list(info.dims)
            assert np.array_equal(vox.astype(np.float32), c), f"lattice {k} of image {i}"
            np.testing.assert_allclose(spacing, list(info.spacing), rtol=1e-6)       # pixdim is f32 in a NIfTI-1 header
    rows = list(csv.reader(open(d / "measures.csv")))
    got_e = np.array([float(x[1]) for x in rows[1:]])
    assert len(got_e) == 650 and np.max(np.abs(got_e - E) / E) < 1e-5                 # six printed digits of the same number
    bbox = json.load(open(d / "bbox.json"))
    for i, rec in enumerate(bbox["images"]):
        assert (rec["points"], rec["pairs"], rec["inliers"], rec["outliers"]) == (counts[i].points, counts[i].pairs, counts[i].inliers, counts[i].outliers)
    hist = np.array([[float(v) for v in r_] for r_ in list(csv.reader(open(d / "histograms.csv")))[1:]])
    for i in range(n_images):
        h = ref.histogram(i)
        assert np.array_equal(hist[:len(h), i], h) and not hist[len(h):, i].any()

    # ---- product run: reported, loose bars (header) ----
    p = read_transforms(runs["product"], n_images)
    rows = list(csv.reader(open(runs["product"] / "measures.csv")))
    pe = np.array([float(x[1]) for x in rows[1:]])
    assert len(pe) == 650
    worst = {"E": float(np.max(np.abs(pe - E) / E)), "matrices": 0.0, "field": 0.0, "raw": 0.0}
    x = ref.xyz().astype(np.float64)
    lo, hi = x.min(axis=0), x.max(axis=0)
    pts = np.stack(np.meshgrid(*[np.linspace(lo[k], hi[k], 20) for k in range(3)], indexing="ij"), axis=-1).reshape(-1, 3)
    for i, (m, lattices) in enumerate(p):
        mr = ref.matrix(i)
        worst["matrices"] = max(worst["matrices"], relerr(np.diag(m)[:3], np.diag(mr)[:3]), relerr(m[:3, 3], mr[:3, 3]))
        assert len(lattices) == ref.num_grids(), "the guard decided differently"
        for k, (dims, origin, spacing, vox) in enumerate(lattices):
            info, c = ref.grid(i, k, _abi.FrogGridInfo())
            assert dims == list(info.dims)
            idx, wt = lattice_taps(pts, info)            # dense lattice over the group's box: what a resampler evaluates
            da = np.einsum("nt,ntk->nk", wt, vox.astype(np.float64)[idx]); db = np.einsum("nt,ntk->nk", wt, c.astype(np.float64)[idx])
            worst["field"] = max(worst["field"], float(np.max(np.abs(da - db))) / max(float(np.max(np.abs(db))), 1e-30))
            worst["raw"] = max(worst["raw"], relerr(vox, c))
    note("config0_product_vs_oracle", " ".join(f"{a} {b:.2e}" for a, b in worst.items()) + f" grids {grids_per_level} half_links {pairs.n_half_links}")
    # measured: E 4.4e-6 (print precision), matrices 6.8e-7, dense field 1.7e-5, raw 1.1e-4; the E / field bars leave room for
    # one threshold flip on 1e4 half-links (header)
    assert worst["E"] < 1e-3 and worst["matrices"] < 1e-5 and worst["field"] < 1e-3 and worst["raw"] < 2e-3

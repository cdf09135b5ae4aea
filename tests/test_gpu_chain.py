"""HIP transform-chain evaluation / Jacobian check (include/frog_chain.h) vs the oracle."""
import os
import subprocess

import numpy as np
import pytest

from frog_amd.chain import Chain, Link, read_transform
from oracle.oracle_api import chain_apply, chain_check
from test_chain import linear_lattice

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def random_chain(rng, n_lattices, amplitude):
    M = np.eye(4); M[:3, :3] = np.diag(rng.uniform(0.8, 1.2, 3)); M[:3, 3] = rng.uniform(-5, 5, 3)
    links = [Link.linear(M)]
    for k in range(n_lattices):
        n = 6 * 2 ** k
        dims = (n + 3, n + 3, n + 4)
        sp = tuple(120.0 / n for _ in range(3))
        links.append(Link.bspline(dims, tuple(-10.0 - s for s in sp), sp,
                                  (amplitude * rng.normal(size=(dims[0] * dims[1] * dims[2], 3))).astype(np.float32)))
    return links


@pytest.mark.parametrize("amplitude", [0.5, 20.0])
def test_apply_and_check_match_oracle(amplitude):
    rng = np.random.default_rng(7)
    links = random_chain(rng, 3, amplitude)
    pts = rng.uniform(-30, 140, (5000, 3))                  # some outside the lattices
    c = Chain(links)
    got, want = c.apply(pts), chain_apply(links, pts)
    assert np.max(np.abs(got - want)) < 1e-9 * max(1.0, np.max(np.abs(want)))
    grid = ((-5.0, -5.0, -5.0), (2.5, 2.5, 2.5), (41, 40, 39))
    n, m = c.check(*grid)
    rn, rm = chain_check(links, *grid)
    # determinants within rounding of zero may fall on either side: allow a handful
    assert abs(n - rn) <= 2 and abs(m - rm) < 1e-9 * max(1.0, abs(rm))
    if amplitude > 1:
        assert n > 100                                       # a wild lattice folds space
    else:
        assert n == 0 and m > 0


def test_closed_form_cases():
    A = np.array([[0.25, 0, 0], [0, 0, -0.5], [0.1, 0.2, 0]])
    L = linear_lattice((9, 8, 10), (-12.0, -10.0, -15.0), (5.0, 4.0, 6.0), A, [1.0, -2.0, 0.5])
    pts = np.random.default_rng(2).uniform(-2, 10, (300, 3))
    assert np.allclose(Chain([L]).apply(pts), pts + pts @ A.T + [1.0, -2.0, 0.5], atol=1e-5)
    n, m = Chain([L]).check((-2, -2, -2), (1, 1, 1), (12, 12, 12))
    assert n == 0 and abs(m - np.linalg.det(np.eye(3) + A)) < 1e-6
    assert np.array_equal(Chain([]).apply(pts), pts)
    with pytest.raises(RuntimeError):
        Chain([Link.bspline((2, 2, 2), (0, 0, 0), (0, 1, 1), np.zeros((8, 3), np.float32))])


def test_chain_written_by_frog_in_both_forms(tmp_path, small_pairs):
    # the chain frog writes (compact sidecars / single JSON) read back and applied: the two forms agree,
    # the diffeomorphism guard (-gd 1) leaves no negative Jacobian on the lattice's own grid, and the
    # chain moves the original keypoints of image i like the solver did
    out = {}
    for sub, extra in (("compact", []), ("single", ["-j"])):
        d = tmp_path / sub
        d.mkdir()
        small_pairs.write(d / "pairs.bin")
        r = subprocess.run([os.path.join(ROOT, "bin", "frog"), "pairs.bin", "-li", "12", "-dl", "2", "-di", "8", "-q", "1"] + extra,
                           cwd=d, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out[sub] = [read_transform(d / "transforms" / f"{i}.json") for i in range(small_pairs.n_images)]
    po = small_pairs.point_offset
    for i in range(small_pairs.n_images):
        links = out["compact"][i]
        assert len(links) == len(out["single"][i]) >= 3
        pts = small_pairs.xyz[po[i]:po[i + 1]].astype(np.float64)
        a, b = Chain(links).apply(pts), Chain(out["single"][i]).apply(pts)
        assert np.max(np.abs(a - b)) < 1e-3                  # two runs: float-atomic order of the lattice only
        assert np.max(np.abs(a - chain_apply(links, pts))) < 1e-9 * np.max(np.abs(a))
        last = links[-1]
        n, m = Chain(links).check(tuple(o + s for o, s in zip(last.origin, last.spacing)), last.spacing,
                                  tuple(d - 3 for d in last.dims))
        assert n == 0 and m > 0


def test_tools_points_transform_and_check_diffeomorphism(tmp_path):
    import ctypes as C
    import json
    from frog_amd import _abi
    lib = _abi.host_lib()
    A = np.diag([-1.5, 0.0, 0.0])
    fold = linear_lattice((10, 10, 10), (-20.0, -20.0, -20.0), (5.0, 5.0, 5.0), A, [0, 0, 0])
    M = np.eye(4); M[:3, 3] = [1.0, 2.0, 3.0]
    (tmp_path / "t.json").write_text(json.dumps({"transforms": [
        {"type": "vtkMatrixToLinearTransform", "matrix": M.ravel().tolist()},
        {"type": "vtkBSplineTransform", "dimensions": list(fold.dims), "origin": list(fold.origin), "spacing": list(fold.spacing),
         "coeffs": fold.coeffs.ravel().tolist()}]}))
    r = subprocess.run([os.path.join(ROOT, "bin", "PointsTransform"), "-p", "1", "1", "1", "-t", "t.json"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "Input point : 1 1 1" in r.stdout, r.stdout + r.stderr
    got = [float(v) for v in r.stdout.split("Output point :")[1].split()[:3]]
    assert np.allclose(got, chain_apply(read_transform(tmp_path / "t.json"), [[1, 1, 1]])[0], atol=1e-4)
    assert np.allclose(got, [2 - 1.5 * 2, 3, 4], atol=1e-4)          # translate, then x - 1.5 x
    # the sampling grid comes from a volume header
    d = (C.c_uint32 * 3)(12, 11, 10); s = (C.c_double * 3)(1.0, 1.0, 1.0); o = (C.c_double * 3)(-6.0, -6.0, -6.0)
    vox = np.zeros(12 * 11 * 10, np.float32)
    assert lib.frog_nifti_write(str(tmp_path / "vol.nii.gz").encode(), d, s, o, 1, vox.ctypes.data_as(_abi.c_float_p)) == 0
    exe = os.path.join(ROOT, "bin", "CheckDiffeomorphism")
    r = subprocess.run([exe, "vol.nii.gz", "t.json"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and f"{12 * 11 * 10} negative jacobian determinant values (100%)" in r.stdout, r.stdout
    r = subprocess.run([exe, "vol.nii.gz", "t.json", "2"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "Resizing image with spacing : 2" in r.stdout and f"{6 * 6 * 5} negative" in r.stdout, r.stdout
    (tmp_path / "id.json").write_text(json.dumps({"transforms": [{"type": "vtkMatrixToLinearTransform", "matrix": np.eye(4).ravel().tolist()}]}))
    r = subprocess.run([exe, "vol.nii.gz", "id.json"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "0 negative jacobian determinant values (0%)" in r.stdout


def test_inverse_and_reslice_match_oracle():
    """The inverse chain (VolumeTransform's -t) and vtkImageReslice's sampling: device against the oracle."""
    from frog_amd.chain import invert
    from oracle.oracle_api import chain_reslice
    from test_chain import smooth_chain
    links = smooth_chain()
    inv = invert(links)
    assert [l.kind for l in inv] == [2, 0] and np.allclose(inv[1].matrix @ links[0].matrix, np.eye(4), atol=1e-12)
    pts = np.random.default_rng(4).uniform(-10, 90, (4000, 3))
    c = Chain(inv)
    got, want = c.apply(pts), chain_apply(inv, pts)
    assert np.abs(got - want).max() < 1e-6                        # two Newton iterations with different 3x3 solvers
    assert np.abs(Chain(links).apply(got) - pts).max() < 2e-3
    n, m = c.check((0.0, 0.0, 0.0), (4.0, 4.0, 4.0), (20, 20, 20))
    rn, rm = chain_check(inv, (0.0, 0.0, 0.0), (4.0, 4.0, 4.0), (20, 20, 20))
    assert n == rn == 0 and abs(m - rm) < 1e-6
    rng = np.random.default_rng(8)
    o, s = (-5.0, 0.0, 2.0), (1.5, 2.0, 1.0)
    for dtype, bg in (("float32", -1.0), ("int16", -1000.0), ("uint8", 0.0), ("float64", 3.5), ("int32", 7.0)):
        vol = rng.uniform(0, 200, (40, 30, 50)).astype(dtype)
        for mode in (0, 1):
            got = c.reslice(vol, o, s, (33, 35, 31), (0.0, 2.0, 4.0), (2.0, 1.5, 1.2), mode, bg)
            want = chain_reslice(inv, vol, o, s, (33, 35, 31), (0.0, 2.0, 4.0), (2.0, 1.5, 1.2), mode, bg)
            assert got.dtype == vol.dtype and got.shape == (31, 35, 33)
            if vol.dtype.kind == "f":
                assert np.allclose(got, want, rtol=0, atol=2e-3 if mode else 0)       # 1e-6 mm x gradient <= 200 / mm
                if mode == 0:
                    assert (got != want).mean() < 1e-3                                  # a sample on a voxel boundary
            else:
                r = np.clip(np.floor(want + 0.5), np.iinfo(vol.dtype).min, np.iinfo(vol.dtype).max)
                assert np.abs(got.astype(np.float64) - r).max() <= 1 and (got != r).mean() < 1e-3
            assert (got == np.array(bg).astype(vol.dtype)).any() and (got != np.array(bg).astype(vol.dtype)).any()


def _write_chain(path, links):
    import json
    out = []
    for l in links:
        if l.kind == 0:
            out.append({"type": "vtkMatrixToLinearTransform", "matrix": l.matrix.ravel().tolist()})
        else:
            out.append({"type": "vtkBSplineTransform", "dimensions": list(l.dims), "origin": list(l.origin), "spacing": list(l.spacing),
                        "coeffs": l.coeffs.ravel().tolist()})
    path.write_text(json.dumps({"transforms": out}))


def test_tools_volume_transform_and_inverse_points(tmp_path):
    """bin/VolumeTransform (tools/VolumeTransform.cxx) end to end: volumes from files, -t inverted, output on the
    reference volume's grid in the source's scalar type; bin/PointsTransform -ti."""
    from frog_amd.chain import invert
    from frog_amd.volume import read_volume, write_volume
    from oracle.oracle_api import chain_reslice
    from test_chain import smooth_chain
    links = smooth_chain()
    _write_chain(tmp_path / "t.json", links)
    z, y, x = np.meshgrid(np.arange(40), np.arange(48), np.arange(56), indexing="ij")
    src = (1000 + 400 * np.sin(x / 6.0) * np.cos(y / 7.0) + 10 * z).astype(np.int16)
    so, ss = (-4.0, -2.0, 0.0), (1.5, 1.5, 2.0)
    write_volume(tmp_path / "src.nii.gz", src, so, ss)
    ro, rs, rd = (0.0, 1.0, 2.0), (2.0, 2.0, 2.5), (36, 30, 28)
    write_volume(tmp_path / "ref.mhd", np.zeros(rd[::-1], np.uint8), ro, rs)
    exe = os.path.join(ROOT, "bin", "VolumeTransform")
    inv = invert(links)
    for args, name, mode, bg in ((["-o", "out.nii.gz"], "out.nii.gz", 1, float(src.min())),
                                 (["-i", "0", "-b", "-5", "-o", "near.mhd"], "near.mhd", 0, -5.0),
                                 ([], "output.mhd", 1, float(src.min()))):
        r = subprocess.run([exe, "src.nii.gz", "ref.mhd", "-t", "t.json"] + args, cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "Transform computed in" in r.stdout and "transformed center :" in r.stdout, r.stdout + r.stderr
        got, o, s = read_volume(tmp_path / name)
        assert got.dtype == np.int16 and got.shape == rd[::-1] and o == ro and s == rs
        want = np.clip(np.floor(chain_reslice(inv, src, so, ss, rd, ro, rs, mode, bg) + 0.5), -32768, 32767)
        assert np.abs(got - want).max() <= 1 and (got != want).mean() < 2e-3
        assert (got == bg).any() and (got != bg).mean() > 0.5
    # -rx mirrors the voxels along x; -ti takes the chain as it is
    r = subprocess.run([exe, "src.nii.gz", "ref.mhd", "-ti", "t.json", "-rx", "1", "-o", "fwd.nii.gz"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    got, _, _ = read_volume(tmp_path / "fwd.nii.gz")
    want = np.clip(np.floor(chain_reslice(links, src, so, ss, rd, ro, rs, 1, float(src.min())) + 0.5), -32768, 32767)
    assert np.abs(got[:, :, ::-1] - want).max() <= 1
    # resampling a volume through T^-1 and looking a point up through T agree: voxel p of the output shows source(T^-1(p))
    exe = os.path.join(ROOT, "bin", "PointsTransform")
    p = np.array([20.0, 15.0, 30.0])
    r = subprocess.run([exe, "-p", *[repr(float(v)) for v in p], "-ti", "t.json"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    back = np.array([float(v) for v in r.stdout.split("Output point :")[1].split()[:3]])
    assert np.abs(chain_apply(links, [back])[0] - p).max() < 2e-3
    # two files: the outer transform is PreMultiply (VTK's default), the last one given acts first
    M = np.eye(4); M[:3, 3] = [10.0, 0.0, 0.0]
    S = np.diag([2.0, 1.0, 1.0, 1.0])
    from frog_amd.chain import Link
    _write_chain(tmp_path / "m.json", [Link.linear(M)]); _write_chain(tmp_path / "s.json", [Link.linear(S)])
    r = subprocess.run([exe, "-p", "1", "1", "1", "-t", "m.json", "-t", "s.json"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert [float(v) for v in r.stdout.split("Output point :")[1].split()[:3]] == [12.0, 1.0, 1.0]     # scale, then translate
    assert subprocess.run([os.path.join(ROOT, "bin", "VolumeTransform"), "src.nii.gz"], cwd=tmp_path, capture_output=True).returncode == 1

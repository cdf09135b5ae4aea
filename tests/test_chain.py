"""Transform-chain oracle (oracle/chain_oracle.cpp) on closed-form cases, and the readers of
transforms/<i>.json (+ .nii.gz sidecars).  VTK is absent, so these cases are what pins the
semantics: PostMultiply order, cubic B-splines reproduce linear functions exactly, Jacobian =
I + gradient of the displacement."""
import ctypes as C
import json

import numpy as np
import pytest

from frog_amd import _abi
from frog_amd.chain import Link, read_transform
from oracle.oracle_api import chain_apply, chain_check


def linear_lattice(dims, origin, spacing, A, b):
    """Coefficients c(x_cp) = A x_cp + b: the spline displacement is then exactly A x + b."""
    g = np.stack(np.meshgrid(*[origin[k] + spacing[k] * np.arange(dims[k]) for k in range(3)], indexing="ij"), -1)
    c = (g @ np.asarray(A, np.float64).T + np.asarray(b, np.float64)).astype(np.float32)
    return Link.bspline(dims, origin, spacing, c.transpose(2, 1, 0, 3).reshape(-1, 3))       # x fastest


def test_identity_linear_and_order():
    pts = np.array([[1.0, 2.0, 3.0], [-4.0, 0.5, 7.0]])
    assert np.array_equal(chain_apply([], pts), pts)
    M = np.diag([2.0, 3.0, 0.5, 1.0]); M[:3, 3] = [1, 2, 3]
    T = np.eye(4); T[:3, 3] = [10, 0, 0]
    out, J = chain_apply([Link.linear(M), Link.linear(T)], pts, jacobian=True)      # listed order: M first, then T
    assert np.allclose(out, pts * [2, 3, 0.5] + [11, 2, 3]) and np.allclose(np.linalg.det(J), 3.0)
    out2 = chain_apply([Link.linear(T), Link.linear(M)], pts)
    assert np.allclose(out2, (pts + [10, 0, 0]) * [2, 3, 0.5] + [1, 2, 3])


def test_bspline_reproduces_linear_displacements_and_their_gradient():
    A = np.array([[0.25, 0.0, 0.0], [0.0, 0.0, -0.5], [0.1, 0.2, 0.0]])
    b = np.array([1.0, -2.0, 0.5])
    L = linear_lattice((9, 8, 10), (-12.0, -10.0, -15.0), (5.0, 4.0, 6.0), A, b)
    rng = np.random.default_rng(1)
    pts = rng.uniform(-2, 10, (200, 3))                      # well inside: all 64 taps exist
    out, J = chain_apply([L], pts, jacobian=True)
    assert np.allclose(out, pts + pts @ A.T + b, atol=1e-5)
    assert np.allclose(J, np.eye(3) + A, atol=1e-6)
    # outside the lattice every tap is missing: BorderModeZero -> identity
    far = np.array([[1e4, 1e4, 1e4]])
    out, J = chain_apply([L], far, jacobian=True)
    assert np.array_equal(out, far) and np.array_equal(J[0], np.eye(3))


def test_negative_determinants_are_counted():
    # displacement -1.5 x along x: Jacobian diag(-0.5, 1, 1) inside the lattice
    L = linear_lattice((10, 10, 10), (-20.0, -20.0, -20.0), (5.0, 5.0, 5.0), np.diag([-1.5, 0, 0]), [0, 0, 0])
    n, m = chain_check([L], (-5, -5, -5), (1, 1, 1), (11, 11, 11))
    assert n == 11 ** 3 and abs(m + 0.5) < 1e-6
    n, m = chain_check([Link.linear(np.eye(4))], (0, 0, 0), (1, 1, 1), (4, 5, 6))
    assert n == 0 and m == 1.0
    n, m = chain_check([Link.linear(np.diag([1.0, -1.0, 1.0, 1.0]))], (0, 0, 0), (1, 1, 1), (4, 5, 6))
    assert n == 120 and m == -1.0


@pytest.mark.parametrize("compact", [False, True])
def test_transform_reader_both_forms(tmp_path, compact):
    # what frog writes (tools/transformIO.h:163-258) -> what its readers rebuild (:375-460)
    lib = _abi.host_lib()
    rng = np.random.default_rng(4)
    M = np.eye(4); M[0, 0] = 1.1; M[:3, 3] = [3, -2, 1]
    dims, origin, spacing = (5, 6, 4), (-10.0, 2.5, 0.0), (12.5, 10.0, 20.0)
    coeffs = rng.normal(size=(5 * 6 * 4, 3)).astype(np.float32)
    t = [{"type": "vtkMatrixToLinearTransform", "matrix": M.ravel().tolist()}]
    if compact:
        d = (C.c_uint32 * 3)(*dims); s = (C.c_double * 3)(*spacing); o = (C.c_double * 3)(*origin)
        assert lib.frog_nifti_write(str(tmp_path / "7.json.0.nii.gz").encode(), d, s, o, 3, coeffs.ctypes.data_as(_abi.c_float_p)) == 0
        t.append({"type": "vtkBSplineTransform", "file": "7.json.0.nii.gz"})
    else:
        t.append({"type": "vtkBSplineTransform", "dimensions": list(dims), "origin": list(origin), "spacing": list(spacing),
                  "coeffs": coeffs.ravel().tolist()})
    (tmp_path / "7.json").write_text(json.dumps({"transforms": t}))
    links = read_transform(tmp_path / "7.json")
    assert [l.kind for l in links] == [0, 1] and np.array_equal(links[0].matrix, M)
    assert links[1].dims == dims and np.allclose(links[1].origin, origin) and np.allclose(links[1].spacing, spacing)
    assert np.array_equal(links[1].coeffs, coeffs)
    (tmp_path / "bad.json").write_text(json.dumps({"transforms": [{"type": "vtkThinPlateSplineTransform"}]}))
    with pytest.raises(ValueError):
        read_transform(tmp_path / "bad.json")


def test_native_transform_reader_matches_python_reader(tmp_path):
    # libfrog_host's reader (the tools use it) against frog_amd.chain.read_transform
    lib = _abi.host_lib()
    rng = np.random.default_rng(9)
    dims, origin, spacing = (4, 5, 6), (1.5, -2.0, 3.25), (10.0, 12.5, 8.0)
    c0 = rng.normal(size=(120, 3)).astype(np.float32); c1 = rng.normal(size=(120, 3)).astype(np.float32)
    d = (C.c_uint32 * 3)(*dims); s = (C.c_double * 3)(*spacing); o = (C.c_double * 3)(*origin)
    assert lib.frog_nifti_write(str(tmp_path / "3.json.1.nii.gz").encode(), d, s, o, 3, c1.ctypes.data_as(_abi.c_float_p)) == 0
    M = np.arange(16, dtype=np.float64).reshape(4, 4) / 7
    (tmp_path / "3.json").write_text(json.dumps({"transforms": [
        {"type": "vtkMatrixToLinearTransform", "matrix": M.ravel().tolist()},
        {"type": "vtkBSplineTransform", "dimensions": list(dims), "origin": list(origin), "spacing": list(spacing),
         "coeffs": c0.ravel().tolist()},
        {"type": "vtkBSplineTransform", "file": "3.json.1.nii.gz"}]}))
    status = C.c_int()
    h = lib.frog_transform_read(str(tmp_path / "3.json").encode(), C.byref(status))
    assert h and status.value == 0 and lib.frog_transform_num_links(h) == 3
    links = lib.frog_transform_links(h)
    want = read_transform(tmp_path / "3.json")
    assert links[0].type == 0 and np.allclose(list(links[0].matrix), M.ravel())
    for k, co in ((1, c0), (2, c1)):
        assert links[k].type == 1 and tuple(links[k].dims) == dims
        assert np.allclose(list(links[k].origin), origin) and np.allclose(list(links[k].spacing), spacing)
        assert np.array_equal(np.ctypeslib.as_array(links[k].coeffs, shape=(120, 3)), co)
        assert np.array_equal(want[k].coeffs, co)
    lib.frog_transform_free(h)
    assert not lib.frog_transform_read(str(tmp_path / "missing.json").encode(), C.byref(status)) and status.value == _abi.FROG_E_IO
    (tmp_path / "bad.json").write_text(json.dumps({"transforms": [{"type": "vtkThinPlateSplineTransform"}]}))
    assert not lib.frog_transform_read(str(tmp_path / "bad.json").encode(), C.byref(status)) and status.value == _abi.FROG_E_INVALID
    # voxel grids: NIfTI and MetaImage headers
    dd, ss, oo = (C.c_uint32 * 3)(), (C.c_double * 3)(), (C.c_double * 3)()
    assert lib.frog_volume_geometry(str(tmp_path / "3.json.1.nii.gz").encode(), dd, ss, oo) == 0
    assert tuple(dd) == dims and np.allclose(list(ss), spacing) and np.allclose(list(oo), origin)
    (tmp_path / "v.mhd").write_text("ObjectType = Image\nNDims = 3\nDimSize = 7 8 9\nElementSpacing = 0.5 0.75 2\nOffset = -1 2 3.5\n")
    assert lib.frog_volume_geometry(str(tmp_path / "v.mhd").encode(), dd, ss, oo) == 0
    assert tuple(dd) == (7, 8, 9) and list(ss) == [0.5, 0.75, 2.0] and list(oo) == [-1.0, 2.0, 3.5]


def smooth_chain(seed=3, amplitude=2.0):
    rng = np.random.default_rng(seed)
    M = np.array([[1.05, 0.08, 0, 4], [-0.06, 0.95, 0.03, -3], [0.02, 0, 1.1, 2], [0, 0, 0, 1.0]])
    dims = (9, 8, 10)
    co = (amplitude * rng.normal(size=(dims[0] * dims[1] * dims[2], 3))).astype(np.float32)
    return [Link.linear(M), Link.bspline(dims, (-25.0, -25.0, -25.0), (15.0, 15.0, 15.0), co)]


def test_inverse_links_undo_the_chain():
    """vtkGeneralTransform::Inverse (VolumeTransform.cxx:55-57): reversed order, inverted matrix, Newton on the
    lattice to VTK's default tolerance of 1e-3."""
    from frog_amd.chain import BSPLINE_INVERSE
    links = smooth_chain()
    inv = [Link(BSPLINE_INVERSE, dims=links[1].dims, origin=links[1].origin, spacing=links[1].spacing, coeffs=links[1].coeffs),
           Link.linear(np.linalg.inv(links[0].matrix))]
    pts = np.random.default_rng(1).uniform(-10, 90, (2000, 3))
    fwd, J = chain_apply(links, pts, jacobian=True)
    back, Ji = chain_apply(inv, fwd, jacobian=True)
    assert np.abs(back - pts).max() < 2e-3                       # the tolerance VTK iterates to
    assert np.abs(np.einsum("nij,njk->nik", Ji, J) - np.eye(3)).max() < 1e-4
    # inverse then forward as well
    assert np.abs(chain_apply(links, chain_apply(inv, pts)) - pts).max() < 2e-3


def test_reslice_oracle_closed_forms():
    """vtkImageReslice as VolumeTransform.cxx:119-136 sets it up."""
    from oracle.oracle_api import chain_reslice
    rng = np.random.default_rng(5)
    vol = rng.uniform(0, 100, (6, 7, 8))                           # [z, y, x]
    o, s = (10.0, -5.0, 3.0), (2.0, 1.5, 3.0)
    # identity chain, same grid: the volume itself, in both modes
    for mode in (0, 1):
        assert np.array_equal(chain_reslice([], vol, o, s, (8, 7, 6), o, s, mode), vol)
    # the chain maps output space to source space: +1 voxel in x reads the neighbour; beyond the half-voxel
    # border the background
    T = np.eye(4); T[0, 3] = s[0]
    out = chain_reslice([Link.linear(T)], vol, o, s, (8, 7, 6), o, s, 1, background=-7.0)
    assert np.allclose(out[:, :, :7], vol[:, :, 1:]) and np.all(out[:, :, 7] == -7.0)
    T[0, 3] = 0.4 * s[0]                                           # 0.4 voxel: the last column is inside the border
    out = chain_reslice([Link.linear(T)], vol, o, s, (8, 7, 6), o, s, 1, background=-7.0)
    assert np.allclose(out[:, :, :7], 0.6 * vol[:, :, :7] + 0.4 * vol[:, :, 1:]) and np.allclose(out[:, :, 7], vol[:, :, 7])
    assert np.array_equal(chain_reslice([Link.linear(T)], vol, o, s, (8, 7, 6), o, s, 0), vol)      # nearest: 0.4 rounds back
    # a linear ramp is reproduced exactly by trilinear interpolation under any affine map (inside the volume)
    z, y, x = np.meshgrid(np.arange(6), np.arange(7), np.arange(8), indexing="ij")
    ramp = 2.0 * (o[0] + s[0] * x) - 1.0 * (o[1] + s[1] * y) + 0.5 * (o[2] + s[2] * z)
    A = np.array([[0.9, 0.1, 0, 1.0], [0, 1.05, 0.05, 0.2], [0.02, 0, 0.95, 0.5], [0, 0, 0, 1]])
    oo, os_, od = (13.0, -3.0, 6.0), (1.0, 1.0, 2.0), (5, 4, 3)
    out = chain_reslice([Link.linear(A)], ramp, o, s, od, oo, os_, 1, background=np.nan)
    zz, yy, xx = np.meshgrid(np.arange(3), np.arange(4), np.arange(5), indexing="ij")
    P = np.stack([oo[0] + os_[0] * xx, oo[1] + os_[1] * yy, oo[2] + os_[2] * zz, np.ones_like(xx)], -1) @ A.T
    assert not np.isnan(out).any() and np.allclose(out, 2.0 * P[..., 0] - P[..., 1] + 0.5 * P[..., 2])


def test_invert_links_is_host_side_and_involutive():
    """frog_chain_invert_links (vtkGeneralTransform::Inverse, VolumeTransform.cxx:55-57) needs no device: reversed
    order, inverted matrices, lattices switched between forward and Newton evaluation; twice = the chain itself."""
    from frog_amd.chain import BSPLINE, BSPLINE_INVERSE, LINEAR, invert
    links = smooth_chain()
    inv = invert(links)
    assert [l.kind for l in inv] == [BSPLINE_INVERSE, LINEAR]
    assert np.allclose(inv[1].matrix @ links[0].matrix, np.eye(4), atol=1e-13)
    assert inv[0].dims == links[1].dims and np.array_equal(inv[0].coeffs, links[1].coeffs)
    back = invert(inv)
    assert [l.kind for l in back] == [LINEAR, BSPLINE] and np.allclose(back[0].matrix, links[0].matrix, atol=1e-13)
    pts = np.random.default_rng(3).uniform(0, 80, (500, 3))
    assert np.abs(chain_apply(inv, chain_apply(links, pts)) - pts).max() < 2e-3
    singular = [Link.linear(np.diag([1.0, 0.0, 1.0, 1.0]))]
    with pytest.raises(RuntimeError):
        invert(singular)
    assert invert([]) == []

"""Transform chains (what frog writes to transforms/<i>.json) applied and checked on the GPU --
ctypes layer over include/frog_chain.h (tools/PointsTransform.cxx, tools/CheckDiffeomorphism.cxx)."""
import ctypes as C
import gzip
import json
import os
import struct

import numpy as np

from . import _abi
from ._abi import check

LINEAR, BSPLINE, BSPLINE_INVERSE = 0, 1, 2


class Link:
    """One transform of a chain: Link.linear(matrix4x4) or Link.bspline(dims, origin, spacing, coeffs[G,3])."""

    def __init__(self, kind, matrix=None, dims=None, origin=None, spacing=None, coeffs=None):
        self.kind = kind
        self.matrix = None if matrix is None else np.ascontiguousarray(matrix, np.float64).reshape(4, 4)
        self.dims = None if dims is None else tuple(int(d) for d in dims)
        self.origin = None if origin is None else tuple(float(v) for v in origin)
        self.spacing = None if spacing is None else tuple(float(v) for v in spacing)
        self.coeffs = None if coeffs is None else np.ascontiguousarray(coeffs, np.float32).reshape(-1, 3)
        if kind != LINEAR and len(self.coeffs) != self.dims[0] * self.dims[1] * self.dims[2]:
            raise ValueError("coefficient count does not match the lattice dimensions")

    @classmethod
    def linear(cls, matrix):
        return cls(LINEAR, matrix=matrix)

    @classmethod
    def bspline(cls, dims, origin, spacing, coeffs):
        return cls(BSPLINE, dims=dims, origin=origin, spacing=spacing, coeffs=coeffs)

    def view(self):
        v = _abi.FrogChainLink()
        v.type = self.kind
        if self.kind == LINEAR:
            v.matrix[:] = self.matrix.ravel().tolist()
        else:
            v.dims[:] = self.dims; v.origin[:] = self.origin; v.spacing[:] = self.spacing
            v.coeffs = self.coeffs.ctypes.data_as(_abi.c_float_p)
        return v


def invert(links):
    """vtkGeneralTransform::Inverse() of a chain: reversed order, inverted matrices, lattices evaluated by
    Newton's method (frog_chain_invert_links)."""
    links = list(links)
    n = len(links)
    src = (_abi.FrogChainLink * max(1, n))(*[l.view() for l in links])
    dst = (_abi.FrogChainLink * max(1, n))()
    check(_abi.hip_lib().frog_chain_invert_links(src, n, dst), "frog_chain_invert_links")
    out = []
    for k in range(n):
        o, v = links[n - 1 - k], dst[k]
        if v.type == LINEAR:
            out.append(Link.linear(np.array(v.matrix[:], np.float64).reshape(4, 4)))
        else:
            out.append(Link(v.type, dims=o.dims, origin=o.origin, spacing=o.spacing, coeffs=o.coeffs))
    return out


def read_nifti_lattice(path):
    """(dims, origin, spacing, voxels[G, components]) of a NIfTI-1 file as the reference's readers use it
    (tools/transformIO.h:439-453: spacing from pixdim, origin from the qform offsets)."""
    raw = open(path, "rb").read()
    if str(path).endswith(".gz"):
        raw = gzip.decompress(raw)
    dim = struct.unpack_from("<8h", raw, 40)
    datatype, = struct.unpack_from("<h", raw, 70)
    if datatype != 16:
        raise ValueError(f"{path}: only FLOAT32 lattices are supported")
    pixdim = struct.unpack_from("<8f", raw, 76)
    vox_offset, = struct.unpack_from("<f", raw, 108)
    qoffset = struct.unpack_from("<3f", raw, 268)
    nx, ny, nz = dim[1:4]
    nc = dim[5] if dim[0] >= 5 else 1
    data = np.frombuffer(raw, "<f4", count=nx * ny * nz * nc, offset=int(vox_offset))
    return (nx, ny, nz), qoffset, pixdim[1:4], data.reshape(nc, nx * ny * nz).T.copy()


def read_transform(path):
    """transforms/<i>.json in either form (tools/transformIO.h:375-460): a list of Links."""
    links = []
    for t in json.load(open(path))["transforms"]:
        if t["type"] == "vtkMatrixToLinearTransform":
            links.append(Link.linear(np.array(t["matrix"], np.float64).reshape(4, 4)))
        elif t["type"] == "vtkBSplineTransform":
            if "file" in t:
                dims, origin, spacing, vox = read_nifti_lattice(os.path.join(os.path.dirname(str(path)), t["file"]))
                links.append(Link.bspline(dims, origin, spacing, vox[:, :3]))
            else:
                links.append(Link.bspline(t["dimensions"], t["origin"], t["spacing"], np.array(t["coeffs"], np.float32).reshape(-1, 3)))
        else:
            raise ValueError(f"Error : transform type {t['type']} not supported")
    return links


class Chain:
    def __init__(self, links, device=0):
        self._lib = _abi.hip_lib()
        self.links = list(links)
        views = (_abi.FrogChainLink * max(1, len(self.links)))(*[l.view() for l in self.links])
        self._h = C.c_void_p()
        check(self._lib.frog_chain_create(views, len(self.links), device, C.byref(self._h)), "frog_chain_create")

    def close(self):
        if self._h:
            self._lib.frog_chain_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def apply(self, points):
        p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        out = np.empty_like(p)
        check(self._lib.frog_chain_apply(self._h, p.ctypes.data_as(_abi.c_double_p), out.ctypes.data_as(_abi.c_double_p), len(p)),
              "frog_chain_apply")
        return out

    def reslice(self, volume, origin, spacing, out_dims, out_origin, out_spacing, interpolation=1, background=0.0):
        """vtkImageReslice as tools/VolumeTransform.cxx:119-136 uses it: `volume` is indexed [z, y, x]; the chain
        maps the output grid's space to the volume's.  Returns an array of the same dtype, shape out_dims[::-1]."""
        src = np.ascontiguousarray(volume)
        if src.dtype.name not in _abi.FROG_V_DTYPES or src.ndim != 3:
            raise ValueError("3-D scalar volume of a supported type expected")
        out = np.empty(tuple(int(v) for v in out_dims[::-1]), src.dtype)
        a, b = _abi.FrogVolume(), _abi.FrogVolume()
        a.dims[:] = src.shape[::-1]; a.spacing[:] = spacing; a.origin[:] = origin
        b.dims[:] = [int(v) for v in out_dims]; b.spacing[:] = out_spacing; b.origin[:] = out_origin
        a.dtype = b.dtype = _abi.FROG_V_DTYPES.index(src.dtype.name)
        a.data = src.ctypes.data; b.data = out.ctypes.data
        check(self._lib.frog_chain_reslice(self._h, C.byref(a), C.byref(b), int(interpolation), float(background)), "frog_chain_reslice")
        return out

    def check(self, origin, spacing, dims):
        """(number of grid nodes with a negative Jacobian determinant, smallest determinant)."""
        o = (C.c_double * 3)(*origin); s = (C.c_double * 3)(*spacing); d = (C.c_uint32 * 3)(*dims)
        n, m = C.c_uint64(), C.c_double()
        check(self._lib.frog_chain_check(self._h, o, s, d, C.byref(n), C.byref(m)), "frog_chain_check")
        return int(n.value), float(m.value)

"""pairs.bin in memory (libfrog_host.so): reader, writer, synthetic groups.

Mirrors ImageGroup::readPairs (registration/imageGroup.cxx:1353-1417) and the
writer in match/match.cpp:675-744 of the reference.
"""
import ctypes as C

import numpy as np

from . import _abi


class Pairs:
    """Owns a frog_pairs handle; exposes the SoA/CSR model as numpy views."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("null frog_pairs handle")
        self._h = C.c_void_p(handle)
        self._lib = _abi.host_lib()
        self.model = _abi.FrogModel()
        self._lib.frog_pairs_model(self._h, C.byref(self.model))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.frog_pairs_free(h)

    # -- constructors -----------------------------------------------------------
    @classmethod
    def read(cls, path):
        lib = _abi.host_lib()
        st = C.c_int(0)
        h = lib.frog_pairs_read(str(path).encode(), C.byref(st))
        if not h:
            # the reference prints "Error : number of pairs is 0" and exit(1)s (imageGroup.cxx:1393-1398)
            raise ValueError(f"cannot read {path}: status {st.value}")
        return cls(h)

    @classmethod
    def synthetic(cls, n_images, points_per_image, pairs_per_block, seed=1, **kw):
        lib = _abi.host_lib()
        p = _abi.FrogSynthParams()
        lib.frog_synth_defaults(C.byref(p))
        p.n_images, p.points_per_image, p.pairs_per_block, p.seed = n_images, points_per_image, pairs_per_block, seed
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        h = lib.frog_synth_generate(C.byref(p))
        if not h:
            raise ValueError("bad synthetic parameters")
        return cls(h)

    @classmethod
    def from_arrays(cls, point_offset, xyz, blocks):
        """blocks: list of (image1, image2, p1 array, p2 array)."""
        lib = _abi.host_lib()
        po = np.ascontiguousarray(point_offset, dtype=np.uint32)
        x = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1)
        b1 = np.array([b[0] for b in blocks], dtype=np.uint16)
        b2 = np.array([b[1] for b in blocks], dtype=np.uint16)
        ptr = np.zeros(len(blocks) + 1, dtype=np.uint64)
        for i, b in enumerate(blocks):
            ptr[i + 1] = ptr[i] + len(b[2])
        p1 = np.concatenate([np.asarray(b[2], dtype=np.uint32) for b in blocks]) if blocks else np.zeros(0, np.uint32)
        p2 = np.concatenate([np.asarray(b[3], dtype=np.uint32) for b in blocks]) if blocks else np.zeros(0, np.uint32)
        h = lib.frog_pairs_from_arrays(
            len(po) - 1, po.ctypes.data_as(_abi.c_u32_p), x.ctypes.data_as(_abi.c_float_p), None,
            len(blocks), b1.ctypes.data_as(C.POINTER(C.c_uint16)), b2.ctypes.data_as(C.POINTER(C.c_uint16)),
            ptr.ctypes.data_as(C.POINTER(C.c_uint64)), p1.ctypes.data_as(_abi.c_u32_p), p2.ctypes.data_as(_abi.c_u32_p))
        if not h:
            raise ValueError("pair indices out of range")
        return cls(h)

    def write(self, path):
        rc = self._lib.frog_pairs_write(self._h, str(path).encode())
        if rc:
            raise IOError(f"cannot write {path}")

    # -- sizes / views ------------------------------------------------------------
    @property
    def n_images(self):
        return self._lib.frog_pairs_num_images(self._h)

    @property
    def n_points(self):
        return self._lib.frog_pairs_num_points(self._h)

    @property
    def n_pairs(self):
        return self._lib.frog_pairs_num_pairs(self._h)

    @property
    def n_half_links(self):
        return 2 * self.n_pairs

    @property
    def n_blocks(self):
        return self._lib.frog_pairs_num_blocks(self._h)

    def _view(self, ptr, ctype, n):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(int(n),))

    @property
    def point_offset(self):
        return self._view(self.model.point_offset, C.c_uint32, self.n_images + 1)

    @property
    def xyz(self):
        return self._view(self.model.xyz, C.c_float, 3 * self.n_points).reshape(-1, 3)

    @property
    def row_ptr(self):
        return self._view(self.model.row_ptr, C.c_uint64, self.n_points + 1)

    @property
    def link_image(self):
        return self._view(self.model.link_image, C.c_uint16, self.n_half_links)

    @property
    def link_point(self):
        return self._view(self.model.link_point, C.c_uint32, self.n_half_links)

    def set_points(self, image, xyz):
        """Move all keypoints of a (fixed) image to their registered position (imageGroup.cxx:1445-1450)."""
        a = np.ascontiguousarray(xyz, np.float32)
        if a.shape != (int(self.point_offset[image + 1]) - int(self.point_offset[image]), 3):
            raise ValueError("one xyz triple per point of the image")
        if self._lib.frog_pairs_set_points(self._h, image, a.ctypes.data_as(_abi.c_float_p)):
            raise ValueError("frog_pairs_set_points")
        self._lib.frog_pairs_model(self._h, C.byref(self.model))

    def append_points(self, image, xyz):
        """Extra link-less points at the end of `image` (how the reference stores landmarks,
        imageGroup.cxx:1185-1201); the model views are refreshed."""
        a = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        rc = self._lib.frog_pairs_append_points(self._h, image, a.ctypes.data_as(_abi.c_float_p), len(a))
        if rc:
            raise ValueError(f"frog_pairs_append_points failed ({rc})")
        self._lib.frog_pairs_model(self._h, C.byref(self.model))

    def block(self, b):
        i1, i2, n = C.c_uint16(), C.c_uint16(), C.c_uint32()
        p1, p2 = _abi.c_u32_p(), _abi.c_u32_p()
        rc = self._lib.frog_pairs_block(self._h, b, C.byref(i1), C.byref(i2), C.byref(n), C.byref(p1), C.byref(p2))
        if rc:
            raise IndexError(b)
        return (i1.value, i2.value,
                np.ctypeslib.as_array(p1, shape=(n.value,)), np.ctypeslib.as_array(p2, shape=(n.value,)))

"""Keypoint matcher (the producer of pairs.bin) -- ctypes layer over include/frog_match.h.

Mirrors the pairing stage of the reference's ``match`` tool (match/match.cpp:255-336 and
the pair loop :616-660): ``Matcher(images).run(jobs)`` = ComputeMatches per image pair on
the GPU.  ``Keypoints`` holds one image's rows (x, y, z, scale, laplacianSign, response,
descriptor...), the layout of a surf3d keypoint file (match.cpp:48-83).
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import check


class Keypoints:
    def __init__(self, xyz, scale, laplacian, response, desc):
        self.xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        self.scale = np.ascontiguousarray(scale, np.float32)
        self.laplacian = np.ascontiguousarray(laplacian, np.float32)
        self.response = np.ascontiguousarray(response, np.float32)
        self.desc = np.ascontiguousarray(desc, np.float32)
        n = len(self.scale)
        if self.desc.ndim != 2 or not (len(self.xyz) == len(self.laplacian) == len(self.response) == len(self.desc) == n):
            raise ValueError("keypoint arrays disagree in length")

    @property
    def n(self):
        return len(self.scale)

    @property
    def dim(self):
        return self.desc.shape[1]

    def view(self):
        p = _abi.c_float_p
        return _abi.FrogKeypoints(self.n, self.dim, self.xyz.ctypes.data_as(p), self.scale.ctypes.data_as(p),
                                  self.laplacian.ctypes.data_as(p), self.response.ctypes.data_as(p),
                                  self.desc.ctypes.data_as(p))

    @classmethod
    def from_rows(cls, rows):
        """rows: [n, 6 + dim] as in a keypoint CSV (match.cpp:60-72)."""
        rows = np.asarray(rows, np.float32).reshape(len(rows), -1)
        return cls(rows[:, 0:3], rows[:, 3], rows[:, 4], rows[:, 5], rows[:, 6:])

    def rows(self):
        return np.concatenate([self.xyz, self.scale[:, None], self.laplacian[:, None], self.response[:, None], self.desc], axis=1)


def synthetic_keypoints(n_images, n_points, dim=48, n_landmarks=None, seed=1, noise=0.05):
    """A group of images observing common landmarks: descriptor = landmark descriptor + noise
    (so that nearest neighbours are mostly the true correspondences), unit-norm like SURF
    descriptors; scales log-uniform in [1, 4]; Laplacian signs +-1; clutter points with
    random descriptors."""
    rng = np.random.default_rng(seed)
    n_landmarks = n_landmarks or n_points
    lm_desc = rng.normal(size=(n_landmarks, dim)).astype(np.float32)
    lm_desc /= np.linalg.norm(lm_desc, axis=1, keepdims=True)
    lm_xyz = rng.uniform(0, 400, size=(n_landmarks, 3)).astype(np.float32)
    lm_scale = np.exp(rng.uniform(0, np.log(4), n_landmarks)).astype(np.float32)
    lm_sign = rng.choice(np.array([-1.0, 1.0], np.float32), n_landmarks)
    out = []
    for _ in range(n_images):
        n_true = int(0.7 * n_points)
        ids = rng.choice(n_landmarks, size=min(n_true, n_landmarks), replace=False)
        d = lm_desc[ids] + noise * rng.normal(size=(len(ids), dim)).astype(np.float32)
        xyz = lm_xyz[ids] + rng.normal(0, 2, size=(len(ids), 3)).astype(np.float32)
        sc = lm_scale[ids] * np.exp(rng.normal(0, 0.05, len(ids))).astype(np.float32)
        sg = lm_sign[ids]
        n_cl = n_points - len(ids)
        d = np.concatenate([d, rng.normal(size=(n_cl, dim)).astype(np.float32)])
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        xyz = np.concatenate([xyz, rng.uniform(0, 400, size=(n_cl, 3)).astype(np.float32)])
        sc = np.concatenate([sc, np.exp(rng.uniform(0, np.log(4), n_cl)).astype(np.float32)])
        sg = np.concatenate([sg, rng.choice(np.array([-1.0, 1.0], np.float32), n_cl)])
        perm = rng.permutation(n_points)
        out.append(Keypoints(xyz[perm], sc[perm], sg[perm], rng.uniform(0, 1, n_points).astype(np.float32), d[perm].astype(np.float32)))
    return out


def all_pairs(n_images):
    """(first, second) for first < second, the job order of match.cpp:616-627."""
    return [(i, j) for i in range(n_images - 1) for j in range(i + 1, n_images)]


def _collect(lib_free, n_jobs, offset, pa, pb):
    total = int(offset[n_jobs])
    a = np.ctypeslib.as_array(pa, shape=(max(total, 1),))[:total].copy()
    b = np.ctypeslib.as_array(pb, shape=(max(total, 1),))[:total].copy()
    lib_free(pa); lib_free(pb)
    off = np.array(offset[:n_jobs + 1], np.uint64)
    return [(a[int(off[k]):int(off[k + 1])], b[int(off[k]):int(off[k + 1])]) for k in range(n_jobs)]


class Matcher:
    """All images' keypoints resident on one GPU; run(jobs) pairs them."""

    def __init__(self, images, device=0):
        self._lib = _abi.hip_lib()
        self.images = list(images)
        views = (_abi.FrogKeypoints * len(self.images))(*[k.view() for k in self.images])
        self._h = C.c_void_p()
        check(self._lib.frog_matcher_create(views, len(self.images), device, C.byref(self._h)), "frog_matcher_create")

    def close(self):
        if self._h:
            self._lib.frog_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, jobs, **options):
        """jobs: [(first, second)].  Returns per job (indices in first, indices in second)."""
        o = _abi.FrogMatchOptions.default(**options)
        n = len(jobs)
        f = (C.c_uint16 * max(n, 1))(*[j[0] for j in jobs])
        s = (C.c_uint16 * max(n, 1))(*[j[1] for j in jobs])
        offset = (C.c_uint64 * (n + 1))()
        pa, pb = _abi.c_u32_p(), _abi.c_u32_p()
        check(self._lib.frog_matcher_run(self._h, f, s, n, C.byref(o), offset, C.byref(pa), C.byref(pb)), "frog_matcher_run")
        return _collect(self._lib.frog_match_free, n, offset, pa, pb)

    def last_stats(self):
        ms, nd = C.c_double(), C.c_double()
        check(self._lib.frog_matcher_last_stats(self._h, C.byref(ms), C.byref(nd)), "frog_matcher_last_stats")
        return ms.value, nd.value

    def last_forms(self):
        """Passes of the last run by form: (exact vector kernel, f32 matrix-core filter, bf16 matrix-core filter)."""
        f = (C.c_uint64 * 3)()
        check(self._lib.frog_matcher_last_forms(self._h, f), "frog_matcher_last_forms")
        return int(f[0]), int(f[1]), int(f[2])


def read_keypoints(path):
    """A surf3d keypoint file (.csv / .csv.gz / .bin) as match.cpp reads it."""
    lib = _abi.host_lib()
    status = C.c_int()
    h = lib.frog_keypoints_read(str(path).encode(), C.byref(status))
    if not h:
        raise RuntimeError(f"cannot read keypoints from {path} (status {status.value})")
    try:
        v = _abi.FrogKeypoints()
        lib.frog_keypoints_view(h, C.byref(v))
        n, d = v.n, v.dim

        def arr(p, shape):
            return np.ctypeslib.as_array(p, shape=shape).copy() if n else np.zeros(shape, np.float32)
        return Keypoints(arr(v.xyz, (n, 3)), arr(v.scale, (n,)), arr(v.laplacian, (n,)), arr(v.response, (n,)),
                         arr(v.desc, (n, d)) if n else np.zeros((0, max(d, 1)), np.float32))
    finally:
        lib.frog_keypoints_free(h)


def write_keypoints(path, kp):
    v = kp.view()
    check(_abi.host_lib().frog_keypoints_write(str(path).encode(), C.byref(v)), "frog_keypoints_write")


def filter_products(cand, query, form, device=0):
    """The matrix-core filter's approximate -|q - c|^2 / 2 for 32 candidates x 32 queries (rows of 48 or 64 floats):
    form 0 = f32 chain, 1 = bf16 splits.  Returns (products[32 candidates][32 queries], bound relative to |q|^2 + |c|^2 on
    -2 x product)."""
    lib = _abi.hip_lib()
    c = np.ascontiguousarray(cand, np.float32)
    q = np.ascontiguousarray(query, np.float32)
    assert c.shape == q.shape and c.shape[0] == 32 and c.shape[1] in (48, 64)
    out = np.empty((32, 32), np.float32)
    bound = C.c_float(0)
    check(lib.frog_match_test_products(device, c.ctypes.data_as(_abi.c_float_p), q.ctypes.data_as(_abi.c_float_p), c.shape[1],
                                       int(form), out.ctypes.data_as(_abi.c_float_p), C.byref(bound)), "frog_match_test_products")
    return out, float(bound.value)

"""Host-side mirror of the reference's ImageGroup for the registration hot path.

``ImageGroup`` keeps the reference's option names and defaults
(registration/imageGroup.h:14-82), its method names and the iteration schedule
of ``ImageGroup::run`` (registration/imageGroup.cxx:31-157); every numeric step
is a call into libfrog_hip.so (include/frog_hip.h).  It exists so tests and
bench.py can drive the C ABI the way the reference's ``main`` drives the class;
the production host is the C++ one in frog_amd/csrc/host (bin/frog).
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import check


class ImageGroup:
    def __init__(self, pairs=None, device=0, image_range=None, **options):
        # imageGroup.h:52-82
        self.linearIterations = 50
        self.deformableLevels = 3
        self.deformableIterations = 200
        self.deformableAlpha = 0.02
        self.linearInitializationAnchor = (0.5, 0.5, 0.5)
        self.statIntervalUpdate = 10
        self.opt = _abi.FrogOptions.default()
        for k, v in options.items():
            if hasattr(self.opt, k):
                setattr(self.opt, k, v)
            elif hasattr(self, k):
                setattr(self, k, v)
            else:
                raise AttributeError(k)
        self._lib = _abi.hip_lib()
        self._ctx = None
        self.measures = []          # E per accepted iteration (imageGroup.h Measure)
        self.gridsPerLevel = []
        self.pairs = None
        self.device = device
        self.image_range = image_range
        if pairs is not None:
            self.readPairs(pairs)

    # -- life cycle ---------------------------------------------------------------
    def readPairs(self, pairs):
        """pairs: a frog_amd.pairs.Pairs (already parsed) or a path to pairs.bin."""
        from .pairs import Pairs
        if not isinstance(pairs, Pairs):
            pairs = Pairs.read(pairs)
        self.pairs = pairs
        self.close()
        ctx = C.c_void_p()
        b, e = self.image_range if self.image_range else (0, pairs.n_images)
        check(self._lib.frog_create(C.byref(pairs.model), C.byref(self.opt), self.device, b, e, C.byref(ctx)),
              "frog_create")
        self._ctx = ctx
        self.image_begin, self.image_end = b, e

    def lattice_reallocations(self):
        """lattice buffers allocated by a set-up after frog_create (0 = the head-room of max_levels_hint held)"""
        n = C.c_int()
        check(self._lib.frog_lattice_reallocations(self._ctx, C.byref(n)), "frog_lattice_reallocations")
        return n.value

    def close(self):
        if self._ctx:
            self._lib.frog_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_images(self):
        return self._lib.frog_num_images(self._ctx)

    @property
    def n_points(self):
        return self._lib.frog_num_points(self._ctx)

    # -- the methods run() calls ------------------------------------------------------
    def setupLinearTransforms(self):
        a = (C.c_float * 3)(*self.linearInitializationAnchor)
        check(self._lib.frog_linear_init(self._ctx, a), "frog_linear_init")

    def transformPoints(self, apply=False):
        check(self._lib.frog_transform_points(self._ctx, int(apply)), "frog_transform_points")

    def updateStats(self):
        check(self._lib.frog_update_stats(self._ctx), "frog_update_stats")

    def updateLinearTransforms(self):
        e = C.c_double()
        check(self._lib.frog_linear_step(self._ctx, C.byref(e)), "frog_linear_step")
        return e.value

    def RANSAC(self, image, iterations=5000, batches=None, inlier_distance=50.0, max_scale=10.0):
        """ImageGroup::RANSAC (imageGroup.cxx:629-716); `batches` defaults to the host's core count as upstream."""
        import os
        o = _abi.FrogRansacOptions(iterations, batches or os.cpu_count() or 1, inlier_distance, max_scale)
        n = C.c_int64()
        check(self._lib.frog_ransac(self._ctx, C.byref(self.pairs.model), image, C.byref(o), C.byref(n)), "frog_ransac")
        return n.value

    def setupDeformableTransforms(self, level):
        info = _abi.FrogGridInfo()
        check(self._lib.frog_deformable_setup(self._ctx, level, C.byref(info)), "frog_deformable_setup")
        return info

    def updateDeformableTransforms(self, alpha):
        e = C.c_double()
        check(self._lib.frog_deformable_step(self._ctx, alpha, C.byref(e)), "frog_deformable_step")
        return e.value

    def countInliers(self):
        arr = (_abi.FrogCounts * self.n_images)()
        check(self._lib.frog_count_inliers(self._ctx, arr), "frog_count_inliers")
        return arr

    def synchronize(self):
        check(self._lib.frog_synchronize(self._ctx), "frog_synchronize")

    def profile_enable(self, on=True):
        check(self._lib.frog_profile_enable(self._ctx, int(on)), "frog_profile_enable")

    def profile_read(self, reset=True):
        """{kernel name: (total ms, launches)} measured with HIP events on the context's stream."""
        arr = (_abi.FrogKernelTime * len(_abi.FROG_K_NAMES))()
        check(self._lib.frog_profile_read(self._ctx, arr, int(reset)), "frog_profile_read")
        return {n: (arr[i].ms_total, arr[i].launches) for i, n in enumerate(_abi.FROG_K_NAMES)}

    # -- run(), imageGroup.cxx:31-157 (no fixed images, no landmarks) ------------------
    def run(self, log=None):
        say = log if log else (lambda *_: None)
        self.measures = []
        self.gridsPerLevel = []
        self.setupLinearTransforms()
        self.transformPoints()
        say("Linear registration")
        for it in range(self.linearIterations):
            if it % self.statIntervalUpdate == 0:
                self.updateStats()
            e = self.updateLinearTransforms()
            self.transformPoints()
            self._measure(e, say)
        self.transformPoints(True)
        for level in range(self.deformableLevels):
            self.setupDeformableTransforms(level)
            self.transformPoints()
            n_grids, alpha, n_diffeo = 1, np.float32(self.deformableAlpha), 0
            it = 0
            while it < self.deformableIterations:
                if it % self.statIntervalUpdate == 0:
                    self.updateStats()
                e = self.updateDeformableTransforms(float(alpha))
                if e < 0:
                    if n_diffeo == 0:
                        alpha = np.float32(alpha / np.float32(2))
                    n_grids += 1
                    self.transformPoints(True)
                    self.setupDeformableTransforms(level)
                    self.transformPoints()
                    n_diffeo = 0
                    continue                      # same iteration index is replayed (:108-113)
                n_diffeo += 1
                self.transformPoints()
                self._measure(e, say)
                it += 1
            self.gridsPerLevel.append(n_grids)
            self.transformPoints(True)
        return self.measures

    def _measure(self, e, say):
        e32 = float(np.float32(e))
        say(f"E = {e32:g}")
        if e32 != e32:
            raise FloatingPointError("Error : NaN")     # imageGroup.cxx:1233-1236 exit(1)
        self.measures.append(e32)

    # -- read-back ---------------------------------------------------------------------
    def points(self):
        n = self.n_points
        xyz = np.empty((n, 3), np.float32)
        xyz2 = np.empty((n, 3), np.float32)
        check(self._lib.frog_get_points(self._ctx, xyz.ctypes.data_as(_abi.c_float_p),
                                        xyz2.ctypes.data_as(_abi.c_float_p)), "frog_get_points")
        return xyz, xyz2

    def set_points2(self, xyz2):
        a = np.ascontiguousarray(xyz2, np.float32)
        check(self._lib.frog_set_points2(self._ctx, a.ctypes.data_as(_abi.c_float_p)), "frog_set_points2")

    def matrix(self, image):
        m = np.empty(16, np.float64)
        check(self._lib.frog_get_linear(self._ctx, image, m.ctypes.data_as(_abi.c_double_p)), "frog_get_linear")
        return m.reshape(4, 4)

    def em(self, image):
        v = np.empty(3, np.float32)
        check(self._lib.frog_get_em(self._ctx, image, v.ctypes.data_as(_abi.c_float_p)), "frog_get_em")
        return v

    def set_em(self, image, c1_c2_ratio):
        v = np.ascontiguousarray(c1_c2_ratio, np.float32)
        check(self._lib.frog_set_em(self._ctx, image, v.ctypes.data_as(_abi.c_float_p)), "frog_set_em")

    def samples(self, image):
        n = C.c_int()
        check(self._lib.frog_get_samples(self._ctx, image, None, None, 0, C.byref(n)), "frog_get_samples")
        s = np.empty(n.value, np.float32)
        o = np.empty(n.value, np.uint32)
        if n.value:
            check(self._lib.frog_get_samples(self._ctx, image, s.ctypes.data_as(_abi.c_float_p),
                                             o.ctypes.data_as(_abi.c_u32_p), n.value, C.byref(n)), "frog_get_samples")
        return s, o

    def histogram(self, image):
        n = C.c_int()
        check(self._lib.frog_get_histogram(self._ctx, image, None, 0, C.byref(n)), "frog_get_histogram")
        h = np.zeros(n.value, np.float32)
        if n.value:
            check(self._lib.frog_get_histogram(self._ctx, image, h.ctypes.data_as(_abi.c_float_p), n.value,
                                               C.byref(n)), "frog_get_histogram")
        return h

    def num_grids(self):
        return self._lib.frog_num_grids(self._ctx)

    def grid(self, image, k):
        info = _abi.FrogGridInfo()
        check(self._lib.frog_get_grid(self._ctx, image, k, C.byref(info), None, 0), "frog_get_grid")
        g = info.dims[0] * info.dims[1] * info.dims[2]
        c = np.empty((g, 3), np.float32)
        check(self._lib.frog_get_grid(self._ctx, image, k, C.byref(info), c.ctypes.data_as(_abi.c_float_p), 3 * g),
              "frog_get_grid")
        return info, c

    def point_sums(self):
        out = np.empty((self.n_points, 4), np.float32)
        check(self._lib.frog_get_point_sums(self._ctx, out.ctypes.data_as(_abi.c_float_p)), "frog_get_point_sums")
        return out

    def errorMap(self, image):
        """saveErrorMaps (imageGroup.cxx:475-567) for one image: (grid info, [G,4] float32).
        Call residualSums() once first (it runs the sweep for all images)."""
        info = _abi.FrogGridInfo()
        check(self._lib.frog_get_error_map(self._ctx, image, C.byref(info), None, 0), "frog_get_error_map")
        g = info.dims[0] * info.dims[1] * info.dims[2]
        out = np.empty((g, 4), np.float32)
        check(self._lib.frog_get_error_map(self._ctx, image, C.byref(info), out.ctypes.data_as(_abi.c_float_p), 4 * g),
              "frog_get_error_map")
        return info, out

    def set_hard_links(self, point, partner, weight2):
        """Landmark constraints (Point::hardLinks): directed links point <- partner, global indices."""
        a = np.ascontiguousarray(point, np.uint64); b = np.ascontiguousarray(partner, np.uint64)
        u64p = C.POINTER(C.c_uint64)
        check(self._lib.frog_set_hard_links(self._ctx, a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), len(a), float(weight2)),
              "frog_set_hard_links")

    def cull_stats_linear(self):
        """(lists built during the linear stage, half-links in the last of them, half-links owned): frog_cull_stats_linear."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self._lib.frog_cull_stats_linear(self._ctx, C.byref(a), C.byref(b), C.byref(c)), "frog_cull_stats_linear")
        return a.value, b.value, c.value

    def cull_stats(self):
        """(lists built, half-links in the last list, half-links owned) of the outlier-culling list
        (frog_hip.h: frog_cull_stats)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self._lib.frog_cull_stats(self._ctx, C.byref(a), C.byref(b), C.byref(c)), "frog_cull_stats")
        return a.value, b.value, c.value

    def cull_ranges(self):
        """(non-empty ranges of the culling list, ranges swept with the lane election) -- frog_test_cull_ranges."""
        a, b = C.c_uint64(), C.c_uint64()
        check(self._lib.frog_test_cull_ranges(self._ctx, C.byref(a), C.byref(b)), "frog_test_cull_ranges")
        return a.value, b.value

    def stray_points(self):
        """Points the scatter found outside their brick since creation (frog_test_stray_points): 0 unless the sort is broken."""
        n = C.c_uint64()
        check(self._lib.frog_test_stray_points(self._ctx, C.byref(n)), "frog_test_stray_points")
        return n.value

    def residualSums(self):
        check(self._lib.frog_residual_sums(self._ctx), "frog_residual_sums")

    def gradient(self, image, n_cp):
        out = np.empty((n_cp, 4), np.float32)
        check(self._lib.frog_get_gradient(self._ctx, image, out.ctypes.data_as(_abi.c_float_p), 4 * n_cp),
              "frog_get_gradient")
        return out


def device_inlier_probability(em, d2, device=0):
    """The half-link sweep's inlier weight for SQUARED distances ``d2`` (f32, what a sweep step has in hand) under the
    mixture ``em`` = (c1, c2, ratio), evaluated on the device: (fast f32 form used for every link, form with the
    reference's promotions at sqrt(d2) used near the threshold)."""
    lib = _abi.hip_lib()
    e = np.ascontiguousarray(em, np.float32)
    dd = np.ascontiguousarray(d2, np.float32)
    fast, exact = np.empty_like(dd), np.empty_like(dd)
    check(lib.frog_test_inlier_probability(device, e.ctypes.data_as(_abi.c_float_p), dd.ctypes.data_as(_abi.c_float_p),
                                           dd.size, fast.ctypes.data_as(_abi.c_float_p),
                                           exact.ctypes.data_as(_abi.c_float_p)), "frog_test_inlier_probability")
    return fast, exact


def device_inlier_weight_pair(em_a, em_b, d2, threshold=0.5, device=0):
    """The deformable sweeps' weight of half-links between images with mixtures ``em_a`` and ``em_b`` for SQUARED distances
    ``d2``, as a sweep step forms it before the threshold band decides: (weight, form) with form 0 = one exponential, inside the
    pair's range, 1 = general form, 2 = one-exponential value below threshold - 1e-4 (dropped as an outlier)."""
    lib = _abi.hip_lib()
    a = np.ascontiguousarray(em_a, np.float32)
    b = np.ascontiguousarray(em_b, np.float32)
    dd = np.ascontiguousarray(d2, np.float32)
    w = np.empty_like(dd)
    form = np.empty(dd.size, np.uint8)
    check(lib.frog_test_inlier_weight_pair(device, a.ctypes.data_as(_abi.c_float_p), b.ctypes.data_as(_abi.c_float_p),
                                           float(threshold), dd.ctypes.data_as(_abi.c_float_p), dd.size,
                                           w.ctypes.data_as(_abi.c_float_p), form.ctypes.data_as(_abi.C.POINTER(_abi.C.c_ubyte))),
          "frog_test_inlier_weight_pair")
    return w, form

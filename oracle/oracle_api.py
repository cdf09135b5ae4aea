"""ctypes binding of the CPU oracle (oracle/libfrog_oracle.so) and of the
reference-Stats build (oracle/_ref/libfrog_refstats.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under frog_amd/ imports this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
fp = C.POINTER(C.c_float)
dp = C.POINTER(C.c_double)
u32p = C.POINTER(C.c_uint32)

_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libfrog_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        L = C.CDLL(path)
        L.frogo_create.restype = C.c_void_p
        L.frogo_create.argtypes = [C.c_void_p, C.c_void_p]
        L.frogo_destroy.argtypes = [C.c_void_p]
        L.frogo_set_threads.argtypes = [C.c_int]
        L.frogo_get_max_threads.restype = C.c_int
        for n in ("frogo_setup_stats", "frogo_update_stats"):
            getattr(L, n).argtypes = [C.c_void_p]
        L.frogo_linear_init.argtypes = [C.c_void_p, fp]
        L.frogo_transform_points.argtypes = [C.c_void_p, C.c_int]
        L.frogo_linear_step.restype = C.c_double
        L.frogo_linear_step.argtypes = [C.c_void_p]
        L.frogo_deformable_setup.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.frogo_deformable_step.restype = C.c_double
        L.frogo_deformable_step.argtypes = [C.c_void_p, C.c_float]
        L.frogo_count_inliers.argtypes = [C.c_void_p, C.c_void_p]
        L.frogo_error_map.argtypes = [C.c_void_p, C.c_uint32, fp, C.c_size_t]; L.frogo_error_map.restype = C.c_int
        L.frogo_run.restype = C.c_int
        L.frogo_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, fp, dp, C.c_int,
                                C.POINTER(C.c_int)]
        L.frogo_set_range.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.frogo_linear_step_local.argtypes = [C.c_void_p, dp]
        L.frogo_bounds_local.argtypes = [C.c_void_p, dp, dp]
        L.frogo_deformable_setup_bounds.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_void_p]
        L.frogo_deformable_phase_a.argtypes = [C.c_void_p, C.c_float, dp, dp]
        L.frogo_deformable_phase_b.restype = C.c_long
        L.frogo_deformable_phase_b.argtypes = [C.c_void_p, dp]
        L.frogo_deformable_phase_c.argtypes = [C.c_void_p]
        L.frogo_num_points.restype = C.c_uint64
        L.frogo_num_points.argtypes = [C.c_void_p]
        for n in ("frogo_get_xyz", "frogo_get_xyz2", "frogo_set_xyz2", "frogo_get_point_sums"):
            getattr(L, n).argtypes = [C.c_void_p, fp]
        L.frogo_get_matrix.argtypes = [C.c_void_p, C.c_uint32, dp]
        L.frogo_get_em.argtypes = [C.c_void_p, C.c_uint32, fp]
        L.frogo_set_em.argtypes = [C.c_void_p, C.c_uint32, fp]
        L.frogo_get_samples.restype = C.c_int
        L.frogo_get_samples.argtypes = [C.c_void_p, C.c_uint32, fp, C.c_int]
        L.frogo_get_sample_ordinals.restype = C.c_int
        L.frogo_get_sample_ordinals.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_int]
        L.frogo_get_histogram.restype = C.c_int
        L.frogo_get_histogram.argtypes = [C.c_void_p, C.c_uint32, fp, C.c_int]
        L.frogo_num_grids.restype = C.c_int
        L.frogo_num_grids.argtypes = [C.c_void_p]
        L.frogo_get_grid.restype = C.c_int
        L.frogo_get_grid.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, fp, C.c_size_t]
        L.frogo_get_gradient.restype = C.c_int
        L.frogo_get_gradient.argtypes = [C.c_void_p, C.c_uint32, fp, C.c_size_t]
        L.frogo_keep_raw_gradient.argtypes = [C.c_void_p, C.c_int]
        L.frogo_get_gradient_raw.restype = C.c_int
        L.frogo_get_gradient_raw.argtypes = [C.c_void_p, C.c_uint32, fp, C.c_size_t]
        _bind_stats(L, "frogo_stats_")
        L.frogo_chipdf.restype = C.c_float
        L.frogo_chipdf.argtypes = [C.c_float]
        L.frogo_bspline_weights_n.restype = None
        L.frogo_bspline_weights_n.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        # as many threads as the process may use (a container's CPU quota counts: the GPU boxes show 256 hardware threads to a
        # 16-CPU share, and 256 OpenMP threads there are throttled together)
        try:
            from frog_amd._abi import usable_cpus
            L.frogo_set_threads(max(1, min(L.frogo_get_max_threads(), usable_cpus())))
        except ImportError:
            pass
        _lib = L
    return _lib


def ref_lib():
    """The reference's own stats.cxx (oracle/_ref); None when it was not built."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libfrog_refstats.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        _bind_stats(L, "refstats_")
        L.refstats_inlier_probability_n.argtypes = [C.c_void_p, fp, C.c_int, fp]
        L.refstats_chipdf.restype = C.c_float
        L.refstats_chipdf.argtypes = [C.c_float]
        _ref = L
    return _ref


_ref_weights = None


def ref_weights_lib():
    """The reference's own vtkBSplineTransformWeights (imageGroup.cxx:221-232), cut out of the file and compiled by
    oracle/Makefile into oracle/_ref/libfrog_refweights.so; None when it was not built."""
    global _ref_weights
    if _ref_weights is None:
        path = os.path.join(_HERE, "_ref", "libfrog_refweights.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.refweights_n.restype = None
        L.refweights_n.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _ref_weights = L
    return _ref_weights


def bspline_weights(f, which="oracle"):
    """The four cubic B-spline weights of every fraction in f (f64): which = "oracle" (the restatement) or "reference"."""
    f = np.ascontiguousarray(f, np.float64)
    out = np.empty((len(f), 4), np.float64)
    L = lib() if which == "oracle" else ref_weights_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libfrog_refweights.so not built")
    (L.frogo_bspline_weights_n if which == "oracle" else L.refweights_n)(f.ctypes.data, len(f), out.ctypes.data)
    return out


def _bind_stats(L, pre):
    g = lambda n: getattr(L, pre + n)
    g("new").restype = C.c_void_p
    g("new").argtypes = [C.c_int, C.c_int, C.c_float]
    g("free").argtypes = [C.c_void_p]
    g("add_slots").argtypes = [C.c_void_p, C.c_int]
    g("reset").argtypes = [C.c_void_p]
    g("add_samples").argtypes = [C.c_void_p, fp, C.c_int]
    g("estimate").argtypes = [C.c_void_p]
    g("inlier_probability").restype = C.c_float
    g("inlier_probability").argtypes = [C.c_void_p, C.c_float]
    g("get_params").argtypes = [C.c_void_p, fp]
    g("set_params").argtypes = [C.c_void_p, fp]
    g("size").restype = C.c_int
    g("size").argtypes = [C.c_void_p]
    g("get_samples").restype = C.c_int
    g("get_samples").argtypes = [C.c_void_p, fp, C.c_int]
    g("histogram").restype = C.c_int
    g("histogram").argtypes = [C.c_void_p, C.c_float, fp, C.c_int]


class Stats:
    """Stats object of either the restatement (which='oracle') or the reference build (which='ref')."""

    def __init__(self, which="oracle", max_size=10000, max_iterations=10000, epsilon=1e-6):
        self.L = lib() if which == "oracle" else ref_lib()
        if self.L is None:
            raise RuntimeError("oracle/_ref not built")
        self.pre = "frogo_stats_" if which == "oracle" else "refstats_"
        self.h = C.c_void_p(self._f("new")(max_size, max_iterations, epsilon))

    def _f(self, n):
        return getattr(self.L, self.pre + n)

    def __del__(self):
        if getattr(self, "h", None):
            self._f("free")(self.h)
            self.h = None

    def add_slots(self, n):
        self._f("add_slots")(self.h, n)

    def reset(self):
        self._f("reset")(self.h)

    def add_samples(self, v):
        v = np.ascontiguousarray(v, np.float32)
        self._f("add_samples")(self.h, v.ctypes.data_as(fp), len(v))

    def estimate(self):
        self._f("estimate")(self.h)

    def params(self):
        o = np.empty(3, np.float32)
        self._f("get_params")(self.h, o.ctypes.data_as(fp))
        return o

    def set_params(self, p):
        p = np.ascontiguousarray(p, np.float32)
        self._f("set_params")(self.h, p.ctypes.data_as(fp))

    def prob(self, d):
        return self._f("inlier_probability")(self.h, float(d))

    def prob_n(self, d):
        """getInlierProbability for an array of distances (reference build only)."""
        d = np.ascontiguousarray(d, np.float32)
        out = np.empty_like(d)
        self.L.refstats_inlier_probability_n(self.h, d.ctypes.data_as(fp), d.size, out.ctypes.data_as(fp))
        return out

    def size(self):
        return self._f("size")(self.h)

    def samples(self):
        n = self.size()
        o = np.empty(n, np.float32)
        self._f("get_samples")(self.h, o.ctypes.data_as(fp), n)
        return o

    def histogram(self, bin=1.0):
        n = self._f("histogram")(self.h, bin, None, 0)
        o = np.empty(n, np.float32)
        self._f("histogram")(self.h, bin, o.ctypes.data_as(fp), n)
        return o


class OracleGroup:
    """frogo_group: the CPU restatement of ImageGroup over a frog_model."""

    def __init__(self, model, options):
        self.L = lib()
        self._model, self._opt = model, options       # keep alive
        self.h = C.c_void_p(self.L.frogo_create(C.byref(model), C.byref(options)))
        self.n_images = model.n_images
        self.P = self.L.frogo_num_points(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.frogo_destroy(self.h)
            self.h = None

    def setup_stats(self):
        self.L.frogo_setup_stats(self.h)

    def linear_init(self, anchor=(0.5, 0.5, 0.5)):
        self.L.frogo_linear_init(self.h, (C.c_float * 3)(*anchor))

    def transform_points(self, apply=False):
        self.L.frogo_transform_points(self.h, int(apply))

    def update_stats(self):
        self.L.frogo_update_stats(self.h)

    def linear_step(self):
        return self.L.frogo_linear_step(self.h)

    def deformable_setup(self, level, info):
        self.L.frogo_deformable_setup(self.h, level, C.byref(info))
        return info

    def deformable_step(self, alpha):
        return self.L.frogo_deformable_step(self.h, alpha)

    def error_map(self, image, n_cp):
        out = np.empty((n_cp, 4), np.float32)
        rc = self.L.frogo_error_map(self.h, image, out.ctypes.data_as(fp), out.size)
        assert rc == 0
        return out

    def set_hard_links(self, point, partner, weight2):
        a = np.ascontiguousarray(point, np.uint64); b = np.ascontiguousarray(partner, np.uint64)
        self.L.frogo_set_hard_links.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t, C.c_float]
        self.L.frogo_set_hard_links.restype = C.c_int
        assert self.L.frogo_set_hard_links(self.h, a.ctypes.data_as(C.POINTER(C.c_uint64)), b.ctypes.data_as(C.POINTER(C.c_uint64)),
                                           len(a), float(weight2)) == 0

    def count_inliers(self, counts_array):
        self.L.frogo_count_inliers(self.h, counts_array)
        return counts_array

    # split phases (disjoint image ranges combined by the caller)
    def set_range(self, b, e):
        self.L.frogo_set_range(self.h, b, e)

    def linear_step_local(self):
        o = np.empty(2, np.float64)
        self.L.frogo_linear_step_local(self.h, o.ctypes.data_as(dp))
        return o

    def bounds_local(self):
        mn, mx = np.empty(3, np.float64), np.empty(3, np.float64)
        self.L.frogo_bounds_local(self.h, mn.ctypes.data_as(dp), mx.ctypes.data_as(dp))
        return mn, mx

    def deformable_setup_bounds(self, level, mn, mx, info):
        mn = np.ascontiguousarray(mn, np.float64); mx = np.ascontiguousarray(mx, np.float64)
        self.L.frogo_deformable_setup_bounds(self.h, level, mn.ctypes.data_as(dp), mx.ctypes.data_as(dp), C.byref(info))
        return info

    def phase_a(self, alpha, gridsum):
        o = np.empty(2, np.float64)
        self.L.frogo_deformable_phase_a(self.h, alpha, gridsum.ctypes.data_as(dp), o.ctypes.data_as(dp))
        return o

    def phase_b(self, gridsum_all):
        g = np.ascontiguousarray(gridsum_all, np.float64)
        return int(self.L.frogo_deformable_phase_b(self.h, g.ctypes.data_as(dp)))

    def phase_c(self):
        self.L.frogo_deformable_phase_c(self.h)

    def ransac(self, image, iterations=5000, batches=8, inlier_distance=50.0, max_scale=10.0):
        self.L.frogo_ransac.restype = C.c_long
        self.L.frogo_ransac.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_float, C.c_float]
        return self.L.frogo_ransac(self.h, image, iterations, batches, inlier_distance, max_scale)

    def run(self, li=50, dl=3, di=200, da=0.02, si=10, anchor=(0.5, 0.5, 0.5)):
        cap = li + dl * di + 8
        E = (C.c_double * cap)()
        ng = (C.c_int * max(dl, 1))()
        n = self.L.frogo_run(self.h, li, dl, di, da, si, (C.c_float * 3)(*anchor), E, cap, ng)
        return np.array(E[:n]), list(ng[:dl])

    def xyz(self):
        o = np.empty((self.P, 3), np.float32)
        self.L.frogo_get_xyz(self.h, o.ctypes.data_as(fp))
        return o

    def xyz2(self):
        o = np.empty((self.P, 3), np.float32)
        self.L.frogo_get_xyz2(self.h, o.ctypes.data_as(fp))
        return o

    def set_xyz2(self, a):
        a = np.ascontiguousarray(a, np.float32)
        self.L.frogo_set_xyz2(self.h, a.ctypes.data_as(fp))

    def matrix(self, image):
        o = np.empty(16, np.float64)
        self.L.frogo_get_matrix(self.h, image, o.ctypes.data_as(dp))
        return o.reshape(4, 4)

    def em(self, image):
        o = np.empty(3, np.float32)
        self.L.frogo_get_em(self.h, image, o.ctypes.data_as(fp))
        return o

    def set_em(self, image, v):
        v = np.ascontiguousarray(v, np.float32)
        self.L.frogo_set_em(self.h, image, v.ctypes.data_as(fp))

    def samples(self, image):
        n = self.L.frogo_get_samples(self.h, image, None, 0)
        s = np.empty(n, np.float32)
        o = np.empty(n, np.uint32)
        self.L.frogo_get_samples(self.h, image, s.ctypes.data_as(fp), n)
        self.L.frogo_get_sample_ordinals(self.h, image, o.ctypes.data_as(u32p), n)
        return s, o

    def histogram(self, image):
        n = self.L.frogo_get_histogram(self.h, image, None, 0)
        o = np.empty(n, np.float32)
        self.L.frogo_get_histogram(self.h, image, o.ctypes.data_as(fp), n)
        return o

    def num_grids(self):
        return self.L.frogo_num_grids(self.h)

    def grid(self, image, k, info):
        self.L.frogo_get_grid(self.h, image, k, C.byref(info), None, 0)
        g = info.dims[0] * info.dims[1] * info.dims[2]
        c = np.empty((g, 3), np.float32)
        self.L.frogo_get_grid(self.h, image, k, C.byref(info), c.ctypes.data_as(fp), 3 * g)
        return info, c

    def point_sums(self):
        o = np.empty((self.P, 4), np.float32)
        self.L.frogo_get_point_sums(self.h, o.ctypes.data_as(fp))
        return o

    def gradient(self, image, n_cp):
        o = np.empty((n_cp, 4), np.float32)
        self.L.frogo_get_gradient(self.h, image, o.ctypes.data_as(fp), 4 * n_cp)
        return o

    def keep_raw_gradient(self, on=True):
        self.L.frogo_keep_raw_gradient(self.h, int(on))

    def gradient_raw(self, image, n_cp):
        """The gradient image as the scatter left it in the last deformable step (keep_raw_gradient() first)."""
        o = np.empty((n_cp, 4), np.float32)
        assert self.L.frogo_get_gradient_raw(self.h, image, o.ctypes.data_as(fp), 4 * n_cp) == 4 * n_cp
        return o


# ---- pairing stage of the reference's `match` tool (oracle/match_oracle.cpp) ---------------
def match_run(images, jobs, threshold=0.22, dist2second=1.0, anat=0.0, sym=0, all=0, threads=None):
    """ComputeMatches (match/match.cpp:255-336) per (first, second) job on the CPU.
    `images`: frog_amd.match.Keypoints.  Returns per job (indices in first, indices in second)."""
    from frog_amd import _abi
    L = lib()
    L.frogo_match_run.restype = C.c_int
    L.frogo_match_run.argtypes = [C.POINTER(_abi.FrogKeypoints), C.c_uint32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16),
                                  C.c_size_t, C.POINTER(_abi.FrogMatchOptions), C.POINTER(C.c_uint64),
                                  C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32))]
    L.frogo_match_free.argtypes = [C.c_void_p]
    L.frogo_match_free.restype = None
    if threads:
        L.frogo_match_set_threads(int(threads))
    views = (_abi.FrogKeypoints * len(images))(*[k.view() for k in images])
    o = _abi.FrogMatchOptions()
    o.threshold, o.dist2second, o.anat, o.sym, o.all = threshold, dist2second, anat, sym, all
    n = len(jobs)
    f = (C.c_uint16 * max(n, 1))(*[j[0] for j in jobs])
    s = (C.c_uint16 * max(n, 1))(*[j[1] for j in jobs])
    offset = (C.c_uint64 * (n + 1))()
    pa, pb = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
    rc = L.frogo_match_run(views, len(images), f, s, n, C.byref(o), offset, C.byref(pa), C.byref(pb))
    assert rc == 0
    total = int(offset[n])
    a = np.ctypeslib.as_array(pa, shape=(max(total, 1),))[:total].copy()
    b = np.ctypeslib.as_array(pb, shape=(max(total, 1),))[:total].copy()
    L.frogo_match_free(pa); L.frogo_match_free(pb)
    return [(a[int(offset[k]):int(offset[k + 1])], b[int(offset[k]):int(offset[k + 1])]) for k in range(n)]


_ref_match = None


def ref_match_lib():
    """The reference's own ComputeMatches / scalar norm / struct Point (match/match.cpp:28-48, :243-251, :255-336), cut out by
    oracle/Makefile into oracle/_ref/libfrog_refmatch.so; None when it was not built."""
    global _ref_match
    if _ref_match is None:
        path = os.path.join(_HERE, "_ref", "libfrog_refmatch.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.refmatch_compute.restype = C.c_long
        L.refmatch_compute.argtypes = ([C.c_uint32] + [C.c_void_p] * 4) * 2 + [C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_float,
                                                                              C.c_int, C.c_void_p, C.c_void_p, C.c_long]
        L.refmatch_norm.restype = C.c_float
        L.refmatch_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        assert L.refmatch_sizeof_point_id() == 4            # INT_PTIDS, the reference's default build
        _ref_match = L
    return _ref_match


def ref_match_run(images, jobs, threshold=0.22, dist2second=1.0, anat=0.0, sym=0, all=0):
    """match_run's interface over the REFERENCE build: per job ComputeMatches(first, second, ...) as main calls it
    (match.cpp:642), with `sym` the reverse direction ComputeMatches(second, first, ..., true) appended (match.cpp:644-648)."""
    L = ref_match_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libfrog_refmatch.so not built")
    f32 = lambda a: np.ascontiguousarray(a, np.float32)

    def one(cand, qry, is_sym):
        arrs = [f32(cand.xyz), f32(cand.scale), f32(cand.laplacian), f32(cand.desc), f32(qry.xyz), f32(qry.scale),
                f32(qry.laplacian), f32(qry.desc)]
        dim = arrs[3].shape[1] if arrs[3].ndim == 2 else arrs[7].shape[1]
        cap = max(1, cand.n * qry.n if all else qry.n)
        a, b = np.empty(cap, np.uint32), np.empty(cap, np.uint32)
        n = L.refmatch_compute(cand.n, *[x.ctypes.data for x in arrs[:4]], qry.n, *[x.ctypes.data for x in arrs[4:]], dim,
                               threshold, dist2second, int(all), anat, int(is_sym), a.ctypes.data, b.ctypes.data, cap)
        assert 0 <= n <= cap
        return a[:n], b[:n]
    out = []
    for first, second in jobs:
        a, b = one(images[first], images[second], False)
        if sym:
            a2, b2 = one(images[second], images[first], True)
            a, b = np.concatenate([a, a2]), np.concatenate([b, b2])
        out.append((a, b))
    return out


# ---- transform chains (oracle/chain_oracle.cpp) --------------------------------------------
def chain_apply(links, points, jacobian=False):
    """Forward evaluation (and Jacobians) of a chain of frog_amd.chain.Link on the CPU."""
    from frog_amd import _abi
    L = lib()
    L.frogo_chain_apply.restype = None
    L.frogo_chain_apply.argtypes = [C.POINTER(_abi.FrogChainLink), C.c_uint32, dp, dp, dp, C.c_size_t]
    views = (_abi.FrogChainLink * max(1, len(links)))(*[l.view() for l in links])
    p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    out = np.empty_like(p)
    jac = np.empty((len(p), 3, 3), np.float64) if jacobian else None
    L.frogo_chain_apply(views, len(links), p.ctypes.data_as(dp), out.ctypes.data_as(dp),
                        jac.ctypes.data_as(dp) if jacobian else None, len(p))
    return (out, jac) if jacobian else out


def chain_check(links, origin, spacing, dims):
    from frog_amd import _abi
    L = lib()
    L.frogo_chain_check.restype = None
    L.frogo_chain_check.argtypes = [C.POINTER(_abi.FrogChainLink), C.c_uint32, dp, dp, C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint64), dp]
    views = (_abi.FrogChainLink * max(1, len(links)))(*[l.view() for l in links])
    o = (C.c_double * 3)(*origin); s = (C.c_double * 3)(*spacing); d = (C.c_uint32 * 3)(*dims)
    n, m = C.c_uint64(), C.c_double()
    L.frogo_chain_check(views, len(links), o, s, d, C.byref(n), C.byref(m))
    return int(n.value), float(m.value)


def chain_reslice(links, src, src_origin, src_spacing, out_dims, out_origin, out_spacing, interpolation=1, background=0.0):
    """vtkImageReslice as VolumeTransform uses it, on the CPU.  `src` is indexed [z, y, x]; so is the result (f64)."""
    from frog_amd import _abi
    L = lib()
    L.frogo_chain_reslice.restype = None
    u3 = C.c_uint32 * 3
    d3 = C.c_double * 3
    L.frogo_chain_reslice.argtypes = [C.POINTER(_abi.FrogChainLink), C.c_uint32, dp, u3, d3, d3, u3, d3, d3, C.c_int, C.c_double, dp]
    views = (_abi.FrogChainLink * max(1, len(links)))(*[l.view() for l in links])
    s = np.ascontiguousarray(src, np.float64)
    out = np.empty(tuple(int(v) for v in out_dims[::-1]), np.float64)
    L.frogo_chain_reslice(views, len(links), s.ctypes.data_as(dp), u3(*s.shape[::-1]), d3(*src_origin), d3(*src_spacing),
                          u3(*out_dims), d3(*out_origin), d3(*out_spacing), int(interpolation), float(background), out.ctypes.data_as(dp))
    return out

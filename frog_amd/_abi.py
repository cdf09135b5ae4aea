"""ctypes view of include/frog_types.h, include/frog_hip.h and include/frog_host.h.

The shared libraries are built in-tree by ``__graft_entry__.build()`` (or
``make -C frog_amd/csrc``).  Loading fails loudly when they are missing: there is
no Python or CPU stand-in for the HIP path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")



def usable_cpus():
    """CPUs this process may run on: the affinity mask capped by the cgroup quota (csrc/common/usable_cpus.h, the same rule).
    On the GPU boxes os.cpu_count() says 256 and cpu.max says 16: 256 OpenMP threads there are throttled together."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def quota(d):
        try:
            with open(os.path.join(d, "cpu.max")) as fh:
                q, p = fh.read().split()[:2]
            return None if q == "max" else float(q) / float(p)
        except (OSError, ValueError):
            pass
        try:
            with open(os.path.join(d, "cpu.cfs_quota_us")) as fh:
                q = float(fh.read())
            with open(os.path.join(d, "cpu.cfs_period_us")) as fh:
                p = float(fh.read())
            return q / p if q > 0 and p > 0 else None
        except (OSError, ValueError):
            return None
    dirs = ["/sys/fs/cgroup", "/sys/fs/cgroup/cpu"]
    try:
        with open("/proc/self/cgroup") as fh:
            for line in fh:
                _, ctrl, path = line.rstrip("\n").split(":", 2)
                root = "/sys/fs/cgroup" if ctrl == "" else ("/sys/fs/cgroup/cpu" if "cpu" in ctrl.split(",") or "cpuacct" in ctrl.split(",") else None)
                while root and len(path) > 1:
                    dirs.append(root + path)
                    path = path.rsplit("/", 1)[0] or "/"
    except (OSError, ValueError):
        pass
    for d in dirs:
        q = quota(d)
        if q and q > 0:
            n = min(n, max(1, math.ceil(q - 1e-9)))
    return max(1, n)


def device_source_hash():
    """sha256 (first 16 hex digits) over the HIP sources of the registration kernels (everything under csrc/device
    except the matcher, the transform chains and the collectives library), in name order: what a PMC measurement under
    profiles/ is keyed on, so that bench.py can tell when the kernels have changed since."""
    import hashlib
    d = os.path.join(_HERE, "csrc", "device")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) and name not in ("comm.hip", "match.hip", "chain.hip"):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_u32_p = C.POINTER(C.c_uint32)


class FrogModel(C.Structure):
    """frog_model (include/frog_types.h)."""
    _fields_ = [("n_images", C.c_uint32),
                ("point_offset", C.c_void_p),
                ("xyz", C.c_void_p),
                ("row_ptr", C.c_void_p),
                ("link_image", C.c_void_p),
                ("link_point", C.c_void_p)]


class FrogOptions(C.Structure):
    """frog_options; defaults = imageGroup.h:52-82, stats.cxx:10-12."""
    _fields_ = [("linear_alpha", C.c_float),
                ("use_scale", C.c_int32),
                ("initial_grid_size", C.c_float),
                ("bounding_box_margin", C.c_float),
                ("inlier_threshold", C.c_float),
                ("guarantee_diffeomorphism", C.c_int32),
                ("max_displacement_ratio", C.c_float),
                ("stats_max_size", C.c_int32),
                ("stats_max_iterations", C.c_int32),
                ("stats_epsilon", C.c_float),
                ("n_fixed_images", C.c_int32),
                ("max_levels_hint", C.c_int32),
                ("reference_order", C.c_int32),
                ("selections_in_background", C.c_int32),
                ("reserved", C.c_int32 * 2)]

    @classmethod
    def default(cls, **kw):
        o = cls(0.5, 1, 100.0, 0.1, 0.5, 1, 0.4, 10000, 10000, 1e-6)
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        return o


class FrogGridInfo(C.Structure):
    _fields_ = [("dims", C.c_int32 * 3),
                ("n_grid", C.c_int32),
                ("origin", C.c_double * 3),
                ("spacing", C.c_double * 3),
                ("bbox", C.c_double * 6)]


class FrogRansacOptions(C.Structure):
    """frog_ransac_options; defaults = imageGroup.h:70-74."""
    _fields_ = [("iterations", C.c_int32), ("batches", C.c_int32),
                ("inlier_distance", C.c_float), ("max_scale", C.c_float)]


class FrogCounts(C.Structure):
    _fields_ = [("points", C.c_int64), ("pairs", C.c_int64),
                ("inliers", C.c_int64), ("outliers", C.c_int64),
                ("c1", C.c_float), ("c2", C.c_float), ("ratio", C.c_float), ("pad_", C.c_float)]


class FrogSynthParams(C.Structure):
    _fields_ = [("n_images", C.c_uint32),
                ("points_per_image", C.c_uint32),
                ("n_landmarks", C.c_uint32),
                ("pairs_per_block", C.c_double),
                ("partners_per_image", C.c_uint32),
                ("outlier_fraction", C.c_float),
                ("noise_sigma", C.c_float),
                ("bump_amplitude", C.c_float),
                ("scale_min", C.c_float),
                ("scale_max", C.c_float),
                ("translation_range", C.c_float),
                ("seed", C.c_uint64)]


class FrogKernelTime(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("launches", C.c_uint64)]


SCHEDULE_MAX_LEVELS, SCHEDULE_MAX_LATTICES, FROG_K_COUNT = 16, 512, 10


class FrogSchedulePlan(C.Structure):
    """frog_schedule_plan (include/frog_host.h)."""
    _fields_ = [("plan_bytes", C.c_uint32), ("result_bytes", C.c_uint32),
                ("warmup_linear", C.c_int32), ("linear", C.c_int32), ("n_levels", C.c_int32),
                ("per_level", C.c_int32 * SCHEDULE_MAX_LEVELS), ("stat_interval", C.c_int32),
                ("deformable_alpha", C.c_float), ("anchor", C.c_float * 3), ("profile", C.c_int32), ("time_comm", C.c_int32),
                ("proxy_xyz2", C.c_void_p), ("proxy_em", C.c_void_p)]


class FrogScheduleLattice(C.Structure):
    _fields_ = [("level", C.c_int32), ("dims", C.c_int32 * 3), ("iterations", C.c_int32), ("setup_host_s", C.c_double)]


class FrogScheduleResult(C.Structure):
    """frog_schedule_result (include/frog_host.h)."""
    _fields_ = [("elapsed_s", C.c_double), ("phase_s", C.c_double * (1 + SCHEDULE_MAX_LEVELS)),
                ("iterations", C.c_int32), ("grids_per_level", C.c_int32 * SCHEDULE_MAX_LEVELS), ("n_lattices", C.c_int32),
                ("lattices", FrogScheduleLattice * SCHEDULE_MAX_LATTICES), ("final_E", C.c_double),
                ("kernels", FrogKernelTime * FROG_K_COUNT),
                ("kernels_by_phase", (FrogKernelTime * FROG_K_COUNT) * (1 + SCHEDULE_MAX_LEVELS)),
                ("comm_ms", C.c_double * 4), ("comm_calls", C.c_uint64 * 4), ("comm_sampled", C.c_uint64 * 4),
                ("replica_hash", C.c_uint64)]


FROG_K_NAMES = ["sweep_linear", "sweep_deformable", "scatter", "lattice", "transform", "stats", "combine", "cull", "sweep_build", "sweep_linear_build"]
FROG_OK, FROG_E_INVALID, FROG_E_NODEVICE, FROG_E_HIP, FROG_E_STATE, FROG_E_NOMEM, FROG_E_IO = range(7)
FROG_BUF_XYZ2, FROG_BUF_EM, FROG_BUF_ENERGY, FROG_BUF_GRIDSUM = range(4)

# name -> (restype, argtypes): every symbol include/frog_hip.h declares
class FrogKeypoints(C.Structure):
    """frog_keypoints (include/frog_match.h): host arrays of one image's keypoints."""
    _fields_ = [("n", C.c_uint32), ("dim", C.c_uint32), ("xyz", c_float_p), ("scale", c_float_p),
                ("laplacian", c_float_p), ("response", c_float_p), ("desc", c_float_p)]


class FrogVolume(C.Structure):
    """frog_volume (include/frog_chain.h)."""
    _fields_ = [("dims", C.c_uint32 * 3), ("spacing", C.c_double * 3), ("origin", C.c_double * 3),
                ("dtype", C.c_int), ("data", C.c_void_p)]


FROG_V_DTYPES = ["uint8", "int8", "uint16", "int16", "uint32", "int32", "float32", "float64"]


class FrogChainLink(C.Structure):
    """frog_chain_link (include/frog_chain.h)."""
    _fields_ = [("type", C.c_int), ("matrix", C.c_double * 16), ("dims", C.c_uint32 * 3), ("origin", C.c_double * 3),
                ("spacing", C.c_double * 3), ("coeffs", c_float_p)]


class FrogMatchOptions(C.Structure):
    _fields_ = [("threshold", C.c_float), ("dist2second", C.c_float), ("anat", C.c_float), ("sym", C.c_int),
                ("all", C.c_int), ("reserved", C.c_int * 3)]

    @classmethod
    def default(cls, **kw):
        o = cls()
        hip_lib().frog_match_options_default(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o


HIP_SYMBOLS = {
    "frog_device_warm": (C.c_int, [C.c_int]),
    "frog_device_count": (C.c_int, []),
    "frog_last_error": (C.c_char_p, []),
    "frog_create": (C.c_int, [C.POINTER(FrogModel), C.POINTER(FrogOptions), C.c_int, C.c_uint32, C.c_uint32,
                              C.POINTER(C.c_void_p)]),
    "frog_destroy": (None, [C.c_void_p]),
    "frog_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "frog_synchronize": (C.c_int, [C.c_void_p]),
    "frog_linear_init": (C.c_int, [C.c_void_p, c_float_p]),
    "frog_transform_points": (C.c_int, [C.c_void_p, C.c_int]),
    "frog_update_stats": (C.c_int, [C.c_void_p]),
    "frog_linear_step": (C.c_int, [C.c_void_p, c_double_p]),
    "frog_ransac": (C.c_int, [C.c_void_p, C.POINTER(FrogModel), C.c_uint32, C.POINTER(FrogRansacOptions),
                              C.POINTER(C.c_int64)]),
    "frog_deformable_setup": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(FrogGridInfo)]),
    "frog_deformable_step": (C.c_int, [C.c_void_p, C.c_float, c_double_p]),
    "frog_count_inliers": (C.c_int, [C.c_void_p, C.POINTER(FrogCounts)]),
    "frog_num_points": (C.c_uint64, [C.c_void_p]),
    "frog_num_images": (C.c_uint32, [C.c_void_p]),
    "frog_get_points": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "frog_set_points2": (C.c_int, [C.c_void_p, c_float_p]),
    "frog_get_linear": (C.c_int, [C.c_void_p, C.c_uint32, c_double_p]),
    "frog_get_em": (C.c_int, [C.c_void_p, C.c_uint32, c_float_p]),
    "frog_set_em": (C.c_int, [C.c_void_p, C.c_uint32, c_float_p]),
    "frog_set_em_rows": (C.c_int, [C.c_void_p, c_float_p, C.c_uint32, C.c_uint32]),
    "frog_get_samples": (C.c_int, [C.c_void_p, C.c_uint32, c_float_p, c_u32_p, C.c_int, C.POINTER(C.c_int)]),
    "frog_get_histogram": (C.c_int, [C.c_void_p, C.c_uint32, c_float_p, C.c_int, C.POINTER(C.c_int)]),
    "frog_num_grids": (C.c_int, [C.c_void_p]),
    "frog_get_grid": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(FrogGridInfo), c_float_p, C.c_size_t]),
    "frog_get_point_sums": (C.c_int, [C.c_void_p, c_float_p]),
    "frog_get_gradient": (C.c_int, [C.c_void_p, C.c_uint32, c_float_p, C.c_size_t]),
    "frog_chain_create": (C.c_int, [C.POINTER(FrogChainLink), C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "frog_chain_destroy": (None, [C.c_void_p]),
    "frog_chain_num_links": (C.c_uint32, [C.c_void_p]),
    "frog_chain_apply": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_size_t]),
    "frog_chain_check": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                   c_double_p]),
    "frog_chain_invert_links": (C.c_int, [C.POINTER(FrogChainLink), C.c_uint32, C.POINTER(FrogChainLink)]),
    "frog_chain_reslice": (C.c_int, [C.c_void_p, C.POINTER(FrogVolume), C.POINTER(FrogVolume), C.c_int, C.c_double]),
    "frog_match_options_default": (None, [C.POINTER(FrogMatchOptions)]),
    "frog_matcher_create": (C.c_int, [C.POINTER(FrogKeypoints), C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "frog_matcher_destroy": (None, [C.c_void_p]),
    "frog_matcher_run": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.c_size_t,
                                   C.POINTER(FrogMatchOptions), C.POINTER(C.c_uint64), C.POINTER(c_u32_p),
                                   C.POINTER(c_u32_p)]),
    "frog_match_free": (None, [C.c_void_p]),
    "frog_matcher_last_stats": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "frog_matcher_last_forms": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "frog_match_test_products": (C.c_int, [C.c_int, c_float_p, c_float_p, C.c_uint32, C.c_int, c_float_p, c_float_p]),
    "frog_get_points2_subset": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, c_float_p]),
    "frog_set_hard_links": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t, C.c_float]),
    "frog_residual_sums": (C.c_int, [C.c_void_p]),
    "frog_get_error_map": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(FrogGridInfo), c_float_p, C.c_size_t]),
    "frog_comm_buffer": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "frog_update_stats_local": (C.c_int, [C.c_void_p]),
    "frog_stats_publish": (C.c_int, [C.c_void_p]),
    "frog_transform_points_local": (C.c_int, [C.c_void_p, C.c_int]),
    "frog_linear_step_local": (C.c_int, [C.c_void_p]),
    "frog_energy_read": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "frog_bounds_local": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "frog_deformable_setup_bounds": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, C.POINTER(FrogGridInfo)]),
    "frog_deformable_phase_a": (C.c_int, [C.c_void_p, C.c_float]),
    "frog_deformable_phase_b": (C.c_int, [C.c_void_p]),
    "frog_deformable_phase_c": (C.c_int, [C.c_void_p, c_double_p]),
    "frog_get_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "frog_comm_unpack_slab": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32]),
    "frog_comm_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "frog_create_seconds": (C.c_int, [C.c_void_p, c_double_p, C.POINTER(C.c_int)]),
    "frog_lattice_reallocations": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "frog_transform_points_slab": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32]),
    "frog_comm_unpack_slab_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32]),
    "frog_step_finish": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "frog_step_speculate": (C.c_int, [C.c_void_p]),
    "frog_cull_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "frog_cull_stats_linear": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "frog_test_stray_points": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "frog_test_cull_ranges": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "frog_test_em_refit": (C.c_int, [C.c_void_p, C.c_int]),
    "frog_test_inlier_probability": (C.c_int, [C.c_int, c_float_p, c_float_p, C.c_size_t, c_float_p, c_float_p]),
    "frog_test_inlier_weight_pair": (C.c_int, [C.c_int, c_float_p, c_float_p, C.c_float, c_float_p, C.c_size_t, c_float_p,
                                              C.POINTER(C.c_ubyte)]),
    "frog_test_bspline_weights": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "frog_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "frog_profile_read": (C.c_int, [C.c_void_p, C.POINTER(FrogKernelTime), C.c_int]),
}

HOST_SYMBOLS = {
    "frog_pairs_read": (C.c_void_p, [C.c_char_p, C.POINTER(C.c_int)]),
    "frog_pairs_write": (C.c_int, [C.c_void_p, C.c_char_p]),
    "frog_pairs_free": (None, [C.c_void_p]),
    "frog_pairs_model": (None, [C.c_void_p, C.POINTER(FrogModel)]),
    "frog_pairs_num_pairs": (C.c_uint64, [C.c_void_p]),
    "frog_pairs_num_points": (C.c_uint64, [C.c_void_p]),
    "frog_pairs_num_images": (C.c_uint32, [C.c_void_p]),
    "frog_pairs_num_blocks": (C.c_uint32, [C.c_void_p]),
    "frog_host_threads": (C.c_int, []),
    "frog_pairs_block": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16),
                                   C.POINTER(C.c_uint32), C.POINTER(c_u32_p), C.POINTER(c_u32_p)]),
    "frog_pairs_from_arrays": (C.c_void_p, [C.c_uint32, c_u32_p, c_float_p, c_float_p, C.c_uint32,
                                            C.POINTER(C.c_uint16), C.POINTER(C.c_uint16),
                                            C.POINTER(C.c_uint64), c_u32_p, c_u32_p]),
    "frog_synth_defaults": (None, [C.POINTER(FrogSynthParams)]),
    "frog_synth_generate": (C.c_void_p, [C.POINTER(FrogSynthParams)]),
    "frog_pairs_append_points": (C.c_int, [C.c_void_p, C.c_uint32, c_float_p, C.c_uint32]),
    "frog_pairs_set_points": (C.c_int, [C.c_void_p, C.c_uint32, c_float_p]),
    "frog_volume_read": (C.c_void_p, [C.c_char_p, C.POINTER(C.c_int)]),
    "frog_volume_free": (None, [C.c_void_p]),
    "frog_volume_view": (None, [C.c_void_p, C.POINTER(FrogVolume)]),
    "frog_volume_range": (C.c_int, [C.POINTER(FrogVolume), c_double_p, c_double_p]),
    "frog_volume_write": (C.c_int, [C.c_char_p, C.POINTER(FrogVolume)]),
    "frog_keypoints_read": (C.c_void_p, [C.c_char_p, C.POINTER(C.c_int)]),
    "frog_keypoints_free": (None, [C.c_void_p]),
    "frog_keypoints_count": (C.c_uint32, [C.c_void_p]),
    "frog_keypoints_view": (None, [C.c_void_p, C.POINTER(FrogKeypoints)]),
    "frog_keypoints_select": (C.c_int, [C.c_void_p, c_u32_p, C.c_uint32]),
    "frog_keypoints_write": (C.c_int, [C.c_char_p, C.POINTER(FrogKeypoints)]),
    "frog_transform_read": (C.c_void_p, [C.c_char_p, C.POINTER(C.c_int)]),
    "frog_transform_free": (None, [C.c_void_p]),
    "frog_transform_num_links": (C.c_uint32, [C.c_void_p]),
    "frog_transform_links": (C.POINTER(FrogChainLink), [C.c_void_p]),
    "frog_volume_geometry": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), c_double_p, c_double_p]),
    "frog_nifti_write": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), c_double_p, c_double_p, C.c_uint32, c_float_p]),
    "frog_run_schedule": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(FrogSchedulePlan), C.POINTER(FrogScheduleResult)]),
}

# include/frog_comm.h (libfrog_comm.so: RCCL; loaded only by hosts that shard over GPUs from C)
COMM_SYMBOLS = {
    "frog_comm_create_rccl": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "frog_comm_create_loopback": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "frog_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "frog_comm_create_rank": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.POINTER(C.c_void_p)]),
    "frog_comm_create_shm": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "frog_comm_set_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "frog_comm_destroy_all": (None, [C.c_int, C.POINTER(C.c_void_p)]),
    "frog_comm_bind": (C.c_int, [C.c_void_p, C.c_void_p, c_u32_p]),
    "frog_comm_all_gather_xyz2": (C.c_int, [C.c_void_p]),
    "frog_comm_slab": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "frog_comm_all_gather_slab": (C.c_int, [C.c_void_p]),
    "frog_comm_gather_points": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32]),
    "frog_comm_all_reduce": (C.c_int, [C.c_void_p, C.c_int]),
    "frog_comm_all_reduce_bounds": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "frog_comm_barrier": (C.c_int, [C.c_void_p]),
    "frog_comm_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "frog_comm_timing_read": (C.c_int, [C.c_void_p, c_double_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
}


def _load(name, symbols):
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C frog_amd/csrc`). frog_amd has no fallback for its native libraries.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for sym, (res, args) in symbols.items():
        fn = getattr(lib, sym)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_hip = None
_host = None
_comm = None


def comm_lib():
    """libfrog_comm.so (maps RCCL): only for hosts that shard over GPUs through include/frog_comm.h."""
    global _comm
    if _comm is None:
        hip_lib()
        _comm = _load("libfrog_comm.so", COMM_SYMBOLS)
    return _comm


def hip_lib():
    global _hip
    if _hip is None:
        # FROG_HIP_LIB: another build of the SAME library (kernel tuning experiments); relative to lib/
        _hip = _load(os.environ.get("FROG_HIP_LIB", "libfrog_hip.so"), HIP_SYMBOLS)
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()                       # libfrog_host.so links against it
        _host = _load("libfrog_host.so", HOST_SYMBOLS)
    return _host


class FrogError(RuntimeError):
    def __init__(self, code, where):
        msg = hip_lib().frog_last_error()
        super().__init__(f"{where}: status {code}: {msg.decode() if msg else ''}")
        self.code = code


def check(code, where):
    if code != FROG_OK:
        raise FrogError(code, where)

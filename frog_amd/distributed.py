"""One process per GPU: images sharded across ranks, collectives over RCCL.

The reference parallelises every loop over images (omp parallel-for,
registration/imageGroup.cxx:239,572,912,1067).  Here each rank owns a contiguous
range of images, balanced by half-link count, and the places where the
reference's loops read ANOTHER image's state become collectives
(``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in tests):

  transformPoints      -> all-gather of the xyz2 rows            (12 B x P)
  updateStats          -> all-reduce(sum) of the EM table, rows of other ranks zero
  updateLinear...      -> all-reduce(sum) of (sum w2 d2, sum w2)
  setupDeformable...   -> all-reduce(max) of the bounding box (as [max, -min])
  updateDeformable...  -> all-reduce(sum) of the proposed-coefficient sums (3 G f64:
                          the cross-image mean of imageGroup.cxx:400-432), then one
                          all-reduce(sum) of (energy sums, oversize count)

The numeric work is behind an *engine* with the split-phase entry points of
include/frog_hip.h; ``HipEngine`` is the only engine this package provides
(no CPU engine: the product path needs the HIP library).
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import check


def plan_shards(row_ptr, point_offset, world_size):
    """Contiguous image ranges, one per rank, balanced by half-link count.

    Every rank gets at least one image; requires n_images >= world_size.
    """
    po = np.asarray(point_offset, dtype=np.int64)
    n_images = len(po) - 1
    if world_size < 1 or n_images < world_size:
        raise ValueError(f"cannot shard {n_images} images over {world_size} ranks")
    links = np.asarray(row_ptr, dtype=np.int64)[po]            # cumulative half-links at image boundaries
    total = int(links[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        i = int(np.searchsorted(links, target, side="left"))
        # nearest boundary, but leave room for the ranks on both sides
        if i > 0 and abs(links[i - 1] - target) <= abs(links[min(i, n_images)] - target):
            i -= 1
        i = max(i, bounds[-1] + 1)
        i = min(i, n_images - (world_size - r))
        bounds.append(i)
    bounds.append(n_images)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


class _DeviceArray:
    """__cuda_array_interface__ view of a raw device pointer owned by libfrog_hip."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


class HipEngine:
    """Split-phase entry points of libfrog_hip.so for one context (one GPU)."""

    def __init__(self, pairs, options, device, image_range):
        import torch
        self._torch = torch
        self._lib = _abi.hip_lib()
        self.pairs = pairs
        self.device = device
        ctx = C.c_void_p()
        b, e = image_range
        check(self._lib.frog_create(C.byref(pairs.model), C.byref(options), device, b, e, C.byref(ctx)), "frog_create")
        self._ctx = ctx
        self.image_begin, self.image_end = b, e
        self.n_images = pairs.n_images
        # run on torch's current stream so that torch.distributed orders against our kernels
        torch.cuda.set_device(device)
        check(self._lib.frog_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "frog_set_stream")
        # A context that owns every image never exchanges coordinates: leave its xyz2 table
        # unexported, so that the library may swap buffers instead of copying (frog_transform_points).
        self.xyz2 = None
        self.pt_begin, self.pt_end = int(pairs.point_offset[b]), int(pairs.point_offset[e])
        if (b, e) != (0, pairs.n_images):
            self.xyz2, _ = self._buffer(_abi.FROG_BUF_XYZ2, "<f4", 3)
        self.em, _ = self._buffer(_abi.FROG_BUF_EM, "<f4", 4)
        self.energy, _ = self._buffer(_abi.FROG_BUF_ENERGY, "<f8", 1)
        self.gridsum = None

    def close(self):
        if self._ctx:
            self._lib.frog_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _buffer(self, which, typestr, width):
        ptr, nbytes, rb, re = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        check(self._lib.frog_comm_buffer(self._ctx, which, C.byref(ptr), C.byref(nbytes), C.byref(rb), C.byref(re)),
              "frog_comm_buffer")
        item = int(typestr[2:])
        n = nbytes.value // (item * width)
        shape = (n, width) if width > 1 else (n,)
        t = self._torch.as_tensor(_DeviceArray(ptr.value, shape, typestr), device=f"cuda:{self.device}")
        return t, (rb.value, re.value)

    # split-phase calls ------------------------------------------------------------
    def linear_init(self, anchor):
        check(self._lib.frog_linear_init(self._ctx, (C.c_float * 3)(*anchor)), "frog_linear_init")

    def transform_points_local(self, apply):
        check(self._lib.frog_transform_points_local(self._ctx, int(apply)), "frog_transform_points_local")

    def update_stats_local(self):
        check(self._lib.frog_update_stats_local(self._ctx), "frog_update_stats_local")

    def stats_publish(self):
        check(self._lib.frog_stats_publish(self._ctx), "frog_stats_publish")

    def linear_step_local(self):
        check(self._lib.frog_linear_step_local(self._ctx), "frog_linear_step_local")

    def linear_step(self):
        """The whole step for a context that owns the whole group (frog_linear_step): returns E."""
        e = C.c_double()
        check(self._lib.frog_linear_step(self._ctx, C.byref(e)), "frog_linear_step")
        return e.value

    def energy_read(self):
        e, nb = C.c_double(), C.c_double()
        check(self._lib.frog_energy_read(self._ctx, C.byref(e), C.byref(nb)), "frog_energy_read")
        return e.value, nb.value

    def bounds_local(self):
        mn, mx = (C.c_double * 3)(), (C.c_double * 3)()
        check(self._lib.frog_bounds_local(self._ctx, mn, mx), "frog_bounds_local")
        return list(mn), list(mx)

    def deformable_setup_bounds(self, level, mins, maxs):
        info = _abi.FrogGridInfo()
        check(self._lib.frog_deformable_setup_bounds(self._ctx, level, (C.c_double * 3)(*mins),
                                                     (C.c_double * 3)(*maxs), C.byref(info)),
              "frog_deformable_setup_bounds")
        self.gridsum, _ = self._buffer(_abi.FROG_BUF_GRIDSUM, "<f8", 1)
        return info

    def unpack_slab(self, slab, slot_rows, rows, self_rank):
        """frog_comm_unpack_slab: the other ranks' rows of a gathered slab (a torch tensor on this device) into xyz2."""
        if getattr(self, "_row_begin_key", None) != tuple(rows):
            rb = [r[0] for r in rows] + [rows[-1][1]]
            self._row_begin = (C.c_uint64 * len(rb))(*rb)
            self._row_begin_key = tuple(rows)
        check(self._lib.frog_comm_unpack_slab(self._ctx, C.c_void_p(slab.data_ptr()), int(slot_rows), len(rows), self._row_begin,
                                              int(self_rank)), "frog_comm_unpack_slab")

    def phase_a(self, alpha):
        check(self._lib.frog_deformable_phase_a(self._ctx, alpha), "frog_deformable_phase_a")

    def phase_b(self):
        check(self._lib.frog_deformable_phase_b(self._ctx), "frog_deformable_phase_b")

    def phase_c(self):
        e = C.c_double()
        check(self._lib.frog_deformable_phase_c(self._ctx, C.byref(e)), "frog_deformable_phase_c")
        return e.value

    def count_inliers(self):
        arr = (_abi.FrogCounts * self.n_images)()
        check(self._lib.frog_count_inliers(self._ctx, arr), "frog_count_inliers")
        return arr

    def make_tensor(self, values, dtype):
        return self._torch.tensor(values, dtype=dtype, device=f"cuda:{self.device}")

    # read-back used by tests / writers
    def matrix(self, image):
        m = np.empty(16, np.float64)
        check(self._lib.frog_get_linear(self._ctx, image, m.ctypes.data_as(_abi.c_double_p)), "frog_get_linear")
        return m.reshape(4, 4)

    def grid(self, image, k):
        info = _abi.FrogGridInfo()
        check(self._lib.frog_get_grid(self._ctx, image, k, C.byref(info), None, 0), "frog_get_grid")
        g = info.dims[0] * info.dims[1] * info.dims[2]
        c = np.empty((g, 3), np.float32)
        check(self._lib.frog_get_grid(self._ctx, image, k, C.byref(info), c.ctypes.data_as(_abi.c_float_p), 3 * g),
              "frog_get_grid")
        return info, c

    def num_grids(self):
        return self._lib.frog_num_grids(self._ctx)

    def points(self):
        """(xyz, xyz2) of every point in the model's order (rows of other ranks: this rank's replica)."""
        n = int(self.pairs.point_offset[-1])
        xyz, xyz2 = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        check(self._lib.frog_get_points(self._ctx, xyz.ctypes.data_as(_abi.c_float_p),
                                        xyz2.ctypes.data_as(_abi.c_float_p)), "frog_get_points")
        return xyz, xyz2

    def set_points2(self, xyz2):
        """xyz2 of every point (the model's order), e.g. another context's replica."""
        a = np.ascontiguousarray(xyz2, np.float32)
        check(self._lib.frog_set_points2(self._ctx, a.ctypes.data_as(_abi.c_float_p)), "frog_set_points2")

    def cull_stats(self):
        """(lists built, half-links in the last list, half-links owned): frog_cull_stats."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self._lib.frog_cull_stats(self._ctx, C.byref(a), C.byref(b), C.byref(c)), "frog_cull_stats")
        return a.value, b.value, c.value

    def cull_stats_kind(self, kind):
        """cull_stats() of the list the sweep `kind` ("sweep_deformable" / "sweep_linear") walks."""
        if kind == "sweep_deformable":
            return self.cull_stats()
        if kind == "sweep_linear" and hasattr(self._lib, "frog_cull_stats_linear"):
            a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
            check(self._lib.frog_cull_stats_linear(self._ctx, C.byref(a), C.byref(b), C.byref(c)), "frog_cull_stats_linear")
            return a.value, b.value, c.value
        return 0, 0, self.cull_stats()[2]

    def profile_enable(self, on=True):
        """True/1: every kernel group; 2: the half-link sweeps only (cheap); False/0: off."""
        check(self._lib.frog_profile_enable(self._ctx, int(on)), "frog_profile_enable")

    def profile_read(self, reset=True):
        arr = (_abi.FrogKernelTime * len(_abi.FROG_K_NAMES))()
        check(self._lib.frog_profile_read(self._ctx, arr, int(reset)), "frog_profile_read")
        return {n: (arr[i].ms_total, arr[i].launches) for i, n in enumerate(_abi.FROG_K_NAMES)}


class NativeComm:
    """The collectives of include/frog_comm.h (libfrog_comm.so: RCCL called from C, on the context's stream) for one
    rank of a one-process-per-GPU run.  The unique id travels over ``torch.distributed``; afterwards an iteration costs a
    handful of ctypes calls instead of three ``torch.distributed`` collectives (tens of microseconds of host time each --
    more than a rank's kernels take at 8 GPUs).  ``create`` returns None when any rank could not set it up or when the
    new communicator fails a known-answer all-reduce / all-gather: the caller then keeps the ``torch.distributed``
    collectives.  Since round 4 ``bench.py`` no longer mixes the two (its native host runs the whole loop in C,
    ``frog_run_schedule``, and its torch host uses ``torch.distributed`` throughout); this class remains for hosts that
    want the Python loop with the C collectives, and for ``tests/test_gpu_cli_and_shards.py``.  Inside a torch process
    libfrog_comm.so's ``librccl.so.1`` resolves to the RCCL build torch ships (same SONAME, already mapped: one instance,
    the image's rccl.h; the entry points used -- unique id, init rank, all-reduce, broadcast, group start/end, destroy --
    have had one ABI since NCCL 2.4)."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle

    @classmethod
    def create(cls, engine, shards, point_offset, rank, world_size, dist, device):
        import os
        import torch
        ok, lib, h = 1, None, C.c_void_p()
        try:
            lib = C.CDLL(os.path.join(_abi.LIB_DIR, "libfrog_comm.so"))
            for name in ("frog_comm_unique_id", "frog_comm_create_rank", "frog_comm_bind", "frog_comm_set_rows",
                         "frog_comm_all_gather_xyz2", "frog_comm_all_reduce", "frog_comm_all_reduce_bounds", "frog_comm_barrier"):
                getattr(lib, name).restype = C.c_int
            lib.frog_comm_destroy_all.restype = None
        except OSError:
            ok = 0
        ident = [None]
        if rank == 0 and ok:
            buf = (C.c_ubyte * 128)()
            ok = int(lib.frog_comm_unique_id(buf) == 0)
            ident = [bytes(buf)] if ok else [None]
        dist.broadcast_object_list(ident, src=0)
        # ncclCommInitRank is itself a collective: every rank must know that every other rank will call it (a rank that
        # could not load the library would otherwise leave the others waiting inside it)
        ready = torch.tensor([1 if (ok and ident[0] is not None) else 0], dtype=torch.int32, device=f"cuda:{device}")
        dist.all_reduce(ready, op=dist.ReduceOp.MIN)
        if int(ready.item()) != 1:
            return None
        if ok and ident[0] is not None:
            idb = (C.c_ubyte * 128).from_buffer_copy(ident[0])
            ok = int(lib.frog_comm_create_rank(world_size, rank, idb, device, C.byref(h)) == 0)
            if ok:
                ib = (C.c_uint32 * (world_size + 1))(*([s[0] for s in shards] + [shards[-1][1]]))
                ok = int(lib.frog_comm_bind(h, engine._ctx, ib) == 0)
            if ok:
                po = np.asarray(point_offset, dtype=np.int64)
                rows = (C.c_uint64 * (world_size + 1))(*([int(po[s[0]]) for s in shards] + [int(po[shards[-1][1]])]))
                ok = int(lib.frog_comm_set_rows(h, rows) == 0)
            if ok:
                ok = int(lib.frog_comm_barrier(h) == 0)          # a first collective on the new communicator, awaited
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{device}")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            if h:
                arr = (C.c_void_p * 1)(h)
                lib.frog_comm_destroy_all(1, arr)
            return None
        self = cls(lib, h)
        # Known-answer check of the two collectives the iterations depend on, before they are adopted: a communicator that
        # comes up but moves wrong data (two RCCL builds in one process, a header / library skew) must not carry a run.
        good = 1
        try:
            good = int(self._known_answers(engine, shards, point_offset, rank, world_size, torch))
        except Exception:                                   # noqa: BLE001 -- any failure here means "do not adopt"
            good = 0
        flag = torch.tensor([good], dtype=torch.int32, device=f"cuda:{device}")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            import sys
            print(f"[frog] rank {rank}: libfrog_comm collectives failed their known-answer check; "
                  f"staying with torch.distributed", file=sys.stderr, flush=True)
            self.close()
            return None
        return self

    def _known_answers(self, engine, shards, point_offset, rank, world_size, torch):
        """all-reduce(sum) of (rank + 1, 2^rank, -rank, 0.5) over the energy buffer and an all-gather in which rank r's
        rows hold r + 1: both results are known in closed form.  Buffers are restored afterwards."""
        keep = engine.energy.clone()
        engine.energy.copy_(torch.tensor([rank + 1.0, 2.0 ** rank, -float(rank), 0.5], dtype=torch.float64,
                                         device=engine.energy.device)[:engine.energy.numel()])
        self.all_reduce(_abi.FROG_BUF_ENERGY)
        n = world_size
        want = torch.tensor([n * (n + 1) / 2.0, 2.0 ** n - 1.0, -n * (n - 1) / 2.0, 0.5 * n], dtype=torch.float64,
                            device=engine.energy.device)[:engine.energy.numel()]
        ok = bool(torch.equal(engine.energy, want))
        engine.energy.copy_(keep)
        if engine.xyz2 is not None:
            po = np.asarray(point_offset, dtype=np.int64)
            rows = [(int(po[b]), int(po[e])) for b, e in shards]
            keep = engine.xyz2.clone()
            engine.xyz2.zero_()
            b, e = rows[rank]
            engine.xyz2[b:e] = float(rank + 1)
            self.all_gather_xyz2()
            for r, (rb, re_) in enumerate(rows):
                ok = ok and bool((engine.xyz2[rb:re_] == float(r + 1)).all())
            engine.xyz2.copy_(keep)
        torch.cuda.synchronize()
        return ok

    def _check(self, rc, what):
        if rc:
            raise RuntimeError(f"{what} failed ({rc}): {_abi.hip_lib().frog_last_error().decode()}")

    def all_gather_xyz2(self):
        self._check(self._lib.frog_comm_all_gather_xyz2(self._h), "frog_comm_all_gather_xyz2")

    def all_reduce(self, which):
        self._check(self._lib.frog_comm_all_reduce(self._h, which), "frog_comm_all_reduce")

    def all_reduce_bounds(self, mn, mx):
        a, b = (C.c_double * 3)(*mn), (C.c_double * 3)(*mx)
        self._check(self._lib.frog_comm_all_reduce_bounds(self._h, a, b), "frog_comm_all_reduce_bounds")
        return list(a), list(b)

    def close(self):
        if self._h:
            arr = (C.c_void_p * 1)(self._h)
            self._lib.frog_comm_destroy_all(1, arr)
            self._h = None


class ShardedImageGroup:
    """ImageGroup's methods (imageGroup.cxx) over image shards, one rank per GPU.

    ``engine`` implements the split-phase calls for this rank's images and exposes
    the collective buffers as torch tensors (``xyz2`` [P,3] f32, ``em`` [nI,4] f32,
    ``energy`` [4] f64, ``gridsum`` [3G] f64 after a lattice exists).  ``shards`` is
    the list of (image_begin, image_end) per rank and ``point_offset`` the model's
    point offsets.  With world_size 1 no collective is issued.
    """

    def __init__(self, engine, shards, point_offset, rank, world_size, group=None, native=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.engine = engine
        self.native = native            # NativeComm: the collectives through libfrog_comm.so instead of torch.distributed
        self.shards = list(shards)
        self.po = np.asarray(point_offset, dtype=np.int64)
        self.rank, self.world_size, self.group = rank, world_size, group
        self.proxy_em = None
        self.linearIterations = 50
        self.deformableLevels = 3
        self.deformableIterations = 200
        self.deformableAlpha = 0.02
        self.linearInitializationAnchor = (0.5, 0.5, 0.5)
        self.statIntervalUpdate = 10
        self.measures = []
        self.gridsPerLevel = []
        self.setup_seconds = []
        self.lattices = []
        # device time of every collective (events on the stream the collectives are ordered against), summed per kind by
        # comm_summary(); costs two event records per collective, so only on request
        self.time_comm = False
        self._comm_events = []
        self._comm_calls = {}
        self.comm_ms = {}
        # ncclAllGather's in-place form (send buffer = this rank's slice of the receive buffer) is what the RCCL path uses;
        # gloo gets a private copy of the rank's rows
        self.inplace_gather = self.multi and self._dist.get_backend(group) == "nccl"

    @property
    def multi(self):
        return self.world_size > 1

    COMM_SAMPLE = 8         # every eighth collective of a kind is timed (two event records each: at 8 GPUs an iteration
                            # is 0.3 ms and has three collectives -- timing all of them would be a tenth of what it measures)

    def _collective(self, kind, fn):
        if not self.time_comm:
            return fn()
        n = self._comm_calls.get(kind, 0)
        self._comm_calls[kind] = n + 1
        if n % self.COMM_SAMPLE:
            return fn()
        a = self._torch.cuda.Event(enable_timing=True); b = self._torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self._comm_events.append((kind, a, b))

    def comm_summary(self):
        """{collective kind: {"sampled_ms": device ms of the timed calls, "sampled": how many were timed, "calls": all of
        them, "avg_ms", "est_total_ms" = avg_ms * calls}} (synchronises)."""
        self._torch.cuda.synchronize()
        for kind, a, b in self._comm_events:
            e = self.comm_ms.setdefault(kind, {"sampled_ms": 0.0, "sampled": 0})
            e["sampled_ms"] += a.elapsed_time(b); e["sampled"] += 1
        self._comm_events = []
        for kind, e in self.comm_ms.items():
            e["calls"] = self._comm_calls.get(kind, 0)
            e["avg_ms"] = e["sampled_ms"] / max(1, e["sampled"])
            e["est_total_ms"] = e["avg_ms"] * e["calls"]
        return self.comm_ms

    # -- the six methods -----------------------------------------------------------
    def setupLinearTransforms(self):
        self.engine.linear_init(self.linearInitializationAnchor)

    def transformPoints(self, apply=False):
        self.engine.transform_points_local(apply)
        if not self.multi:
            return
        if self.native:
            self._collective("all_gather_xyz2", self.native.all_gather_xyz2)
            return
        rows = [(int(self.po[ib]), int(self.po[ie])) for ib, ie in self.shards]
        if len({e - b for b, e in rows}) == 1 and rows[0][0] == 0 and all(a[1] == b[0] for a, b in zip(rows, rows[1:])):
            # equal shards: one all-gather (ncclAllGather over xGMI) straight into the replica
            b, e = rows[self.rank]
            # in place on RCCL: the rank's rows are already where ncclAllGather wants them (sendbuff == recvbuff +
            # rank * sendcount, the form FSDP uses for its flat parameters); a private copy for other backends
            mine = self.engine.xyz2[b:e] if self.inplace_gather else self.engine.xyz2[b:e].clone()
            self._collective("all_gather_xyz2", lambda: self._dist.all_gather_into_tensor(
                self.engine.xyz2[:rows[-1][1]], mine, group=self.group))
            return
        # ragged shards: still ONE collective -- every rank contributes its rows padded to the longest
        # shard, the gathered slab is unpacked into the replica (world_size small device copies).  One
        # all-gather instead of world_size broadcasts: on xGMI each collective launch costs tens of
        # microseconds, comparable to a rank's whole share of an iteration at 8 GPUs.
        xyz2 = self.engine.xyz2
        longest = max(e - b for b, e in rows)
        key = (longest, xyz2.dtype, xyz2.device)
        if getattr(self, "_gather_key", None) != key:
            self._gather_in = self._torch.zeros((longest,) + tuple(xyz2.shape[1:]), dtype=xyz2.dtype, device=xyz2.device)
            self._gather_out = self._torch.empty((self.world_size * longest,) + tuple(xyz2.shape[1:]), dtype=xyz2.dtype,
                                                 device=xyz2.device)
            self._gather_key = key
        b, e = rows[self.rank]
        self._gather_in[:e - b].copy_(xyz2[b:e])
        self._collective("all_gather_xyz2", lambda: self._dist.all_gather_into_tensor(self._gather_out, self._gather_in, group=self.group))
        if hasattr(self.engine, "unpack_slab") and self.world_size <= 64:
            # one launch for all the other ranks' rows (a copy per rank is world_size - 1 launches and as many trips
            # through the Python / torch dispatch per iteration: at 8 ranks more host time than a rank's kernels take)
            self.engine.unpack_slab(self._gather_out, longest, rows, self.rank)
            return
        for r, (rb, re_) in enumerate(rows):
            if r != self.rank and re_ > rb:
                xyz2[rb:re_].copy_(self._gather_out[r * longest:r * longest + (re_ - rb)])

    def updateStats(self):
        self.engine.update_stats_local()
        if self.proxy_em is not None and not self.multi:
            # single-process proxy of one rank (bench.py --shard-of): the other ranks' rows of the mixture table, which the
            # all-reduce below would bring, from a table handed in
            b, e = self.engine.image_begin, self.engine.image_end       # (the engine runs on torch's current stream: ordered)
            self.engine.em[:b] = self.proxy_em[:b]
            self.engine.em[e:] = self.proxy_em[e:]
        if self.multi and self.native:
            self._collective("all_reduce_em", lambda: self.native.all_reduce(_abi.FROG_BUF_EM))
        elif self.multi:
            self._collective("all_reduce_em", lambda: self._dist.all_reduce(self.engine.em, op=self._dist.ReduceOp.SUM, group=self.group))
        self.engine.stats_publish()

    def updateLinearTransforms(self):
        if not self.multi and hasattr(self.engine, "linear_step") and (self.engine.image_begin, self.engine.image_end) == (0, self.engine.n_images):
            return self.engine.linear_step()        # one rank, every image: the entry point that also queues the transform
        self.engine.linear_step_local()
        if self.multi:
            self._reduce_energy()
        return self.engine.energy_read()[0]

    def _reduce_energy(self):
        if self.native:
            self._collective("all_reduce_energy", lambda: self.native.all_reduce(_abi.FROG_BUF_ENERGY))
        else:
            self._collective("all_reduce_energy", lambda: self._dist.all_reduce(self.engine.energy, op=self._dist.ReduceOp.SUM, group=self.group))

    def setupDeformableTransforms(self, level):
        import time
        t0 = time.perf_counter()
        info = self._setup(level)
        self.setup_seconds.append(time.perf_counter() - t0)      # host side only: the device work is queued, not awaited
        # one entry per lattice: its size and the accepted iterations taken on it (bench.py prices an iteration's
        # algorithmic bytes with the lattices that were really built)
        dims = [int(d) for d in getattr(info, "dims", ())]       # (test engines return their own kind of record)
        self.lattices.append({"level": int(level), "dims": dims,
                              "control_points": dims[0] * dims[1] * dims[2] if len(dims) == 3 else 0, "iterations": 0})
        return info

    def _setup(self, level):
        mn, mx = self.engine.bounds_local()
        if self.multi and self.native:
            mn, mx = self.native.all_reduce_bounds(mn, mx)
        elif self.multi:
            t = self.engine.make_tensor(list(mx) + [-v for v in mn], self._torch.float64)
            self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)
            v = t.cpu().tolist()
            mx, mn = v[:3], [-x for x in v[3:]]
        return self.engine.deformable_setup_bounds(level, mn, mx)

    def updateDeformableTransforms(self, alpha):
        self.engine.phase_a(alpha)
        if self.multi and self.native:
            self._collective("all_reduce_gridsum", lambda: self.native.all_reduce(_abi.FROG_BUF_GRIDSUM))
        elif self.multi:
            self._collective("all_reduce_gridsum", lambda: self._dist.all_reduce(self.engine.gridsum, op=self._dist.ReduceOp.SUM, group=self.group))
        self.engine.phase_b()
        if self.multi:
            self._reduce_energy()
        return self.engine.phase_c()

    def countInliers(self):
        return self.engine.count_inliers()

    def _global_rank(self, r):
        if self.group is None:
            return r
        return self._dist.get_global_rank(self.group, r)

    # -- run(), imageGroup.cxx:31-157 ------------------------------------------------
    def run(self):
        self.measures, self.gridsPerLevel = [], []
        self.setupLinearTransforms()
        self.transformPoints()
        for it in range(self.linearIterations):
            if it % self.statIntervalUpdate == 0:
                self.updateStats()
            e = self.updateLinearTransforms()
            self.transformPoints()
            self.measures.append(float(np.float32(e)))
        self.transformPoints(True)
        for level in range(self.deformableLevels):
            self.gridsPerLevel.append(self.run_level(level, self.deformableIterations))
        return self.measures

    def run_level(self, level, iterations):
        """One deformable level with the regrid / alpha-halving state machine (:78-128)."""
        self.setupDeformableTransforms(level)
        self.transformPoints()
        n_grids, alpha, n_diffeo, it = 1, np.float32(self.deformableAlpha), 0, 0
        while it < iterations:
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
                continue
            n_diffeo += 1
            self.transformPoints()
            self.measures.append(float(np.float32(e)))
            self.lattices[-1]["iterations"] += 1
            it += 1
        self.transformPoints(True)
        return n_grids

#!/usr/bin/env python3
"""bench.py -- registration iterations/s of the FROG groupwise hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2]): 100 synthetic images x 20 000 keypoints,
~50 M pairs (10^8 half-links), default solver flags (-g 100, -gd 1, -si 10).
One "step" = one registration iteration of ImageGroup::run's loops
(registration/imageGroup.cxx:54-66 and :88-121): updateStats every 10th
iteration, update{Linear,Deformable}Transforms, transformPoints.  The K timed
steps keep the reference's default mix 50 : 200 : 200 : 200 (linear : level 0 :
level 1 : level 2), i.e. n_lin = round(K*50/650) linear iterations followed by
three deformable levels sharing the rest; lattice set-up, re-basing and any
regrid the diffeomorphism guard triggers are inside the timed region.  Warm-up =
linear set-up + W linear iterations.  Inputs are resident in HBM before the
timed region starts.

Hosts.  The timed loop is C (`frog_run_schedule`, include/frog_host.h: the loop bodies of ImageGroup::run over the C ABI,
collectives from C through libfrog_comm.so) -- host "native".  The Python loop over `torch.distributed`
(frog_amd/distributed.py) is host "torch", kept as the fallback and for the single-GPU proxies (--shard-of).

N > 1.  The process that is started (by hand, or as a rank by torch.distributed.run) never touches the GPU: it runs a
sequence of ATTEMPTS, each a set of fresh child processes (one per rank it is responsible for) under a timeout:
  preflight   RCCL known answers (all-reduce, in-place all-gather on the library's own buffers) + latencies
  native      the timed schedule through the C loop, collectives over RCCL (host-staged shared memory if the preflight failed)
  torch       the timed schedule through the Python loop over torch.distributed
The fastest attempt whose ranks end with bit-identical replicas is the line; every attempt's outcome is in it
(`hosts_tried`).  A hang is a timed-out child with its stderr in the line, not a silent time-out of the whole run.

Prints ONE JSON line on rank 0 (see README / DESIGN.md section 6).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# BASELINE.json configs[1], [2] and [4] as generator / solver parameters.  configs[3] is [2]
# sharded over 8 GPUs: `--gpus 8`.
CONFIGS = {
    2: dict(images=20, points=20000, pairs_per_block=10526.0, partners=0, levels=0, what="20 images, ~2 M pairs, linear only"),
    3: dict(images=100, points=20000, pairs_per_block=10101.0, partners=0, levels=3, what="100 images, ~50 M pairs, linear + 3 deformable levels"),
    5: dict(images=500, points=20000, pairs_per_block=16667.0, partners=60, levels=5,
            what="500 images (~60 partner images each), ~2.5e8 pairs, linear + 5 deformable levels, -gd 1"),
}
COMM_KINDS = ["all_gather_xyz2", "all_reduce_em", "all_reduce_energy", "all_reduce_gridsum"]


def schedule(k, levels=3):
    """Split K timed iterations into (linear, [level0, ...]) in the reference's default mix 50 : 200 per level."""
    if levels == 0:
        return k, []
    n_lin = max(1, int(round(k * 50.0 / (50.0 + 200.0 * levels)))) if k > 1 else k
    rest = k - n_lin
    per = [rest // levels] * levels
    per[-1] += rest - levels * (rest // levels)
    return n_lin, per


def lattice_reallocations(lib, ctx):
    """lattice buffers a set-up had to allocate inside the timed loops (0 = max_levels_hint's head-room held; include/frog_hip.h)"""
    n = C.c_int()
    return n.value if lib.frog_lattice_reallocations(ctx, C.byref(n)) == 0 else None


def cpu_baseline(pairs, n_lin, per_level, stat_interval, repeats=3):
    """Oracle (CPU restatement, OpenMP over images like the reference) on a bounded
    sample of the same workload: `repeats` stats refreshes, linear iterations and
    deformable iterations per level (about ten seconds of CPU work on the box's cores;
    the mean of each kind), extrapolated to the timed schedule."""
    from frog_amd import _abi
    from oracle.oracle_api import OracleGroup, lib
    # as many threads as CPUs the process may use (a container's quota counts: 256 threads on a 16-CPU share are throttled
    # together and the baseline reads five times too slow), OMP_NUM_THREADS respected when it asks for fewer
    cores = min(lib().frogo_get_max_threads(), _abi.usable_cpus())
    lib().frogo_set_threads(cores)
    ref = OracleGroup(pairs.model, _abi.FrogOptions.default())
    ref.setup_stats()
    ref.linear_init()
    ref.transform_points()

    def timed(fn):
        t = time.perf_counter(); fn(); return time.perf_counter() - t
    t_stats, t_lin = [], []
    for _ in range(repeats):
        t_stats.append(timed(ref.update_stats))
        t_lin.append(timed(lambda: (ref.linear_step(), ref.transform_points())))
    ref.transform_points(True)
    t_def = []
    for level in range(len(per_level)):
        if per_level[level] == 0:
            t_def.append(0.0)
            continue
        ref.deformable_setup(level, _abi.FrogGridInfo())
        ref.transform_points()
        ts = [timed(lambda: (ref.deformable_step(0.02), ref.transform_points())) for _ in range(repeats)]
        t_def.append(sum(ts) / len(ts))
        ref.transform_points(True)
    t_stats, t_lin = sum(t_stats) / len(t_stats), sum(t_lin) / len(t_lin)
    refreshes = -(-n_lin // stat_interval) + sum(-(-n // stat_interval) for n in per_level)
    total = n_lin * t_lin + sum(n * t for n, t in zip(per_level, t_def)) + refreshes * t_stats
    k = n_lin + sum(per_level)
    return {"value": k / total, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"{repeats} updateStats + {repeats} linear iterations + {repeats} deformable iterations per level on the "
                      "same pairs, timed with the oracle (oracle/frog_oracle.cpp, OpenMP over images), the mean of each kind "
                      "extrapolated to the timed schedule",
            "seconds": {"stats_refresh": t_stats, "linear_iteration": t_lin, "deformable_iteration": t_def}}


# ---- the line -------------------------------------------------------------------------------------------------------------

def roofline_and_iteration(prof, cull_def, cull_lin, l_own, p_own, i_own, n_lin, lattices, elapsed, levels):
    """`roofline` (dominant kernel) and `iteration` objects of the line from the live kernel times and the lists' statistics.
    prof: {kernel group: [ms, launches]}; cull_*: (lists built, half-links in the last list, half-links owned);
    lattices: [{"level", "dims", "iterations"}]."""
    from frog_amd import _abi
    dom = "sweep_deformable" if prof["sweep_deformable"][1] else "sweep_linear"
    ms, launches = prof[dom]
    # Units one launch processes: with a culling list the steady-state sweep WALKS the listed
    # half-links only (the others are decided by a certified distance bound and never touched); the one launch per list
    # that writes the list ("sweep_build" / "sweep_linear_build") walks every half-link.  `achieved` / `frac` price every
    # launch at the half-links it walked (20 B each: 8 B link + 12 B gathered xyz2) + 12 B per owned point, over the time
    # of ALL those launches, list-writing ones included.  The same launches priced at the reference's bytes for all L
    # half-links (what upstream's loop touches per iteration) are `frac_algorithmic_equiv`: a saving of work,
    # not a bandwidth.
    listed = cull_def if dom == "sweep_deformable" else cull_lin
    build_name = {"sweep_deformable": "sweep_build", "sweep_linear": "sweep_linear_build"}[dom]
    bms, bl = prof.get(build_name, [0.0, 0])
    walked = float(listed[1]) if listed[0] else float(l_own)
    all_ms, all_launches = ms + bms, launches + bl
    walked_bytes = launches * (20.0 * walked + 12.0 * p_own) + bl * (20.0 * l_own + 12.0 * p_own)
    alg_bytes = 20.0 * l_own + 12.0 * p_own
    achieved = walked_bytes / (all_ms * 1e-3) / 1e9 if all_launches else 0.0
    # `bound`: what the counters say bounds the kernel (profiles/) -- the fused sweep is held by the
    # texture path, vector-ALU issue and LDS together (each ~70 % busy), while its HBM traffic is a quarter of the peak;
    # `frac` stays the fraction of the HBM roofline at the algorithmic bytes of what it walks, as the metric asks.
    roofline = {"bound": "ta+valu+lds", "bound_note": "on-chip units (texture path, vector ALU, LDS ~70 % busy each); frac is against the "
                "HBM roofline at the algorithmic bytes of the half-links walked",
                "frac_rule": "r03: walked half-links, list-writing launches included",
                "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": None,
                "avg_launch_ms": all_ms / all_launches if all_launches else None, "launches": int(all_launches),
                "walked_half_links_per_steady_launch": walked, "half_links_owned": l_own, "points_owned": p_own,
                "bytes_per_steady_launch": 20.0 * walked + 12.0 * p_own,
                "steady_launches": {"launches": int(launches), "avg_launch_ms": ms / launches if launches else None},
                "frac_algorithmic_equiv": (alg_bytes * all_launches / (all_ms * 1e-3) / 1e9 / 8000.0) if all_launches else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "timed_launches": "HIP events on the launch's own dispatch packet, live in the timed region: every launch (--kernel-times) / "
                                  "every list-writing launch and one steady launch in four, avg_launch_ms = launches x the mean of the "
                                  "timed ones (default; FROG_BENCH_PROFILE=2 times every sweep: 1.6 % slower line, same averages)"}
    if bl:
        roofline["list_writing_launches"] = {"launches": int(bl), "avg_launch_ms": bms / bl, "walks": "every half-link"}
    if listed[0]:
        roofline["culling"] = {"lists_built": int(listed[0]), "listed_half_links": int(listed[1]),
                               "listed_fraction": listed[1] / max(listed[2], 1)}
    # Whole-iteration fraction = what the metric pays for: the algorithmic bytes of every timed iteration (SURVEY 8d:
    # B_lin = 20 L + 36 P, B_def = 20 L + 48 P + 104 I G with the I G of the lattices that were really built) over the
    # timed region's wall time.  `..._walked` prices the half-links at the ones the sweeps walked.
    b_lin = 20.0 * l_own + 36.0 * p_own
    def_iters = [(la["dims"][0] * la["dims"][1] * la["dims"][2], la["iterations"]) for la in lattices]
    b_total = n_lin * b_lin + sum(n * (20.0 * l_own + 48.0 * p_own + 104.0 * i_own * g) for g, n in def_iters)
    walked_lin = float(cull_lin[1]) if cull_lin[0] else float(l_own)
    walked_def = float(cull_def[1]) if cull_def[0] else float(l_own)
    b_walked = n_lin * (20.0 * walked_lin + 36.0 * p_own) + sum(n * (20.0 * walked_def + 48.0 * p_own + 104.0 * i_own * g) for g, n in def_iters)
    iteration = {"algorithmic_bytes": b_total, "elapsed_s": elapsed, "iteration_frac": b_total / elapsed / 8e12,
                 "iteration_frac_walked": b_walked / elapsed / 8e12,
                 "lattices": [{"level": la["level"], "dims": la["dims"], "iterations": la["iterations"]} for la in lattices],
                 "formula": "(n_lin (20 L + 36 P) + sum over lattices n (20 L + 48 P + 104 I G)) / elapsed / 8 TB/s, per rank"}
    # HBM bytes per launch of that kernel from the PMC counters (FETCH_SIZE, WRITE_SIZE collected in their own
    # rocprofv3 passes by scripts/profile_bench.sh and corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE
    # doubled on gfx950).  Counters cannot be read from inside this process, so the committed measurement of
    # the same workload is reported; null when there is none for this workload / shard size, or when the device
    # sources have changed since it was taken (profiles/hbm_traffic.json "measured_at").
    src_hash = _abi.device_source_hash()
    roofline["device_source_hash"] = src_hash
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
            doc = json.load(fh)
        # one entry per workload (cfg 3, cfg 5, ...: scripts/merge_traffic.py), keyed on the workload AND on the device sources
        # the counters were collected with: stale after any kernel change
        for tr in doc.get("entries", [doc]):
            if (tr.get("kernel") == dom and tr.get("half_links_owned", tr.get("half_links_per_launch")) == l_own
                    and tr.get("measured_at") == src_hash):
                roofline["traffic"] = tr["traffic_bytes_per_launch"]
                roofline["traffic_source"] = tr.get("source")
                roofline["traffic_measured_at"] = tr.get("measured_at")
                break
    except (OSError, ValueError, KeyError):
        pass
    return roofline, iteration


def workload_of(args):
    cfg = CONFIGS[args.config]
    images = args.images or cfg["images"]
    points = args.points or cfg["points"]
    ppb = args.pairs_per_block or cfg["pairs_per_block"]
    levels = cfg["levels"] if args.levels is None else args.levels
    steps = args.steps
    if steps is None:
        steps = {2: 50, 3: 650, 5: 130}[args.config] if args.levels is None else 50 + 200 * levels
    partners = cfg["partners"] if images == cfg["images"] else 0
    return images, points, ppb, levels, steps, partners


def make_line(args, world, k, elapsed, host, collectives, pairs, levels, n_lin, per_level, grids, final_e, roofline, iteration,
              prof, phase_k, phase_s, setup_seconds):
    images, points = pairs.n_images, int(pairs.point_offset[1] - pairs.point_offset[0])
    return {
        "metric": "registration iterations/sec (linear+deformable)",
        "value": k / elapsed,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": k,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / k,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "host": host,
        "collectives": collectives,
        "config": {"workload": f"BASELINE.json configs[{ {2: 1, 3: 2, 5: 4}[args.config] }]: {images} images x {points} keypoints, "
                               f"{pairs.n_pairs} pairs ({pairs.n_half_links} half-links), linear + {levels} deformable "
                               f"levels, -g 100 -gd 1 -si 10",
                   "schedule": {"linear": n_lin, "deformable_per_level": per_level},
                   "parallelism": f"images sharded over {world} GPU(s)", "grids_per_level": grids,
                   "final_E": final_e,
                   **({"mode": "-exact 1 (frog_options::reference_order): the reference's own order and arithmetic, bit-equal to the oracle"}
                      if getattr(args, "exact", False) else {})},
        "roofline": roofline,
        "iteration": iteration,
        "kernels_ms": {n: {"total_ms": v[0], "launches": int(v[1])} for n, v in prof.items() if v[1]},
        "kernels_ms_by_phase": phase_k,
        "phase_iterations_per_s": {
            "linear": n_lin / phase_s["linear"],
            **{f"level{l}": per_level[l] / phase_s[f"level{l}"] for l in range(levels) if per_level[l]}},
        "setup_seconds": setup_seconds,
    }


# ---- host "native": the C loop ------------------------------------------------------------------------------------------

class Rendezvous:
    """What the ranks of one attempt exchange outside the collectives: small files in a directory all of them see
    (one node).  No torch, no sockets."""

    def __init__(self, directory, attempt, rank, world):
        self.dir, self.attempt, self.rank, self.world = directory, attempt, rank, world

    def path(self, name):
        return os.path.join(self.dir, f"a{self.attempt}_{name}")

    def put(self, name, data):
        tmp = self.path(name) + f".tmp{os.getpid()}"
        with open(tmp, "wb") as fh:
            fh.write(data)
        os.replace(tmp, self.path(name))

    def get(self, name, timeout=120.0):
        t0 = time.time()
        while not os.path.exists(self.path(name)):
            if time.time() - t0 > timeout:
                raise TimeoutError(f"rank {self.rank}: {name} never appeared in {self.dir}")
            self.check_peers()
            time.sleep(0.005)
        with open(self.path(name), "rb") as fh:
            return fh.read()

    def fail(self, why):
        """Tell the other ranks of this attempt that this one is gone (they stop waiting for it at once)."""
        try:
            self.put(f"failed_{self.rank}", str(why).encode()[:2000])
        except OSError:
            pass

    def check_peers(self):
        for r in range(self.world):
            if r != self.rank and os.path.exists(self.path(f"failed_{r}")):
                raise RuntimeError(f"rank {r} of this attempt failed: " + open(self.path(f"failed_{r}"), "rb").read().decode(errors="replace")[:500])

    def all_ready(self, what, timeout=120.0):
        """Every rank has reached `what` (or one has failed): called in front of a call that blocks in a collective --
        ncclCommInitRank waits for every rank, for ever -- so that a rank that could not even create its context costs the
        others a second, not the attempt's whole time-out."""
        self.put(f"{what}_{self.rank}", b"1")
        for r in range(self.world):
            self.get(f"{what}_{r}", timeout)

    def gather_json(self, name, obj, timeout=120.0):
        """Every rank contributes obj; returns the list in rank order (on every rank)."""
        self.put(f"{name}_{self.rank}", json.dumps(obj).encode())
        return [json.loads(self.get(f"{name}_{r}", timeout)) for r in range(self.world)]


def create_native_comm(rdv, transport, ctx, shards, point_offset, device):
    """The rank's communicator of include/frog_comm.h, bound to ctx: RCCL (ncclCommInitRank, the id through the
    rendezvous directory) or the host-staged shared-memory one."""
    from frog_amd import _abi
    cl = _abi.comm_lib()
    h = C.c_void_p()
    rank, world = rdv.rank, rdv.world
    rdv.all_ready("context")            # every rank has its context (on its own device): now the communicator
    if transport == "rccl":
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            _abi.check(cl.frog_comm_unique_id(buf), "frog_comm_unique_id")
            rdv.put("rccl_id", bytes(buf))
        ident = (C.c_ubyte * 128).from_buffer_copy(rdv.get("rccl_id"))
        _abi.check(cl.frog_comm_create_rank(world, rank, ident, device, C.byref(h)), "frog_comm_create_rank")
    elif transport == "shm":
        name = f"frogbench_{os.path.basename(rdv.dir)}_{rdv.attempt}".encode()
        _abi.check(cl.frog_comm_create_shm(world, rank, name, device, C.byref(h)), "frog_comm_create_shm")
    else:
        raise SystemExit(f"unknown transport {transport}")
    ib = (C.c_uint32 * (world + 1))(*([s[0] for s in shards] + [shards[-1][1]]))
    _abi.check(cl.frog_comm_bind(h, ctx, ib), "frog_comm_bind")
    rows = (C.c_uint64 * (world + 1))(*([int(point_offset[s[0]]) for s in shards] + [int(point_offset[shards[-1][1]])]))
    _abi.check(cl.frog_comm_set_rows(h, rows), "frog_comm_set_rows")
    _abi.check(cl.frog_comm_barrier(h), "frog_comm_barrier")
    return cl, h


def native_plan(args, n_lin, per_level, profile, time_comm, warmup=None):
    from frog_amd import _abi
    plan = _abi.FrogSchedulePlan()
    plan.plan_bytes, plan.result_bytes = C.sizeof(_abi.FrogSchedulePlan), C.sizeof(_abi.FrogScheduleResult)
    plan.warmup_linear = args.warmup if warmup is None else warmup
    plan.linear = n_lin
    plan.n_levels = len(per_level)
    for l, n in enumerate(per_level):
        plan.per_level[l] = n
    plan.stat_interval = 10
    plan.deformable_alpha = 0.02
    plan.anchor[0] = plan.anchor[1] = plan.anchor[2] = 0.5
    plan.profile = int(os.environ.get("FROG_BENCH_PROFILE", profile))        # experiment switch (0: no event on any launch, 2: on every sweep)
    plan.time_comm = int(time_comm)
    return plan


def run_native(args, rank, world, local_rank, transport, rdv):
    """One rank of the timed schedule through frog_run_schedule.  Returns the line on rank 0 (None elsewhere)."""
    from frog_amd import _abi
    from frog_amd.pairs import Pairs
    from frog_amd.distributed import plan_shards
    import numpy as np
    lib, host = _abi.hip_lib(), _abi.host_lib()
    if lib.frog_device_count() < 1:
        raise SystemExit("no HIP device: bench.py measures the HIP path only")
    images, points, ppb, levels, steps, partners_per_image = workload_of(args)
    t0 = time.perf_counter()
    pairs = Pairs.synthetic(images, points, ppb, seed=1, partners_per_image=partners_per_image)
    t_gen = time.perf_counter() - t0
    n_lin, per_level = schedule(steps, levels)
    opts = _abi.FrogOptions.default(max_levels_hint=levels)      # what a host knows before it starts: -dl
    if args.exact:
        opts.reference_order = 1
    proxy = None
    if args.shard_of:
        if world != 1:
            raise SystemExit("--shard-of is a single-process proxy")
        shards = plan_shards(pairs.row_ptr, pairs.point_offset, args.shard_of[1])
        shards = [shards[args.shard_of[0]]]
        args.kernel_times = os.environ.get("FROG_PROXY_SWEEPS_ONLY") != "1"      # (the proxy's point is the per-kernel table)
        if args.proxy_partners == "registered":
            # what the other ranks would hand over: the whole group registered by one context (default schedule), its final
            # coordinates and mixtures
            full = C.c_void_p()
            _abi.check(lib.frog_create(C.byref(pairs.model), C.byref(opts), local_rank, 0, pairs.n_images, C.byref(full)), "frog_create")
            fl, fp = 50, [200] * levels             # the reference's default schedule, -li 50 -di 200
            res = _abi.FrogScheduleResult()
            _abi.check(host.frog_run_schedule(full, None, C.byref(native_plan(args, fl, fp, 0, False, warmup=0)), C.byref(res)), "frog_run_schedule")
            n_pts = int(pairs.point_offset[-1])
            xyz2 = np.empty((n_pts, 3), np.float32)
            _abi.check(lib.frog_get_points(full, None, xyz2.ctypes.data_as(_abi.c_float_p)), "frog_get_points")
            em = np.zeros((pairs.n_images, 4), np.float32)
            for i in range(pairs.n_images):
                _abi.check(lib.frog_get_em(full, i, em[i].ctypes.data_as(_abi.c_float_p)), "frog_get_em")
            lib.frog_destroy(full)
            proxy = (xyz2, em)
    else:
        shards = plan_shards(pairs.row_ptr, pairs.point_offset, world)
    t0 = time.perf_counter()
    ctx = C.c_void_p()
    b, e = shards[rank]
    _abi.check(lib.frog_create(C.byref(pairs.model), C.byref(opts), local_rank, b, e, C.byref(ctx)), "frog_create")
    t_create = time.perf_counter() - t0
    cs, nsel = (C.c_double * 3)(), C.c_int()
    _abi.check(lib.frog_create_seconds(ctx, cs, C.byref(nsel)), "frog_create_seconds")
    create_breakdown = {"layout_build_host": round(cs[0], 4), "allocate_upload": round(cs[1], 4), "selection_replay": round(cs[2], 4),
                        "selections_replayed": int(nsel.value),
                        "note": "frog_create: host-side layout build (all cores) / device allocations + uploads / reservoir selections "
                                "of the whole run replayed ahead on the side stream (one block per image: latency-bound, off the timed region)"}
    cl, comm = None, None
    if world > 1:
        cl, comm = create_native_comm(rdv, transport, ctx, shards, pairs.point_offset, local_rank)
    # mode 3: HIP events on every list-writing sweep and on one steady sweep in four (frog_profile_enable): an event-carrying
    # launch starts ~6 us late and holds its successor back -- with all of them timed that was 1.6 % of this line
    plan = native_plan(args, n_lin, per_level, 1 if args.kernel_times else 3, world > 1)
    if proxy is not None:
        plan.proxy_xyz2 = proxy[0].ctypes.data
        plan.proxy_em = proxy[1].ctypes.data
    res = _abi.FrogScheduleResult()
    if os.environ.get("FROG_BENCH_PREWARM"):         # experiment: N census passes (state untouched) right before the schedule
        counts = (_abi.FrogCounts * pairs.n_images)()
        _abi.check(lib.frog_linear_init(ctx, (C.c_float * 3)(0.5, 0.5, 0.5)), "frog_linear_init")
        _abi.check(lib.frog_transform_points(ctx, 0), "frog_transform_points")
        for _ in range(int(os.environ["FROG_BENCH_PREWARM"])):
            _abi.check(lib.frog_count_inliers(ctx, counts), "frog_count_inliers")
    t_call = time.perf_counter()
    _abi.check(host.frog_run_schedule(ctx, comm, C.byref(plan), C.byref(res)), "frog_run_schedule")
    if os.environ.get("FROG_BENCH_TRACE_HOST"):
        print(f"[bench] create done -> schedule call {t_call - t0 - t_create:.4f} s; schedule call {time.perf_counter() - t_call:.4f} s of which timed {res.elapsed_s:.4f} s", file=sys.stderr)

    def cull(fn):
        a, bb, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _abi.check(fn(ctx, C.byref(a), C.byref(bb), C.byref(c)), "frog_cull_stats")
        return a.value, bb.value, c.value
    cull_def, cull_lin = cull(lib.frog_cull_stats), cull(lib.frog_cull_stats_linear)
    prof = {n: [res.kernels[i].ms_total, int(res.kernels[i].launches)] for i, n in enumerate(_abi.FROG_K_NAMES)}
    tags = ["linear"] + [f"level{l}" for l in range(levels)]
    phase_s = {t: res.phase_s[i] for i, t in enumerate(tags) if i == 0 or per_level[i - 1]}
    was_proxy = bool(args.shard_of)
    if args.shard_of and not args.kernel_times:
        args.shard_of = None            # no per-kernel table to print: the plain line
    if args.kernel_times:
        phase_k = {t: {n: {"ms": round(res.kernels_by_phase[i][j].ms_total, 4), "launches": int(res.kernels_by_phase[i][j].launches)}
                       for j, n in enumerate(_abi.FROG_K_NAMES) if res.kernels_by_phase[i][j].launches}
                   for i, t in enumerate(tags) if t in phase_s}
    else:
        phase_k = {"all": {n: {"ms": round(v[0], 4), "launches": v[1]} for n, v in prof.items() if v[1]}}
    lattices = [{"level": int(la.level), "dims": [int(d) for d in la.dims], "iterations": int(la.iterations)}
                for la in res.lattices[:res.n_lattices]]
    comm_ms = None
    if world > 1:
        comm_ms = {}
        for j, kind in enumerate(COMM_KINDS):
            if res.comm_calls[j]:
                avg = res.comm_ms[j] / max(1, res.comm_sampled[j])
                comm_ms[kind] = {"sampled_ms": res.comm_ms[j], "sampled": int(res.comm_sampled[j]), "calls": int(res.comm_calls[j]),
                                 "avg_ms": avg, "est_total_ms": avg * res.comm_calls[j]}
    mine = {"elapsed": res.elapsed_s, "hash": int(res.replica_hash), "kernel_ms": sum(v[0] for v in prof.values()),
            "sweep_ms": prof["sweep_deformable"][0] + prof["sweep_linear"][0] + prof["sweep_build"][0] + prof["sweep_linear_build"][0],
            "comm_est_ms": sum(v["est_total_ms"] for v in (comm_ms or {}).values()), "final_E": res.final_E}
    everyone = rdv.gather_json("result", mine) if world > 1 else [mine]
    elapsed = max(x["elapsed"] for x in everyone)
    line = None
    if rank == 0:
        po, rp = pairs.point_offset, pairs.row_ptr
        p_own = int(po[e]) - int(po[b])
        l_own = int(rp[int(po[e])]) - int(rp[int(po[b])])
        roofline, iteration = roofline_and_iteration(prof, cull_def, cull_lin, l_own, p_own, e - b, n_lin, lattices, elapsed, levels)
        k = n_lin + sum(per_level)
        collectives = "none" if world == 1 else {"rccl": "libfrog_comm (RCCL from C)", "shm": "libfrog_comm (host-staged shared memory)"}[transport]
        line = make_line(args, world, k, elapsed, "native (frog_run_schedule, C loop)", collectives, pairs, levels, n_lin, per_level,
                         [int(g) for g in res.grids_per_level[:levels]], res.final_E, roofline, iteration, prof, phase_k, phase_s,
                         {"generate": t_gen, "create": t_create, "create_breakdown": create_breakdown,
                          "lattice_setups": [la.setup_host_s for la in res.lattices[:res.n_lattices]],
                          "lattice_reallocations_in_the_loops": lattice_reallocations(lib, ctx)})
        if args.shard_of:
            line["proxy"] = (f"rank {args.shard_of[0]} of {args.shard_of[1]} on one GPU: owns images {shards[0]}, no collective, "
                             + ("other ranks' coordinates and mixtures: those a full run of the default schedule ends with, standing still"
                                if proxy is not None else "other ranks' coordinates static where the set-up left them, their mixtures zero")
                             + "; `value` is NOT the metric")
            line["proxy_ms_per_iteration"] = {ph: {n: v["ms"] / max(1, (n_lin if ph == "linear" else per_level[int(ph[5:])]))
                                                   for n, v in ks.items()} for ph, ks in phase_k.items()}
        if world > 1:
            line["comm_ms"] = comm_ms
            # every collective issued inside the timed region over its iterations: 2 per deformable iteration + 1 per linear one
            # + 1 per statistics refresh and a few per lattice set-up since round 5 (include/frog_hip.h frog_comm_mode; 3 + 2 before)
            line["collectives_per_iteration"] = round(sum(v["calls"] for v in comm_ms.values()) / max(1, k), 3)
            line["replicas_identical"] = len({x["hash"] for x in everyone}) == 1
            line["ranks"] = {"elapsed_s": [x["elapsed"] for x in everyone], "kernel_ms_total": [x["kernel_ms"] for x in everyone],
                             "sweep_ms_total": [x["sweep_ms"] for x in everyone], "comm_est_ms_total": [x["comm_est_ms"] for x in everyone],
                             "note": "per rank over the timed region: wall time, sum of the bracketed kernels' device time "
                                     "(all groups with --kernel-times, else the half-link sweeps), estimated device time of the collectives"}
        if world == 1 and not args.no_cpu_baseline and args.config == 3 and not was_proxy:
            line["cpu_baseline"] = cpu_baseline(pairs, n_lin, per_level, 10)
    # (with the CPU baseline, i.e. on the full default line: the A/B scripts pass --no-cpu-baseline and get neither)
    end_to_end_pending = (world == 1 and rank == 0 and line is not None and not was_proxy and not args.no_end_to_end
                          and (args.end_to_end or (not args.no_cpu_baseline and args.config == 3)))
    if comm:
        if world > 1:
            rdv.gather_json("done", {"rank": rank})          # nobody tears its communicator down while another still reduces
        arr = (C.c_void_p * 1)(comm)
        cl.frog_comm_destroy_all(1, arr)
    lib.frog_destroy(ctx)
    if (world == 1 and rank == 0 and line is not None and not was_proxy and not args.exact and not args.no_exact_mode
            and (args.exact_mode or (not args.no_cpu_baseline and args.config == 3))):
        line["exact_mode"] = exact_mode(args, pairs, levels, n_lin, per_level, local_rank)
    if end_to_end_pending:
        line["end_to_end"] = end_to_end(pairs, levels)
    return line, (world == 1 or len({x["hash"] for x in everyone}) == 1)


def exact_mode(args, pairs, levels, n_lin, per_level, local_rank):
    """The same schedule once more through bin/frog -exact 1 = frog_options::reference_order: the device path in the reference's
    own order and arithmetic, whose every per-point sum, gradient image, lattice, matrix and coordinate is bit-equal to the oracle
    (tests/test_gpu_reference_order.py) -- the mode inside north_star's literal "transform parameters within 1e-4".  Measured after
    the timed region, like cpu_baseline; same frog_run_schedule loop, same warm-up, no kernel events.  When the line's schedule is
    a short one (the round-end driver's 20 steps: a lattice's chains are built for five iterations), the mode is also timed over the
    reference's default schedule for the configuration (`default_schedule`: what DESIGN.md 2c and the README quote)."""
    from frog_amd import _abi
    lib, host = _abi.hip_lib(), _abi.host_lib()

    def run(lin, per):
        opts = _abi.FrogOptions.default(max_levels_hint=levels)
        opts.reference_order = 1
        ctx = C.c_void_p()
        t0 = time.perf_counter()
        _abi.check(lib.frog_create(C.byref(pairs.model), C.byref(opts), local_rank, 0, pairs.n_images, C.byref(ctx)), "frog_create")
        t_create = time.perf_counter() - t0
        try:
            res = _abi.FrogScheduleResult()
            _abi.check(host.frog_run_schedule(ctx, None, C.byref(native_plan(args, lin, per, 0, False)), C.byref(res)), "frog_run_schedule")
        finally:
            lib.frog_destroy(ctx)
        k = lin + sum(per)
        tags = ["linear"] + [f"level{l}" for l in range(levels)]
        return {"value": k / res.elapsed_s, "unit": "iterations/s", "ms_per_step": 1e3 * res.elapsed_s / k, "steps": k,
                "final_E": res.final_E, "grids_per_level": [int(g) for g in res.grids_per_level[:levels]],
                "phase_iterations_per_s": {t: (lin if i == 0 else per[i - 1]) / res.phase_s[i]
                                           for i, t in enumerate(tags) if (lin if i == 0 else per[i - 1]) and res.phase_s[i] > 0},
                "create_s": t_create}
    try:
        out = run(n_lin, per_level)
        out["mode"] = ("bin/frog -exact 1 (frog_options::reference_order): the reference's own order and arithmetic on the device, "
                       "bit-equal to the oracle; the per-lattice chain builds (DESIGN.md 2c) are inside the timed region")
        d_lin, d_per = schedule(50 + 200 * levels, levels)           # -li 50 -di 200 per level
        if args.config == 3 and (d_lin, list(d_per)) != (n_lin, list(per_level)):
            out["default_schedule"] = run(d_lin, d_per)
        return out
    except Exception as exc:                # noqa: BLE001 -- an add-on: the measured line is printed whatever happens here
        return {"error": f"{type(exc).__name__}: {exc}"}


def end_to_end(pairs, levels):
    """`bin/frog pairs.bin` as a user runs it, wall clock of the whole process: read pairs.bin, frog_create, the reference's default
    schedule (-li 50 -di 200; -dl as the configuration), write transforms/ + the csv / json reports.  After the timed region, in a
    child process, with the context above already destroyed.  pairs.bin goes to a temporary directory (memory-backed when
    /dev/shm exists: the file's 0.4 GB are then read at memory speed -- a disk would add its own time)."""
    import re, shutil, subprocess, tempfile
    exe = os.path.join(ROOT, "bin", "frog")
    if not os.path.exists(exe):
        return {"error": "bin/frog not built"}
    try:
        d = tempfile.mkdtemp(prefix="frog_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    except OSError as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}
    # the child is a plain user's run: no profiler or preloaded library of a wrapped parent, and the [timing] lines of the host
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROF", "HSA_TOOLS_"))}
    env["FROG_TIMING"] = "1"
    try:
        t0 = time.perf_counter()
        pairs.write(os.path.join(d, "pairs.bin"))
        t_write = time.perf_counter() - t0
        size = os.path.getsize(os.path.join(d, "pairs.bin"))
        cmd = [exe, "pairs.bin", "-q", "1", "-dl", str(levels)]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=900, env=env)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": f"bin/frog exited {r.returncode}", "stderr": r.stderr[-500:]}
        out = {"command": "bin/frog pairs.bin -q 1 -dl %d" % levels, "wall_s": round(wall, 3), "pairs_bin_bytes": size,
               "pairs_bin_written_in_s": round(t_write, 3)}
        m = re.search(r"Iteration loops : (\d+) iterations in ([0-9.eE+-]+)s", r.stdout)
        if m:
            out["iterations"] = int(m.group(1)); out["loops_s"] = float(m.group(2))
            out["iterations_per_s_of_the_whole_process"] = round(int(m.group(1)) / wall, 1)
        m = re.search(r"Total time : ([0-9.eE+-]+)s", r.stdout)
        if m:
            out["total_time_printed_s"] = float(m.group(1))
        out["host_timing_s"] = {k.strip(): float(v) for k, v in re.findall(r"\[timing\] ([^:\n]+?) *: ([0-9.eE+-]+)s", r.stdout)}
        out["note"] = ("whole process, default schedule: reading pairs.bin, frog_create (layout build, upload, selection replay), the "
                       "iteration loops, error maps, transforms/ and reports; `value` above is the loops alone (SURVEY 8d)")
        return out
    except (subprocess.TimeoutExpired, OSError) as exc:       # an add-on: the measured line is printed whatever happens here
        return {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ---- preflight: RCCL known answers and latencies on the library's own buffers ---------------------------------------------

def run_preflight(rank, world, local_rank, transport, rdv):
    from frog_amd import _abi
    from frog_amd.pairs import Pairs
    from frog_amd.distributed import plan_shards
    import numpy as np
    lib = _abi.hip_lib()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipDeviceSynchronize.argtypes = []
    H2D, D2H = 1, 2
    pairs = Pairs.synthetic(2 * world, 4000, 1500, seed=3)
    shards = plan_shards(pairs.row_ptr, pairs.point_offset, world)
    ctx = C.c_void_p()
    _abi.check(lib.frog_create(C.byref(pairs.model), C.byref(_abi.FrogOptions.default()), local_rank, shards[rank][0], shards[rank][1],
                               C.byref(ctx)), "frog_create")
    cl, comm = create_native_comm(rdv, transport, ctx, shards, pairs.point_offset, local_rank)

    def buf(which):
        p, nb, rb, re = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        _abi.check(lib.frog_comm_buffer(ctx, which, C.byref(p), C.byref(nb), C.byref(rb), C.byref(re)), "frog_comm_buffer")
        return p, nb.value, rb.value, re.value
    ok = True
    # all-reduce(sum) of (rank + 1, 2^rank, -rank, 0.5) on FROG_BUF_ENERGY
    p, nb, _, _ = buf(_abi.FROG_BUF_ENERGY)
    src = np.array([rank + 1.0, 2.0 ** rank, -float(rank), 0.5])
    _abi.check(lib.frog_synchronize(ctx), "frog_synchronize")
    hip.hipMemcpy(p, src.ctypes.data_as(C.c_void_p), 32, H2D)
    _abi.check(cl.frog_comm_all_reduce(comm, _abi.FROG_BUF_ENERGY), "frog_comm_all_reduce")
    _abi.check(lib.frog_synchronize(ctx), "frog_synchronize")
    got = np.zeros(4)
    hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), p, 32, D2H)
    n = world
    want = np.array([n * (n + 1) / 2.0, 2.0 ** n - 1.0, -n * (n - 1) / 2.0, 0.5 * n])
    ok = ok and bool(np.array_equal(got, want))
    # in-place all-gather: rank r's rows hold r + 1
    p, nb, rb, re = buf(_abi.FROG_BUF_XYZ2)
    n_pts = nb // 12
    mine = np.zeros((n_pts, 3), np.float32)
    mine[rb:re] = rank + 1.0
    hip.hipMemcpy(p, mine.ctypes.data_as(C.c_void_p), nb, H2D)
    _abi.check(cl.frog_comm_all_gather_xyz2(comm), "frog_comm_all_gather_xyz2")
    _abi.check(lib.frog_synchronize(ctx), "frog_synchronize")
    got = np.empty((n_pts, 3), np.float32)
    hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), p, nb, D2H)
    po = pairs.point_offset
    for r, (ib, ie) in enumerate(shards):
        ok = ok and bool((got[int(po[ib]):int(po[ie])] == r + 1.0).all())
    # latencies: 50 awaited small all-reduces, 20 gathers
    lat = {}
    for name, fn, reps in (("all_reduce_32B_us", lambda: cl.frog_comm_all_reduce(comm, _abi.FROG_BUF_ENERGY), 50),
                           (f"all_gather_{nb // 1024}KB_us", lambda: cl.frog_comm_all_gather_xyz2(comm), 20)):
        _abi.check(cl.frog_comm_barrier(comm), "frog_comm_barrier")
        t0 = time.perf_counter()
        for _ in range(reps):
            _abi.check(fn(), name)
        _abi.check(lib.frog_synchronize(ctx), "frog_synchronize")
        lat[name] = 1e6 * (time.perf_counter() - t0) / reps
    # the calls the native host's loop makes (round 5: two collectives per deformable iteration, one per linear one): a linear
    # iteration whose transform goes straight into the gather's slab, the step's sums in the slots' trailers -- every rank must
    # read the same finite E and end with the same replica of the coordinates
    import zlib
    flow = {"E": None, "crc": None}
    try:
        _abi.check(lib.frog_comm_mode(ctx, 1), "frog_comm_mode")
        _abi.check(lib.frog_linear_init(ctx, (C.c_float * 3)(0.5, 0.5, 0.5)), "frog_linear_init")
        _abi.check(cl.frog_comm_gather_points(comm, 0, 0, 0), "frog_comm_gather_points")
        _abi.check(lib.frog_update_stats_local(ctx), "frog_update_stats_local")
        _abi.check(cl.frog_comm_all_reduce(comm, _abi.FROG_BUF_EM), "frog_comm_all_reduce")
        _abi.check(lib.frog_stats_publish(ctx), "frog_stats_publish")
        _abi.check(lib.frog_linear_step_local(ctx), "frog_linear_step_local")
        _abi.check(cl.frog_comm_gather_points(comm, 0, 0, 0xB), "frog_comm_gather_points")
        e = C.c_double()
        _abi.check(lib.frog_step_finish(ctx, C.byref(e)), "frog_step_finish")
        xyz2 = np.empty((n_pts, 3), np.float32)
        _abi.check(lib.frog_get_points(ctx, None, xyz2.ctypes.data_as(_abi.c_float_p)), "frog_get_points")
        flow = {"E": e.value, "crc": zlib.crc32(xyz2.tobytes())}
    except Exception as exc:            # reported, and the attempt counts as failed
        flow["error"] = str(exc)[:300]
        ok = False
    everyone = rdv.gather_json("preflight", {"ok": ok, "lat": lat, "flow": flow})
    flows = [x.get("flow", {}) for x in everyone]
    same_flow = (all(f.get("E") is not None and np.isfinite(f["E"]) for f in flows)
                 and len({f.get("E") for f in flows}) == 1 and len({f.get("crc") for f in flows}) == 1)
    if not same_flow:
        ok = False
        everyone = [dict(x, ok=False) for x in everyone]
    arr = (C.c_void_p * 1)(comm)
    rdv.gather_json("done", {"rank": rank})
    cl.frog_comm_destroy_all(1, arr)
    lib.frog_destroy(ctx)
    good = all(x["ok"] for x in everyone)
    line = ({"preflight": transport, "known_answers": good, "latencies_rank0": lat, "linear_iteration_E": flows[0].get("E"),
             "replicas_identical": len({f.get("crc") for f in flows}) == 1} if rank == 0 else None)
    return line, good


# ---- host "torch": the Python loop over torch.distributed (frog_amd/distributed.py) -------------------------------------

def run_torch(args, rank, world, local_rank, backend, rdv):
    import torch
    import torch.distributed as dist
    from frog_amd import _abi
    from frog_amd.pairs import Pairs
    from frog_amd.distributed import HipEngine, ShardedImageGroup, plan_shards

    if _abi.hip_lib().frog_device_count() < 1:
        raise SystemExit("no HIP device: bench.py measures the HIP path only")
    torch.cuda.set_device(local_rank)
    if world > 1:
        store = f"file://{rdv.path('torch_store')}" if rdv else None
        kw = dict(init_method=store, rank=rank, world_size=world) if store else {}
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"), **kw)
        else:
            dist.init_process_group(backend=backend, **kw)
    images, points, ppb, levels, steps, partners_per_image = workload_of(args)
    t0 = time.perf_counter()
    pairs = Pairs.synthetic(images, points, ppb, seed=1, partners_per_image=partners_per_image)
    t_gen = time.perf_counter() - t0
    shards = plan_shards(pairs.row_ptr, pairs.point_offset, world)
    opts = _abi.FrogOptions.default(max_levels_hint=levels)
    t0 = time.perf_counter()
    engine = HipEngine(pairs, opts, local_rank, shards[rank])
    t_create = time.perf_counter() - t0
    grp = ShardedImageGroup(engine, shards, pairs.point_offset, rank, world)
    grp.time_comm = world > 1
    n_lin, per_level = schedule(steps, levels)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    grp.setupLinearTransforms()
    grp.transformPoints()
    it = 0
    for _ in range(args.warmup):
        if it % grp.statIntervalUpdate == 0:
            grp.updateStats()
        grp.updateLinearTransforms()
        grp.transformPoints()
        it += 1
    engine.profile_enable(1 if args.kernel_times else 3)
    phase_s, phase_k = {}, {}
    prof = {n: [0.0, 0] for n in _abi.FROG_K_NAMES}

    def take(tag):
        cur = engine.profile_read(reset=True)
        phase_k[tag] = {n: {"ms": round(v[0], 4), "launches": int(v[1])} for n, v in cur.items() if v[1]}
        for n, v in cur.items():
            prof[n][0] += v[0]; prof[n][1] += v[1]
    sync()
    t_start = time.perf_counter()
    tp = t_start
    e = 0.0
    for _ in range(n_lin):
        if it % grp.statIntervalUpdate == 0:
            grp.updateStats()
        e = grp.updateLinearTransforms()
        grp.transformPoints()
        it += 1
    grp.transformPoints(True)

    def phase_end(tag, t_phase):
        if args.kernel_times:
            torch.cuda.synchronize()
        phase_s[tag] = time.perf_counter() - t_phase
        if args.kernel_times:
            take(tag)
    phase_end("linear", tp)
    grids = []
    for level in range(levels):
        if per_level[level] == 0:
            continue
        tp = time.perf_counter()
        grids.append(grp.run_level(level, per_level[level]))
        phase_end(f"level{level}", tp)
    sync()
    elapsed = time.perf_counter() - t_start
    if not args.kernel_times:
        take("all")
    engine.profile_enable(False)
    if grp.time_comm:
        grp.comm_summary()
    if grp.measures:
        e = grp.measures[-1]
    replicas_identical = True
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        if backend != "nccl":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        grp.transformPoints(True)
        torch.cuda.synchronize()
        chk = torch.stack([engine.xyz2.double().sum(), engine.xyz2.double().abs().sum(), engine.em.double().sum()])
        if backend != "nccl":
            chk = chk.cpu()
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi)) and bool(torch.isfinite(chk).all())
    if levels == 0:
        grp.transformPoints(True)
    line = None
    if rank == 0:
        own_b, own_e = shards[rank]
        po, rp = pairs.point_offset, pairs.row_ptr
        p_own = int(po[own_e]) - int(po[own_b])
        l_own = int(rp[int(po[own_e])]) - int(rp[int(po[own_b])])
        roofline, iteration = roofline_and_iteration(prof, engine.cull_stats_kind("sweep_deformable"), engine.cull_stats_kind("sweep_linear"),
                                                     l_own, p_own, own_e - own_b, n_lin, grp.lattices, elapsed, levels)
        line = make_line(args, world, n_lin + sum(per_level), elapsed, "torch (Python loop, frog_amd/distributed.py)",
                         "none" if world == 1 else f"torch.distributed/{backend}", pairs, levels, n_lin, per_level, grids, e,
                         roofline, iteration, prof, phase_k, phase_s, {"generate": t_gen, "create": t_create, "lattice_setups": grp.setup_seconds})
        if grp.comm_ms:
            line["comm_ms"] = grp.comm_ms
        if world > 1:
            line["replicas_identical"] = replicas_identical
        if world == 1 and not args.no_cpu_baseline and args.config == 3:
            line["cpu_baseline"] = cpu_baseline(pairs, n_lin, per_level, grp.statIntervalUpdate)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return line, replicas_identical


# ---- the launcher: attempts of fresh children -----------------------------------------------------------------------------

def child_command(argv, host, transport, directory, attempt):
    return [sys.executable, os.path.abspath(__file__), *argv, "--child", host, "--child-transport", transport,
            "--child-dir", directory, "--child-attempt", str(attempt)]


def attempts_plan(rehearsal):
    """[(host, transport)] in the order they are tried; FROG_BENCH_HOSTS="native,torch" etc. restricts the measured ones."""
    want = [h.strip() for h in os.environ.get("FROG_BENCH_HOSTS", "preflight,native,torch").split(",") if h.strip()]
    t_native, t_torch = ("shm", "gloo") if rehearsal else ("rccl", "nccl")
    plan = []
    for h in want:
        if h == "preflight":
            plan.append(("preflight", t_native))
        elif h == "native":
            plan.append(("native", t_native))
        elif h == "torch":
            plan.append(("torch", t_torch))
        elif h == "native-shm":
            plan.append(("native", "shm"))
    return plan


def orchestrate(args, argv, n, my_ranks, directory):
    """Runs the attempts; every participating parent (one under a plain `python bench.py --gpus N`, N under
    torch.distributed.run) walks the same list and reads the same outcome files, so all take the same decisions.
    The parent of rank 0 prints the line.  Never touches the GPU itself."""
    rehearsal = os.environ.get("FROG_BENCH_BACKEND", "nccl") != "nccl"       # one-GPU boxes: every rank on device 0, host-staged collectives
    os.makedirs(directory, exist_ok=True)
    env_base = dict(os.environ)
    env_base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    if "OMP_NUM_THREADS" not in os.environ or os.environ.get("FROG_BENCH_UNDER_TORCHRUN") == "1":
        # torch.distributed.run sets OMP_NUM_THREADS=1 when it is unset; frog_create builds its layout on the host threads
        from frog_amd._abi import usable_cpus
        env_base["OMP_NUM_THREADS"] = str(max(1, usable_cpus() // n))
    # a healthy attempt takes 5-20 s (the torch host up to two minutes more on a fresh box: its first `import torch`); the
    # worst case of a whole run -- preflight passes, the native RCCL attempt hangs, fallback -- stays near five minutes
    timeouts = {"preflight": float(os.environ.get("FROG_BENCH_PREFLIGHT_TIMEOUT", "90")),
                "native": float(os.environ.get("FROG_BENCH_ATTEMPT_TIMEOUT", "150")),
                "torch": float(os.environ.get("FROG_BENCH_ATTEMPT_TIMEOUT", "240"))}
    plan = attempts_plan(rehearsal)
    tried, lines = [], []
    rccl_ok, rccl_why = True, "the RCCL preflight failed"
    k = 0
    queue = list(plan)
    while queue:
        host, transport = queue.pop(0)
        if not rccl_ok and transport in ("rccl", "nccl"):
            tried.append({"host": host, "transport": transport, "ok": False, "skipped": rccl_why})
            continue
        t0 = time.time()
        procs = {}
        for r in my_ranks:
            env = dict(env_base, RANK=str(r), WORLD_SIZE=str(n), LOCAL_RANK=str(0 if rehearsal else r))
            for v in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE"):
                env.pop(v, None)        # the children rendezvous through files in `directory`, not through the launcher's store
            out = open(os.path.join(directory, f"a{k}_r{r}.out"), "w")
            err = open(os.path.join(directory, f"a{k}_r{r}.err"), "w")
            procs[r] = (subprocess.Popen(child_command(argv, host, transport, directory, k), env=env, stdout=out, stderr=err), out, err)
        for r, (p, out, err) in procs.items():
            left = max(1.0, timeouts[host] - (time.time() - t0))
            try:
                rc = p.wait(timeout=left)
            except subprocess.TimeoutExpired:
                p.kill()                # this exact child, nothing else
                p.wait()
                rc = -9
            out.close(); err.close()
            tmp = os.path.join(directory, f"a{k}_rc{r}.tmp")
            with open(tmp, "w") as fh:
                fh.write(str(rc))
            os.replace(tmp, os.path.join(directory, f"a{k}_rc{r}"))
        # the outcome of the attempt = every rank's return code (other parents write theirs)
        rcs, t_wait = {}, time.time()
        while len(rcs) < n and time.time() - t_wait < timeouts[host] + 60.0:
            for r in range(n):
                path = os.path.join(directory, f"a{k}_rc{r}")
                if r not in rcs and os.path.exists(path):
                    rcs[r] = int(open(path).read().strip() or "1")
            if len(rcs) < n:
                time.sleep(0.05)
        ok = len(rcs) == n and all(v == 0 for v in rcs.values())
        rec = {"host": host, "transport": transport, "ok": ok, "seconds": round(time.time() - t0, 2),
               "return_codes": [rcs.get(r) for r in range(n)]}
        line = None
        try:
            for raw in open(os.path.join(directory, f"a{k}_r0.out")):
                if raw.lstrip().startswith("{"):
                    line = json.loads(raw)
        except (OSError, ValueError):
            pass
        if host == "preflight":
            if line:
                rec.update({kk: line[kk] for kk in ("known_answers", "latencies_rank0", "linear_iteration_E", "replicas_identical") if kk in line})
            ok = ok and bool(line and line.get("known_answers"))
            rec["ok"] = ok
            if not ok:
                rccl_ok = False
                if not rehearsal and ("native", "shm") not in queue:
                    queue.append(("native", "shm"))     # still a multi-GPU run, host-staged: a curve and a diagnosis instead of nothing
        elif ok and line and line.get("replicas_identical", True):
            rec.update({"value": line["value"], "replicas_identical": line.get("replicas_identical")})
            lines.append(line)
        else:
            rec["ok"] = False
            if line:
                rec["replicas_identical"] = line.get("replicas_identical")
            if transport in ("rccl", "nccl") and (any(v == -9 for v in rcs.values()) or len(rcs) < n):
                # a rank had to be killed at the time-out: RCCL hangs on this box, and the other RCCL host would sit out its
                # own time-out the same way -- go to the host-staged transport at once
                rec["timed_out"] = True
                rccl_ok, rccl_why = False, "the previous RCCL attempt hung until its time-out"
                if not rehearsal and ("native", "shm") not in queue and not lines:
                    queue.append(("native", "shm"))
        if not rec["ok"] and "skipped" not in rec:
            for r in sorted(set(my_ranks) & {rr for rr, v in rcs.items() if v != 0} or set(my_ranks[:1])):
                try:
                    tail = open(os.path.join(directory, f"a{k}_r{r}.err")).read()[-1500:]
                    rec.setdefault("stderr_tail", {})[str(r)] = tail
                except OSError:
                    pass
        tried.append(rec)
        k += 1
        if not queue and not lines and not rehearsal and not any(t["transport"] == "shm" for t in tried):
            queue.append(("native", "shm"))             # every RCCL attempt failed, however the preflight went: the last resort
    if 0 not in my_ranks:
        return 0 if lines else 1
    if not lines:
        sys.stderr.write(json.dumps({"error": "no attempt produced a valid line", "hosts_tried": tried}, indent=1) + "\n")
        return 1
    best = max(lines, key=lambda d: d["value"])
    best["hosts_tried"] = tried
    print(json.dumps(best), flush=True)
    return 0


def run_directory(n):
    """A directory every participating parent derives alike: under torch.distributed.run from its run id and port, otherwise
    fresh."""
    import tempfile
    port, run_id = os.environ.get("MASTER_PORT"), os.environ.get("TORCHELASTIC_RUN_ID")
    if "WORLD_SIZE" in os.environ and port:
        return os.path.join(tempfile.gettempdir(), f"frog_bench_{os.getuid()}_{port}_{run_id or 'x'}_{os.getppid()}")
    return tempfile.mkdtemp(prefix="frog_bench_")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed iterations; default = the reference's schedule for the configuration (650 for config 3: "
                         "-li 50 -dl 3 -di 200), 130 for config 5")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="BASELINE.json configuration: 3 (default, the one the metric is quoted on), 2 (linear only), 5 (500 images)")
    ap.add_argument("--levels", type=int, default=None, help="deformable levels (default: the configuration's)")
    ap.add_argument("--images", type=int, default=None)
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--pairs-per-block", type=float, default=None)
    ap.add_argument("--host", choices=("native", "torch"), default="native",
                    help="N = 1: which loop drives the timed region (native: frog_run_schedule in C, the default; torch: the Python "
                         "loop).  N > 1 tries both (FROG_BENCH_HOSTS restricts)")
    ap.add_argument("--shard-of", type=int, nargs=2, metavar=("R", "N"), default=None,
                    help="single-GPU proxy of rank R of an N-GPU run: this process owns shard R of N, no collective is "
                         "issued, the other ranks' images stand still (where, see --proxy-partners); prints per-phase kernel "
                         "times of that rank's share (not a metric line)")
    ap.add_argument("--proxy-partners", choices=("registered", "static"), default="registered",
                    help="--shard-of: the other ranks' coordinates and mixtures are those a full single-context run of the "
                         "default schedule ends with (registered: every image in the common frame, as a real rank sees its "
                         "partners -- partner points lie where the own points are, the sweeps' gathers are local) or stay "
                         "where the set-up left them (static: unregistered partners, mixtures of other images zero)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the bin/frog whole-process run after the timed region (N = 1)")
    ap.add_argument("--end-to-end", action="store_true", help="run it even with --no-cpu-baseline / another --config")
    ap.add_argument("--exact", action="store_true",
                    help="time bin/frog -exact 1 = frog_options::reference_order instead of the product kernels: the device path in the "
                         "reference's own order and arithmetic, bit-equal to the oracle (DESIGN.md 2a); the line says so in config.mode")
    ap.add_argument("--no-exact-mode", action="store_true", help="skip the -exact 1 run of the same schedule after the timed region (N = 1)")
    ap.add_argument("--exact-mode", action="store_true", help="run it even with --no-cpu-baseline / another --config")
    ap.add_argument("--kernel-times", action="store_true",
                    help="HIP-event times of every kernel group, not only of the half-link sweeps (costs ~6 %% of the rate)")
    ap.add_argument("--child", choices=("preflight", "native", "torch"), default=None, help=argparse.SUPPRESS)
    ap.add_argument("--child-transport", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--child-dir", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--child-attempt", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    argv = [a for a in sys.argv[1:]]

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if args.child:                                  # a rank of one attempt
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        rdv = Rendezvous(args.child_dir, args.child_attempt, rank, world)
        try:
            # the device of this rank: LOCAL_RANK, unless the launcher gave every process a view of its own GPU only
            # (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank: one visible device, index 0)
            from frog_amd import _abi
            n_dev = _abi.hip_lib().frog_device_count()
            if n_dev >= 1 and local_rank >= n_dev:
                sys.stderr.write(f"[bench] rank {rank}: LOCAL_RANK {local_rank} but {n_dev} visible device(s): using device {local_rank % n_dev}\n")
                local_rank %= n_dev
            if args.child == "preflight":
                line, good = run_preflight(rank, world, local_rank, args.child_transport, rdv)
            elif args.child == "native":
                line, good = run_native(args, rank, world, local_rank, args.child_transport, rdv)
            else:
                line, good = run_torch(args, rank, world, local_rank, args.child_transport, rdv)
        except BaseException as exc:                # noqa: BLE001 -- whatever it was, the other ranks must not wait for this one
            rdv.fail(f"{type(exc).__name__}: {exc}")
            raise
        if line is not None:
            print(json.dumps(line), flush=True)
        raise SystemExit(0 if good else 3)

    if args.gpus > 1:
        if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
            # a rank started by torch.distributed.run: this process stays off the GPU and runs its rank of every attempt as a child
            if world != args.gpus:
                raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
            os.environ["FROG_BENCH_UNDER_TORCHRUN"] = "1"
            raise SystemExit(orchestrate(args, argv, world, [rank], run_directory(world)))
        if os.environ.get("FROG_BENCH_LAUNCH_DRY_RUN") == "1":
            d = "<run directory>"
            print(json.dumps({"attempts": [{"host": h, "transport": t, "command": child_command(argv, h, t, d, i)}
                                           for i, (h, t) in enumerate(attempts_plan(os.environ.get("FROG_BENCH_BACKEND", "nccl") != "nccl"))],
                              "ranks": list(range(args.gpus))}), flush=True)
            raise SystemExit(0)
        # plain `python bench.py --gpus N`: this process is the launcher of every rank (nothing has touched the GPU yet)
        directory = run_directory(args.gpus)
        try:
            rc = orchestrate(args, argv, args.gpus, list(range(args.gpus)), directory)
        finally:
            if os.environ.get("FROG_BENCH_KEEP_DIR") != "1":        # (under a launcher the other ranks' parents may still be reading it)
                import shutil
                shutil.rmtree(directory, ignore_errors=True)
        raise SystemExit(rc)

    if world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.host == "torch":
        if args.shard_of:
            raise SystemExit("--shard-of runs through the C loop (--host native)")
        line, _ = run_torch(args, 0, 1, local_rank, "nccl", None)
    else:
        line, _ = run_native(args, 0, 1, local_rank, None, None)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

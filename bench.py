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

Prints ONE JSON line on rank 0 (see README / DESIGN.md section 6).
"""
import argparse
import ctypes as C
import json
import os
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


def schedule(k, levels=3):
    """Split K timed iterations into (linear, [level0, ...]) in the reference's default mix 50 : 200 per level."""
    if levels == 0:
        return k, []
    n_lin = max(1, int(round(k * 50.0 / (50.0 + 200.0 * levels)))) if k > 1 else k
    rest = k - n_lin
    per = [rest // levels] * levels
    per[-1] += rest - levels * (rest // levels)
    return n_lin, per


def cpu_baseline(pairs, n_lin, per_level, stat_interval, repeats=3):
    """Oracle (CPU restatement, OpenMP over images like the reference) on a bounded
    sample of the same workload: `repeats` stats refreshes, linear iterations and
    deformable iterations per level (about ten seconds of CPU work on the box's cores;
    the mean of each kind), extrapolated to the timed schedule."""
    from frog_amd import _abi
    from oracle.oracle_api import OracleGroup, lib
    cores = lib().frogo_get_max_threads()
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


def free_port():
    """A TCP port free on 127.0.0.1 right now (the rendezvous port handed to torch.distributed.run)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(n, argv, port=None):
    """`python bench.py --gpus N ...` without a launcher around it: the command that starts the N ranks, one process
    per GPU, as children of this one (the form the driver uses itself for N > 1, README / DESIGN section 6)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__), *argv]


def self_launch(n, argv):
    """Start the ranks as FRESH child processes and pass rank 0's line through.  Called before anything in this process
    has touched the GPU (before `import torch` and before libfrog_hip is loaded): the parent only waits.  Never
    os.exec*: the children are ordinary subprocesses and this process exits with their return code."""
    import subprocess
    cmd = launch_command(n, argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    # torch.distributed.run sets OMP_NUM_THREADS=1 when it is unset; frog_create builds its layout on the host threads
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    if os.environ.get("FROG_BENCH_LAUNCH_DRY_RUN") == "1":
        print(json.dumps({"launch": cmd, "env": {k: env[k] for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS")}}), flush=True)
        return 0
    # stdout carries ONE line: rank 0's metric line.  Whatever else the children print there (gloo's connection notices in
    # rehearsals, library banners) goes to stderr.
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        out = sys.stdout if line.lstrip().startswith('{"metric"') else sys.stderr
        out.write(line)
        out.flush()
    return proc.wait()


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
    ap.add_argument("--kernel-times", action="store_true",
                    help="HIP-event times of every kernel group, not only of the half-link sweeps (costs ~6 %% of the rate)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (nothing has touched the GPU yet)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from frog_amd import _abi
    from frog_amd.pairs import Pairs
    from frog_amd.distributed import HipEngine, NativeComm, ShardedImageGroup, plan_shards

    if _abi.hip_lib().frog_device_count() < 1:
        raise SystemExit("no HIP device: bench.py measures the HIP path only")
    # Rehearsal hook for boxes with a single GPU: FROG_BENCH_BACKEND=gloo puts every rank on
    # device 0 and moves the collectives through gloo.  The driver's runs use RCCL ("nccl").
    backend = os.environ.get("FROG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend)

    cfg = CONFIGS[args.config]
    images = args.images or cfg["images"]
    points = args.points or cfg["points"]
    ppb = args.pairs_per_block or cfg["pairs_per_block"]
    levels = cfg["levels"] if args.levels is None else args.levels
    if args.steps is None:
        args.steps = {2: 50, 3: 650, 5: 130}[args.config] if args.levels is None else 50 + 200 * levels
    if args.config != 3 or args.shard_of:
        args.no_cpu_baseline = True        # the host baseline is timed on the configuration the metric is quoted on
    t0 = time.perf_counter()
    pairs = Pairs.synthetic(images, points, ppb, seed=1, partners_per_image=cfg["partners"] if images == cfg["images"] else 0)
    t_gen = time.perf_counter() - t0
    if args.shard_of:
        if world != 1:
            raise SystemExit("--shard-of is a single-process proxy")
        shards = plan_shards(pairs.row_ptr, pairs.point_offset, args.shard_of[1])
        shards = [shards[args.shard_of[0]]]
    else:
        shards = plan_shards(pairs.row_ptr, pairs.point_offset, world)
    opts = _abi.FrogOptions.default()
    partners = None
    if args.shard_of and args.proxy_partners == "registered":
        # what the other ranks would hand over: the whole group registered by one context (default schedule), its final
        # coordinates and mixtures (both engines number the points alike: the numbering depends on the model only)
        full = HipEngine(pairs, opts, local_rank, (0, pairs.n_images))
        gfull = ShardedImageGroup(full, [(0, pairs.n_images)], pairs.point_offset, 0, 1)
        gfull.deformableLevels = levels
        gfull.run()
        torch.cuda.synchronize()
        partners = (full.points()[1], full.em.clone())
        torch.cuda.synchronize()
        del gfull
        full.close()
    t0 = time.perf_counter()
    engine = HipEngine(pairs, opts, local_rank, shards[rank])
    t_create = time.perf_counter() - t0
    # N > 1 over RCCL: torch.distributed carries the collectives (torch's own RCCL).  FROG_NATIVE_COMM=1 issues them from C
    # instead (libfrog_comm.so, include/frog_comm.h, on the context's stream: ncclCommInitRank with an id carried by
    # torch.distributed) -- opt-in until a run with >= 2 GPUs has validated it; it is adopted only if every rank sets it up
    # AND it passes a known-answer all-reduce and all-gather (NativeComm.create)
    native = None
    if world > 1 and backend == "nccl" and os.environ.get("FROG_NATIVE_COMM", "0") == "1":
        native = NativeComm.create(engine, shards, pairs.point_offset, rank, world, dist, local_rank)
    grp = ShardedImageGroup(engine, shards, pairs.point_offset, rank, world, native=native)
    # per-collective device time in the line ("comm_ms", every eighth call of a kind); FROG_BENCH_TIME_COMM=1 also in rehearsals
    grp.time_comm = world > 1 and (backend == "nccl" or os.environ.get("FROG_BENCH_TIME_COMM") == "1")
    if args.shard_of:
        args.kernel_times = True

    n_lin, per_level = schedule(args.steps, levels)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up: linear set-up + W linear iterations --------------------------------
    grp.setupLinearTransforms()
    grp.transformPoints()
    if partners is not None:
        own_b, own_e = shards[rank]
        pb, pe = int(pairs.point_offset[own_b]), int(pairs.point_offset[own_e])
        mixed = engine.points()[1]              # the model's point order on both sides
        mixed[:pb] = partners[0][:pb]
        mixed[pe:] = partners[0][pe:]
        engine.set_points2(mixed)
        grp.proxy_em = partners[1]              # stands in for the all-reduce of the mixture table at every refresh
    it = 0
    for _ in range(args.warmup):
        if it % grp.statIntervalUpdate == 0:
            grp.updateStats()
        grp.updateLinearTransforms()
        grp.transformPoints()
        it += 1

    # ---- timed region: exactly K iterations --------------------------------------------
    # live HIP-event timing of the dominant kernel (the half-link sweeps); every kernel group with --kernel-times
    engine.profile_enable(1 if args.kernel_times else 2)
    phase_s = {}
    phase_k = {}
    prof = {n: [0.0, 0] for n in _abi.FROG_K_NAMES}

    def take(tag):
        # HIP-event kernel times of the phase just finished (the events are already complete:
        # the phase ended with a device synchronisation)
        cur = engine.profile_read(reset=True)
        phase_k[tag] = {n: {"ms": round(v[0], 4), "launches": int(v[1])} for n, v in cur.items() if v[1]}
        for n, v in cur.items():
            prof[n][0] += v[0]; prof[n][1] += v[1]
    sync()
    t_start = time.perf_counter()
    tp = t_start
    for _ in range(n_lin):
        if it % grp.statIntervalUpdate == 0:
            grp.updateStats()
        e = grp.updateLinearTransforms()
        grp.transformPoints()
        it += 1
    grp.transformPoints(True)
    # Phase boundaries: with --kernel-times the device is drained and the per-phase kernel times are read; otherwise nothing
    # is waited for inside the timed region that the registration itself does not wait for (every step ends with the host
    # reading its scalars, every level starts with the bounds read-back, so the host clock is at most one transform behind
    # the device): a drain + read per phase cost 50-90 us of idle GPU each, 2 % of a 20-step run.
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

    replicas_identical = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # after the timed region: every rank holds a replica of all transformed coordinates (the all-gather's product) and
        # of the mixture table; they must be the same bits on every rank, whatever carried the collectives
        grp.transformPoints(True)
        torch.cuda.synchronize()
        chk = torch.stack([engine.xyz2.double().sum(), engine.xyz2.double().abs().sum(), engine.em.double().sum()])
        if backend != "nccl":
            chk = chk.cpu()
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi)) and bool(torch.isfinite(chk).all())

    k = n_lin + sum(per_level)
    own_b, own_e = shards[rank]
    po, rp = pairs.point_offset, pairs.row_ptr
    p_own = int(po[own_e]) - int(po[own_b])
    l_own = int(rp[int(po[own_e])]) - int(rp[int(po[own_b])])
    # dominant kernel: the half-link sweep of the deformable step
    dom = "sweep_deformable" if prof["sweep_deformable"][1] else "sweep_linear"
    if levels == 0:
        grp.transformPoints(True)
    ms, launches = prof[dom]
    # Units one launch processes: with a culling list the steady-state sweep WALKS the listed
    # half-links only (the others are decided by a certified distance bound and never touched); the one launch per list
    # that writes the list ("sweep_build" / "sweep_linear_build") walks every half-link.  `achieved` / `frac` price every
    # launch at the half-links it walked (20 B each: 8 B link + 12 B gathered xyz2) + 12 B per owned point, over the time
    # of ALL those launches, list-writing ones included.  The same launches priced at the reference's bytes for all L
    # half-links (what upstream's loop touches per iteration) are `frac_algorithmic_equiv`: a saving of work,
    # not a bandwidth.
    listed = engine.cull_stats_kind(dom)                   # (lists built, half-links in the last list, half-links owned)
    build_name = {"sweep_deformable": "sweep_build", "sweep_linear": "sweep_linear_build"}[dom]
    bms, bl = prof.get(build_name, [0.0, 0])
    walked = float(listed[1]) if listed[0] else float(l_own)
    all_ms, all_launches = ms + bms, launches + bl
    walked_bytes = launches * (20.0 * walked + 12.0 * p_own) + bl * (20.0 * l_own + 12.0 * p_own)
    alg_bytes = 20.0 * l_own + 12.0 * p_own
    achieved = walked_bytes / (all_ms * 1e-3) / 1e9 if all_launches else 0.0
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": None,
                "avg_launch_ms": all_ms / all_launches if all_launches else None, "launches": int(all_launches),
                "walked_half_links_per_steady_launch": walked, "half_links_owned": l_own, "points_owned": p_own,
                "bytes_per_steady_launch": 20.0 * walked + 12.0 * p_own,
                "steady_launches": {"launches": int(launches), "avg_launch_ms": ms / launches if launches else None},
                "frac_algorithmic_equiv": (alg_bytes * all_launches / (all_ms * 1e-3) / 1e9 / 8000.0) if all_launches else None,
                "algorithmic_bytes_per_launch": alg_bytes}
    if bl:
        roofline["list_writing_launches"] = {"launches": int(bl), "avg_launch_ms": bms / bl, "walks": "every half-link"}
    if listed[0]:
        roofline["culling"] = {"lists_built": int(listed[0]), "listed_half_links": int(listed[1]),
                               "listed_fraction": listed[1] / max(listed[2], 1)}
    # Whole-iteration fraction = what the metric pays for: the algorithmic bytes of every timed iteration (SURVEY 8d:
    # B_lin = 20 L + 36 P, B_def = 20 L + 48 P + 104 I G with the I G of the lattices that were really built) over the
    # timed region's wall time.  `..._walked` prices the half-links at the ones the sweeps walked.
    i_own = own_e - own_b
    b_lin = 20.0 * l_own + 36.0 * p_own
    def_iters = [(la["control_points"], la["iterations"]) for la in grp.lattices]
    b_total = n_lin * b_lin + sum(n * (20.0 * l_own + 48.0 * p_own + 104.0 * i_own * g) for g, n in def_iters)
    lin_cull = engine.cull_stats_kind("sweep_linear")
    walked_lin = float(lin_cull[1]) if lin_cull[0] else float(l_own)
    walked_def = float(engine.cull_stats_kind("sweep_deformable")[1]) if engine.cull_stats_kind("sweep_deformable")[0] else float(l_own)
    b_walked = n_lin * (20.0 * walked_lin + 36.0 * p_own) + sum(n * (20.0 * walked_def + 48.0 * p_own + 104.0 * i_own * g) for g, n in def_iters)
    iteration = {"algorithmic_bytes": b_total, "elapsed_s": elapsed, "iteration_frac": b_total / elapsed / 8e12,
                 "iteration_frac_walked": b_walked / elapsed / 8e12,
                 "lattices": [{"level": la["level"], "dims": la["dims"], "iterations": la["iterations"]} for la in grp.lattices],
                 "formula": "(n_lin (20 L + 36 P) + sum over lattices n (20 L + 48 P + 104 I G)) / elapsed / 8 TB/s, per rank"}
    # HBM bytes per launch of that kernel from the PMC counters (FETCH_SIZE, WRITE_SIZE collected in their own
    # rocprofv3 passes by scripts/profile_bench.sh and corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE
    # doubled on gfx950).  Counters cannot be read from inside this process, so the committed measurement of
    # the same workload is reported; null when there is none for this workload / shard size, or when the device
    # sources have changed since it was taken (profiles/hbm_traffic.json "measured_at").
    src_hash = _abi.device_source_hash()
    roofline["device_source_hash"] = src_hash
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")) as fh:
            tr = json.load(fh)
        # keyed on the workload AND on the device sources the counters were collected with: stale after any kernel change
        if (tr.get("kernel") == dom and tr.get("half_links_per_launch") == l_own
                and tr.get("measured_at") == src_hash):
            roofline["traffic"] = tr["traffic_bytes_per_launch"]
            roofline["traffic_source"] = tr.get("source")
            roofline["traffic_measured_at"] = tr.get("measured_at")
    except (OSError, ValueError, KeyError):
        pass

    if rank == 0:
        out = {
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
            "collectives": ("none" if world == 1 else "libfrog_comm (RCCL from C)" if native else f"torch.distributed/{backend}"),
            "config": {"workload": f"BASELINE.json configs[{ {2: 1, 3: 2, 5: 4}[args.config] }]: {images} images x {points} keypoints, "
                                   f"{pairs.n_pairs} pairs ({pairs.n_half_links} half-links), linear + {levels} deformable "
                                   f"levels, -g 100 -gd 1 -si 10",
                       "schedule": {"linear": n_lin, "deformable_per_level": per_level},
                       "parallelism": f"images sharded over {world} GPU(s)", "grids_per_level": grids,
                       "final_E": e},
            "roofline": roofline,
            "iteration": iteration,
            "kernels_ms": {n: {"total_ms": v[0], "launches": int(v[1])} for n, v in prof.items() if v[1]},
            "kernels_ms_by_phase": phase_k,
            "phase_iterations_per_s": {
                "linear": n_lin / phase_s["linear"],
                **{f"level{l}": per_level[l] / phase_s[f"level{l}"] for l in range(levels) if per_level[l]}},
            "setup_seconds": {"generate": t_gen, "create": t_create, "lattice_setups": grp.setup_seconds},
        }
        if args.shard_of:
            out["proxy"] = (f"rank {args.shard_of[0]} of {args.shard_of[1]} on one GPU: owns images {shards[0]}, no collective, "
                            + ("other ranks' coordinates and mixtures: those a full run of the default schedule ends with, standing still"
                               if partners is not None else "other ranks' coordinates static where the set-up left them, their mixtures zero")
                            + "; `value` is NOT the metric")
            out["proxy_ms_per_iteration"] = {ph: {n: v["ms"] / max(1, (n_lin if ph == "linear" else per_level[int(ph[5:])]))
                                                   for n, v in ks.items()} for ph, ks in phase_k.items()}
        if grp.comm_ms:
            out["comm_ms"] = grp.comm_ms
        if replicas_identical is not None:
            out["replicas_identical"] = replicas_identical
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pairs, n_lin, per_level, grp.statIntervalUpdate)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if replicas_identical is False:
        # the line above says so; a run whose ranks disagree about the coordinates is not a measurement
        raise SystemExit("replicas_identical is false: the ranks' replicas of xyz2 / the EM table differ")


if __name__ == "__main__":
    main()

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


def cpu_baseline(pairs, n_lin, per_level, stat_interval):
    """Oracle (CPU restatement, OpenMP over images like the reference) on a bounded
    sample of the same workload: one stats refresh, one linear iteration, one
    deformable iteration per level; extrapolated to the timed schedule."""
    from frog_amd import _abi
    from oracle.oracle_api import OracleGroup, lib
    cores = lib().frogo_get_max_threads()
    ref = OracleGroup(pairs.model, _abi.FrogOptions.default())
    ref.setup_stats()
    ref.linear_init()
    ref.transform_points()
    t = time.perf_counter(); ref.update_stats(); t_stats = time.perf_counter() - t
    t = time.perf_counter(); ref.linear_step(); ref.transform_points(); t_lin = time.perf_counter() - t
    ref.transform_points(True)
    t_def = []
    for level in range(len(per_level)):
        if per_level[level] == 0:
            t_def.append(0.0)
            continue
        ref.deformable_setup(level, _abi.FrogGridInfo())
        ref.transform_points()
        t = time.perf_counter(); ref.deformable_step(0.02); ref.transform_points(); t_def.append(time.perf_counter() - t)
        ref.transform_points(True)
    refreshes = -(-n_lin // stat_interval) + sum(-(-n // stat_interval) for n in per_level)
    total = n_lin * t_lin + sum(n * t for n, t in zip(per_level, t_def)) + refreshes * t_stats
    k = n_lin + sum(per_level)
    return {"value": k / total, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": "1 updateStats + 1 linear iteration + 1 deformable iteration per level on the same "
                      "pairs, timed with the oracle (oracle/frog_oracle.cpp, OpenMP over images) and "
                      "extrapolated to the timed schedule",
            "seconds": {"stats_refresh": t_stats, "linear_iteration": t_lin, "deformable_iteration": t_def}}


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
                    help="single-GPU proxy of rank R of an N-GPU run: this process owns shard R of N, the other ranks' "
                         "coordinates stay where the set-up left them, no collective is issued; prints per-phase kernel "
                         "times of that rank's share (not a metric line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-times", action="store_true",
                    help="HIP-event times of every kernel group, not only of the half-link sweeps (costs ~6 %% of the rate)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
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
    t0 = time.perf_counter()
    engine = HipEngine(pairs, opts, local_rank, shards[rank])
    t_create = time.perf_counter() - t0
    # N > 1 over RCCL: the collectives are issued from C (libfrog_comm.so, include/frog_comm.h) on the context's stream --
    # ncclCommInitRank with an id carried by torch.distributed; FROG_NATIVE_COMM=0, or any rank failing to set it up,
    # keeps them in torch.distributed
    native = None
    if world > 1 and backend == "nccl" and os.environ.get("FROG_NATIVE_COMM", "1") != "0":
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
    alg_bytes = 20.0 * l_own + 12.0 * p_own      # 8 B link + 12 B gathered xyz2 per half-link, 12 B own xyz2 per point
    achieved = alg_bytes / (ms / launches * 1e-3) / 1e9 if launches else 0.0
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": None,
                "avg_launch_ms": ms / launches if launches else None, "launches": int(launches),
                "algorithmic_bytes_per_launch": alg_bytes, "half_links_per_launch": l_own}
    # Certified outlier culling (frog_hip.h): the deformable sweep decides EVERY half-link, most false matches by a
    # distance bound instead of an evaluation.  `achieved` above prices the launch at the reference's bytes for all L
    # half-links; `frac_listed` prices it at the half-links it actually walked.
    lists_built, listed, owned = engine.cull_stats()
    if lists_built and launches:
        roofline["culling"] = {"lists_built": int(lists_built), "listed_half_links": int(listed),
                               "listed_fraction": listed / max(owned, 1),
                               "frac_listed": (20.0 * listed + 12.0 * p_own) / (ms / launches * 1e-3) / 1e9 / 8000.0}
        # the launches that walked every half-link AND wrote a list (one per list; a different kernel instantiation, its own
        # row in the rocprofv3 summary) are not in `avg_launch_ms` above: they are reported here
        bms, bl = prof.get("sweep_build", [0.0, 0])
        if bl:
            roofline["culling"]["list_writing_launches"] = {"launches": int(bl), "avg_launch_ms": bms / bl}
    # HBM bytes per launch of that kernel from the PMC counters (FETCH_SIZE, WRITE_SIZE collected in their own
    # rocprofv3 passes by scripts/profile_bench.sh and corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE
    # doubled on gfx950).  Counters cannot be read from inside this process, so the committed measurement of
    # the same workload is reported; null when there is none for this workload / shard size, or when the device
    # sources have changed since it was taken (profiles/hbm_traffic.json "measured_at").
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")) as fh:
            tr = json.load(fh)
        # keyed on the workload AND on the device sources the counters were collected with: stale after any kernel change
        if (tr.get("kernel") == dom and tr.get("half_links_per_launch") == l_own
                and tr.get("measured_at") == _abi.device_source_hash()):
            roofline["traffic"] = tr["traffic_bytes_per_launch"]
            roofline["traffic_source"] = tr.get("source")
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
            "kernels_ms": {n: {"total_ms": v[0], "launches": int(v[1])} for n, v in prof.items() if v[1]},
            "kernels_ms_by_phase": phase_k,
            "phase_iterations_per_s": {
                "linear": n_lin / phase_s["linear"],
                **{f"level{l}": per_level[l] / phase_s[f"level{l}"] for l in range(levels) if per_level[l]}},
            "setup_seconds": {"generate": t_gen, "create": t_create, "lattice_setups": grp.setup_seconds},
        }
        if args.shard_of:
            out["proxy"] = (f"rank {args.shard_of[0]} of {args.shard_of[1]} on one GPU: owns images {shards[0]}, no collective, "
                            f"other ranks' coordinates static; `value` is NOT the metric")
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


if __name__ == "__main__":
    main()

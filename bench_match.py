#!/usr/bin/env python3
"""bench_match.py -- the keypoint matcher (producer of pairs.bin) on one MI355X.

Auxiliary bench (the repo's headline metric is bench.py's).  Workload: the matching stage of
the same pipeline configuration as bench.py -- N images x 20 000 keypoints x 48-float
descriptors, every image pair (first < second) through ComputeMatches (match/match.cpp:255-336),
default flags of run.sh (-d 1, -d2 1).  Keypoints are resident in HBM before the timed region;
the timed region covers all pairing kernels, the per-query decisions and the return of the
pair lists to the host.

Prints ONE JSON line.  `roofline`: the dominant kernel is the matrix-core filter of frog_amd/csrc/device/match.hip.  Which form
ran is asked of the library (frog_matcher_last_forms), not re-derived from the environment.  `frac` = the FLOPs the instructions
PERFORM over the dense peak of the unit they run on: since round 5 the filter is three products of bf16 (hi, lo) splits
(match_mfma16_kernel: 3 (D + 2) terms padded to a multiple of 16 per (query, candidate) pair, v_mfma_f32_32x32x16_bf16) against
the dense bf16 peak of 2.5 PFLOP/s -- about 0.15: the pass is not bound by the matrix cores but by its tile loop (operand fetch ->
LDS -> barrier with one four-wavefront block per CU) and the two kernels around the filter, and `bound` says so.
`f32_equivalent` keeps the figure of rounds 1-5 as a labelled second field: ALGORITHMIC FLOPs (one (D + 2)-term product per pair
that passes the sign and scale tests, 2 FLOP per term) over the f32-input MFMA peak (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz =
157 TFLOP/s), the unit an exact-f32 product would need -- a speed-up over that unit, not a utilisation of anything.
FROG_MATCH_F32=1 runs the f32 chain, whose `frac` is against that f32 peak.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=100)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--dim", type=int, default=48)
    ap.add_argument("--jobs", type=int, default=0, help="image pairs to match (0 = all)")
    ap.add_argument("--threshold", type=float, default=1.0)
    ap.add_argument("--cpu-jobs", type=int, default=2, help="image pairs timed with the CPU oracle (0 = skip)")
    args = ap.parse_args()

    import numpy as np
    from frog_amd.match import Matcher, all_pairs, synthetic_keypoints

    t0 = time.perf_counter()
    imgs = synthetic_keypoints(args.images, args.points, dim=args.dim, seed=1)
    t_gen = time.perf_counter() - t0
    jobs = all_pairs(args.images)
    if args.jobs:
        jobs = jobs[:args.jobs]
    t0 = time.perf_counter()
    m = Matcher(imgs)
    t_upload = time.perf_counter() - t0
    m.run(jobs[:2], threshold=args.threshold)                      # warm-up
    t0 = time.perf_counter()
    res = m.run(jobs, threshold=args.threshold)
    elapsed = time.perf_counter() - t0
    ms, nd = m.last_stats()
    n_pairs = int(sum(len(a) for a, _ in res))
    # Algorithmic FLOPs: one (D + 2)-term product (2 FLOP per term) per (query, candidate) pair that passes the sign and scale
    # tests.  f32-input MFMA peak: 64 FLOP/clk/SIMD (MI355X_MICROARCH.md) = 256 CUs x 4 SIMDs x 64 x 2.4 GHz = 157.3 TFLOP/s.
    flops = 2.0 * (args.dim + 2) * nd
    peak_f32 = 256 * 4 * 64 * 2.4e9
    n_valu, n_f32, n_bf16 = m.last_forms()
    form = "bf16" if n_bf16 >= max(n_f32, n_valu) and n_bf16 else ("f32" if n_f32 >= n_valu and n_f32 else "valu")
    dp = 48 if args.dim <= 48 else 64
    k3 = (3 * (dp + 2) + 15) // 16 * 16
    if form == "bf16":
        issued_flops, peak, kernel, bound = 2.0 * k3 * nd, 2.5e15, "match_mfma16_kernel (v_mfma_f32_32x32x16_bf16 x 3 splits)", "tile-loop"
    elif form == "f32":
        issued_flops, peak, kernel, bound = flops, peak_f32, "match_mfma_kernel (v_mfma_f32_32x32x2_f32)", "mfma"
    else:
        issued_flops, peak, kernel, bound = flops, 157.3e12 / 2, "match_kernel (vector ALU, exact)", "valu"
    out = {
        "metric": "image pairs matched/sec (20 000 x 20 000 keypoints, 48-D)",
        "value": len(jobs) / elapsed, "unit": "image pairs/s", "n_gpus": 1, "higher_is_better": True,
        "seconds": elapsed, "kernel_seconds": ms * 1e-3, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.images} images x {args.points} keypoints x {args.dim} floats, {len(jobs)} image pairs, "
                               f"-d {args.threshold} -d2 1", "matches": n_pairs,
                   "candidate_pairs": float(sum(imgs[a].n * imgs[b].n for a, b in jobs)), "distances_evaluated": nd},
        "roofline": {"bound": bound, "kernel": kernel, "achieved": issued_flops / (ms * 1e-3) / 1e12,
                     "peak": peak / 1e12, "unit": "TFLOP/s", "frac": issued_flops / (ms * 1e-3) / peak, "traffic": None,
                     "frac_rule": "r06: FLOPs the issued matrix instructions perform / dense peak of the unit they run on (bf16: 2.5 PFLOP/s); "
                                  "rounds 1-5 quoted f32_equivalent.frac here",
                     "bound_note": "bf16 form: a third of the kernel is matrix work, the rest its tile loop (fetch -> LDS -> barrier, one "
                                   "four-wavefront block per CU) and the per-candidate tests: DESIGN.md section 10" if form == "bf16" else None,
                     "passes_by_form": {"valu": n_valu, "f32_mfma": n_f32, "bf16_mfma": n_bf16},
                     "f32_equivalent": {"frac": flops / (ms * 1e-3) / peak_f32, "achieved_tflops": flops / (ms * 1e-3) / 1e12,
                                        "peak_tflops": peak_f32 / 1e12,
                                        "rule": "algorithmic FLOPs (2 (D + 2) per pair that passes the filters) / f32-input MFMA peak: "
                                                "how many exact-f32 units the pass is worth, not a utilisation"},
                     "note": "ms = all kernels of the run (range search, MFMA filter, exact verification)"},
        "setup_seconds": {"generate": t_gen, "upload": t_upload},
    }
    if args.cpu_jobs:
        from oracle.oracle_api import lib, match_run
        sample = jobs[:args.cpu_jobs]
        from frog_amd._abi import usable_cpus
        cores = min(lib().frogo_match_get_max_threads(), usable_cpus())      # the CPUs the process may use, not the machine's
        t0 = time.perf_counter()
        ref = match_run(imgs, sample, threshold=args.threshold, threads=cores)
        t_cpu = time.perf_counter() - t0
        same = all(np.array_equal(a, c) and np.array_equal(b, d) for (a, b), (c, d) in zip(res[:len(sample)], ref))
        out["cpu_baseline"] = {"value": len(sample) / t_cpu, "unit": "image pairs/s", "cores": cores,
                               "kind": "port", "sample": f"the first {len(sample)} image pairs with the oracle "
                               "(oracle/match_oracle.cpp; one thread per image pair, as upstream's omp loop)",
                               "identical_pairs": bool(same)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

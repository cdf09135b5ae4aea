"""Generates tests/golden/schedule_golden.npz: the CPU oracle (oracle/frog_oracle.cpp) run over BASELINE.json configs[2] at its
FULL size (100 images x 20 000 keypoints, 1e8 half-links) through the reference's whole default schedule (-li 50 -dl 3 -di 200,
-g 100 -gd 1 -si 10: ImageGroup::run, imageGroup.cxx:31-157).  Data only: what the run ends with and the energy it printed at
every iteration --

  E            f64[650]   the energy of every accepted iteration (measures.csv's column)
  grids        i32[3]     lattices per level (the guard's regrids)
  matrices     f64[100,4,4]
  em           f32[100,3] (c1, c2, ratio) of the last refresh
  inliers      i64[100]   countInliers' census after the run
  dims/origin/spacing     of every lattice, creation order
  sha_grid     u8[n_lattices, n_images, 32]: one sha256 per (lattice, image) over the f32 coefficient array -- equality of ALL
               coefficients without storing them
  images       the images whose coefficients are stored, nodes node_stride[k] apart: coeff_<k> f32[len(images), ceil(G/stride), 3]
  max_coeff    f32[n_lattices] max |c| over ALL images (the scale deviations are quoted against)
  sha_xyz2, xyz2_sample   final coordinates: hash of all of them, every POINT_STRIDE-th point

The oracle took 1 220 s for this on the build container's eight CPUs (the file records it as oracle_seconds), which is why the GPU
suite compares against the stored run instead (tests/test_gpu_schedule_golden.py); tests/test_schedule_golden.py holds the file
against the oracle as it is built now for the first iterations.  With --config5: BASELINE.json configs[4] (500 images, 4.6e8
half-links, five levels) over 20 + 5 x 40 iterations into schedule_golden_cfg5.npz (2 500 s, 35 GB); with --config2: configs[1]
(20 images, linear only, 50 iterations) into schedule_golden_cfg2.npz (seconds).
Usage: python tests/golden/make_schedule_golden.py [--config5 | --config2] [threads]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from frog_amd import _abi                   # noqa: E402
from frog_amd.pairs import Pairs            # noqa: E402
from oracle import oracle_api               # noqa: E402

CFG5 = "--config5" in sys.argv[1:]
CFG2 = "--config2" in sys.argv[1:]
if CFG2:
    # BASELINE.json configs[1]: 20 images, 2 M pairs, linear only (-dl 0): the reference's own CPU-runnable case
    IMAGES, NODES_KEPT, NODE_STRIDE, POINT_STRIDE = [], None, 1, 20
    LI, DL, DI = 50, 0, 0
    NAME = "schedule_golden_cfg2.npz"
elif CFG5:
    # BASELINE.json configs[4]: 500 images, ~60 partner images each, five levels; the schedule of scripts/parity_reference_order.py
    # --config5 (20 + 5 x 40: the default 50 + 5 x 200 would keep the oracle busy for four hours)
    IMAGES = list(range(0, 500, 71))
    NODES_KEPT = 1500                       # per (lattice, image): the stride is ceil(G / NODES_KEPT)
    POINT_STRIDE = 500
    LI, DL, DI = 20, 5, 40
    NAME = "schedule_golden_cfg5.npz"
else:
    IMAGES = list(range(0, 100, 18))        # 0, 18, ..., 90
    NODES_KEPT = None
    NODE_STRIDE = 4
    POINT_STRIDE = 100
    LI, DL, DI = 50, 3, 200
    NAME = "schedule_golden.npz"


def main():
    args = [a for a in sys.argv[1:] if a not in ("--config5", "--config2")]
    if args:
        oracle_api.lib().frogo_set_threads(int(args[0]))
    if CFG2:
        pairs = Pairs.synthetic(20, 20000, 10526, seed=1)       # bench.py --config 2
    elif CFG5:
        pairs = Pairs.synthetic(500, 20000, 16667, seed=1, partners_per_image=60)     # bench.py --config 5
    else:
        pairs = Pairs.synthetic(100, 20000, 10101, seed=1)      # bench.py's workload (CONFIGS[3])
    ref = oracle_api.OracleGroup(pairs.model, _abi.FrogOptions.default())
    t0 = time.time()
    E, grids = ref.run(LI, DL, DI)
    seconds = time.time() - t0
    print(f"oracle run: {seconds:.0f} s, {len(E)} energies, grids per level {grids}, final E {E[-1]!r}", flush=True)
    n_img = pairs.n_images
    out = {"E": np.asarray(E, np.float64), "grids": np.asarray(grids, np.int32),
           "matrices": np.stack([ref.matrix(i) for i in range(n_img)]),
           "em": np.stack([ref.em(i) for i in range(n_img)]),
           "images": np.asarray(IMAGES, np.int32), "point_stride": np.int32(POINT_STRIDE),
           "schedule": np.asarray([LI, DL, DI], np.int32), "n_half_links": np.int64(pairs.n_half_links),
           "oracle_seconds": np.float64(seconds)}
    counts = (_abi.FrogCounts * n_img)()
    ref.count_inliers(counts)
    out["inliers"] = np.asarray([counts[i].inliers for i in range(n_img)], np.int64)
    out["outliers"] = np.asarray([counts[i].outliers for i in range(n_img)], np.int64)
    n_grids = ref.num_grids()
    dims, origin, spacing, shas, max_coeff, strides = [], [], [], [], [], []
    for k in range(n_grids):
        sha_k, mx, kept = [], 0.0, []
        for i in range(n_img):
            info, c = ref.grid(i, k, _abi.FrogGridInfo())
            if i == 0:
                dims.append(list(info.dims)); origin.append(list(info.origin)); spacing.append(list(info.spacing))
                strides.append(NODE_STRIDE if NODES_KEPT is None else -(-len(c) // NODES_KEPT))
            sha_k.append(hashlib.sha256(np.ascontiguousarray(c, np.float32).tobytes()).digest())
            mx = max(mx, float(np.abs(c).max()))
            if i in IMAGES:
                kept.append(c[::strides[-1]].copy())
        shas.append(sha_k); max_coeff.append(mx)
        out[f"coeff_{k}"] = np.stack(kept).astype(np.float32)
    out["dims"] = np.asarray(dims, np.int32); out["origin"] = np.asarray(origin, np.float64); out["spacing"] = np.asarray(spacing, np.float64)
    out["node_stride"] = np.asarray(strides, np.int32)
    out["sha_grid"] = np.frombuffer(b"".join(b"".join(row) for row in shas), np.uint8).reshape(n_grids, n_img, 32)
    if n_grids == 0:
        out["dims"] = np.zeros((0, 3), np.int32); out["origin"] = np.zeros((0, 3)); out["spacing"] = np.zeros((0, 3))
    out["max_coeff"] = np.asarray(max_coeff, np.float32)
    xyz2 = ref.xyz2()
    out["sha_xyz2"] = np.frombuffer(hashlib.sha256(xyz2.tobytes()).digest(), np.uint8)
    out["xyz2_sample"] = xyz2[::POINT_STRIDE].copy()
    path = os.path.join(ROOT, "tests", "golden", NAME)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

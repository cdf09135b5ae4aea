#!/usr/bin/env python3
"""Timings of the two stages around the solver that were added last, GPU vs the CPU oracle:
 * RANSAC of one image against 19 fixed ones (20 000 keypoints per image, ~10 500 pairs per image pair),
   5 000 candidates in 8 batches (imageGroup.cxx:629-804);
 * reslicing a 256^3 int16 volume onto a 256^3 grid through the INVERSE of a 1 + 7 link chain, trilinear
   (tools/VolumeTransform.cxx)."""
import os
import sys
import time

import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # two OpenMP runtimes live in this process (product: libomp, oracle: libgomp)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frog_amd import _abi                                     # noqa: E402
from frog_amd.chain import Chain, Link, invert                # noqa: E402
from frog_amd.image_group import ImageGroup                   # noqa: E402
from frog_amd.pairs import Pairs                              # noqa: E402
from oracle.oracle_api import OracleGroup, chain_reslice, lib  # noqa: E402


# ---- RANSAC
pairs = Pairs.synthetic(20, 20000, 10526, seed=1, scale_min=1.0, scale_max=1.0)
n = pairs.n_images
po, rp = pairs.point_offset, pairs.row_ptr
links = int(rp[po[n]]) - int(rp[po[n - 1]])
g = ImageGroup(pairs, n_fixed_images=n - 1)
g.setupLinearTransforms(); g.transformPoints()
g.RANSAC(n - 1, iterations=80, batches=8)                     # warm-up (module load)
g.setupLinearTransforms(); g.transformPoints()
t0 = time.perf_counter(); a = g.RANSAC(n - 1, iterations=5000, batches=8); t_gpu = time.perf_counter() - t0
ref = OracleGroup(pairs.model, _abi.FrogOptions.default(n_fixed_images=n - 1))
ref.setup_stats(); ref.linear_init(); ref.transform_points()
t0 = time.perf_counter(); b = ref.ransac(n - 1, iterations=5000, batches=8); t_cpu = time.perf_counter() - t0
print(f"RANSAC: image with {links} half-links, 5000 candidates: GPU {t_gpu * 1e3:.1f} ms ({5000 * links / t_gpu / 1e9:.1f} G link tests/s), "
      f"best census {a}; CPU oracle (8 threads: one per batch) {t_cpu:.2f} s, census {b}; "
      f"matrices agree to {np.abs(g.matrix(n - 1) - ref.matrix(n - 1)).max():.1e}")

threads = lib().frogo_get_max_threads()

# ---- reslice
rng = np.random.default_rng(1)
M = np.eye(4); M[:3, 3] = [3, -2, 1]
chain = [Link.linear(M)]
for k in (4, 4, 8, 8, 16, 16, 16):
    dims = (k + 3, k + 3, k + 3)
    sp = tuple(400.0 / k for _ in range(3))
    chain.append(Link.bspline(dims, tuple(-s for s in sp), sp, (1.0 * rng.normal(size=(dims[0] ** 3, 3))).astype(np.float32)))
inv = invert(chain)
z, y, x = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")
vol = (1000 + 500 * np.sin(x / 9.0) * np.cos(y / 11.0) + 2 * z).astype(np.int16)
o, s = (0.0, 0.0, 0.0), (400 / 256,) * 3
c = Chain(inv)
c.reslice(vol[:8], o, s, (256, 256, 8), o, s, 1, 0.0)
t0 = time.perf_counter(); got = c.reslice(vol, o, s, (256, 256, 256), o, s, 1, 0.0); t_gpu = time.perf_counter() - t0
t0 = time.perf_counter(); want = chain_reslice(inv, vol, o, s, (256, 256, 16), (0.0, 0.0, 100.0), s, 1, 0.0); t_cpu = (time.perf_counter() - t0) * 16
sub = c.reslice(vol, o, s, (256, 256, 16), (0.0, 0.0, 100.0), s, 1, 0.0)
diff = np.abs(sub - np.clip(np.floor(want + 0.5), -32768, 32767))
print(f"reslice 256^3 int16 through the inverse of 1 + 7 links: GPU {t_gpu * 1e3:.0f} ms incl. host<->device copies "
      f"({256 ** 3 / t_gpu / 1e6:.0f} M voxels/s); CPU oracle ({threads} threads, 16 slices x 16) {t_cpu:.1f} s; "
      f"largest difference on the checked slab {int(diff.max())} grey level(s), {(diff > 0).mean() * 100:.3f} % of voxels")

#!/usr/bin/env python3
"""Jacobian check of a 1 + 7 link chain on a 256^3 grid: GPU (include/frog_chain.h) vs the CPU oracle."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frog_amd.chain import Chain, Link          # noqa: E402
from oracle.oracle_api import chain_check, lib   # noqa: E402

rng = np.random.default_rng(1)
M = np.eye(4); M[:3, 3] = [3, -2, 1]
links = [Link.linear(M)]
for n in (4, 4, 8, 8, 16, 16, 16):
    dims = (n + 3, n + 3, n + 3)
    sp = tuple(400.0 / n for _ in range(3))
    links.append(Link.bspline(dims, tuple(-s for s in sp), sp, (2.0 * rng.normal(size=(dims[0] ** 3, 3))).astype(np.float32)))
grid = ((0.0, 0.0, 0.0), (400 / 256,) * 3, (256, 256, 256))
c = Chain(links)
c.check(*grid)
t0 = time.perf_counter(); n, m = c.check(*grid); t_gpu = time.perf_counter() - t0
sub = (grid[0], grid[1], (256, 256, 32))
t0 = time.perf_counter(); rn, rm = chain_check(links, *sub); t_cpu = (time.perf_counter() - t0) * 8
print(f"GPU: {n} negative of {256 ** 3}, min det {m:.4f}, {t_gpu * 1e3:.1f} ms = {256 ** 3 / t_gpu / 1e9:.2f} G nodes/s; "
      f"CPU oracle ({lib().frogo_get_max_threads()} threads, 1/8 of the grid x 8): {t_cpu:.2f} s")

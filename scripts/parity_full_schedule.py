#!/usr/bin/env python3
"""cfg 3 (100 images x 20 000 keypoints, 1e8 half-links) over the reference's FULL default schedule (-li 50 -dl 3 -di 200,
regrids as the diffeomorphism guard asks for them), HIP path and oracle free-running from the same pairs: the parity numbers
of tests/test_gpu_round3.py::test_config3_free_running_schedule_against_the_oracle (which runs 10 + 3 x 10 iterations) for a
whole registration.  Takes ~6 minutes of oracle time on the GPU box; writes a JSON summary.

    python3 scripts/parity_full_schedule.py [out.json] [linear] [per_level]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np                                               # noqa: E402

from frog_amd import _abi                                        # noqa: E402
from frog_amd.image_group import ImageGroup                      # noqa: E402
from frog_amd.pairs import Pairs                                 # noqa: E402
from oracle.oracle_api import OracleGroup                        # noqa: E402
from lattice_util import lattice_deviation, node_weights        # noqa: E402


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_full_schedule.json"
    li = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    di = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    dl = 3
    t0 = time.time()
    pairs = Pairs.synthetic(100, 20000, 10101, seed=1)
    g = ImageGroup(pairs)
    ref = OracleGroup(pairs.model, _abi.FrogOptions.default())
    ref.setup_stats()
    po = np.asarray(pairs.point_offset)
    g.setupLinearTransforms(); ref.linear_init()
    g.transformPoints(); ref.transform_points()
    e_dev, e_ref = [], []
    for it in range(li):
        if it % 10 == 0:
            g.updateStats(); ref.update_stats()
        e_dev.append(g.updateLinearTransforms()); e_ref.append(ref.linear_step())
        g.transformPoints(); ref.transform_points()
        if it % 10 == 9:
            print(f"[{time.time() - t0:6.0f}s] linear {it + 1}/{li}  E {e_dev[-1]:.6f} / {e_ref[-1]:.6f}", flush=True)
    worst_m = 0.0
    for i in range(pairs.n_images):
        m, mr = g.matrix(i), ref.matrix(i)
        worst_m = max(worst_m, relerr(np.diag(m)[:3], np.diag(mr)[:3]), relerr(m[:3, 3], mr[:3, 3]))
    g.transformPoints(True); ref.transform_points(True)
    snapshots, levels, grids, rejects = [], [], [], 0
    for level in range(dl):
        def setup():
            info = g.setupDeformableTransforms(level)
            rinfo = ref.deformable_setup(level, _abi.FrogGridInfo())
            assert list(info.dims) == list(rinfo.dims), f"lattice dimensions differ at level {level}"
            snapshots.append(ref.xyz().copy()); levels.append(level)
            g.transformPoints(); ref.transform_points()
        setup()
        alpha, nd, it, n_g = np.float32(0.02), 0, 0, 1
        while it < di:
            if it % 10 == 0:
                g.updateStats(); ref.update_stats()
            e, er = g.updateDeformableTransforms(float(alpha)), ref.deformable_step(float(alpha))
            assert (e < 0) == (er < 0), f"guard decisions differ at level {level}, iteration {it}"
            if e < 0:
                rejects += 1
                if nd == 0:
                    alpha = np.float32(alpha / np.float32(2))
                n_g += 1
                g.transformPoints(True); ref.transform_points(True)
                setup()
                nd = 0
                continue
            nd += 1
            g.transformPoints(); ref.transform_points()
            e_dev.append(e); e_ref.append(er)
            it += 1
            if it % 25 == 0:
                print(f"[{time.time() - t0:6.0f}s] level {level} iteration {it}/{di}  lattices {n_g}  E {e:.6f} / {er:.6f}", flush=True)
        grids.append(n_g)
        g.transformPoints(True); ref.transform_points(True)
    e_dev, e_ref = np.array(e_dev), np.array(e_ref)
    res = {"workload": f"100 images x 20000 keypoints, {pairs.n_half_links} half-links, -li {li} -dl {dl} -di {di}",
           "iterations": int(len(e_dev)), "guard_rejections": rejects, "grids_per_level": grids,
           "E_max_rel_dev": float(np.max(np.abs(e_dev - e_ref) / e_ref)), "E_final": [float(e_dev[-1]), float(e_ref[-1])],
           "matrices_max_rel_dev": worst_m, "lattices": []}
    for k in range(ref.num_grids()):
        w = node_weights(ref, k, po, snapshots[k])
        worst = {"lattice": k, "level": levels[k], "raw": 0.0, "weighted": 0.0, "field": 0.0}
        for i in range(pairs.n_images):
            d = lattice_deviation(g, ref, k, i, snapshots[k][po[i]:po[i + 1]], w)
            for key in ("raw", "weighted", "field"):
                worst[key] = max(worst[key], d[key])
            worst["weak_nodes"], worst["nodes"] = d["weak"], d["nodes"]
        res["lattices"].append(worst)
        print(f"[{time.time() - t0:6.0f}s] lattice {k}: {worst}", flush=True)
    res["final_xyz_rel_dev"] = relerr(g.points()[0], ref.xyz())
    res["seconds"] = time.time() - t0
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

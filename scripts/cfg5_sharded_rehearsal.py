#!/usr/bin/env python3
"""BASELINE.json configs[4]'s group (500 images x 20 000 keypoints, ~60 partner images each, 4.6e8 half-links) through the
C++ multi-GPU host with EIGHT contexts on one GPU (`bin/frog -ngl 8`: shards of 62-63 images, every collective of
include/frog_comm.h, host-staged) against the one-context run of the same binary -- what a one-GPU box can execute of the
configuration's 8-way form.  Levels 0-2 only (`-dl 3`): the files `frog` writes for levels 3-4 of 500 images are tens of GB.

    python3 scripts/cfg5_sharded_rehearsal.py [out.json]
"""
import csv
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np                                               # noqa: E402

from frog_amd.pairs import Pairs                                 # noqa: E402


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cfg5_sharded_rehearsal.json"
    t0 = time.time()
    work = tempfile.mkdtemp(prefix="frog_cfg5_")
    one, eight = os.path.join(work, "one"), os.path.join(work, "eight")
    os.makedirs(one); os.makedirs(eight)
    pairs = Pairs.synthetic(500, 20000, 16667, seed=1, partners_per_image=60)
    n_half = pairs.n_half_links
    pairs.write(os.path.join(one, "pairs.bin"))
    del pairs
    os.symlink(os.path.join(one, "pairs.bin"), os.path.join(eight, "pairs.bin"))
    print(f"[{time.time() - t0:5.0f}s] pairs.bin written ({os.path.getsize(os.path.join(one, 'pairs.bin')) / 1e9:.2f} GB, {n_half} half-links)", flush=True)
    flags = ["-li", "3", "-dl", "3", "-di", "3", "-j", "-q", "1"]
    times = {}
    for cwd, extra in ((one, []), (eight, ["-ngl", "8"])):
        t = time.time()
        r = subprocess.run([os.path.join(ROOT, "bin", "frog"), "pairs.bin", *flags, *extra], cwd=cwd, capture_output=True, text=True,
                           env=dict(os.environ, FROG_CHECK_REPLICAS="1", FROG_TIMING="1"))
        times[os.path.basename(cwd)] = time.time() - t
        print(f"[{time.time() - t0:5.0f}s] bin/frog {' '.join(extra)} rc={r.returncode}", flush=True)
        if r.returncode != 0:
            print(r.stdout[-3000:], r.stderr[-3000:])
            raise SystemExit(1)
        if extra:
            assert "Images sharded over 8 contexts" in r.stdout and "Replicas identical : yes (8 contexts" in r.stdout, r.stdout[-2000:]
    ea = np.array([float(x[1]) for x in list(csv.reader(open(os.path.join(one, "measures.csv"))))[1:]])
    eb = np.array([float(x[1]) for x in list(csv.reader(open(os.path.join(eight, "measures.csv"))))[1:]])
    worst_m = worst_c = 0.0
    for i in range(500):
        ta = json.load(open(os.path.join(one, "transforms", f"{i}.json")))["transforms"]
        tb = json.load(open(os.path.join(eight, "transforms", f"{i}.json")))["transforms"]
        assert len(ta) == len(tb)
        worst_m = max(worst_m, relerr(ta[0]["matrix"], tb[0]["matrix"]))
        for x, y in zip(ta[1:], tb[1:]):
            assert x["dimensions"] == y["dimensions"]
            worst_c = max(worst_c, relerr(x["coeffs"], y["coeffs"]))
    ba, bb = json.load(open(os.path.join(one, "bbox.json"))), json.load(open(os.path.join(eight, "bbox.json")))
    res = {"workload": f"500 images x 20000 keypoints, {n_half} half-links, {' '.join(flags)}", "contexts": 8,
           "replicas_identical": True, "iterations": int(len(ea)), "E_max_rel_dev": float(np.max(np.abs(ea - eb) / eb)),
           "matrices_max_rel_dev": worst_m, "lattices_max_rel_dev": worst_c,
           "half_pairs": [ba["halfPairs"], bb["halfPairs"]], "inliers": [ba["inliers"], bb["inliers"]],
           "seconds_process": times, "seconds_total": time.time() - t0}
    assert len(ea) == len(eb) and res["E_max_rel_dev"] < 1e-5 and worst_m < 1e-6 and worst_c < 1e-5
    assert ba["halfPairs"] == bb["halfPairs"] == n_half and abs(ba["inliers"] - bb["inliers"]) <= 2
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res), flush=True)
    subprocess.run(["rm", "-rf", work])


if __name__ == "__main__":
    main()

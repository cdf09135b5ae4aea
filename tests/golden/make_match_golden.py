"""Generates tests/golden/match_golden.json from the reference build of ComputeMatches
(oracle/_ref/libfrog_refmatch.so: `make -C oracle ref`, needs /root/reference).  Data only: three small keypoint sets (f32 bit
patterns, base64) and the pair lists the reference's own function returns for them under each option set."""
import base64
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from frog_amd.match import Keypoints, synthetic_keypoints       # noqa: E402
from oracle import oracle_api                                    # noqa: E402

DIM = 16
imgs = synthetic_keypoints(3, 160, dim=DIM, seed=20261005, noise=0.08)
imgs[2] = Keypoints.from_rows(imgs[2].rows()[:57])
# exact duplicates and near-ties, so that the strict `<` of the nearest / second-nearest update and the NaN of sqrt(0 / 0) matter
rows = imgs[1].rows()
rows[:12, 6:] = imgs[0].rows()[:12, 6:]
rows[12:20, 6:] = imgs[0].rows()[30:38, 6:] + np.float32(1e-3)
rows[:20, 3:5] = imgs[0].rows()[:20, 3:5]
imgs[1] = Keypoints.from_rows(rows)
JOBS = [(0, 1), (0, 2), (1, 2), (2, 0)]
OPTIONS = [dict(threshold=0.22), dict(threshold=1.0), dict(threshold=0.8, dist2second=0.8), dict(threshold=1.0, anat=60.0),
           dict(threshold=1.0, sym=1), dict(threshold=1e10), dict(threshold=3e19),
           dict(all=1, threshold=0.7), dict(all=1, threshold=0.7, sym=1), dict(all=1, threshold=1.0, anat=80.0),
           dict(all=1, threshold=1e10), dict(all=1, threshold=0.0)]
b64 = lambda a: base64.b64encode(np.ascontiguousarray(a, "<f4").tobytes()).decode()
out = {"source": "oracle/_ref/libfrog_refmatch.so = /root/reference/match/match.cpp:28-48, :243-251, :255-336 compiled as they are "
                 "(g++ -O2 -DINT_PTIDS, scalar norm)",
       "dim": DIM,
       "images": [{"n": k.n, "xyz": b64(k.xyz), "scale": b64(k.scale), "laplacian": b64(k.laplacian), "desc": b64(k.desc)} for k in imgs],
       "jobs": JOBS, "cases": []}
for o in OPTIONS:
    r = oracle_api.ref_match_run(imgs, JOBS, **o)
    out["cases"].append({"options": o, "pairs": [[a.tolist(), b.tolist()] for a, b in r]})
    print(o, [len(a) for a, _ in r])
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "match_golden.json"), "w"))

"""BASELINE.json configs[2] (100 images x 20 000 keypoints, 1e8 half-links) over the reference's whole default schedule
(-li 50 -dl 3 -di 200): the product path against FROG_REFERENCE_ORDER=1, both on the device and free-running.
tests/test_gpu_reference_order.py holds reference-order mode bit-equal to the CPU oracle; what this run measures is therefore
what re-association (fast weight, partner-group sums, tiled scatter, fused multiply-adds) does to a real schedule.
Writes gpurun_out/parity_reference_order.json (copied to profiles/ per round).  Usage: parity_reference_order.py [--tag NAME] [--config5] [li dl di]"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from frog_amd.pairs import Pairs
import test_gpu_reference_order as T


class Env:                       # the two calls of monkeypatch the test helper uses
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k): os.environ.pop(k, None)


tag = ""
if "--tag" in sys.argv:                      # a suffix of the output file (scripts/bisect_parity.sh: one file per cell)
    k = sys.argv.index("--tag"); tag = "_" + sys.argv[k + 1]; del sys.argv[k:k + 2]
argv = [a for a in sys.argv[1:] if a != "--config5"]
cfg5 = "--config5" in sys.argv[1:]          # BASELINE.json configs[4]: 500 images, ~60 partners each, five levels
li, dl, di = ([int(x) for x in argv[:3]] + ([20, 5, 40] if cfg5 else [50, 3, 200])[len(argv):])[:3]
if cfg5:
    pairs = Pairs.synthetic(500, 20000, 16667, seed=1, partners_per_image=60)
    images = range(0, 500, 71)
else:
    pairs = Pairs.synthetic(100, 20000, 10101, seed=1)
    images = range(0, 100, 9)
t0 = time.time()
r = T.fast_against_reference_order(pairs, li, dl, di, Env(), images)
r["seconds"] = time.time() - t0
r["schedule"] = {"li": li, "dl": dl, "di": di}
r["workload"] = "%d images x 20 000 keypoints, %d half-links" % (pairs.n_images, pairs.n_half_links)
out = os.path.join(ROOT, "gpurun_out", ("parity_reference_order_cfg5%s.json" if cfg5 else "parity_reference_order%s.json") % tag)
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as fh:
    json.dump(r, fh, indent=1)
print(json.dumps(r, indent=1))

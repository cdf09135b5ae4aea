"""Diagnostic: the 40-image cfg-5-shaped group over a longer schedule, product path vs FROG_REFERENCE_ORDER=1."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from frog_amd.pairs import Pairs
import test_gpu_reference_order as T


class Env:
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k): os.environ.pop(k, None)


li, dl, di = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (20, 5, 40)
pairs = Pairs.synthetic(40, 20000, 16667, seed=2, partners_per_image=20)
r = T.fast_against_reference_order(pairs, li, dl, di, Env(), range(40))
print({k: v for k, v in r.items() if k != "lattices"})
for k, d in enumerate(r["lattices"]):
    print(k, {a: (f"{b:.2e}" if isinstance(b, float) else b) for a, b in d.items()})

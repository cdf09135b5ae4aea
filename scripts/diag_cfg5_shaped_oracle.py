"""Diagnostic: reference-order mode vs the oracle on the 40-image cfg-5-shaped group, longer schedule (arbiter)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from frog_amd.pairs import Pairs
import test_gpu_reference_order as T


class Env:
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k): os.environ.pop(k, None)


li, dl, di = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (20, 5, 12)
pairs = Pairs.synthetic(40, 20000, 16667, seed=2, partners_per_image=20)
print(T.run_equal(pairs, li, dl, di, Env(), images=range(0, 40, 7)), flush=True)

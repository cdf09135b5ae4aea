"""Reference-order mode, fast forms (round 6) against the literal forms (FROG_REF_LITERAL=1), in lockstep on one group with
np.array_equal after every step: per-point sums, gradient images, coordinates, matrices, mixtures, lattices.  Prints the first
quantity that differs.  Usage: diag_exact_forms.py [--config5 | --config3] [li dl di]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from frog_amd.pairs import Pairs
import test_gpu_reference_order as T

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--config5" in sys.argv:
    pairs, images, default = Pairs.synthetic(500, 20000, 16667, seed=1, partners_per_image=60), list(range(0, 500, 71)), [20, 5, 40]
else:
    pairs, images, default = Pairs.synthetic(100, 20000, 10101, seed=1), list(range(0, 100, 9)), [10, 3, 12]
li, dl, di = ([int(x) for x in argv[:3]] + default[len(argv):])[:3]
os.environ["FROG_REFERENCE_ORDER"] = "1"
fast = T.Side(pairs)
os.environ["FROG_REF_LITERAL"] = "1"
lit = T.Side(pairs)
del os.environ["FROG_REF_LITERAL"], os.environ["FROG_REFERENCE_ORDER"]
counters = {"steps": 0}
inner = T.equality_checker(images, counters)


def check(tag, sides, e=None, infos=None):
    try:
        inner(tag, sides, e, infos)
    except AssertionError as exc:
        print("FIRST DIFFERENCE at", tag, ":", exc, flush=True)
        a, b = sides
        kind = tag if isinstance(tag, str) else tag[0]
        if kind == "step":
            sa, sb = a.point_sums(), b.point_sums()
            bad = np.nonzero(np.any(sa != sb, axis=1))[0]
            print("  per-point sums differ at", len(bad), "points; first:", bad[:5], sa[bad[:3]], sb[bad[:3]])
            po = np.asarray(pairs.point_offset)
            print("  images of those points:", np.unique(np.searchsorted(po, bad[:1000], side="right") - 1)[:20])
        print("  energies", e)
        raise SystemExit(1)
    if not isinstance(tag, str) and tag[0] in ("linear", "deformable"):
        print(tag, "equal", e, flush=True)


T.lockstep([fast, lit], li, dl, di, check)
print("equal over the whole schedule", counters)

"""Lockstep GPU-vs-oracle run printing where the two diverge (diagnostic, not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frog_amd import _abi
from frog_amd.pairs import Pairs
from frog_amd.image_group import ImageGroup
from oracle.oracle_api import OracleGroup

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)

pairs = Pairs.synthetic(6, 3000, 1500, seed=7)
g = ImageGroup(pairs); ref = OracleGroup(pairs.model, _abi.FrogOptions.default()); ref.setup_stats()
inject = len(sys.argv) > 1 and sys.argv[1] == "inject"
g.setupLinearTransforms(); ref.linear_init(); g.transformPoints(); ref.transform_points()
def stats(tag):
    g.updateStats(); ref.update_stats()
    d = max(rel(g.em(i), ref.em(i)) for i in range(6))
    same = all(np.array_equal(g.samples(i)[0], ref.samples(i)[0]) for i in range(6))
    print(f"{tag}: em rel diff {d:.2e} samples identical {same}")
    if inject:
        for i in range(6): g.set_em(i, ref.em(i))
for it in range(30):
    if it % 10 == 0: stats(f"lin {it}")
    e = g.updateLinearTransforms(); er = ref.linear_step()
    g.transformPoints(); ref.transform_points()
print("after linear: xyz2 rel", rel(g.points()[1], ref.xyz2()), "E", e, er)
g.transformPoints(True); ref.transform_points(True)
for level in range(3):
    g.setupDeformableTransforms(level); ref.deformable_setup(level, _abi.FrogGridInfo())
    g.transformPoints(); ref.transform_points()
    alpha = 0.02; nd = 0; it = 0
    while it < 40:
        if it % 10 == 0: stats(f"L{level} it {it}")
        e = g.updateDeformableTransforms(alpha); er = ref.deformable_step(alpha)
        if (e < 0) != (er < 0): print("REJECT MISMATCH", e, er); break
        if e < 0:
            if nd == 0: alpha /= 2
            g.transformPoints(True); ref.transform_points(True)
            g.setupDeformableTransforms(level); ref.deformable_setup(level, _abi.FrogGridInfo())
            g.transformPoints(); ref.transform_points(); nd = 0
            print(f"  regrid at L{level} it {it}")
            continue
        nd += 1
        g.transformPoints(); ref.transform_points()
        if it % 10 == 9:
            k = g.num_grids() - 1
            d = max(rel(g.grid(i, k)[1], ref.grid(i, k, _abi.FrogGridInfo())[1]) for i in range(6))
            print(f"  L{level} it {it}: coeff rel {d:.2e} xyz2 rel {rel(g.points()[1], ref.xyz2()):.2e} E rel {abs(e-er)/er:.2e}")
        it += 1
    g.transformPoints(True); ref.transform_points(True)

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from frog_amd import _abi
from frog_amd.image_group import ImageGroup
from oracle.oracle_api import OracleGroup
from test_gpu_parity import ragged_pairs
pairs = ragged_pairs()
opt = dict(stats_max_size=500)
g = ImageGroup(pairs, **opt); ref = OracleGroup(pairs.model, _abi.FrogOptions.default(**opt)); ref.setup_stats()
g.setupLinearTransforms(); ref.linear_init(); g.transformPoints(); ref.transform_points()
for it in range(8):
    if it % 5 == 0:
        g.updateStats(); ref.update_stats()
        print("em", [float(np.max(np.abs(g.em(i) - ref.em(i)) / np.abs(ref.em(i)))) for i in range(pairs.n_images)])
        print("nsamples", [len(g.samples(i)[0]) for i in range(pairs.n_images)], [len(ref.samples(i)[0]) for i in range(pairs.n_images)])
    e = g.updateLinearTransforms(); er = ref.linear_step()
    g.transformPoints(); ref.transform_points()
    dm = [float(np.max(np.abs(g.matrix(i)[:3] - ref.matrix(i)[:3]))) for i in range(pairs.n_images)]
    print(it, e, er, "matrix abs diff", ["%.2e" % x for x in dm])

import sys; sys.path.insert(0, '.')
from frog_amd.pairs import Pairs
from frog_amd.image_group import ImageGroup
pairs = Pairs.synthetic(100, 20000, 10101.0, seed=1)
g = ImageGroup(pairs)
g.setupLinearTransforms(); g.transformPoints()
for it in range(5):
    if it % 10 == 0: g.updateStats()
    g.updateLinearTransforms(); g.transformPoints()
g.transformPoints(True); g.setupDeformableTransforms(0); g.transformPoints(); g.updateStats()
for it in range(3):
    g.updateDeformableTransforms(0.02); g.transformPoints()
print("cfg3 cull ranges (non-empty, with election):", g.cull_ranges(), "stats", g.cull_stats())

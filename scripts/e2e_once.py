import sys, json; sys.path.insert(0, '.')
import bench
from frog_amd.pairs import Pairs
pairs = Pairs.synthetic(100, 20000, 10101, seed=1)
print(json.dumps(bench.end_to_end(pairs, 3), indent=1))

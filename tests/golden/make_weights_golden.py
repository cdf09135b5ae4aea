"""Generates tests/golden/weights_golden.json from the reference build of vtkBSplineTransformWeights
(oracle/_ref/libfrog_refweights.so: `make -C oracle ref`, needs /root/reference).  Data only: fractions and the four
weights the reference's own function returns for them, as f64 bit patterns."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_api       # noqa: E402

rng = np.random.default_rng(20261004)
f = np.concatenate([
    np.array([0.0, 1.0 - 2.0 ** -53, 1.0 - 2.0 ** -24, 2.0 ** -24, 2.0 ** -53, 5e-324, 1e-300, 0.5, 0.25, 0.75, 1.0 / 3.0,
              2.0 / 3.0, 7.6e-6, 1e-4]),
    rng.random(150, dtype=np.float32).astype(np.float64),       # the scatter's fractions: f32 widened (imageGroup.cxx:311)
    rng.random(150),                                             # the transform's: f64
    rng.random(40) * 1e-6, 1.0 - rng.random(40) * 1e-6])
w = oracle_api.bspline_weights(f, "reference")
out = {"source": "oracle/_ref/libfrog_refweights.so = /root/reference/registration/imageGroup.cxx:221-232 compiled as it is (g++ -O2)",
       "f_bits": [format(int(v), "016x") for v in f.view(np.uint64)],
       "weights_bits": [[format(int(v), "016x") for v in row] for row in w.view(np.uint64)]}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "weights_golden.json"), "w"))
print(len(f), "fractions written")

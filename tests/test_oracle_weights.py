"""D2 of SURVEY.md 8(a), pinned: the oracle's cubic B-spline weights against the REFERENCE's own
vtkBSplineTransformWeights (imageGroup.cxx:221-232) -- the one self-contained function of imageGroup.cxx, cut out of the
file where it lies under /root/reference and compiled by `make -C oracle ref` into oracle/_ref/libfrog_refweights.so
(oracle/ref_weights_api.cpp; nothing of the reference enters the tree).  Bit for bit, and against a committed fixture
generated from that build (tests/golden/make_weights_golden.py), so that the comparison also runs where the reference tree
and oracle/_ref are absent."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_api

HERE = os.path.dirname(os.path.abspath(__file__))


def fractions():
    """Every kind of fraction the scatter and the transforms produce: f32 fractions widened to f64 (imageGroup.cxx:311),
    f64 fractions of an f64 quotient (vtkBSplineTransform), the ends of [0, 1), denormals, values an ulp from a face."""
    rng = np.random.default_rng(5)
    f32 = rng.random(200000, dtype=np.float32).astype(np.float64)
    f64 = rng.random(200000)
    edge = np.array([0.0, 1.0 - 2.0 ** -53, 1.0 - 2.0 ** -24, 2.0 ** -24, 2.0 ** -53, 5e-324, 1e-300, 0.5, 0.25, 0.75,
                     1.0 / 3.0, 2.0 / 3.0, 7.6e-6, 1e-4])
    tiny = (rng.random(2000) * 1e-6)
    near_one = 1.0 - rng.random(2000) * 1e-6
    return np.concatenate([edge, f32, f64, tiny, near_one])


def test_oracle_weights_equal_the_reference_build_bit_for_bit():
    if oracle_api.ref_weights_lib() is None:
        pytest.skip("oracle/_ref/libfrog_refweights.so not built (reference tree absent)")
    f = fractions()
    got = oracle_api.bspline_weights(f, "oracle")
    want = oracle_api.bspline_weights(f, "reference")
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    # what the function is: a partition of unity of non-negative weights
    assert np.max(np.abs(want.sum(axis=1) - 1.0)) < 1e-15 and want.min() > -1e-16


def test_oracle_weights_equal_the_golden_fixture():
    """tests/golden/weights_golden.json: fractions and the reference build's weights as hex bit patterns."""
    fx = json.load(open(os.path.join(HERE, "golden", "weights_golden.json")))
    f = np.array([int(h, 16) for h in fx["f_bits"]], np.uint64).view(np.float64)
    want = np.array([[int(h, 16) for h in row] for row in fx["weights_bits"]], np.uint64)
    got = oracle_api.bspline_weights(f, "oracle")
    assert np.array_equal(got.view(np.uint64), want)

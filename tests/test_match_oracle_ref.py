"""SURVEY.md 8(f) row 2, pinned: the pairing oracle (oracle/match_oracle.cpp) against the REFERENCE's own ComputeMatches, scalar
`norm` and `struct Point` (match/match.cpp:255-336, :243-251, :28-48) -- the std-only pieces of match.cpp, cut out of the file
where it lies under /root/reference and compiled as they are by `make -C oracle ref` into oracle/_ref/libfrog_refmatch.so
(oracle/ref_match_api.cpp; nothing of the reference enters the tree).  Identical pair lists on every option set the GPU tests
use, and against a committed fixture generated from that build (tests/golden/make_match_golden.py), so that the comparison also
runs where the reference tree and oracle/_ref are absent."""
import base64
import json
import os

import numpy as np
import pytest

from frog_amd.match import Keypoints, all_pairs, synthetic_keypoints
from oracle import oracle_api

HERE = os.path.dirname(os.path.abspath(__file__))

OPTIONS = [dict(threshold=0.22), dict(threshold=1.0), dict(threshold=0.6, dist2second=0.8), dict(threshold=1.0, anat=30.0),
           dict(threshold=1.0, sym=1), dict(threshold=1e10, dist2second=1.0), dict(threshold=3e19),
           dict(all=1, threshold=0.9), dict(all=1, threshold=0.9, sym=1), dict(all=1, threshold=1.2, anat=40.0),
           dict(all=1, threshold=0.3), dict(all=1, threshold=1e10), dict(all=1, threshold=0.0)]


def same(got, want):
    assert len(got) == len(want)
    for k, ((ga, gb), (wa, wb)) in enumerate(zip(got, want)):
        assert np.array_equal(ga, wa) and np.array_equal(gb, wb), f"job {k}: {len(ga)} vs {len(wa)} pairs"


def need_ref():
    if oracle_api.ref_match_lib() is None:
        pytest.skip("oracle/_ref/libfrog_refmatch.so not built (reference tree absent)")


@pytest.mark.parametrize("opts", OPTIONS)
def test_oracle_pairs_equal_the_reference_build(opts):
    need_ref()
    imgs = synthetic_keypoints(4, 700, seed=11)                 # ragged sizes, as the GPU tests' groups
    imgs[2] = Keypoints.from_rows(imgs[2].rows()[:333])
    imgs[3] = Keypoints.from_rows(imgs[3].rows()[:33])
    jobs = all_pairs(4) + [(3, 0)]
    want = oracle_api.ref_match_run(imgs, jobs, **opts)
    same(oracle_api.match_run(imgs, jobs, **opts), want)
    if opts["threshold"] >= 0.6:
        assert sum(len(a) for a, _ in want) > 100


@pytest.mark.parametrize("dim", [8, 48, 50, 64, 100, 128])
def test_oracle_pairs_equal_the_reference_build_descriptor_lengths_ties_and_empty_images(dim):
    need_ref()
    rng = np.random.default_rng(dim)
    imgs = synthetic_keypoints(3, 400, dim=dim, seed=dim)
    rows_c, rows_q = imgs[0].rows(), imgs[1].rows()
    rows_c[rng.integers(0, 400, 60), 6:] = rows_c[rng.integers(0, 400, 60), 6:]          # duplicate candidates: first one wins
    rows_q[:40, 6:] = rows_c[rng.integers(0, 400, 40), 6:]                                # exact copies: d1 = 0
    rows_q[:40, 3:5] = 1.0; rows_c[:, 3:5] = 1.0
    e = imgs[2]
    imgs = [Keypoints.from_rows(rows_c), Keypoints.from_rows(rows_q), Keypoints(e.xyz[:0], e.scale[:0], e.laplacian[:0], e.response[:0], e.desc[:0])]
    jobs = [(0, 1), (1, 0), (0, 2), (2, 1)]
    for opts in (dict(threshold=1.2), dict(threshold=2.0, dist2second=1.5), dict(all=1, threshold=1.0, sym=1)):
        same(oracle_api.match_run(imgs, jobs, **opts), oracle_api.ref_match_run(imgs, jobs, **opts))


def test_oracle_norm_equals_the_reference_build_bit_for_bit():
    """The squared distance every comparison is made on: f32, summed in dimension order (match.cpp:243-251)."""
    need_ref()
    L = oracle_api.ref_match_lib()
    rng = np.random.default_rng(3)
    for dim in (1, 7, 48, 64, 129):
        a = rng.normal(size=(200, dim)).astype(np.float32) * np.float32(10.0) ** rng.integers(-3, 4, (200, 1)).astype(np.float32)
        b = rng.normal(size=(200, dim)).astype(np.float32)
        want = np.array([L.refmatch_norm(a[k].ctypes.data, b[k].ctypes.data, dim) for k in range(200)], np.float32)
        acc = np.zeros(200, np.float32)
        for d in range(dim):
            t = a[:, d] - b[:, d]
            acc = acc + t * t
        assert np.array_equal(acc.view(np.uint32), want.view(np.uint32))


def golden():
    fx = json.load(open(os.path.join(HERE, "golden", "match_golden.json")))
    f = lambda s, shape: np.frombuffer(base64.b64decode(s), "<f4").reshape(shape).copy()
    imgs = [Keypoints(f(i["xyz"], (i["n"], 3)), f(i["scale"], (i["n"],)), f(i["laplacian"], (i["n"],)),
                      np.zeros(i["n"], np.float32), f(i["desc"], (i["n"], fx["dim"]))) for i in fx["images"]]
    jobs = [tuple(j) for j in fx["jobs"]]
    return imgs, jobs, fx["cases"]


def test_oracle_pairs_equal_the_golden_fixture():
    """tests/golden/match_golden.json: three keypoint sets and the reference build's pair lists under twelve option sets."""
    imgs, jobs, cases = golden()
    assert len(cases) == 12
    for case in cases:
        want = [(np.array(a, np.uint32), np.array(b, np.uint32)) for a, b in case["pairs"]]
        same(oracle_api.match_run(imgs, jobs, **case["options"]), want)

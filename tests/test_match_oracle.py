"""Pairing oracle (oracle/match_oracle.cpp) against hand-worked cases of
ComputeMatches (match/match.cpp:255-336) -- and, since round 6, the same cases against the REFERENCE's own ComputeMatches
(oracle/_ref/libfrog_refmatch.so: the std-only pieces of match.cpp compiled as they are, oracle/ref_match_api.cpp), so the
hand-worked answers are checked against upstream's code too.  tests/test_match_oracle_ref.py compares the two on whole
synthetic groups and on a committed fixture."""
import numpy as np
import pytest

from frog_amd.match import Keypoints, all_pairs
from oracle import oracle_api


@pytest.fixture(params=["oracle", "reference"])
def match_run(request):
    if request.param == "oracle":
        return oracle_api.match_run
    if oracle_api.ref_match_lib() is None:
        pytest.skip("oracle/_ref/libfrog_refmatch.so not built (reference tree absent)")
    return oracle_api.ref_match_run


def kp(desc, scale=None, sign=None, xyz=None):
    desc = np.asarray(desc, np.float32)
    n = len(desc)
    return Keypoints(np.zeros((n, 3), np.float32) if xyz is None else xyz,
                     np.ones(n, np.float32) if scale is None else scale,
                     np.ones(n, np.float32) if sign is None else sign,
                     np.zeros(n, np.float32), desc)


def test_nearest_and_ratio_test(match_run):
    # image 0 = candidates, image 1 = queries (ComputeMatches(points2 = first, points1 = second));
    # all values exactly representable, so the f32 arithmetic is exact
    cand = kp([[0, 0], [1, 0], [0, 4]])
    qry = kp([[0.125, 0], [0.5, 0], [0, 3.5]])
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=10.0, dist2second=1.0)
    # query 0 -> cand 0 (d1 1/64, d2 49/64); query 1: d1 == d2 = 1/4 -> sqrt(1) < 1 fails;
    # query 2 -> cand 2 (d1 1/4, d2 12.25)
    assert a.tolist() == [0, 2] and b.tolist() == [0, 2]
    # ratio test sqrt(d1/d2) < dist2second: query 0 1/7 = .1428..., query 2 .5/3.5 = .1428...
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=10.0, dist2second=0.14)
    assert b.tolist() == []
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=10.0, dist2second=0.15)
    assert b.tolist() == [0, 2]
    # the distance threshold is on sqrt(d1), strict
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=0.5, dist2second=1.0)
    assert b.tolist() == [0]                     # query 2: sqrt(1/4) = .5 is not < .5
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=0.125, dist2second=1.0)
    assert b.tolist() == []
    # ties keep the FIRST candidate that attains the minimum (strict <), and d2 == d1 then
    cand2 = kp([[1, 0], [1, 0], [0, 0]])
    (a, b), = match_run([cand2, kp([[1, 0.5]])], [(0, 1)], threshold=10.0, dist2second=1.5)
    assert a.tolist() == [0]                     # d1 = d2 = 1/4 at candidates 0 and 1: sqrt(1) < 1.5
    # an exact duplicate: d1 = d2 = 0, sqrt(0/0) is NaN and NaN < x is false -> no pair
    (a, b), = match_run([cand2, kp([[1, 0]])], [(0, 1)], threshold=10.0, dist2second=1.5)
    assert a.tolist() == []


def test_single_candidate_has_no_second(match_run):
    cand = kp([[0, 0]])
    qry = kp([[0.3, 0.4]])
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=0.6, dist2second=0.5)
    assert a.tolist() == [0] and b.tolist() == [0]          # d2 == FLT_MAX accepts whatever the ratio


def test_sign_and_scale_filters(match_run):
    cand = kp([[0, 0], [0, 0.01]], scale=np.array([1.0, 1.0], np.float32), sign=np.array([1.0, -1.0], np.float32))
    qry = kp([[0, 0.01], [0, 0.01], [0, 0.01]], scale=np.array([1.0, 1.31, 1 / 1.31], np.float32),
             sign=np.array([-1.0, 1.0, 1.0], np.float32))
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=10.0)
    # query 0 (sign -1) only sees cand 1; queries 1 and 2 are outside the 1.3 scale ratio of cand 0
    assert a.tolist() == [1] and b.tolist() == [0]
    # the ratio compares f32 quotients with the double 1.3: 1.3f / 1 = 1.2999999523 is NOT > 1.3
    qry = kp([[0, 0]] * 2, scale=np.array([np.float32(1.3), np.nextafter(np.float32(1.3), np.float32(2))], np.float32))
    (a, b), = match_run([kp([[0, 0]]), qry], [(0, 1)], threshold=10.0)
    assert b.tolist() == [0]


def test_anatomical_test_and_sym(match_run):
    xyz_c = np.array([[0, 0, 0], [100, 0, 0]], np.float32)
    cand = kp([[0, 0], [0, 0.2]], xyz=xyz_c)
    qry = kp([[0, 0.19]], xyz=np.array([[1, 0, 0]], np.float32))
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=10.0, anat=5.0)
    assert a.tolist() == [0] and b.tolist() == [0]          # cand 1 is nearer in descriptor space but 99 mm away
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=10.0)
    assert a.tolist() == [1]
    # -sym appends the reverse direction: for each keypoint i of `first`, (i, its match in `second`)
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=10.0, sym=1)
    assert a.tolist() == [1, 0, 1] and b.tolist() == [0, 0, 0]


def test_stale_match_variable_quirk(match_run):
    # `match` is declared outside the query loop (match.cpp:259): a query without any candidate that
    # still passes the tests (needs sqrt(FLT_MAX) < threshold) re-emits the previous query's match
    cand = kp([[0, 0], [1, 1]], sign=np.array([1.0, 1.0], np.float32))
    qry = kp([[1, 1], [5, 5]], sign=np.array([1.0, -1.0], np.float32))
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=3e19)
    assert a.tolist() == [1, 1] and b.tolist() == [0, 1]
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=10.0)
    assert a.tolist() == [1] and b.tolist() == [0]


def test_job_order_and_all_pairs():
    assert all_pairs(3) == [(0, 1), (0, 2), (1, 2)]


def test_match_all_pushes_the_running_match_variable(match_run):
    """matchAll (match.cpp:297-302), worked by hand.  Candidates in 1-D at 0, 10, 3, 0.25, 9; threshold 1 (on sqrt(dist)).
    query 0 at 0:   cand 0: |0| < 1 -> push `match` (= 0, the initial value, :259); cand 1: 10 far -> d1 = 100, match = 1;
                    cand 2: 3 far, 9 < 100 -> match = 2; cand 3: 0.25 < 1 -> push 2; cand 4: 9 far, 81 > 9 -> match stays 2.
    query 1 at 9.5: cand 0: far, d1 = 90.25, match = 0; cand 1: 0.5 < 1 -> push 0; cand 2: far, 42.25 -> match = 2;
                    cand 3: far, 85.56 no; cand 4: 0.5 < 1 -> push 2.
    query 2 at 0.5: sign -1, sees nothing: no push, `match` stays 2.
    query 3 at 3.5: cand 0: far 12.25 -> match 0; cand 1: far 42.25; cand 2: 0.5 < 1 -> push 0; cand 3: far 10.56 -> match 3;
                    cand 4: far.
    The second-nearest test does not apply (:318)."""
    cand = kp([[0.0], [10.0], [3.0], [0.25], [9.0]])
    qry = kp([[0.0], [9.5], [0.5], [3.5]], sign=np.array([1, 1, -1, 1], np.float32))
    (a, b), = match_run([cand, qry], [(0, 1)], threshold=1.0, dist2second=0.0, all=1)
    assert a.tolist() == [0, 2, 0, 2, 0] and b.tolist() == [0, 0, 1, 1, 3]
    # carried over from query to query: a query whose first candidate is within the threshold pushes the previous query's match
    qry2 = kp([[9.5], [0.1]])
    (a, b), = match_run([cand, qry2], [(0, 1)], threshold=1.0, all=1)
    assert a.tolist() == [0, 2, 2, 2] and b.tolist() == [0, 0, 1, 1]      # query 1: cand 0 within -> 2 (query 0's last), cand 3 within -> still 2 (cand 1, 2 far: 98.01, 8.41 -> match 2)
    # -sym: the reverse direction appended with the pair turned round (make_pair(i, match), :299)
    (a, b), = match_run([cand, qry2], [(0, 1)], threshold=1.0, all=1, sym=1)
    assert a.tolist()[:4] == [0, 2, 2, 2] and b.tolist()[:4] == [0, 0, 1, 1]
    # reverse: queries = cand (5), candidates = qry2 (9.5, 0.1): q0 at 0: c0 far (match 0), c1 0.1 within -> push 0; q1 at 10: c0 0.5
    # within -> push 0; c1 far -> match 1; q2 at 3: both far, match -> c1 (8.41 < 42.25) = 1; q3 at 0.25: c0 far -> match 0, c1 0.15
    # within -> push 0; q4 at 9: c0 0.5 within -> push 0 (carried), c1 far -> match 1
    assert a.tolist()[4:] == [0, 1, 3, 4] and b.tolist()[4:] == [0, 0, 0, 0]

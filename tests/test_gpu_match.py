"""HIP matcher (include/frog_match.h) vs the pairing oracle: identical pair lists
(index work: bit-exact), through the C ABI."""
import numpy as np
import pytest

from frog_amd.match import Keypoints, Matcher, all_pairs, synthetic_keypoints
from oracle.oracle_api import match_run

pytestmark = pytest.mark.gpu


def same(got, want):
    assert len(got) == len(want)
    for k, ((ga, gb), (wa, wb)) in enumerate(zip(got, want)):
        assert np.array_equal(ga, wa) and np.array_equal(gb, wb), f"job {k}: {len(ga)} vs {len(wa)} pairs"


@pytest.mark.parametrize("opts", [dict(threshold=0.22), dict(threshold=1.0), dict(threshold=0.6, dist2second=0.8),
                                  dict(threshold=1.0, anat=30.0), dict(threshold=1.0, sym=1),
                                  dict(threshold=1e10, dist2second=1.0)])
def test_pairs_identical_to_oracle(opts):
    # ragged sizes: not multiples of the 256-query blocks or the 32-candidate tiles
    imgs = synthetic_keypoints(4, 1500, seed=11)
    imgs[2] = Keypoints.from_rows(imgs[2].rows()[:777])
    imgs[3] = Keypoints.from_rows(imgs[3].rows()[:33])
    jobs = all_pairs(4)
    m = Matcher(imgs)
    got = m.run(jobs, **opts)
    same(got, match_run(imgs, jobs, **opts))
    ms, nd = m.last_stats()
    assert ms > 0 and 0 < nd <= sum(imgs[a].n * imgs[b].n for a, b in jobs) * (2 if opts.get("sym") else 1)
    assert sum(len(a) for a, _ in got) > 0 or opts["threshold"] < 0.3


@pytest.mark.parametrize("dim", [8, 48, 50, 64, 100, 128])
def test_descriptor_lengths(dim):
    imgs = synthetic_keypoints(2, 600, dim=dim, seed=dim)
    got = Matcher(imgs).run([(0, 1), (1, 0)], threshold=1.2)
    same(got, match_run(imgs, [(0, 1), (1, 0)], threshold=1.2))
    assert len(got[0][0]) > 100


def test_ties_duplicates_and_range_boundaries():
    # duplicate descriptors: the FIRST candidate attaining the minimum wins (strict <) also when
    # the duplicates sit in different candidate ranges (the kernel splits candidates over blocks)
    rng = np.random.default_rng(5)
    base = synthetic_keypoints(2, 3000, seed=9)
    rows_c, rows_q = base[0].rows(), base[1].rows()
    rows_c[:, 4] = 1.0; rows_q[:, 4] = 1.0            # one Laplacian sign
    rows_c[:, 3] = 1.0; rows_q[:, 3] = 1.0            # one scale
    dup = rng.integers(0, 3000, 400)
    rows_c[rng.integers(0, 3000, 400), 6:] = rows_c[dup, 6:]              # candidates duplicated far apart
    rows_q[:200, 6:] = rows_c[rng.integers(0, 3000, 200), 6:] + np.float32(0.01)
    imgs = [Keypoints.from_rows(rows_c), Keypoints.from_rows(rows_q)]
    for opts in (dict(threshold=2.0, dist2second=1.5), dict(threshold=2.0, dist2second=1.0)):
        same(Matcher(imgs).run([(0, 1)], **opts), match_run(imgs, [(0, 1)], **opts))


def test_scale_ratio_boundary_and_filters():
    # scales straddling the 1.3 ratio by single ulps on both sides, both signs
    n = 512
    rng = np.random.default_rng(2)
    f = np.float32
    sc_c = rng.uniform(0.5, 8, n).astype(f)
    steps = rng.integers(-3, 4, n)
    sc_q = (sc_c * f(1.3)).astype(f)
    for _ in range(3):
        sc_q = np.where(steps > 0, np.nextafter(sc_q, f(100)), np.where(steps < 0, np.nextafter(sc_q, f(0)), sc_q)).astype(f)
        steps = steps - np.sign(steps)
    inv = rng.random(n) < 0.5
    sc_q = np.where(inv, (sc_c / f(1.3)).astype(f), sc_q).astype(f)
    d = rng.normal(size=(n, 48)).astype(f)
    cand = Keypoints(rng.uniform(0, 100, (n, 3)), sc_c, rng.choice(np.array([-1, 1], f), n), np.zeros(n, f), d)
    # every query's nearest descriptor is its own candidate: whether it survives is the scale test alone
    qry = Keypoints(cand.xyz, sc_q, cand.laplacian, np.zeros(n, f), d + f(1e-3))
    imgs = [cand, qry]
    got = Matcher(imgs).run([(0, 1)], threshold=0.5)
    want = match_run(imgs, [(0, 1)], threshold=0.5)
    same(got, want)
    assert 0 < len(got[0][0]) < n


def test_empty_image_single_candidate_and_stale_match_quirk():
    f = np.float32
    empty = Keypoints(np.zeros((0, 3), f), np.zeros(0, f), np.zeros(0, f), np.zeros(0, f), np.zeros((0, 48), f))
    imgs = synthetic_keypoints(2, 300, seed=4)
    one = Keypoints.from_rows(imgs[0].rows()[:1])
    group = [imgs[0], imgs[1], empty, one]
    jobs = [(0, 2), (2, 1), (3, 1), (1, 3), (0, 1)]
    for opts in (dict(threshold=1.0), dict(threshold=3e19), dict(threshold=3e19, sym=1)):
        same(Matcher(group).run(jobs, **opts), match_run(group, jobs, **opts))


def test_argument_validation():
    imgs = synthetic_keypoints(2, 50, seed=1)
    bad = Keypoints.from_rows(imgs[0].rows())
    bad.scale[3] = 0.0
    with pytest.raises(RuntimeError):
        Matcher([imgs[0], bad])
    with pytest.raises(RuntimeError):
        Matcher([imgs[0], synthetic_keypoints(1, 50, dim=32)[0]])
    with pytest.raises(RuntimeError):
        Matcher(imgs).run([(0, 5)])

"""HIP matcher (include/frog_match.h) vs the pairing oracle: identical pair lists
(index work: bit-exact), through the C ABI."""
import numpy as np
import pytest

from frog_amd.match import Keypoints, Matcher, all_pairs, synthetic_keypoints
from oracle.oracle_api import match_run

pytestmark = pytest.mark.gpu


def note(name, value):
    """Numbers worth keeping from a GPU run (gpurun_out/test_numbers.txt travels back with the call)."""
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "test_numbers.txt"), "a") as fh:
            fh.write(f"{name} {value}\n")


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


@pytest.mark.parametrize("opts", [dict(threshold=0.9), dict(threshold=0.9, sym=1), dict(threshold=1.2, anat=40.0),
                                  dict(threshold=0.3), dict(threshold=1e10), dict(threshold=0.0)])
def test_match_all_identical_to_oracle(opts):
    """matchAll (match.cpp:297-302): one pair per candidate within the threshold, carrying upstream's `match` variable --
    the nearest candidate so far BEYOND the threshold, in the caller's candidate order, carried over from query to query."""
    imgs = synthetic_keypoints(4, 1500, seed=13)
    imgs[2] = Keypoints.from_rows(imgs[2].rows()[:777])
    imgs[3] = Keypoints.from_rows(imgs[3].rows()[:33])
    jobs = all_pairs(4) + [(3, 0)]
    m = Matcher(imgs)
    got = m.run(jobs, all=1, **opts)
    want = match_run(imgs, jobs, all=1, **opts)
    same(got, want)
    n = sum(len(a) for a, _ in want)
    assert n == 0 if opts["threshold"] == 0.0 else (n > 100 or opts["threshold"] < 0.9)
    if opts["threshold"] == 1e10:           # every candidate that passes the filters is "within": the carried value is 0
        assert all((b == 0).all() if opts.get("sym") else (a == 0).all() for a, b in got)


@pytest.mark.parametrize("opts", [dict(threshold=1.0), dict(threshold=0.6, dist2second=0.8), dict(threshold=1.0, anat=30.0, sym=1),
                                  dict(threshold=1e10), dict(all=1, threshold=0.9, sym=1), dict(all=1, threshold=1.2, anat=40.0)])
def test_pairs_identical_to_the_reference_build(opts):
    """The device's pair lists against the REFERENCE's own ComputeMatches (oracle/_ref/libfrog_refmatch.so: match.cpp:255-336
    with its scalar norm and struct Point, compiled as they are; oracle/ref_match_api.cpp) -- no restatement in between."""
    from oracle import oracle_api
    if oracle_api.ref_match_lib() is None:
        pytest.skip("oracle/_ref/libfrog_refmatch.so not built (reference tree absent)")
    imgs = synthetic_keypoints(4, 1500, seed=17)
    imgs[2] = Keypoints.from_rows(imgs[2].rows()[:777])
    imgs[3] = Keypoints.from_rows(imgs[3].rows()[:33])
    jobs = all_pairs(4) + [(3, 0)]
    want = oracle_api.ref_match_run(imgs, jobs, **opts)
    same(Matcher(imgs).run(jobs, **opts), want)
    assert sum(len(a) for a, _ in want) > 100


def test_pairs_identical_to_the_golden_fixture_of_the_reference_build():
    """tests/golden/match_golden.json (generated from the reference build by tests/golden/make_match_golden.py)."""
    from test_match_oracle_ref import golden
    imgs, jobs, cases = golden()
    m = Matcher(imgs)
    for case in cases:
        want = [(np.array(a, np.uint32), np.array(b, np.uint32)) for a, b in case["pairs"]]
        same(m.run(jobs, **case["options"]), want)


@pytest.mark.parametrize("dim", [8, 100])
def test_match_all_descriptor_lengths_and_empty_images(dim):
    imgs = synthetic_keypoints(3, 300, dim=dim, seed=dim)
    e = imgs[2]
    imgs[2] = Keypoints(e.xyz[:0], e.scale[:0], e.laplacian[:0], e.response[:0], e.desc[:0])
    jobs = [(0, 1), (1, 0), (0, 2), (2, 1)]
    same(Matcher(imgs).run(jobs, all=1, threshold=1.0, sym=1), match_run(imgs, jobs, all=1, threshold=1.0, sym=1))


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


def test_matrix_core_filter_corner_cases(monkeypatch):
    """The MFMA filter (match.hip) must leave the pair lists of the exact vector kernel untouched: descriptor norms far
    from 1 (the bound scales with |q|^2 + |c|^2), hundreds of identical candidates (more half tiles inside the bound
    than the scan kernel lists: its slow path), near-ties closer than the bound, and the two paths against each other."""
    rng = np.random.default_rng(11)
    f = np.float32
    base = synthetic_keypoints(2, 2500, seed=13)
    cases = []
    for scale in (f(1e-3), f(37.0)):                                  # tiny and large norms
        cases.append([Keypoints(k.xyz, k.scale, k.laplacian, k.response, k.desc * scale) for k in base])
    rows_c, rows_q = base[0].rows(), base[1].rows()
    rows_c[:, 3] = 1.0; rows_q[:, 3] = 1.0; rows_c[:, 4] = 1.0; rows_q[:, 4] = 1.0
    rows_c[rng.permutation(2500)[:1500], 6:] = rows_c[7, 6:]           # 1 500 identical candidates, spread over every tile
    rows_q[:300, 6:] = rows_c[7, 6:] + f(1e-4) * rng.normal(size=(300, 48)).astype(f)
    cases.append([Keypoints.from_rows(rows_c), Keypoints.from_rows(rows_q)])
    rows_c, rows_q = base[0].rows(), base[1].rows()
    rows_c[:, 3] = 1.0; rows_q[:, 3] = 1.0; rows_c[:, 4] = 1.0; rows_q[:, 4] = 1.0
    for k in range(400):                                               # candidates 1e-6 apart: far inside the bound of 6e-5
        rows_c[(5 * k + 1) % 2500, 6:] = rows_c[(5 * k) % 2500, 6:] + f(1e-6) * rng.normal(size=48).astype(f)
    rows_q[:400, 6:] = rows_c[(5 * np.arange(400)) % 2500, 6:] + f(3e-4) * rng.normal(size=(400, 48)).astype(f)
    cases.append([Keypoints.from_rows(rows_c), Keypoints.from_rows(rows_q)])
    # a tight anatomical window: most keypoints keep no candidate or a single one (threshold of the filter = -inf)
    cases.append(base)
    for imgs in cases:
        for opts in (dict(threshold=1e9, dist2second=1.0), dict(threshold=1e9, dist2second=0.9, sym=1)) if imgs is not base else (dict(threshold=1e9, anat=6.0),):
            want = match_run(imgs, [(0, 1)], **opts)
            monkeypatch.delenv("FROG_MATCH_VALU", raising=False)
            got = Matcher(imgs).run([(0, 1)], **opts)
            monkeypatch.setenv("FROG_MATCH_VALU", "1")
            vec = Matcher(imgs).run([(0, 1)], **opts)
            monkeypatch.delenv("FROG_MATCH_VALU")
            same(got, want); same(vec, want)
            assert len(want[0][0]) > 100 or imgs is base


def test_bf16_filter_products_within_the_bound():
    """match.hip match_mfma16_kernel: -|q - c|^2 / 2 from three products of bf16 (hi, lo) splits on the bf16 matrix cores.
    The verification stage allows |-2 P - d| <= 2^-12 (|q|^2 + |c|^2); the derivation (5.3e-5 on P, 1.2e-4 on -2 P) assumes
    something about the instruction's internal f32 accumulation that the ISA document does not state, so the products are
    MEASURED here against f64, through the kernels' own operand builder and instruction sequence: unit-norm descriptors,
    norms of 1e-3 and 37, near-identical pairs (the distance is what is left of a cancellation of three terms of size S/2),
    descriptors of mixed magnitudes, one-hot descriptors, both descriptor lengths -- and the f32 chain beside it."""
    from frog_amd.match import filter_products
    rng = np.random.default_rng(5)
    f = np.float32
    worst = {0: 0.0, 1: 0.0}
    for dim in (48, 64):
        cases = []
        for scale in (1.0, 1e-3, 37.0):
            c = rng.normal(size=(32, dim)); c /= np.linalg.norm(c, axis=1, keepdims=True)
            q = rng.normal(size=(32, dim)); q /= np.linalg.norm(q, axis=1, keepdims=True)
            cases.append(((c * scale).astype(f), (q * scale).astype(f)))
        c = np.abs(rng.normal(size=(32, dim))); c /= np.linalg.norm(c, axis=1, keepdims=True)          # SURF-like: non-negative
        cases.append((c.astype(f), (c + 1e-4 * rng.normal(size=(32, dim))).astype(f)))                 # near-identical
        cases.append((c.astype(f), c[::-1].astype(f)))
        m = (rng.normal(size=(32, dim)) * 10.0 ** rng.integers(-6, 3, size=(32, dim))).astype(f)       # mixed magnitudes
        cases.append((m, (m * (1 + 1e-3 * rng.normal(size=m.shape))).astype(f)))
        hot = np.zeros((32, dim), f); hot[np.arange(32), rng.integers(0, dim, 32)] = f(3.0)
        cases.append((hot, hot[rng.permutation(32)]))
        ones = np.full((32, dim), f(1.0) + f(2.0) ** -9, f)                                            # every entry half way between two bf16 values
        cases.append((ones, (ones * f(1.001)).astype(f)))
        for cand, query in cases:
            d = ((cand.astype(np.float64)[:, None, :] - query.astype(np.float64)[None, :, :]) ** 2).sum(axis=2)
            S = (cand.astype(np.float64) ** 2).sum(axis=1)[:, None] + (query.astype(np.float64) ** 2).sum(axis=1)[None, :]
            for form in (0, 1):
                P, bound = filter_products(cand, query, form)
                err = np.abs(-2.0 * P.astype(np.float64) - d) / S
                worst[form] = max(worst[form], float(err.max()))
                assert err.max() <= 0.5 * bound, (dim, form, float(err.max()), bound)      # half the bound: the other half is head-room
    note("filter_products_max_error_over_S_f32_bf16", f"{worst[0]:.3e} {worst[1]:.3e} (bounds 3.05e-05 2.44e-04)")


def test_bf16_and_f32_filters_give_the_exact_kernels_pairs(monkeypatch):
    """The default filter is the bf16 one; FROG_MATCH_F32=1 keeps the f32 chain, FROG_MATCH_VALU=1 no filter at all: the three
    pair lists are the oracle's, on a group with duplicates and near-ties and for 64-value descriptors (13 steps of K = 16)."""
    base = synthetic_keypoints(3, 3000, seed=29)
    rows = [k.rows() for k in base]
    rng = np.random.default_rng(3)
    rows[1][:500, 6:] = rows[0][rng.integers(0, 3000, 500), 6:] + np.float32(2e-4) * rng.normal(size=(500, 48)).astype(np.float32)
    rows[2][:300, 6:] = rows[0][7, 6:]
    imgs = [Keypoints.from_rows(r) for r in rows]
    wide = [Keypoints(k.xyz, k.scale, k.laplacian, k.response, np.concatenate([k.desc, k.desc[:, :16] * np.float32(0.5)], axis=1)) for k in imgs]
    for group in (imgs, wide):
        jobs = [(0, 1), (0, 2), (1, 2)]
        opts = dict(threshold=1e9, dist2second=0.95, sym=1)
        want = match_run(group, jobs, **opts)
        for env in ({}, {"FROG_MATCH_F32": "1"}, {"FROG_MATCH_VALU": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got = Matcher(group).run(jobs, **opts)
            for k in env:
                monkeypatch.delenv(k)
            same(got, want)


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


def _read_pairs_bin(path):
    """pairs.bin as match writes it (match.cpp:684-742)."""
    import struct
    raw = open(path, "rb").read()
    off = 0
    n_img, = struct.unpack_from("<H", raw, off); off += 2
    images = []
    for _ in range(n_img):
        ln, = struct.unpack_from("<H", raw, off); off += 2
        name = raw[off:off + ln].decode(); off += ln
        rigid = struct.unpack_from("<3d", raw, off); off += 24
        npts, = struct.unpack_from("<I", raw, off); off += 4
        pts = np.frombuffer(raw, "<f4", count=6 * npts, offset=off).reshape(npts, 6); off += 24 * npts
        images.append((name, rigid, pts))
    blocks = []
    while off < len(raw):
        i, j, size = struct.unpack_from("<HHI", raw, off); off += 8
        pr = np.frombuffer(raw, "<u4", count=2 * size, offset=off).reshape(size, 2); off += 8 * size
        blocks.append((i, j, pr))
    return images, blocks


def test_match_cli_end_to_end(tmp_path):
    import os
    import subprocess
    from frog_amd.match import write_keypoints
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    imgs = synthetic_keypoints(4, 900, seed=21)
    names = []
    for i, (kp, ext) in enumerate(zip(imgs, ["csv.gz", "csv", "csv.gz", "csv.gz"])):
        p = tmp_path / f"points{i}.{ext}"
        write_keypoints(p, kp)
        names.append(str(p))
    # list file: absolute paths, optional rigid translation (match.cpp:459-494)
    (tmp_path / "list.txt").write_text("".join(f"{n},0,0,{5.0 * i}\n" for i, n in enumerate(names)))
    exe = os.path.join(root, "bin", "match")
    r = subprocess.run([exe, "list.txt", "-o", "pairs.bin", "-d", "1", "-np", "700", "-sp", "0.05"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Pairing..." in r.stdout and "Nb Match :" in r.stdout and "48 values per descriptor" in r.stdout
    images, blocks = _read_pairs_bin(tmp_path / "pairs.bin")
    assert [im[0] for im in images] == [os.path.basename(n) for n in names]
    assert [im[1][2] for im in images] == [0.0, 5.0, 10.0, 15.0]
    # the pruning of match.cpp:566-581: response >= sp, then the np best responses (partial_sort, descending)
    pruned = []
    for kp, (_, _, pts) in zip(imgs, images):
        rows = kp.rows()
        rows = rows[rows[:, 5] >= np.float32(0.05)]
        rows = rows[np.argsort(-rows[:, 5], kind="stable")][:700]
        assert len(pts) == len(rows) and np.array_equal(pts[:, 5], rows[:, 5]) and np.array_equal(pts[:, :3], rows[:, :3])
        pruned.append(Keypoints.from_rows(rows))
    want = match_run(pruned, all_pairs(4), threshold=1.0)
    assert [(b[0], b[1]) for b in blocks] == all_pairs(4)
    for (i, j, pr), (wa, wb) in zip(blocks, want):
        assert np.array_equal(pr[:, 0], wa) and np.array_equal(pr[:, 1], wb)
    assert sum(len(b[2]) for b in blocks) == int(r.stdout.split("Nb Match :")[1].split()[0]) > 500
    # -sym and -targ
    r = subprocess.run([exe, "list.txt", "-o", "sym.bin", "-d", "1", "-sym", "-targ", "2"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    _, blocks = _read_pairs_bin(tmp_path / "sym.bin")
    jobs = [(0, 2), (1, 2)]
    want = match_run(imgs, jobs, threshold=1.0, sym=1)
    assert [(b[0], b[1]) for b in blocks] == jobs
    for (i, j, pr), (wa, wb) in zip(blocks, want):
        assert np.array_equal(pr[:, 0], wa) and np.array_equal(pr[:, 1], wb)
    # the pairs file feeds frog as is
    r = subprocess.run([os.path.join(root, "bin", "frog"), "pairs.bin", "-li", "3", "-dl", "0", "-q", "1"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "pairs read" in r.stdout, r.stdout[-1500:]
    # -all (match.cpp:406: a flag that, like -p, swallows the next word)
    r = subprocess.run([exe, "list.txt", "-all", "x", "-o", "all.bin", "-d", "0.9", "-sym"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    _, blocks = _read_pairs_bin(tmp_path / "all.bin")
    want = match_run(imgs, all_pairs(4), threshold=0.9, sym=1, all=1)
    assert [(b[0], b[1]) for b in blocks] == all_pairs(4)
    for (i, j, pr), (wa, wb) in zip(blocks, want):
        assert np.array_equal(pr[:, 0], wa) and np.array_equal(pr[:, 1], wb)
    assert sum(len(b[2]) for b in blocks) > 100
    # -anat with -transformPrefix (match.cpp:517-558, :278-290): the anatomical-distance test is taken on positions
    # moved by <prefix><image>.json.  Image 1 is written 150 mm off; its transform brings it back.
    import json
    two = [imgs[0], Keypoints(imgs[1].xyz + np.float32([150, 0, 0]), imgs[1].scale, imgs[1].laplacian, imgs[1].response, imgs[1].desc)]
    sub = tmp_path / "anat"; sub.mkdir()
    for i, kp in enumerate(two):
        write_keypoints(sub / f"points{i}.csv.gz", kp)
    (sub / "list.txt").write_text("".join(f"{sub}/points{i}.csv.gz\n" for i in range(2)))
    for i, tx in enumerate((0.0, -150.0)):
        M = np.eye(4); M[0, 3] = tx
        (sub / f"t{i}.json").write_text(json.dumps({"transforms": [{"type": "vtkMatrixToLinearTransform", "matrix": M.ravel().tolist()}]}))
    want = match_run(imgs[:2], [(0, 1)], threshold=1.0, anat=20.0)[0]
    r = subprocess.run([exe, "list.txt", "-o", "far.bin", "-d", "1", "-anat", "20"], cwd=sub, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    far = _read_pairs_bin(sub / "far.bin")[1][0][2]
    r = subprocess.run([exe, "list.txt", "-o", "near.bin", "-d", "1", "-anat", "20", "-transformPrefix", str(sub / "t")], cwd=sub,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Reading transform" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    images, blocks = _read_pairs_bin(sub / "near.bin")
    assert np.array_equal(blocks[0][2][:, 0], want[0]) and np.array_equal(blocks[0][2][:, 1], want[1])
    assert len(want[0]) > 300 and len(far) < len(want[0]) // 10
    assert np.array_equal(images[1][2][:, :3], two[1].xyz)                      # pairs.bin keeps the original coordinates
    assert subprocess.run([exe, "list.txt", "-transformPrefix", "missing"], cwd=sub, capture_output=True).returncode == 1
    assert subprocess.run([exe], capture_output=True, text=True).stdout.startswith("Usage : match pointFiles.txt")

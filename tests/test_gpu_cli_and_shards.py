"""bin/frog end to end against the oracle, and the sharded (2 ranks, one GPU, gloo)
run against the single-context run."""
import csv
import json
import os
import socket
import subprocess

import numpy as np
import pytest

from frog_amd import _abi
from frog_amd.pairs import Pairs
from oracle.oracle_api import OracleGroup

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL = 1e-4


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)


def test_cli_outputs_match_oracle(tmp_path, small_pairs):
    small_pairs.write(tmp_path / "pairs.bin")
    r = subprocess.run([os.path.join(ROOT, "bin", "frog"), "pairs.bin", "-li", "20", "-dl", "2", "-di", "15", "-j"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Linear registration" in r.stdout and "Total time" in r.stdout and "half pairs" in r.stdout

    ref = OracleGroup(small_pairs.model, _abi.FrogOptions.default())
    # histograms_linear.csv is written before the deformable stage: replay run() by hand
    ref.setup_stats(); ref.linear_init(); ref.transform_points()
    E = []
    for it in range(20):
        if it % 10 == 0:
            ref.update_stats()
        E.append(ref.linear_step()); ref.transform_points()
    ref.transform_points(True)
    hist_lin = [ref.histogram(i) for i in range(small_pairs.n_images)]
    for level in range(2):
        ref.deformable_setup(level, _abi.FrogGridInfo()); ref.transform_points()
        alpha, nd, it = np.float32(0.02), 0, 0
        while it < 15:
            if it % 10 == 0:
                ref.update_stats()
            e = np.float32(ref.deformable_step(float(alpha)))
            if e < 0:
                if nd == 0:
                    alpha = np.float32(alpha / np.float32(2))
                ref.transform_points(True); ref.deformable_setup(level, _abi.FrogGridInfo()); ref.transform_points()
                nd = 0
                continue
            nd += 1; ref.transform_points(); E.append(float(e)); it += 1
        ref.transform_points(True)

    rows = list(csv.reader(open(tmp_path / "measures.csv")))
    assert rows[0] == ["Iteration", " E", " landmarkAv", " landmarkMax", " landmarkSTD"]
    got_e = np.array([float(x[1]) for x in rows[1:]])
    assert len(got_e) == len(E) and np.max(np.abs(got_e - np.array(E)) / np.array(E)) < 1e-3   # 6 printed digits

    rows = list(csv.reader(open(tmp_path / "histograms_linear.csv")))
    assert rows[0] == [f"image {i}" for i in range(small_pairs.n_images)]
    got = np.array([[float(v) for v in r_] for r_ in rows[1:]])
    for i, h in enumerate(hist_lin):
        # bins are bit-exact on identical coordinates; after 20 iterations the coordinates agree to
        # f32 rounding, so a sample sitting on a bin edge may move: compare counts with that slack
        assert got[:, i].sum() == h.sum()
        n = max(len(h), got.shape[0])
        a = np.zeros(n); a[:got.shape[0]] = got[:, i]
        b = np.zeros(n); b[:len(h)] = h
        assert np.abs(a - b).sum() <= 8

    for i in range(small_pairs.n_images):
        t = json.load(open(tmp_path / "transforms" / f"{i}.json"))["transforms"]
        assert t[0]["type"] == "vtkMatrixToLinearTransform" and len(t) == 1 + ref.num_grids()
        m = np.array(t[0]["matrix"]).reshape(4, 4)
        assert relerr(np.diag(m)[:3], np.diag(ref.matrix(i))[:3]) < REL and relerr(m[:3, 3], ref.matrix(i)[:3, 3]) < REL
        for k in range(ref.num_grids()):
            info, c = ref.grid(i, k, _abi.FrogGridInfo())
            assert t[1 + k]["type"] == "vtkBSplineTransform" and t[1 + k]["dimensions"] == list(info.dims)
            assert relerr(np.array(t[1 + k]["coeffs"]).reshape(-1, 3), c) < REL
    # bbox.json = countInliers' records (imageGroup.cxx:1036-1058, what FROG.py's consumers read) + the bounding box
    # (:1493-1511), every field.  Both sides are free-running here, so coordinates agree to rounding, not bits: the
    # integer fields that do not depend on them are equal, the census may differ by the few links that sit on the
    # threshold (its exactness on identical inputs is test_gpu_round2.py::test_census_is_exact_on_identical_inputs).
    bbox = json.load(open(tmp_path / "bbox.json"))
    rc = ref.count_inliers((_abi.FrogCounts * small_pairs.n_images)())
    assert sorted(bbox) == ["bbox", "halfPairs", "images", "inliers", "outlierRatio", "outliers"]
    assert bbox["halfPairs"] == small_pairs.n_half_links and len(bbox["images"]) == small_pairs.n_images
    tot_in = tot_out = 0
    for i, rec in enumerate(bbox["images"]):
        assert sorted(rec) == ["EMStats", "inliers", "outliers", "pairs", "points"]
        assert rec["points"] == rc[i].points and rec["pairs"] == rc[i].pairs
        assert rec["inliers"] + rec["outliers"] == rec["pairs"] and abs(rec["inliers"] - rc[i].inliers) <= 2
        for k, want in (("c1", rc[i].c1), ("c2", rc[i].c2), ("ratio", rc[i].ratio)):
            assert abs(rec["EMStats"][k] - want) <= 1e-4 * abs(want), (i, k)
        tot_in += rec["inliers"]; tot_out += rec["outliers"]
    assert bbox["inliers"] == tot_in and bbox["outliers"] == tot_out
    assert abs(bbox["outlierRatio"] - tot_out / bbox["halfPairs"]) < 1e-12
    x = ref.xyz().astype(np.float64)
    assert relerr(bbox["bbox"], [x.min(axis=0), x.max(axis=0)]) < 1e-6
    assert os.path.exists(tmp_path / "histograms.csv")


def test_cli_default_surface_sidecars_error_maps_and_pairs_csv(tmp_path, small_pairs):
    # default mode (no -j): coefficients in <i>.json.<n>.nii.gz sidecars named by "file"
    # (tools/transformIO.h:196-208), errorMaps/<i>.nii.gz (imageGroup.cxx:475-567), and with
    # -wp 1 pairs.csv.gz (:924-986).  The -j run of the same input is the cross-check.
    import gzip
    from nifti_util import read_nifti
    exe = os.path.join(ROOT, "bin", "frog")
    args = ["pairs.bin", "-li", "12", "-dl", "2", "-di", "8", "-q", "1"]
    for sub, extra in (("compact", ["-wp", "1"]), ("single", ["-j"])):
        d = tmp_path / sub
        d.mkdir()
        small_pairs.write(d / "pairs.bin")
        r = subprocess.run([exe] + args + extra, cwd=d, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    n_grids = None
    for i in range(small_pairs.n_images):
        tc = json.load(open(tmp_path / "compact" / "transforms" / f"{i}.json"))["transforms"]
        ts = json.load(open(tmp_path / "single" / "transforms" / f"{i}.json"))["transforms"]
        assert len(tc) == len(ts) and tc[0] == ts[0] or relerr(tc[0]["matrix"], ts[0]["matrix"]) < 1e-12
        n_grids = len(tc) - 1
        for k in range(1, len(tc)):
            assert tc[k]["type"] == "vtkBSplineTransform" and set(tc[k]) == {"type", "file"}
            assert tc[k]["file"] == f"{i}.json.{k - 1}.nii.gz"
            h, vox = read_nifti(tmp_path / "compact" / "transforms" / tc[k]["file"])
            assert list(h["dim"][1:4]) == ts[k]["dimensions"] and h["dim"][5] == 3
            np.testing.assert_allclose(h["pixdim"][1:4], ts[k]["spacing"], rtol=1e-6)
            np.testing.assert_allclose(h["qoffset"], ts[k]["origin"], rtol=1e-6, atol=1e-4)
            # two runs differ by the order of the lattice's float atomics only
            assert relerr(vox, np.array(ts[k]["coeffs"]).reshape(-1, 3)) < REL
        h, vox = read_nifti(tmp_path / "compact" / "errorMaps" / f"{i}.nii.gz")
        assert list(h["dim"][1:4]) == ts[-1]["dimensions"] and h["dim"][5] == 4
        assert (vox[:, 3] >= 0).all() and vox[:, 3].sum() > 0
        assert (vox[vox[:, 3] == 0][:, :3] == 0).all()
    assert n_grids >= 2
    rows = gzip.open(tmp_path / "compact" / "pairs.csv.gz", "rt").read().split("\n")
    assert len(rows) == small_pairs.n_half_links and all(len(r_.split(",")) == 6 for r_ in rows[:50])
    dist = np.array([float(r_.split(",")[4]) for r_ in rows])
    prob = np.array([float(r_.split(",")[5]) for r_ in rows])
    assert (np.diff(dist) >= 0).all() and ((prob >= 0) & (prob <= 1)).all()
    assert not os.path.exists(tmp_path / "single" / "pairs.csv.gz")


def test_cli_usage_and_bad_input(tmp_path):
    exe = os.path.join(ROOT, "bin", "frog")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage : frog inputPairs.bin [options]" in r.stdout
    (tmp_path / "bad.bin").write_bytes(b"\x02\x00garbage")
    r = subprocess.run([exe, "bad.bin"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from frog_amd import _abi
from frog_amd.pairs import Pairs
from frog_amd.distributed import HipEngine, ShardedImageGroup, plan_shards
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share GPU 0; gloo moves the tensors
pairs = Pairs.synthetic(int(os.environ.get("FROG_TEST_IMAGES", "6")), 3000, 1500, seed=7)
shards = plan_shards(pairs.row_ptr, pairs.point_offset, world)
eng = HipEngine(pairs, _abi.FrogOptions.default(), 0, shards[rank])
g = ShardedImageGroup(eng, shards, pairs.point_offset, rank, world)
g.linearIterations, g.deformableLevels, g.deformableIterations = 12, 2, 12
E = g.run()
torch.cuda.synchronize()
b, e = shards[rank]
res = {"E": E, "grids": g.gridsPerLevel, "range": [b, e],
       "matrix": {i: eng.matrix(i).tolist() for i in range(b, e)},
       "coeff": {i: [eng.grid(i, k)[1].tolist() for k in range(eng.num_grids())] for i in range(b, e)},
       "xyz2": eng.points()[1].tolist()}
json.dump(res, open(sys.argv[2] + f".{rank}", "w"))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def _launch(tmp_path, world, tag, images=6, extra_env=None):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FROG_TEST_IMAGES=str(images), **(extra_env or {}))
        procs.append(subprocess.Popen(["python", str(script), ROOT, str(tmp_path / tag)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [json.load(open(str(tmp_path / tag) + f".{r}")) for r in range(world)]


def test_two_ranks_match_one_rank(tmp_path):
    one = _launch(tmp_path, 1, "one")[0]
    two = _launch(tmp_path, 2, "two")
    assert two[0]["E"] == two[1]["E"] and two[0]["grids"] == two[1]["grids"] == one["grids"]
    assert np.max(np.abs(np.array(two[0]["E"]) - np.array(one["E"])) / np.array(one["E"])) < 1e-6
    assert two[0]["range"][1] == two[1]["range"][0]
    # replicas identical on both ranks and equal to the unsharded coordinates (f64 sum order only)
    assert np.array_equal(np.array(two[0]["xyz2"]), np.array(two[1]["xyz2"]))
    assert relerr(two[0]["xyz2"], one["xyz2"]) < 1e-6
    for res in two:
        for i, m in res["matrix"].items():
            assert relerr(m, one["matrix"][i]) < 1e-6
        for i, grids in res["coeff"].items():
            for k, c in enumerate(grids):
                assert relerr(c, one["coeff"][i][k]) < 1e-5


def test_three_ranks_with_ragged_shards_match_one_rank(tmp_path):
    """Seven images over three ranks: shards of 2 / 2 / 3 images, i.e. unequal row counts -- the all-gather of the coordinates
    goes through the padded slab and frog_comm_unpack_slab (one launch for all the other ranks' rows), as the 100 images of
    cfg 3 do over 8 ranks."""
    one = _launch(tmp_path, 1, "one7", images=7)[0]
    three = _launch(tmp_path, 3, "three7", images=7)
    sizes = sorted(r["range"][1] - r["range"][0] for r in three)
    assert sizes == [2, 2, 3]
    assert three[0]["E"] == three[1]["E"] == three[2]["E"] and three[0]["grids"] == one["grids"]
    assert np.max(np.abs(np.array(three[0]["E"]) - np.array(one["E"])) / np.array(one["E"])) < 1e-6
    assert np.array_equal(np.array(three[0]["xyz2"]), np.array(three[1]["xyz2"]))
    assert np.array_equal(np.array(three[0]["xyz2"]), np.array(three[2]["xyz2"]))
    assert relerr(three[0]["xyz2"], one["xyz2"]) < 1e-6
    for res in three:
        for i, grids in res["coeff"].items():
            for k, c in enumerate(grids):
                assert relerr(c, one["coeff"][i][k]) < 1e-5


def test_ragged_ranks_with_a_zero_skin_list(tmp_path):
    """The same three ranks with culling lists that have NO skin (FROG_CULL_SKIN=1.0,0.0, also for the linear stage): any
    movement of any point -- this rank's or another's -- invalidates the list, so a rank that failed to notice another rank's
    points moving (their displacement is measured while their rows are unpacked from the gathered slab) would sweep a stale
    list and drift away from the one-rank run, whose results do not depend on the skin at all."""
    zero = {"FROG_CULL_SKIN": "1.0,0.0", "FROG_CULL_SKIN_LINEAR": "1.0,0.0"}
    one = _launch(tmp_path, 1, "one7d", images=7)[0]
    three = _launch(tmp_path, 3, "three7z", images=7, extra_env=zero)
    assert three[0]["E"] == three[1]["E"] == three[2]["E"] and three[0]["grids"] == one["grids"]
    assert np.max(np.abs(np.array(three[0]["E"]) - np.array(one["E"])) / np.array(one["E"])) < 1e-6
    assert np.array_equal(np.array(three[0]["xyz2"]), np.array(three[2]["xyz2"]))
    assert relerr(three[0]["xyz2"], one["xyz2"]) < 1e-6


def test_cli_validation_landmarks(tmp_path):
    # -l <dir>: one file per image (sorted names), "name,x,y,z" lines with x and y inverted (-il 1 is
    # the default); landmarks become link-less points of their image (imageGroup.cxx:1161-1227), the
    # per-iteration measures and the two landmark reports come from their xyz2 (:1229-1351)
    pairs = Pairs.synthetic(5, 2500, 1200, seed=13)
    pairs.write(tmp_path / "pairs.bin")
    rng = np.random.default_rng(3)
    po = np.asarray(pairs.point_offset)
    ld = tmp_path / "landmarks"
    ld.mkdir()
    # anatomical correspondences: keypoints of image 0 whose nearest matched partners in every other image
    # sit at the same common-space place (the partners of a true match), slightly jittered
    rp, li, lp = np.asarray(pairs.row_ptr), np.asarray(pairs.link_image), np.asarray(pairs.link_point)
    xyz_all = np.asarray(pairs.xyz)
    chosen = []
    for p0 in range(po[1]):
        links = {int(li[l]): int(lp[l]) for l in range(rp[p0], rp[p0 + 1])}
        if set(links) == {1, 2, 3, 4}:
            chosen.append((p0, links))
        if len(chosen) == 3:
            break
    assert len(chosen) == 3
    extra = []
    for i in range(5):
        pick = np.array([xyz_all[p0] if i == 0 else xyz_all[po[i] + links[i]] for p0, links in chosen], np.float32)
        pick = pick + rng.normal(0, 0.5, pick.shape).astype(np.float32)
        names = ["apex", "base", "carina"]
        if i == 0:                                            # one image may hold a landmark twice
            names = names + ["apex"]; pick = np.concatenate([pick, pick[:1] + np.float32(0.25)])
        extra.append((names, pick.astype(np.float32)))
        with open(ld / f"img{i:02d}.csv", "w") as f:
            f.write("# name,x,y,z\n")
            for n_, p in zip(names, pick):
                f.write(f"{n_},{-p[0]:.9g},{-p[1]:.9g},{p[2]:.9g}\n")
    r = subprocess.run([os.path.join(ROOT, "bin", "frog"), "pairs.bin", "-l", "landmarks", "-li", "12", "-dl", "1", "-di", "6", "-j"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    n_land = sum(len(n_) for n_, _ in extra)
    assert f", {n_land} landmarks:max=" in r.stdout

    # the same group in the oracle: landmarks appended as link-less points
    ref_pairs = Pairs.synthetic(5, 2500, 1200, seed=13)
    for i, (_, pick) in enumerate(extra):
        ref_pairs.append_points(i, pick)
    rpo = np.asarray(ref_pairs.point_offset)
    assert list(np.diff(rpo)) == [2500 + len(n_) for n_, _ in extra]
    groups = {}
    for i, (names, _) in enumerate(extra):
        for k, n_ in enumerate(names):
            groups.setdefault(n_, []).append(rpo[i] + 2500 + k)
    order = sorted(groups)                                   # std::map iterates in key order

    def measure(xyz2):
        d = []
        for n_ in order:
            pts = xyz2[groups[n_]].astype(np.float32)
            c = np.zeros(3, np.float32)
            for p in pts:
                c += p / np.float32(len(pts))
            d += list(np.sqrt(((pts - c) ** 2).sum(1)))
        d = np.array(d, np.float64)
        return d.mean(), d.max(), np.sqrt((d * d).mean() - d.mean() ** 2), d

    ref = OracleGroup(ref_pairs.model, _abi.FrogOptions.default())
    ref.setup_stats(); ref.linear_init(); ref.transform_points()
    want = []
    for it in range(12):
        if it % 10 == 0:
            ref.update_stats()
        ref.linear_step(); ref.transform_points(); want.append(measure(ref.xyz2())[:3])
    ref.transform_points(True)
    ref.deformable_setup(0, _abi.FrogGridInfo()); ref.transform_points()
    for it in range(6):
        if it % 10 == 0:
            ref.update_stats()
        assert ref.deformable_step(0.02) >= 0
        ref.transform_points(); want.append(measure(ref.xyz2())[:3])
    ref.transform_points(True)
    rows = list(csv.reader(open(tmp_path / "measures.csv")))[1:]
    assert len(rows) == len(want)
    got = np.array([[float(v) for v in r_[2:5]] for r_ in rows])
    assert np.max(np.abs(got - np.array(want)) / np.maximum(np.array(want), 1e-3)) < 2e-3        # 6 printed digits, f32 coordinates
    assert got[-1, 0] < got[0, 0]                               # registration brings the landmarks together
    # reports
    final = measure(ref.xyz2())[3]
    lines = [l.split(",") for l in open(tmp_path / "distances.txt").read().split()]
    assert [l[1] for l in lines] == [n_ for n_ in order for _ in groups[n_]]
    assert np.allclose([float(l[0]) for l in lines], final, rtol=2e-3, atol=1e-3)
    tl = json.load(open(tmp_path / "transformedLandmarks.json"))
    assert sorted(tl) == order and [e["image"] for e in tl["apex"]] == [0, 0, 1, 2, 3, 4] and len(tl["base"]) == 5
    assert np.allclose(tl["carina"][0]["xyz"], ref.xyz2()[groups["carina"][0]], rtol=1e-4, atol=1e-3)
    bbox = json.load(open(tmp_path / "bbox.json"))
    assert not bbox.get("images") or bbox["images"][0]["points"] == 2504
    # -lc: the same landmarks as constraints pull their copies together much harder than the matches alone
    r = subprocess.run([os.path.join(ROOT, "bin", "frog"), "pairs.bin", "-lc", "landmarks", "-li", "12", "-dl", "1", "-di", "6", "-j",
                        "-mf", "measures_lc.csv"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lc = np.array([[float(v) for v in r_[2:5]] for r_ in list(csv.reader(open(tmp_path / "measures_lc.csv")))[1:]])
    assert np.allclose(lc[:12], got[:12], rtol=1e-5)          # the linear stage ignores hard links
    assert lc[-1, 0] < 0.95 * got[-1, 0]


def test_cli_incremental_registration_with_fixed_images(tmp_path):
    """The workflow of tools/register.py:88-94: register a group, then register an image against it with the
    group's images fixed at their transforms (`-fi n -fd dir`, RANSAC by default)."""
    from frog_amd.chain import Chain, read_transform
    from frog_amd.pairs import Pairs
    # images that differ by translations and smooth bumps only: a similarity + one lattice level can register them
    small_pairs = Pairs.synthetic(6, 3000, 1500, seed=11, scale_min=1.0, scale_max=1.0)
    n = small_pairs.n_images
    first = tmp_path / "group"; second = tmp_path / "added"
    first.mkdir(); second.mkdir()
    small_pairs.write(first / "pairs.bin")
    small_pairs.write(second / "pairs.bin")
    exe = os.path.join(ROOT, "bin", "frog")
    r = subprocess.run([exe, "pairs.bin", "-li", "30", "-dl", "1", "-di", "20", "-q", "1"], cwd=first, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([exe, "pairs.bin", "-j", "-fd", str(first / "transforms"), "-fi", str(n - 1), "-dl", "1", "-di", "20",
                        "-ri", "800", "-rb", "4", "-q", "1"], cwd=second, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"RANSAC registration for image {n - 1}: " in r.stdout and "Reading transforms in directory" in r.stdout
    assert "Linear registration, iteration" not in r.stdout                       # :40-49 replaces the linear loop
    # only the moving image gets outputs (:481, :1464); bbox.json lists {} for the fixed ones (:995-1000)
    assert sorted(os.listdir(second / "transforms")) == [f"{n - 1}.json"]
    assert sorted(os.listdir(second / "errorMaps")) == [f"{n - 1}.nii.gz"]
    bbox = json.load(open(second / "bbox.json"))
    assert bbox["images"][:n - 1] == [{}] * (n - 1) and bbox["images"][n - 1]["pairs"] > 0
    assert bbox["RANSAC"][0]["image"] == n - 1 and bbox["RANSAC"][0]["threshold"] == 50

    # the oracle on the same inputs: fixed images moved by the first run's transforms
    po = small_pairs.point_offset
    xyz = np.array(small_pairs.xyz, np.float32)
    for i in range(n - 1):
        p = xyz[po[i]:po[i + 1]]
        for link in read_transform(first / "transforms" / f"{i}.json"):           # float between links, as vtkGeneralTransform
            p = Chain([link]).apply(p.astype(np.float64)).astype(np.float32)
        xyz[po[i]:po[i + 1]] = p
    blocks = [small_pairs.block(b) for b in range(small_pairs.n_blocks)]
    moved = Pairs.from_arrays(po, xyz, blocks)
    ref = OracleGroup(moved.model, _abi.FrogOptions.default(n_fixed_images=n - 1))
    ref.setup_stats(); ref.linear_init(); ref.transform_points()
    best = ref.ransac(n - 1, iterations=800, batches=4)
    assert bbox["RANSAC"][0]["inliers"] == best
    ref.transform_points(); ref.update_stats(); ref.transform_points(True)
    ref.deformable_setup(0, _abi.FrogGridInfo()); ref.transform_points()
    E = []
    for it in range(20):
        if it % 10 == 0:
            ref.update_stats()
        E.append(ref.deformable_step(0.02)); ref.transform_points()
    rows = list(csv.reader(open(second / "measures.csv")))
    got_e = np.array([float(x[1]) for x in rows[1:]])
    assert len(got_e) == len(E) and np.max(np.abs(got_e - np.array(E)) / np.array(E)) < 1e-3
    t = json.load(open(second / "transforms" / f"{n - 1}.json"))["transforms"]
    m = np.array(t[0]["matrix"]).reshape(4, 4)
    assert np.allclose(m, ref.matrix(n - 1), rtol=1e-6, atol=1e-6)
    assert abs(np.linalg.det(m[:3, :3])) > 0 and not np.allclose(m[:3, :3], np.diag(np.diag(m[:3, :3])))   # a rotation, unlike the linear stage
    info, c = ref.grid(n - 1, 0, _abi.FrogGridInfo())
    assert relerr(np.array(t[1]["coeffs"]).reshape(-1, 3), c) < REL
    # and the point of it all: the added image lands where the group registration had put it
    rec = bbox["images"][n - 1]
    print("inliers", rec["inliers"], "of", rec["pairs"], "E", got_e[0], got_e[-1])
    assert rec["inliers"] > 0.5 * rec["pairs"] and got_e[-1] < got_e[0]
    own = np.array(small_pairs.xyz[po[n - 1]:po[n]], np.float64)
    a = Chain(read_transform(first / "transforms" / f"{n - 1}.json")).apply(own)
    b = Chain(read_transform(second / "transforms" / f"{n - 1}.json")).apply(own)
    print("median distance", np.median(np.linalg.norm(a - b, axis=1)), np.median(np.linalg.norm(a - own, axis=1)))
    assert np.median(np.linalg.norm(a - b, axis=1)) < 5.0 < np.median(np.linalg.norm(a - own, axis=1))


def _run_frog(cwd, *flags, env=None):
    r = subprocess.run([os.path.join(ROOT, "bin", "frog"), "pairs.bin", "-li", "12", "-dl", "2", "-di", "10", "-j", "-q", "1", *flags],
                       cwd=cwd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def _compare_runs(a, b, n_images, tol=1e-6):
    ea = np.array([float(x[1]) for x in list(csv.reader(open(a / "measures.csv")))[1:]])
    eb = np.array([float(x[1]) for x in list(csv.reader(open(b / "measures.csv")))[1:]])
    assert len(ea) == len(eb) and np.max(np.abs(ea - eb) / eb) < 1e-5       # six printed digits
    for i in range(n_images):
        ta = json.load(open(a / "transforms" / f"{i}.json"))["transforms"]
        tb = json.load(open(b / "transforms" / f"{i}.json"))["transforms"]
        assert len(ta) == len(tb)
        assert relerr(ta[0]["matrix"], tb[0]["matrix"]) < tol
        for x, y in zip(ta[1:], tb[1:]):
            assert x["dimensions"] == y["dimensions"] and relerr(x["coeffs"], y["coeffs"]) < 10 * tol
    ba, bb = json.load(open(a / "bbox.json")), json.load(open(b / "bbox.json"))
    assert ba["halfPairs"] == bb["halfPairs"] and abs(ba["inliers"] - bb["inliers"]) <= 2
    assert relerr(ba["bbox"], bb["bbox"]) < tol
    ha = list(csv.reader(open(a / "histograms.csv"))); hb = list(csv.reader(open(b / "histograms.csv")))
    assert ha[0] == hb[0] and len(ha) == len(hb)


def test_cli_sharded_over_three_contexts_on_one_gpu(tmp_path, small_pairs):
    """bin/frog -ngl 3: the C++ multi-GPU host (one thread per rank, image shards, the collectives of
    include/frog_comm.h) with its ranks sharing this box's one GPU and host-staged collectives.  Everything but the
    transport is what -ng 3 runs: same shards, same split phases, same files -- which must agree with the one-context run
    (sums over ranks associate differently: 1e-6, not bits)."""
    one, three = tmp_path / "one", tmp_path / "three"
    for d in (one, three):
        d.mkdir()
        small_pairs.write(d / "pairs.bin")
    _run_frog(one)
    out = _run_frog(three, "-ngl", "3")
    assert "Images sharded over 3 contexts" in out
    _compare_runs(one, three, small_pairs.n_images)
    assert os.path.exists(three / "errorMaps" / "5.nii.gz")


def test_cli_sharded_eight_ways_on_one_gpu(tmp_path):
    """BASELINE.json configs[3] is configs[2] sharded 8 ways: the 8-rank control flow (ragged shards of 2-3 images,
    eight contexts, every collective of include/frog_comm.h) on a group that fits the test, ranks sharing the one GPU."""
    pairs = Pairs.synthetic(20, 2000, 700, seed=21)
    one, eight = tmp_path / "one", tmp_path / "eight"
    for d in (one, eight):
        d.mkdir()
        pairs.write(d / "pairs.bin")
    _run_frog(one)
    out = _run_frog(eight, "-ngl", "8")
    assert "Images sharded over 8 contexts" in out
    _compare_runs(one, eight, pairs.n_images)


def test_cli_sharded_over_two_gpus_rccl(tmp_path, small_pairs):
    """bin/frog -ng 2: the same over RCCL.  Needs two devices; the round-end driver's GPU box has one."""
    if _abi.hip_lib().frog_device_count() < 2:
        pytest.skip("needs >= 2 HIP devices (RCCL refuses two ranks on one device); -ngl covers the control flow")
    one, two = tmp_path / "one", tmp_path / "two"
    for d in (one, two):
        d.mkdir()
        small_pairs.write(d / "pairs.bin")
    _run_frog(one)
    _run_frog(two, "-ng", "2")
    _compare_runs(one, two, small_pairs.n_images)


def test_cli_sharded_host_over_a_one_device_rccl_communicator(tmp_path, small_pairs):
    """The C++ multi-GPU host (runSharded) with ONE rank on a real RCCL communicator (ncclCommInitAll of one device):
    every collective of include/frog_comm.h is issued on the context's stream exactly as with N ranks -- what a box with
    a single GPU can execute of `bin/frog -ng N`.  Same files as the plain single-context run."""
    one, sharded = tmp_path / "one", tmp_path / "sharded"
    for d in (one, sharded):
        d.mkdir()
        small_pairs.write(d / "pairs.bin")
    _run_frog(one)
    # FROG_COMM_EXERCISE_SINGLE_RANK: the one-rank communicator does not return early -- ncclAllGather in place on the slab,
    # ncclAllReduce on the proposal sums (+ the energy sums behind them) and on the mixture table really run, on one rank
    out = _run_frog(sharded, "-ng", "1", env=dict(os.environ, FROG_SHARDED_ALWAYS="1", FROG_COMM_EXERCISE_SINGLE_RANK="1"))
    assert "Images sharded over 1 GPUs" in out
    _compare_runs(one, sharded, small_pairs.n_images)
    # the flow of rounds 2-4 (three collectives, in-place gather of FROG_BUF_XYZ2 through the slab) over the same communicator
    three = tmp_path / "three"
    three.mkdir()
    small_pairs.write(three / "pairs.bin")
    _run_frog(three, "-ng", "1", env=dict(os.environ, FROG_SHARDED_ALWAYS="1", FROG_COMM_EXERCISE_SINGLE_RANK="1", FROG_THREE_COLLECTIVES="1"))
    _compare_runs(one, three, small_pairs.n_images)


def test_native_communicator_single_rank_over_rccl():
    """libfrog_comm.so's one-process-per-GPU form (frog_comm_unique_id / frog_comm_create_rank: ncclGetUniqueId,
    ncclCommInitRank) with ONE rank -- all the box allows -- through a real RCCL: the id, the communicator, bind, rows, a
    first awaited all-reduce (frog_comm_barrier), and the collectives' single-rank paths.  In a child process, as a host
    would load it."""
    import sys
    code = r"""
import ctypes as C, os, sys
sys.path.insert(0, sys.argv[1])
from frog_amd import _abi
from frog_amd.pairs import Pairs
from frog_amd.distributed import HipEngine
pairs = Pairs.synthetic(4, 600, 300, seed=3)
eng = HipEngine(pairs, _abi.FrogOptions.default(), 0, (0, 4))
lib = C.CDLL(os.path.join(_abi.LIB_DIR, "libfrog_comm.so"))
buf = (C.c_ubyte * 128)()
assert lib.frog_comm_unique_id(buf) == 0
h = C.c_void_p()
assert lib.frog_comm_create_rank(1, 0, buf, 0, C.byref(h)) == 0, _abi.hip_lib().frog_last_error()
assert lib.frog_comm_bind(h, eng._ctx, (C.c_uint32 * 2)(0, 4)) == 0
assert lib.frog_comm_set_rows(h, (C.c_uint64 * 2)(0, 2400)) == 0
assert lib.frog_comm_barrier(h) == 0, _abi.hip_lib().frog_last_error()
assert lib.frog_comm_all_gather_xyz2(h) == 0 and lib.frog_comm_all_reduce(h, _abi.FROG_BUF_ENERGY) == 0
# the padded gather with a step's scalars in its trailer, for real on the one rank (FROG_COMM_EXERCISE_SINGLE_RANK)
import numpy as np
eng.linear_init((0.5, 0.5, 0.5))
assert lib.frog_comm_gather_points(h, 0, 0, 0) == 0, _abi.hip_lib().frog_last_error()
before = eng.points()[1].copy() if hasattr(eng, "points") else None
eng.update_stats_local(); eng.stats_publish()
assert _abi.hip_lib().frog_comm_mode(eng._ctx, 1) == 0
assert _abi.hip_lib().frog_linear_step_local(eng._ctx) == 0
assert lib.frog_comm_gather_points(h, 0, 0, 0xB) == 0, _abi.hip_lib().frog_last_error()
E = C.c_double()
assert _abi.hip_lib().frog_step_finish(eng._ctx, C.byref(E)) == 0 and E.value > 0, _abi.hip_lib().frog_last_error()
arr = (C.c_void_p * 1)(h); lib.frog_comm_destroy_all.restype = None; lib.frog_comm_destroy_all(1, arr)
print("native comm ok")
"""
    r = subprocess.run([sys.executable, "-c", code, ROOT], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, FROG_COMM_EXERCISE_SINGLE_RANK="1"))
    assert r.returncode == 0 and "native comm ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_native_host_three_ranks_over_shared_memory_matches_one_rank(tmp_path):
    """bench.py's launcher and the C loop (frog_run_schedule) with three ranks on the one GPU, collectives through the
    cross-process shared-memory communicator of libfrog_comm (what a box with one GPU can rehearse of `--gpus N`): the
    preflight's known answers hold, the ranks' replicas are bit-identical and the schedule ends on the energy of the
    one-rank run."""
    import sys
    args = ["--images", "14", "--points", "4000", "--pairs-per-block", "1500", "--steps", "26", "--warmup", "3", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads(r1.stdout.strip().splitlines()[-1])
    env.update(FROG_BENCH_BACKEND="gloo", FROG_BENCH_HOSTS="preflight,native")
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", *args], capture_output=True, text=True, timeout=900, env=env)
    assert r3.returncode == 0, r3.stderr[-3000:]
    three = json.loads(r3.stdout.strip().splitlines()[-1])
    assert [(t["host"], t["transport"], t["ok"]) for t in three["hosts_tried"]] == [("preflight", "shm", True), ("native", "shm", True)]
    assert three["hosts_tried"][0]["known_answers"] is True
    assert three["n_gpus"] == 3 and three["replicas_identical"] is True
    assert three["config"]["grids_per_level"] == one["config"]["grids_per_level"]
    assert abs(three["config"]["final_E"] - one["config"]["final_E"]) <= 1e-6 * one["config"]["final_E"]
    # two collectives per deformable iteration, one per linear one (round 5): the all-reduce of the energy sums and the oversize
    # count is gone from the loop -- they ride on the proposal sums' all-reduce and on the coordinate gather
    assert set(three["comm_ms"]) == {"all_gather_xyz2", "all_reduce_em", "all_reduce_gridsum"}
    assert three["collectives_per_iteration"] < 2.6        # 26 steps: 2 + 3 x 8, three refreshes, four set-ups
    assert len(three["ranks"]["elapsed_s"]) == 3


def test_bench_as_ranks_of_torch_distributed_run(tmp_path):
    """The way the round-end driver starts N > 1: `python -m torch.distributed.run ... bench.py --gpus N`.  Every rank process
    stays off the GPU and runs its rank of each attempt as a fresh child; the ranks' parents agree through the run directory.
    Two ranks on the one GPU over the shared-memory communicator: one line, from rank 0, the one-rank run's energy."""
    import sys
    args = ["--images", "12", "--points", "3000", "--pairs-per-block", "1200", "--steps", "26", "--warmup", "3", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads(r1.stdout.strip().splitlines()[-1])
    env.update(FROG_BENCH_BACKEND="gloo", FROG_BENCH_HOSTS="native")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *args],
                        capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    lines = [l for l in r2.stdout.splitlines() if l.lstrip().startswith('{"metric"')]
    assert len(lines) == 1, r2.stdout[-2000:]
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["replicas_identical"] is True and two["hosts_tried"][0]["ok"] is True
    assert abs(two["config"]["final_E"] - one["config"]["final_E"]) <= 1e-6 * one["config"]["final_E"]

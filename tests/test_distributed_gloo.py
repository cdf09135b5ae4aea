"""N>1 path on CPU: shard planning and the collective wiring of
frog_amd.distributed.ShardedImageGroup, world_size 2 over gloo.

The engine here is a test double (numpy/torch CPU tensors, toy arithmetic with the
same data flow as libfrog_hip's split phases): it checks that every rank ends up
with complete replicas, that partial sums are combined exactly once, and that the
regrid state machine stays in lockstep across ranks.  Numerics of the real engine
are covered by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from frog_amd.distributed import ShardedImageGroup, plan_shards
from frog_amd.pairs import Pairs


def test_plan_shards_balances_half_links():
    p = Pairs.synthetic(12, 200, 60, seed=4)
    for w in (1, 2, 3, 5, 12):
        shards = plan_shards(p.row_ptr, p.point_offset, w)
        assert len(shards) == w and shards[0][0] == 0 and shards[-1][1] == 12
        assert all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
        assert all(e > b for b, e in shards)
        po, rp = p.point_offset, p.row_ptr
        loads = [int(rp[po[e]]) - int(rp[po[b]]) for b, e in shards]
        assert sum(loads) == p.n_half_links
        if w <= 3:
            assert max(loads) < 1.6 * p.n_half_links / w
    with pytest.raises(ValueError):
        plan_shards(p.row_ptr, p.point_offset, 13)


def test_plan_shards_skewed_images():
    # one heavy image first: every rank still owns at least one image
    xyz = np.zeros((40, 3), np.float32)
    blocks = [(0, j, np.arange(10), np.zeros(10, int)) for j in range(1, 4)]
    p = Pairs.from_arrays([0, 10, 20, 30, 40], xyz, blocks)
    assert plan_shards(p.row_ptr, p.point_offset, 4) == [(0, 1), (1, 2), (2, 3), (3, 4)]
    assert plan_shards(p.row_ptr, p.point_offset, 2)[0] == (0, 1)


class ToyEngine:
    """Same buffers and phase boundaries as HipEngine, trivial arithmetic."""

    def __init__(self, point_offset, shard, n_images, reject_first=True):
        self.po = point_offset
        self.ib, self.ie = shard
        self.n_images = n_images
        P = int(point_offset[-1])
        self.xyz = torch.arange(P * 4, dtype=torch.float32).reshape(P, 4) % 97
        self.xyz2 = torch.full((P, 4), -1.0)
        self.em = torch.zeros(n_images, 4)
        self.energy = torch.zeros(4, dtype=torch.float64)
        self.gridsum = None
        self.G = 0
        self.scale = 1.0
        self.coeff = None
        self.reject_first = reject_first
        self.rejected = 0
        self.log = []

    def rows(self):
        return slice(int(self.po[self.ib]), int(self.po[self.ie]))

    def linear_init(self, anchor):
        self.log.append("init")

    def transform_points_local(self, apply):
        r = self.rows()
        self.xyz2[r] = self.xyz[r] * self.scale
        if apply:
            self.xyz[r] = self.xyz2[r]

    def update_stats_local(self):
        self.em.zero_()
        for i in range(self.ib, self.ie):
            self.em[i, 0] = float(self.xyz2[int(self.po[i]):int(self.po[i + 1]), 0].sum()) + 1 + i

    def stats_publish(self):
        self.log.append(("em", self.em[:, 0].clone()))

    def linear_step_local(self):
        # needs the complete replica: uses every row of xyz2 and every em row
        assert (self.xyz2[:, 0] >= 0).all() and (self.em[:, 0] > 0).all()
        r = self.rows()
        self.energy[:] = 0
        self.energy[0] = float(self.xyz2[r, 1].double().sum())
        self.energy[1] = float(r.stop - r.start)
        self.scale *= 0.5

    def energy_read(self):
        return float(torch.sqrt(self.energy[0] / self.energy[1])), float(self.energy[2])

    def bounds_local(self):
        r = self.rows()
        x = self.xyz[r, :3].double()
        return x.min(dim=0).values.tolist(), x.max(dim=0).values.tolist()

    def deformable_setup_bounds(self, level, mins, maxs):
        self.bounds = (tuple(mins), tuple(maxs))
        self.G = 5 + level
        self.gridsum = torch.zeros(3 * self.G, dtype=torch.float64)
        self.coeff = torch.zeros(self.ie - self.ib, 3 * self.G, dtype=torch.float64)
        return (level, self.bounds)

    def phase_a(self, alpha):
        self.prop = self.coeff + alpha * (1 + torch.arange(self.ib, self.ie, dtype=torch.float64))[:, None]
        self.gridsum[:] = self.prop.sum(dim=0)
        self.energy[:] = 0
        self.energy[0] = float(self.ie - self.ib)
        self.energy[1] = float(self.ie - self.ib)

    def phase_b(self):
        mean = self.gridsum / self.n_images
        self.prop = self.prop - mean
        big = 1.0 if (self.reject_first and self.rejected == 0 and self.ib == 0) else 0.0   # only rank 0 sees it
        self.energy[2] = big

    def phase_c(self):
        e, nbig = self.energy_read()
        if nbig > 0:
            self.rejected += 1
            return -1.0
        self.coeff = self.prop
        return e

    def make_tensor(self, values, dtype):
        return torch.tensor(values, dtype=dtype)

    def count_inliers(self):
        return None


class OracleEngine:
    """Engine double backed by the CPU oracle's split phases: real arithmetic, CPU tensors.
    Lives in tests/ only -- the product has no CPU engine."""

    def __init__(self, pairs, shard, guarantee=True):
        from frog_amd import _abi
        from oracle.oracle_api import OracleGroup
        self._abi = _abi
        self.opt = _abi.FrogOptions.default()
        self.g = OracleGroup(pairs.model, self.opt)
        self.g.setup_stats()
        self.ib, self.ie = shard
        self.g.set_range(self.ib, self.ie)
        self.n_images = pairs.n_images
        self.po = np.asarray(pairs.point_offset).astype(np.int64)
        P = int(self.po[-1])
        self.xyz2 = torch.zeros(P, 3)
        self.em = torch.zeros(self.n_images, 4)
        for i in range(self.n_images):
            self.em[i, :3] = torch.tensor([10.0, 300.0, 0.5])
        self.energy = torch.zeros(4, dtype=torch.float64)
        self.gridsum = None

    def rows(self):
        return slice(int(self.po[self.ib]), int(self.po[self.ie]))

    def _push(self):
        self.g.set_xyz2(self.xyz2.numpy())
        for i in range(self.n_images):
            self.g.set_em(i, self.em[i, :3].numpy())

    def linear_init(self, anchor):
        self.g.linear_init(anchor)

    def transform_points_local(self, apply):
        self.g.transform_points(apply)
        r = self.rows()
        self.xyz2[r] = torch.from_numpy(self.g.xyz2()[r])

    def update_stats_local(self):
        self._push()
        self.g.update_stats()
        self.em.zero_()
        for i in range(self.ib, self.ie):
            self.em[i, :3] = torch.from_numpy(self.g.em(i))

    def stats_publish(self):
        pass

    def linear_step_local(self):
        self._push()
        self.energy.zero_()
        self.energy[:2] = torch.from_numpy(self.g.linear_step_local())

    def energy_read(self):
        return float(np.sqrt(self.energy[0] / self.energy[1])), float(self.energy[2])

    def bounds_local(self):
        mn, mx = self.g.bounds_local()
        return mn.tolist(), mx.tolist()

    def deformable_setup_bounds(self, level, mins, maxs):
        info = self.g.deformable_setup_bounds(level, mins, maxs, self._abi.FrogGridInfo())
        self.gridsum = torch.zeros(3 * info.dims[0] * info.dims[1] * info.dims[2], dtype=torch.float64)
        return info

    def phase_a(self, alpha):
        self._push()
        self.energy.zero_()
        self.energy[:2] = torch.from_numpy(self.g.phase_a(alpha, self.gridsum.numpy()))

    def phase_b(self):
        self.energy[2] = float(self.g.phase_b(self.gridsum.numpy()))

    def phase_c(self):
        e, nbig = self.energy_read()
        if self.opt.guarantee_diffeomorphism and nbig > 0:
            return -1.0
        self.g.phase_c()
        return e

    def make_tensor(self, values, dtype):
        return torch.tensor(values, dtype=dtype)

    def count_inliers(self):
        return None


def _oracle_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from frog_amd import _abi
        p = Pairs.synthetic(5, 500, 260, seed=12)
        shards = plan_shards(p.row_ptr, p.point_offset, world)
        eng = OracleEngine(p, shards[rank])
        g = ShardedImageGroup(eng, shards, p.point_offset, rank, world)
        g.linearIterations, g.deformableLevels, g.deformableIterations = 14, 2, 12
        E = g.run()
        b, e = shards[rank]
        out[rank] = {"E": E, "grids": g.gridsPerLevel, "range": (b, e),
                     "matrix": {i: eng.g.matrix(i) for i in range(b, e)},
                     "coeff": {i: [eng.g.grid(i, k, _abi.FrogGridInfo())[1] for k in range(eng.g.num_grids())]
                               for i in range(b, e)},
                     "xyz2": eng.xyz2.numpy().copy()}
    finally:
        dist.destroy_process_group()


def test_sharded_oracle_matches_unsharded_oracle_gloo():
    """Numerics of the decomposition: two gloo ranks, each an oracle restricted to its image
    range, driven by the PRODUCT's ShardedImageGroup, against the plain oracle run()."""
    from frog_amd import _abi
    from oracle.oracle_api import OracleGroup
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_oracle_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    p = Pairs.synthetic(5, 500, 260, seed=12)
    ref = OracleGroup(p.model, _abi.FrogOptions.default())
    E, grids = ref.run(li=14, dl=2, di=12)
    a, b = out[0], out[1]
    assert a["E"] == b["E"] and a["grids"] == b["grids"] == grids
    assert np.allclose(a["E"], E, rtol=1e-6)          # E is carried as f32; f64 partial sums differ in order only
    assert np.array_equal(a["xyz2"], b["xyz2"])
    assert np.max(np.abs(a["xyz2"] - ref.xyz2())) <= 1e-5 * np.max(np.abs(ref.xyz2()))
    for res in (a, b):
        for i, m in res["matrix"].items():
            assert np.allclose(m, ref.matrix(i), rtol=1e-9, atol=1e-12)
        for i, grids_i in res["coeff"].items():
            for k, c in enumerate(grids_i):
                rc = ref.grid(i, k, _abi.FrogGridInfo())[1]
                assert np.max(np.abs(c - rc)) <= 1e-5 * max(np.max(np.abs(rc)), 1e-6)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out, n_images=5):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = Pairs.synthetic(n_images, 60, 30, seed=8)
        shards = plan_shards(p.row_ptr, p.point_offset, world)
        eng = ToyEngine(np.asarray(p.point_offset).copy(), shards[rank], p.n_images)
        g = ShardedImageGroup(eng, shards, p.point_offset, rank, world)
        g.linearIterations, g.deformableLevels, g.deformableIterations, g.statIntervalUpdate = 3, 2, 4, 2
        measures = g.run()
        out[rank] = {"measures": measures, "grids": g.gridsPerLevel, "xyz2": eng.xyz2.clone(),
                     "em": [e[1] for e in eng.log if isinstance(e, tuple)], "bounds": eng.bounds,
                     "coeff_sum": eng.coeff.sum(dim=0), "rejected": eng.rejected}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [5, 4])      # 5: ragged shards (broadcasts); 4: equal shards (one all-gather)
def test_sharded_run_world_size_2_gloo(n_images):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, n_images), nprocs=world, join=True)
    a, b = out[0], out[1]
    # single-process run of the same toy problem = ground truth
    p = Pairs.synthetic(n_images, 60, 30, seed=8)
    eng = ToyEngine(np.asarray(p.point_offset).copy(), (0, p.n_images), p.n_images)
    g = ShardedImageGroup(eng, [(0, p.n_images)], p.point_offset, 0, 1)
    g.linearIterations, g.deformableLevels, g.deformableIterations, g.statIntervalUpdate = 3, 2, 4, 2
    ref = g.run()
    # identical control flow and energies on both ranks, equal to the unsharded run
    assert a["measures"] == b["measures"] and a["grids"] == b["grids"] == g.gridsPerLevel
    assert np.allclose(a["measures"], ref, rtol=1e-12)
    # the rejection raised on rank 0 only reached rank 1 through the all-reduce
    assert a["rejected"] == b["rejected"] == 1 and a["grids"][0] == 2
    # replicas are complete and identical
    assert torch.equal(a["xyz2"], b["xyz2"]) and torch.equal(a["xyz2"], eng.xyz2)
    assert len(a["em"]) == len(b["em"]) > 0
    for x, y in zip(a["em"], b["em"]):
        assert torch.equal(x, y) and (x > 0).all()
    assert a["bounds"] == b["bounds"] == eng.bounds
    # cross-image mean removed using ALL images: coefficients sum to zero over the group
    assert torch.allclose(a["coeff_sum"] + b["coeff_sum"], torch.zeros_like(a["coeff_sum"]), atol=1e-12)

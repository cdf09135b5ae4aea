"""Round-4 regression tests of the C ABI's state handling (all on the GPU, through the C ABI)."""
import numpy as np
import pytest

from frog_amd.image_group import ImageGroup
from frog_amd.pairs import Pairs

pytestmark = pytest.mark.gpu


def _sequence(pairs, **opt):
    """A linear stage, a deformable level, then a SECOND linear stage on the same context (the C ABI allows it):
    frog_linear_init must leave nothing of the first stages behind -- neither a list built for another criterion nor a
    speculative transform of the old matrices."""
    g = ImageGroup(pairs, **opt)
    g.setupLinearTransforms(); g.transformPoints()
    x_init = g.points()[1].copy()
    for it in range(12):
        if it % 10 == 0:
            g.updateStats()
        g.updateLinearTransforms(); g.transformPoints()
    g.transformPoints(True)
    g.setupDeformableTransforms(0); g.transformPoints()
    g.updateStats()
    for it in range(5):
        assert g.updateDeformableTransforms(0.02) >= 0
        g.transformPoints()
    # second linear stage, WITHOUT a statistics refresh before its first steps
    g.setupLinearTransforms(); g.transformPoints()
    x_again = g.points()[1].copy()
    es = []
    for it in range(3):
        es.append(g.updateLinearTransforms()); g.transformPoints()
    mats = np.stack([g.matrix(i) for i in range(pairs.n_images)])
    # a linear step's speculative transform must not survive a linear_init
    g.updateLinearTransforms()
    g.setupLinearTransforms(); g.transformPoints()
    x_third = g.points()[1].copy()
    return x_init, x_again, x_third, np.array(es), mats, g.points()[1].copy()


def test_a_second_linear_stage_after_a_deformable_level(monkeypatch):
    """after a deformable stage the list and the cut-offs follow the threshold criterion; a linear step that
    walked that list would leave out links with non-zero weight.  With the lists on, the second linear stage must give what
    it gives without any list (f64 re-association aside), and the coordinates after frog_linear_init + transformPoints are
    the initial transform's, not a speculative transform of older matrices."""
    pairs = Pairs.synthetic(8, 3000, 1200, seed=5)
    monkeypatch.setenv("FROG_CULL", "0")
    a = _sequence(pairs)
    monkeypatch.delenv("FROG_CULL")
    b = _sequence(pairs)
    for x_init, x_again, x_third, es, mats, x in (a, b):
        # the points were re-based in between, so the second stage's initial transform acts on other coordinates: what must
        # hold is that init + transform is reproducible, i.e. the third equals the second
        assert np.array_equal(x_again, x_third)
    assert np.max(np.abs(a[3] - b[3]) / a[3]) <= 1e-13
    assert np.max(np.abs(a[4] - b[4])) <= 1e-12 * np.max(np.abs(a[4]))
    assert np.max(np.abs(a[5].astype(np.float64) - b[5])) <= 1.2e-7 * np.max(np.abs(a[5]))


def test_the_c_loop_and_the_python_loop_end_on_the_same_bits():
    """frog_run_schedule (the C loop bench.py times) against the ImageGroup mirror driving the same C ABI from Python: the
    same schedule -- warm-up, linear iterations, three levels with their refreshes and any regrid -- ends on the same
    energy, the same lattices per level and the same coordinates, bit for bit (both are hosts of the same device path)."""
    import ctypes as C
    from frog_amd import _abi
    pairs = Pairs.synthetic(10, 3000, 1300, seed=4)
    lib, host = _abi.hip_lib(), _abi.host_lib()
    opts = _abi.FrogOptions.default(max_levels_hint=3)
    ctx = C.c_void_p()
    _abi.check(lib.frog_create(C.byref(pairs.model), C.byref(opts), 0, 0, pairs.n_images, C.byref(ctx)), "frog_create")
    plan = _abi.FrogSchedulePlan()
    plan.plan_bytes, plan.result_bytes = C.sizeof(_abi.FrogSchedulePlan), C.sizeof(_abi.FrogScheduleResult)
    plan.warmup_linear, plan.linear, plan.n_levels = 4, 11, 3
    for l, n in enumerate((14, 13, 12)):
        plan.per_level[l] = n
    plan.stat_interval, plan.deformable_alpha = 10, 0.02
    plan.anchor[0] = plan.anchor[1] = plan.anchor[2] = 0.5
    res = _abi.FrogScheduleResult()
    _abi.check(host.frog_run_schedule(ctx, None, C.byref(plan), C.byref(res)), "frog_run_schedule")
    assert res.iterations == 11 + 14 + 13 + 12 and res.n_lattices >= 3 and res.elapsed_s > 0
    n = int(pairs.point_offset[-1])
    xyz_c = np.empty((n, 3), np.float32)
    _abi.check(lib.frog_get_points(ctx, xyz_c.ctypes.data_as(_abi.c_float_p), None), "frog_get_points")
    lib.frog_destroy(ctx)

    g = ImageGroup(pairs, max_levels_hint=3)
    g.setupLinearTransforms(); g.transformPoints()
    it, e = 0, 0.0
    for _ in range(4 + 11):
        if it % 10 == 0:
            g.updateStats()
        e = g.updateLinearTransforms(); g.transformPoints()
        it += 1
    g.transformPoints(True)
    grids = []
    for level, n_it in enumerate((14, 13, 12)):
        g.setupDeformableTransforms(level); g.transformPoints()
        alpha, nd, k, ng = np.float32(0.02), 0, 0, 1
        while k < n_it:
            if k % 10 == 0:
                g.updateStats()
            ee = g.updateDeformableTransforms(float(alpha))
            if ee < 0:
                if nd == 0:
                    alpha = np.float32(alpha / np.float32(2))
                ng += 1
                g.transformPoints(True); g.setupDeformableTransforms(level); g.transformPoints()
                nd = 0
                continue
            nd += 1; g.transformPoints(); e = ee; k += 1
        grids.append(ng)
        g.transformPoints(True)
    assert list(res.grids_per_level[:3]) == grids
    assert float(np.float32(e)) == res.final_E
    assert np.array_equal(g.points()[0], xyz_c)

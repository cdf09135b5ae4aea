"""A second, independent restatement of one linear and one deformable iteration, written in
numpy straight from the formulas of SURVEY.md Appendix D (D.1 promotions, D.2 linear step, D.3
deformable step, D.4 lattice geometry), compared with the C++ oracle on a small group.  It is
not the reference either (imageGroup.cxx cannot be built here: parity stays unpinned), but two
restatements that were written separately and agree to f32 rounding bound the room for a
transcription error in the oracle the GPU path is judged against."""
import numpy as np

from frog_amd import _abi
from oracle.oracle_api import OracleGroup

f32 = np.float32


def inlier_probability(d, c1, c2, ratio):
    """stats.h:84-92 + chipdf stats.h:10-16, with the promotions of Appendix D.1 (d: f32 array)."""
    eps = f32(1e-10)

    def chipdf(x):                                  # x f32 -> f32
        c = f32(0.797884560802865)
        x2 = (x * x).astype(f32)
        return ((c * x2).astype(f32).astype(np.float64) * np.exp(-0.5 * x2.astype(np.float64))).astype(f32)
    a1 = (f32(c1) + eps).astype(f32); a2 = (f32(c2) + eps).astype(f32)
    x1 = ((f32(ratio) * chipdf((d / a1).astype(f32))).astype(f32) / a1).astype(f32)
    x2 = ((1.0 - np.float64(f32(ratio))) * chipdf((d / a2).astype(f32)).astype(np.float64) / np.float64(a2)).astype(f32)
    p = (x1 / ((x1 + x2).astype(f32) + eps).astype(f32)).astype(f32)
    return np.where(d.astype(np.float64) < 0.1, f32(1), p).astype(f32)


def links_of(pairs):
    """(image1, point1 global, image2, point2 global) of every half-link."""
    po = np.asarray(pairs.point_offset, np.int64)
    rp = np.asarray(pairs.row_ptr, np.int64)
    p1 = np.repeat(np.arange(po[-1]), np.diff(rp))
    i1 = np.searchsorted(po, p1, side="right") - 1
    i2 = np.asarray(pairs.link_image, np.int64)
    p2 = po[i2] + np.asarray(pairs.link_point, np.int64)
    return i1, p1, i2, p2


def weights(pairs, xyz2, em):
    i1, p1, i2, p2 = links_of(pairs)
    diff = (xyz2[p2] - xyz2[p1]).astype(f32)
    d2 = ((diff[:, 0] * diff[:, 0]).astype(f32) + (diff[:, 1] * diff[:, 1]).astype(f32)).astype(f32)
    d2 = (d2 + (diff[:, 2] * diff[:, 2]).astype(f32)).astype(f32)
    d = np.sqrt(d2).astype(f32)
    pa, pb = np.empty(len(d), f32), np.empty(len(d), f32)
    for img in range(pairs.n_images):               # per-image parameters: evaluate image by image
        pa[i1 == img] = inlier_probability(d[i1 == img], *em[img])
        pb[i2 == img] = inlier_probability(d[i2 == img], *em[img])
    return i1, p1, p2, diff, d2, d, np.minimum(pa, pb)


def test_one_linear_step_matches_appendix_d2(tiny_pairs):
    g = OracleGroup(tiny_pairs.model, _abi.FrogOptions.default())
    g.setup_stats(); g.linear_init(); g.transform_points(); g.update_stats()
    xyz2 = g.xyz2().astype(f32)
    em = [g.em(i) for i in range(tiny_pairs.n_images)]
    before = [g.matrix(i).copy() for i in range(tiny_pairs.n_images)]
    i1, p1, p2, diff, d2, d, w = weights(tiny_pairs, xyz2, em)
    ww = (w * w).astype(f32)
    E = np.sqrt(np.sum(((ww * d).astype(f32) * d).astype(f32).astype(np.float64)) / np.sum(ww.astype(np.float64)))
    got_E = g.linear_step()
    assert abs(got_E - E) / E < 1e-6
    la = 0.5
    for img in range(tiny_pairs.n_images):
        sel = i1 == img
        ws = w[sel].astype(np.float64)
        pa, pb, df = xyz2[p1[sel]], xyz2[p2[sel]], diff[sel]
        sW = ws.sum()
        M = before[img]
        for k in range(3):
            wk = w[sel]
            sDisp = (wk * df[:, k]).astype(f32).astype(np.float64).sum()
            sPosA = (wk * pa[:, k]).astype(f32).astype(np.float64).sum()
            sPosB = (wk * pb[:, k]).astype(f32).astype(np.float64).sum()
            sPosA2 = ((wk * pa[:, k]).astype(f32) * pa[:, k]).astype(f32).astype(np.float64).sum()
            sPosB2 = ((wk * pb[:, k]).astype(f32) * pb[:, k]).astype(f32).astype(np.float64).sum()
            new_scale = f32(((sW * sPosB2 - sPosB ** 2) / (sW * sPosA2 - sPosA ** 2)) ** (0.5 * la))
            scale = f32(M[k, k])
            want_scale = np.float64(f32(scale * new_scale))
            want_t = np.float64(f32(M[k, 3])) + la * sDisp / sW + sPosA * np.float64(f32(1) - new_scale) / sW
            assert abs(g.matrix(img)[k, k] - want_scale) < 2e-6 * abs(want_scale)
            assert abs(g.matrix(img)[k, 3] - want_t) < 1e-5 * max(1.0, abs(want_t))


def bspline_weights(f):
    F3 = f ** 3 / 6
    F0 = (f * f - f) / 2 - F3 + 1.0 / 6
    F2 = f + F0 - 2 * F3
    F1 = 1 - F0 - F2 - F3
    return np.stack([F0, F1, F2, F3], -1)


def test_one_deformable_step_matches_appendix_d3(tiny_pairs):
    n_img = tiny_pairs.n_images
    g = OracleGroup(tiny_pairs.model, _abi.FrogOptions.default())
    g.setup_stats(); g.linear_init(); g.transform_points()
    for it in range(10):
        if it % 10 == 0:
            g.update_stats()
        g.linear_step(); g.transform_points()
    g.transform_points(True)
    info = g.deformable_setup(1, _abi.FrogGridInfo())
    g.transform_points(); g.update_stats()
    xyz, xyz2 = g.xyz().astype(f32), g.xyz2().astype(f32)
    em = [g.em(i) for i in range(n_img)]
    dims, origin, spacing = list(info.dims), np.array(info.origin), np.array(info.spacing)
    # D.4: geometry from the 1.2x bounding box of the moving points
    mn, mx = xyz.min(0).astype(np.float64), xyz.max(0).astype(np.float64)
    ctr, half = (mn + mx) / 2, (mx - mn) / 2 * float(f32(1) + f32(2) * f32(0.1))     # `1 + 2 * boundingBoxMargin` with a float member: f32 arithmetic (imageGroup.h:29, .cxx:168)
    size = 100.0 / 2
    n = np.maximum(1, np.round(2 * half / size)).astype(int)
    assert dims == list(n + 3)
    assert np.allclose(spacing, 2 * half / n, rtol=1e-12) and np.allclose(origin, ctr - half - 2 * half / n, rtol=1e-9, atol=1e-9)

    alpha = f32(0.02)
    i1, p1, p2, diff, d2, d, w = weights(tiny_pairs, xyz2, em)
    inl = w >= f32(0.5)
    w2 = (w * w).astype(f32)
    E = np.sqrt(np.sum((w2 * d2).astype(f32)[inl].astype(np.float64)) / np.sum(w2[inl].astype(np.float64)))
    got_E = g.deformable_step(float(alpha))
    assert got_E > 0 and abs(got_E - E) / E < 1e-6
    # per-point f32 sums in link order (np.add.at on f32 accumulates sequentially in index order)
    P = len(xyz)
    sums = np.zeros((P, 4), f32)
    contrib = np.concatenate([(w2[:, None] * diff).astype(f32), w2[:, None]], 1)[inl]
    np.add.at(sums, p1[inl], contrib)
    assert np.max(np.abs(sums - g.point_sums())) <= 2e-5 * np.max(np.abs(sums))
    # scatter + control-point step + mean removal, per D.3
    G = dims[0] * dims[1] * dims[2]
    po = np.asarray(tiny_pairs.point_offset, np.int64)
    prop = np.zeros((n_img, G, 3))
    for img in range(n_img):
        grad = np.zeros((dims[2], dims[1], dims[0], 4))
        for p in range(po[img], po[img + 1]):
            if sums[p, 3] == 0:
                continue
            coord = ((xyz[p].astype(np.float64) - origin) / spacing).astype(f32)
            i0 = np.floor(coord).astype(int)
            F = bspline_weights((coord - i0.astype(f32)).astype(np.float64))
            wgt = F[2][:, None, None] * F[1][None, :, None] * F[0][None, None, :]
            sl = (slice(i0[2] - 1, i0[2] + 3), slice(i0[1] - 1, i0[1] + 3), slice(i0[0] - 1, i0[0] + 3))
            grad[sl] += wgt[..., None] * sums[p].astype(np.float64)
        grad = grad.reshape(G, 4)
        gw = grad[:, 3]
        step = np.where(gw[:, None] > 0, float(alpha) * grad[:, :3] / np.where(gw > 0, gw, 1)[:, None], 0.0)
        prop[img] = step                                # coefficients start at zero on a new lattice
    prop -= prop.mean(axis=0, keepdims=True)            # :400-432, no fixed image
    for img in range(n_img):
        _, c = g.grid(img, 0, _abi.FrogGridInfo())
        scale = max(np.max(np.abs(prop[img])), 1e-12)
        assert np.max(np.abs(c - prop[img])) < 2e-4 * scale, img

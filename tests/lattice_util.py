"""Comparison of B-spline lattices between the HIP path and the oracle (shared by the GPU parity tests).

`compare_lattice` applies the criterion of tests/test_gpu_round2.py::test_parity_sweep to one lattice of one image:
coefficients of well supported control points within 1e-4 of the lattice's largest coefficient, weakly supported ones
within 1e-2, and the displacement field at every point of the image within 1e-4 of its maximum."""
import numpy as np

from frog_amd import _abi

REL = 1e-4

# A control point's step is alpha * g / gw = alpha * (weighted mean of the per-point ratios sDisp_p / sWeight_p over the
# points p in its support, weights w_p * sWeight_p, w_p = product of three cubic basis values).  Two implementations whose
# point coordinates differ by k f32 ulps (6e-8 * 300 mm = 2e-5 mm, i.e. dt = k * 2e-5 / spacing = k * 2..8e-7 in lattice
# units) disagree on a TAIL weight (1 - t)^3 / 6 by 3 dt / (1 - t) relative, and the weighted mean moves by that times
# the spread of the ratios (several mm) -- against max|c| of a lattice of ~0.1 mm per step.  The less a control point is
# supported (sum of the basis weights of its image's points: `support`), the more of its weight sits in such tails and
# the less the data determine it: measured, with identical per-point sums and coordinates differing in the last bits,
# 1.4e-4 of max|c| at a support of 1e-3 and 3e-4 at 2e-2, while the displacement field the coefficients define agrees to
# 4e-6.  A max-norm over raw coefficients therefore measures the reference's own f32 conditioning on the rim of the box,
# not parity.  The comparison weights every control point by how much the image's points determine it:
#   * |c - c_ref| * min(1, support) <= 1e-4 max|c_ref|      (support >= 1: the plain 1e-4 bar; below: relaxed in proportion)
#   * control points without any support (value = -group mean): the plain 1e-4 bar;
#   * every control point, unweighted: 1e-2 of max|c_ref| -- they still have to be the same numbers;
#   * and the quantity the coefficients exist for, the displacement field at EVERY point of the image: 1e-4 of its maximum.
RIM_REL = 1e-2


def bspline_weights(f):
    """imageGroup.cxx:221-232, vectorised; f: fractions in [0, 1)."""
    f2 = f * f
    F3 = f2 * f / 6.0
    F0 = (f2 - f) * 0.5 - F3 + 1.0 / 6.0
    F2 = f + F0 - 2.0 * F3
    F1 = 1.0 - F0 - F2 - F3
    return np.stack([F0, F1, F2, F3], axis=-1)


def lattice_taps(xyz, info):
    """Control-point indices [n, 64] and basis weights [n, 64] of the points xyz [n, 3] on the lattice `info`."""
    dims = np.array(list(info.dims)); origin = np.array(list(info.origin)); spacing = np.array(list(info.spacing))
    q = (xyz.astype(np.float64) - origin) / spacing
    fl = np.floor(q)
    w = [bspline_weights(q[:, k] - fl[:, k]) for k in range(3)]
    i0 = fl.astype(np.int64) - 1
    idx = np.empty((len(xyz), 64), np.int64); wt = np.empty((len(xyz), 64))
    t = 0
    for k in range(4):
        for j in range(4):
            for i in range(4):
                idx[:, t] = (i0[:, 0] + i) + dims[0] * ((i0[:, 1] + j) + dims[1] * (i0[:, 2] + k))
                wt[:, t] = w[0][:, i] * w[1][:, j] * w[2][:, k]
                t += 1
    return idx, wt


def compare_lattice(g, ref, k, i, pts):
    """Lattice k of image i on both sides; pts = the reference's re-based coordinates of the image's points the
    lattice acts on.  Returns (largest support-weighted deviation of a coefficient / max|c_ref|, largest deviation of the
    displacement field over all points / max displacement, control points with a support below 1, control points)."""
    info, c = g.grid(i, k)
    rinfo, rc = ref.grid(i, k, _abi.FrogGridInfo())
    assert list(info.dims) == list(rinfo.dims)
    idx, wt = lattice_taps(pts, rinfo)
    support = np.zeros(len(rc)); np.add.at(support, idx.ravel(), wt.ravel())
    scale = max(float(np.max(np.abs(rc))), 1e-30)
    err = np.max(np.abs(c.astype(np.float64) - rc), axis=1) / scale
    weight = np.where(support == 0.0, 1.0, np.minimum(1.0, support))
    dev_c = float(np.max(err * weight))
    assert float(np.max(err)) <= RIM_REL, f"lattice {k} image {i}: coefficients off by {float(np.max(err)):.2e}"
    ok = support >= 1.0
    disp = np.einsum("nt,ntk->nk", wt, c.astype(np.float64)[idx])
    rdisp = np.einsum("nt,ntk->nk", wt, rc.astype(np.float64)[idx])
    dev_d = float(np.max(np.abs(disp - rdisp))) / max(float(np.max(np.abs(rdisp))), 1e-30)
    return dev_c, dev_d, int(np.count_nonzero(~ok)), len(rc)



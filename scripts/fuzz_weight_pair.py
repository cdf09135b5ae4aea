"""fuzz_weight_pair.py -- on the GPU box: the deformable sweeps' weight (k_links.hip.h inlier_weight_pair, through frog_test_inlier_weight_pair)
against the reference build of stats.cxx for 400 random pairs of mixtures (c1 1e-2..1e2, c2/c1 0.5..1e3, ratios incl. 1e-6 and 1 - 1e-6) and
thresholds 0.01..0.99: the largest deviation of a VALUE (bound 2^-16) and whether any dropped link is an inlier in the reference build.
Test infrastructure (uses oracle/_ref)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from frog_amd.image_group import device_inlier_weight_pair
from oracle.oracle_api import Stats, ref_lib
assert ref_lib() is not None
rng = np.random.default_rng(123)
bound = 2.0 ** -16
worst = 0.0; worst_at = None; nforms = np.zeros(3, np.int64); bad_drop = 0
for it in range(400):
    def mix():
        c1 = float(np.float32(10.0 ** rng.uniform(-2, 2)))
        c2 = float(np.float32(c1 * 10.0 ** rng.uniform(-0.3, 3)))
        r = float(np.float32(rng.choice([rng.uniform(0.01, 0.99), 10.0 ** rng.uniform(-6, -2), 1 - 10.0 ** rng.uniform(-6, -2)])))
        return (c1, c2, r)
    ma, mb = mix(), mix()
    thr = float(rng.choice([0.5, 0.5, 0.1, 0.9, 0.01, 0.99]))
    ra, rb = Stats("ref"), Stats("ref"); ra.set_params(list(ma)); rb.set_params(list(mb))
    cs = min(ma[0], mb[0])
    d = np.concatenate([cs * np.geomspace(0.01, 80.0, 20000), rng.uniform(0, 5 * max(ma[1], mb[1]), 5000), np.linspace(0, 0.3, 301)]).astype(np.float32)
    d2 = (d * d).astype(np.float32)
    root = np.sqrt(d2)
    want = np.minimum(ra.prob_n(root), rb.prob_n(root)).astype(np.float64)
    k = int(np.argmin(np.abs(want - thr)))
    dense = (d[k] + np.arange(-2000, 2000) * np.spacing(d[k])).astype(np.float32)
    d2 = np.concatenate([d2, (dense * dense).astype(np.float32)]); root = np.sqrt(d2)
    want = np.minimum(ra.prob_n(root), rb.prob_n(root)).astype(np.float64)
    w, form = device_inlier_weight_pair(ma, mb, d2, thr)
    v = form < 2
    if v.any():
        fin = np.isfinite(w[v])
        if not fin.all(): print("non-finite", ma, mb, thr)
        dev = np.abs(w[v].astype(np.float64) - want[v])
        if dev.max() > worst: worst, worst_at = float(dev.max()), (ma, mb, thr, float(d2[v][int(np.argmax(dev))]), int(form[v][int(np.argmax(dev))]))
    nb = int(np.count_nonzero(want[form == 2] >= thr))
    if nb: bad_drop += nb; print("BAD DROP", ma, mb, thr, nb, want[form == 2].max())
    nforms += np.bincount(form, minlength=3)[:3]
print("worst", worst, worst_at, "bound", bound, "forms", nforms.tolist(), "bad drops", bad_drop)

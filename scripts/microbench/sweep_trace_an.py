"""Per-block phase timeline of the fused deformable sweep (builds with -DFROG_SWEEP_TRACE, FROG_SWEEP_TRACE_FILE=path):
usage: sweep_trace_an.py trace.bin"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8, 8).astype(np.int64)      # [block][wave][slot]
live = (a[:, 0, 6] > 0) & (a[:, :, 3] > 0).all(axis=1)        # blocks whose eight wavefronts all stamped their first step
a = a[live]
t0 = a[:, :, 0].min()
us = lambda x: x * 10e-3                                                              # 100 MHz wall clock
ph = ["entry->tile", "tile->barrier", "barrier->first step", "first step->walk end", "walk end->barrier", "barrier->end"]
d = np.diff(a[:, :, :7], axis=2)
print("blocks", len(a), "span us", us(a[:, :, 6].max() - t0))
for k, name in enumerate(ph):
    print(f"  {name:24s} mean {us(d[:, :, k].mean()):7.2f} us   p90 {us(np.percentile(d[:, :, k], 90)):7.2f}   max-over-waves mean {us(d[:, :, k].max(axis=1).mean()):7.2f}")
life = us(a[:, :, 6].max(axis=1) - a[:, :, 0].min(axis=1))
walk = us(a[:, :, 4] - a[:, :, 3])
print("block life us: mean %.2f p50 %.2f p90 %.2f" % (life.mean(), np.median(life), np.percentile(life, 90)))
print("walk per wave us: mean %.2f; slowest wave of a block mean %.2f; records per range mean %.0f" % (walk.mean(), walk.max(axis=1).mean(), a[:, :, 7].mean()))
steps = np.maximum(1, (a[:, :, 7] + 63) // 64)
print("walk us per step: %.3f" % (walk.sum() / steps.sum()))
start = us(a[:, :, 0].min(axis=1) - t0); end = us(a[:, :, 6].max(axis=1) - t0)
T = int(end.max() * 10) + 1
occ = np.zeros(T + 1); np.add.at(occ, (start * 10).astype(int), 1); np.add.at(occ, (end * 10).astype(int), -1); occ = np.cumsum(occ)
print("concurrent blocks: mean %.0f max %d; deciles" % (occ[:T].mean(), occ.max()), [int(b.mean()) for b in np.array_split(occ[:T], 10)])
# where the spread inside a block comes from: records per range against walk time, wave by wave
rec = a[:, :, 7].astype(float)
print("records per range: mean %.0f  std over the 8 waves of a block (mean) %.0f  max/mean in a block %.3f" % (rec.mean(), rec.std(axis=1).mean(), (rec.max(axis=1) / rec.mean(axis=1)).mean()))
print("walk us: std over the waves of a block (mean) %.2f   max/mean in a block %.3f" % (walk.std(axis=1).mean(), (walk.max(axis=1) / walk.mean(axis=1)).mean()))
print("corr(walk, records) over all waves %.3f; inside a block (mean of per-block corr) %.3f" % (np.corrcoef(walk.ravel(), rec.ravel())[0, 1], np.nanmean([np.corrcoef(walk[b], rec[b])[0, 1] for b in range(0, len(a), 7)])))
print("per wave index: mean records", rec.mean(axis=0).round(0), " mean walk us", walk.mean(axis=0).round(2))
print("                mean entry->first step us", us(a[:, :, 3] - a[:, :, 0]).mean(axis=0).round(2), " wait at the last barrier us", us(a[:, :, 5] - a[:, :, 4]).mean(axis=0).round(2))

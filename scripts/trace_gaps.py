"""Timeline of a rocprofv3 --kernel-trace run: per phase (between lattice set-ups), where the time of the stream goes --
kernels by name and the idle gaps between them.  usage: trace_gaps.py <dir with p_kernel_trace.csv> [first_n_rows]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1] + "/p_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("frog::", "")
# keep the timed region: from the first deformable sweep's preceding linear sweeps ... simply print everything after the last warm-up linear sweep
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    out.append(((s - t0) / 1e3, name(r), (e - s) / 1e3, gap))
    prev_end = max(prev_end or 0, e)
lim = int(sys.argv[2]) if len(sys.argv) > 2 else len(out)
for t, n, d, g in out[:lim]:
    flag = "  <== gap" if g > 20 else ""
    print(f"{t:12.1f} us  {n[:44]:44s} {d:9.1f} us   gap {g:8.1f}{flag}")
tot = collections.Counter(); cnt = collections.Counter(); gaps = 0.0
for t, n, d, g in out:
    tot[n] += d; cnt[n] += 1; gaps += g if g > 0 else 0
print("---- totals (us)")
for n, v in tot.most_common(): print(f"{n[:50]:50s} {v:10.1f}  x{cnt[n]}")
print("gaps total", round(gaps, 1), "span", round(out[-1][0] + out[-1][2] - out[0][0], 1))

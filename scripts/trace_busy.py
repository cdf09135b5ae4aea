"""trace_busy.py <dir with p_kernel_trace.csv> -- of a rocprofv3 --kernel-trace run: the span of the kernels, the time at least one
kernel was running (union of the intervals), and the mean number of kernels running -- is the device waiting for the host?"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/p_kernel_trace.csv")))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "match_" in r["Kernel_Name"] or len(sys.argv) > 2)
span = iv[-1][1] - iv[0][0]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
total = sum(e - s for s, e in iv)
print(f"kernels {len(iv)}  span {span/1e3:.1f} us  some kernel running {busy/1e3:.1f} us = {busy/span:.3f}  sum of durations {total/1e3:.1f} us = {total/span:.2f} kernels at a time on average")

#!/usr/bin/env python3
"""Per-kernel HBM traffic of the finest level from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, CSV output):
for every kernel the launches with the LARGEST grid (the finest lattice) -- average bytes per launch, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950 -- beside the average duration of the same launches from the kernel trace.
usage: level_traffic.py TRACE_DIR FETCH_DIR WRITE_DIR [kernel-name substrings ...]"""
import collections
import csv
import glob
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("frog::", "")


def rows(d, suffix):
    for f in glob.glob(f"{d}/**/*_{suffix}.csv", recursive=True):
        yield from csv.DictReader(open(f))


trace, fetch, write = sys.argv[1:4]
want = sys.argv[4:] or ["lattice_step_kernel", "scatter_kernel", "transform_bspline", "sweep_kernel<1"]
dur = collections.defaultdict(list)
for r in rows(trace, "kernel_trace"):
    dur[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
val = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}
for d in (fetch, write):
    for r in rows(d, "counter_collection"):
        if r["Counter_Name"] in val:
            val[r["Counter_Name"]][(short(r["Kernel_Name"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
print(f"{'kernel':44s} {'grid':>10s} {'launches':>8s} {'avg ms':>9s} {'fetch GB':>9s} {'write GB':>9s} {'GB total':>9s} {'TB/s':>6s}")
for name in sorted({k[0] for k in dur}):
    if not any(w in name for w in want):
        continue
    grid = max(g for (n, g) in dur if n == name)
    ms = sum(dur[(name, grid)]) / len(dur[(name, grid)]) / 1e6
    f = val["FETCH_SIZE"].get((name, grid)); w = val["WRITE_SIZE"].get((name, grid))
    fg = 2.0 * sum(f) / len(f) * 1024 / 1e9 if f else float("nan")
    wg = sum(w) / len(w) * 1024 / 1e9 if w else float("nan")
    print(f"{name[:44]:44s} {grid:10d} {len(dur[(name, grid)]):8d} {ms:9.3f} {fg:9.2f} {wg:9.2f} {fg + wg:9.2f} {(fg + wg) / ms:6.2f}")

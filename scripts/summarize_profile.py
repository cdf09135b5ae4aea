#!/usr/bin/env python3
"""Condense rocprofv3 output directories (kernel stats + PMC passes) into one text summary.

usage: summarize_profile.py OUT.txt TRACE_DIR [PMC_DIR ...]
"""
import collections
import csv
import glob
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("frog::", "")


def main():
    out, trace, pmcs = sys.argv[1], sys.argv[2], [a for a in sys.argv[3:] if not a.isdigit()]
    lines = []
    for f in glob.glob(f"{trace}/**/*_kernel_stats.csv", recursive=True):
        lines.append(f"# rocprofv3 --kernel-trace --stats  ({f})")
        lines.append(f"{'kernel':38s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
        for r in csv.DictReader(open(f)):
            lines.append(f"{short(r['Name'])[:38]:38s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} "
                         f"{float(r['AverageNs'])/1e3:10.1f} {float(r['MinNs'])/1e3:9.1f} {float(r['MaxNs'])/1e3:9.1f} "
                         f"{float(r['Percentage']):6.2f}")
    for d in pmcs:
        for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
            acc = collections.defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                k = (short(r["Kernel_Name"]), r["Counter_Name"])
                acc[k][0] += float(r["Counter_Value"])
                acc[k][1] += 1
            lines.append("")
            lines.append(f"# rocprofv3 --pmc  ({f}) -- average counter value per launch")
            for (k, c), (v, n) in sorted(acc.items()):
                lines.append(f"{k[:38]:38s} {c:14s} {v / n:16.1f}  launches {n}")
    # HBM traffic of the dominant kernel (deformable sweep = sweep_kernel<1, ...>), per launch, for bench.py's
    # roofline.traffic: FETCH_SIZE and WRITE_SIZE are in KB; FETCH_SIZE counts 64 B per 128-B request on gfx950
    # (MI355X_MICROARCH.md, HBM section) and is doubled
    fetch = write = None
    for d in pmcs:
        for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
            tot = collections.defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                # the launch that also writes the culling list (last template argument true: once per list) is not the steady state
                # (template arguments: MODE, EMD_LDS, WIDE, BUILD, FUSED)
                nm = short(r["Kernel_Name"]).strip()
                targs = [a.strip() for a in nm[nm.find("<") + 1:nm.rfind(">")].split(",")] if "<" in nm else []
                if (nm.startswith("sweep_kernel<1") and not (len(targs) > 3 and targs[3] == "true")
                        and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE")):
                    tot[r["Counter_Name"]][0] += float(r["Counter_Value"]); tot[r["Counter_Name"]][1] += 1
            if "FETCH_SIZE" in tot: fetch = tot["FETCH_SIZE"][0] / tot["FETCH_SIZE"][1]
            if "WRITE_SIZE" in tot: write = tot["WRITE_SIZE"][0] / tot["WRITE_SIZE"][1]
    if fetch is not None and write is not None:
        import json
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from frog_amd._abi import device_source_hash
        json.dump({"kernel": "sweep_deformable", "measured_at": device_source_hash(), "fetch_size_kb": fetch, "write_size_kb": write,
                   "traffic_bytes_per_launch": 2.0 * fetch * 1024.0 + write * 1024.0,
                   "half_links_owned": int(sys.argv[-1]) if sys.argv[-1].isdigit() else None,      # = roofline.half_links_owned of the bench line
                   "half_links_per_launch": int(sys.argv[-1]) if sys.argv[-1].isdigit() else None,  # the key's name until round 3
                   "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950)"},
                  open(out.replace("_bench_n1.txt", "_hbm_traffic.json"), "w"), indent=1)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

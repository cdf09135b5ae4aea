#!/usr/bin/env python3
"""merge_traffic.py OUT.json IN.json [IN.json ...] -- the per-workload HBM-traffic measurements scripts/summarize_profile.py writes
(gpurun_out/TAG_hbm_traffic.json: one per profiled configuration) into the one file bench.py reads (profiles/hbm_traffic.json:
{"entries": [...]}, one entry per (kernel, half_links_owned))."""
import json
import sys

out, ins = sys.argv[1], sys.argv[2:]
entries = {}
for f in ins:
    d = json.load(open(f))
    for e in d.get("entries", [d]):
        entries[(e.get("kernel"), e.get("half_links_owned"))] = e
json.dump({"entries": list(entries.values())}, open(out, "w"), indent=1)
print(len(entries), "entries ->", out)

#!/bin/bash
# ab_blocked.sh [VAR] -- A/B of a lattice switch (default FROG_LATTICE_BLOCKED; FROG_LATTICE_SPARSE) = 0 / 1 on one box: cfg 3 (650 steps, per-kernel times) and cfg 5
set -e
cd "$(dirname "$0")/.."
VAR=${1:-FROG_LATTICE_BLOCKED}
for v in 0 1 0 1; do
  env $VAR=$v python3 bench.py --no-cpu-baseline --no-end-to-end --kernel-times > gpurun_out/abl_cfg3_$v.json 2>/dev/null
  python3 - $v <<'PY'
import json, sys
v = sys.argv[1]
a = json.load(open(f"gpurun_out/abl_cfg3_{v}.json"))
k = a["kernels_ms_by_phase"]
print("cfg3 switch=%s %7.1f it/s E %s" % (v, a["value"], a["config"]["final_E"]), {ph: {n: round(x["ms"] / x["launches"], 4) for n, x in ks.items() if n in ("scatter", "lattice", "transform")} for ph, ks in k.items() if ph != "linear"}, flush=True)
PY
done
for v in 0 1; do
  env $VAR=$v python3 bench.py --config 5 --kernel-times > gpurun_out/abl_cfg5_$v.json 2>/dev/null
  python3 - $v <<'PY'
import json, sys
v = sys.argv[1]
a = json.load(open(f"gpurun_out/abl_cfg5_{v}.json"))
k = a["kernels_ms_by_phase"]
print("cfg5 switch=%s %7.1f it/s E %s" % (v, a["value"], a["config"]["final_E"]), a["phase_iterations_per_s"], flush=True)
print("     ", {ph: {n: round(x["ms"] / x["launches"], 3) for n, x in ks.items() if n in ("scatter", "lattice", "transform", "sweep_deformable")} for ph, ks in k.items() if ph != "linear"}, flush=True)
PY
done

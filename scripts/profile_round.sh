#!/bin/bash
# profile_round.sh TAG -- on the GPU box: the round-end set, one after the other (copy gpurun_out/TAG_* into profiles/):
#   profile_bench.sh (cfg 3: the line with cpu_baseline, exact_mode and end_to_end, kernel times, the driver's 20-step command,
#   rocprofv3 stats and counter passes), bench.py --exact, the rank-of-eight proxy with and without brackets, 3- and 5-rank
#   rehearsals on the one GPU (host-staged transport), profile_cfg5.sh, cfg5_level4_traffic.sh, profile_match.sh
set -e
TAG=${1:-round}
cd "$(dirname "$0")/.."
scripts/profile_bench.sh $TAG > gpurun_out/${TAG}_profile_bench.log 2>&1
python3 bench.py --exact --no-cpu-baseline --no-end-to-end > gpurun_out/${TAG}_bench_exact_n1.json 2> gpurun_out/${TAG}_exact.err
python3 bench.py --exact --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/${TAG}_bench_exact_n1_steps20.json 2>> gpurun_out/${TAG}_exact.err
python3 bench.py --shard-of 0 8 --steps 130 > gpurun_out/${TAG}_proxy_cfg3.json 2> /dev/null
FROG_PROXY_SWEEPS_ONLY=1 python3 bench.py --shard-of 0 8 --steps 130 > gpurun_out/${TAG}_proxy_cfg3_sweeps_only.json 2> /dev/null
FROG_BENCH_BACKEND=gloo FROG_BENCH_HOSTS=native python3 bench.py --gpus 3 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_n3_rehearsal.json 2> gpurun_out/${TAG}_n3.err || true
FROG_BENCH_BACKEND=gloo FROG_BENCH_HOSTS=native python3 bench.py --gpus 5 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_n5_rehearsal.json 2> gpurun_out/${TAG}_n5.err || true
scripts/profile_cfg5.sh $TAG > gpurun_out/${TAG}_profile_cfg5.log 2>&1
scripts/cfg5_level4_traffic.sh $TAG > gpurun_out/${TAG}_level4.log 2>&1
scripts/profile_match.sh $TAG > gpurun_out/${TAG}_profile_match.log 2>&1
python3 - $TAG <<'PY'
import json, sys
t = sys.argv[1]
def line(f):
    try: return json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: return {"error": str(e)}
for name in ("bench_n1", "bench_n1_steps20", "bench_n1_kernel_times", "bench_exact_n1", "bench_exact_n1_steps20", "proxy_cfg3", "proxy_cfg3_sweeps_only",
             "bench_n3_rehearsal", "bench_n5_rehearsal", "bench_cfg5", "bench_match"):
    d = line(f"gpurun_out/{t}_{name}.json")
    print(name, d.get("value"), d.get("config", {}).get("final_E"), d.get("replicas_identical"), (d.get("roofline") or {}).get("frac"), (d.get("exact_mode") or {}).get("value"), (d.get("end_to_end") or {}).get("wall_s"))
PY

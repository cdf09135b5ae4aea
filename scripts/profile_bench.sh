#!/bin/bash
# profile_bench.sh TAG -- on the GPU box: bench line + rocprofv3 kernel stats + HBM/L2 counter passes,
# condensed into gpurun_out/TAG_bench_n1.{json,txt} (copy the two files into profiles/ to keep them).
set -e
TAG=${1:-prof}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/$TAG
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --no-cpu-baseline --kernel-times > $O/bench_kernel_times.json 2>> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2>> $O/bench.err     # the round-end driver's command line
ARGS="bench.py --steps 65 --warmup 10 --no-cpu-baseline --kernel-times"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python3 $ARGS > $O/under_rocprof.json 2> $O/trace.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/fetch.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/write.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/tcc -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/tcc.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $O/tcp -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/tcp.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/sq -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/sq.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum -d $O/ta -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/ta.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $O/lat -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/lat.log
# the same sweep without the outlier-culling list (every half-link walked), for the before / after of the counters
export FROG_CULL=0
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/nocull_trace -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/nocull_trace.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/nocull_sq -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/nocull_sq.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum -d $O/nocull_ta -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/nocull_ta.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum -d $O/nocull_lat -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/nocull_lat.log
unset FROG_CULL
python3 scripts/summarize_profile.py gpurun_out/${TAG}_nocull_n1.txt $O/nocull_trace $O/nocull_sq $O/nocull_ta $O/nocull_lat > /dev/null
python3 scripts/summarize_profile.py gpurun_out/${TAG}_bench_n1.txt $O/trace $O/fetch $O/write $O/tcc $O/tcp $O/sq $O/ta $O/lat $(python3 -c "import json; print(json.load(open('$O/bench.json'))['roofline']['half_links_owned'])") > /dev/null
cp $O/bench.json gpurun_out/${TAG}_bench_n1.json
cp $O/bench_kernel_times.json gpurun_out/${TAG}_bench_n1_kernel_times.json
cp $O/bench_steps20.json gpurun_out/${TAG}_bench_n1_steps20.json
cp $O/under_rocprof.json gpurun_out/${TAG}_bench_n1_under_rocprof.json
# roctx ranges of the kernel groups (FROG_ROCTX=1): a marker trace next to the kernel trace
FROG_ROCTX=1 timeout -k 10 200 rocprofv3 --kernel-trace --marker-trace -d $O/roctx -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/roctx.log || true
python3 - <<PY > gpurun_out/${TAG}_roctx_ranges.txt || true
import csv, glob, collections
files = glob.glob("$O/roctx/**/p_marker_api_trace.csv", recursive=True)
tot = collections.Counter(); cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Function") or r.get("Name") or ""
        if name.startswith("frog:"):
            tot[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; cnt[name] += 1
print("# FROG_ROCTX=1 bench.py --steps 20 --warmup 5 under rocprofv3 --marker-trace: roctx ranges (host-side enqueue time, us)")
for n, v in tot.most_common(): print(f"{n:40s} {v:10.1f} us  x{cnt[n]}")
print("files", files)
PY
python3 scripts/trace_gaps.py $(dirname $(find $O/roctx -name p_kernel_trace.csv | head -1)) > gpurun_out/${TAG}_timeline_steps20.txt || true
head -12 gpurun_out/${TAG}_bench_n1.txt
python3 -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_n1.json')); print(d['value'], d['roofline'], d['cpu_baseline']['value'])"

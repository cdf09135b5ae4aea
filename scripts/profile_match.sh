#!/bin/bash
# profile_match.sh TAG -- on the GPU box: bench_match.py line + rocprofv3 kernel stats + SQ counters
set -e
TAG=${1:-match}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/$TAG
mkdir -p $O
python3 bench_match.py --images 100 --cpu-jobs 24 > $O/bench.json 2> $O/bench.err
ARGS="bench_match.py --images 30 --cpu-jobs 0"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python3 $ARGS > $O/under_rocprof.json 2> $O/trace.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/sq -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/sq.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $O/lds -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/lds.log
python3 scripts/summarize_profile.py gpurun_out/${TAG}_bench_match.txt $O/trace $O/sq $O/lds > /dev/null
cp $O/bench.json gpurun_out/${TAG}_bench_match.json
cat gpurun_out/${TAG}_bench_match.txt | head -30
cut -c1-1500 $O/bench.json

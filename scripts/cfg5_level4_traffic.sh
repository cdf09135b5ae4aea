#!/bin/bash
# cfg5_level4_traffic.sh TAG [ENV=V ...] -- on the GPU box: kernel trace + FETCH_SIZE + WRITE_SIZE passes (separate runs) of
# bench.py --config 5 --steps 65, and the finest level's per-kernel traffic (scripts/level_traffic.py) into gpurun_out/TAG_cfg5_level4_traffic.txt
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for kv in "$@"; do export "$kv"; done
O=gpurun_out/${TAG}_l4
mkdir -p $O
ARGS="bench.py --config 5 --steps 65"
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/trace -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/trace.log
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/fetch.log
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/write.log
python3 scripts/level_traffic.py $O/trace $O/fetch $O/write | tee gpurun_out/${TAG}_cfg5_level4_traffic.txt
rm -rf $O/trace $O/fetch $O/write

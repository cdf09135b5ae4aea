#!/bin/bash
# retake_traffic.sh TAG -- on the GPU box: only the FETCH_SIZE / WRITE_SIZE passes of the deformable sweep, for cfg 3 and cfg 5, into
# gpurun_out/TAG_hbm_traffic.json and gpurun_out/TAG_cfg5_hbm_traffic.json (merge with scripts/merge_traffic.py).  profiles/hbm_traffic.json
# is keyed to the hash of the device sources: after a change that does not touch the sweep this re-measures the two entries without
# the whole profile set.
set -e
TAG=${1:-retake}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for CFG in 3 5; do
  O=gpurun_out/${TAG}_t$CFG
  mkdir -p $O
  if [ $CFG = 3 ]; then ARGS="bench.py --steps 65 --warmup 10 --no-cpu-baseline --kernel-times"; OUT=gpurun_out/${TAG}_bench_n1.txt; else ARGS="bench.py --config 5 --steps 65 --kernel-times"; OUT=gpurun_out/${TAG}_cfg5_bench_n1.txt; fi
  python3 $ARGS > $O/bench.json 2> $O/bench.err
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/trace.log
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/fetch.log
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/write.log
  python3 scripts/summarize_profile.py $OUT $O/trace $O/fetch $O/write $(python3 -c "import json; print(json.load(open('$O/bench.json'))['roofline']['half_links_owned'])") > /dev/null
done
ls gpurun_out/${TAG}*hbm_traffic.json

#!/bin/bash
# profile_cfg5.sh TAG -- on the GPU box: BASELINE.json configs[4] (500 images, five levels) on one GPU: the bench line, per-phase
# kernel times, rocprofv3 kernel stats and the FETCH_SIZE / WRITE_SIZE passes of the deformable sweep
# (gpurun_out/TAG_cfg5_hbm_traffic.json: merge into profiles/hbm_traffic.json with scripts/merge_traffic.py).
set -e
TAG=${1:-prof}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/${TAG}_cfg5
mkdir -p $O
python3 bench.py --config 5 > $O/bench.json 2> $O/bench.err
python3 bench.py --config 5 --kernel-times > $O/bench_kernel_times.json 2>> $O/bench.err
ARGS="bench.py --config 5 --steps 65 --kernel-times"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python3 $ARGS > $O/under_rocprof.json 2> $O/trace.log
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/fetch.log
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/write.log
python3 scripts/summarize_profile.py gpurun_out/${TAG}_cfg5_bench_n1.txt $O/trace $O/fetch $O/write $(python3 -c "import json; print(json.load(open('$O/bench.json'))['roofline']['half_links_owned'])") > /dev/null
cp $O/bench.json gpurun_out/${TAG}_bench_cfg5.json
cp $O/bench_kernel_times.json gpurun_out/${TAG}_bench_cfg5_kernel_times.json
head -30 gpurun_out/${TAG}_cfg5_bench_n1.txt
python3 -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_cfg5.json')); print(d['value'], d['roofline']['frac'], d['iteration']['iteration_frac'])"

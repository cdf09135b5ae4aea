#!/bin/bash
# pmc_match.sh -- on the GPU box: MFMA / wait / memory counters of match_mfma_kernel (bench_match.py --images 30) in two rocprofv3 --pmc passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_match_pmc; mkdir -p $O
ARGS="bench_match.py --images 30 --cpu-jobs 0"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY -d $O/a -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/a.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_SALU SQ_WAVES -d $O/b -o p --output-format csv -- python3 $ARGS > /dev/null 2> $O/b.log
python3 - <<PY
import csv,glob,collections
for d in ("a","b"):
    for f in glob.glob(f"$O/{d}/**/*_counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda:[0.0,0])
        for r in csv.DictReader(open(f)):
            if "match_mfma" in r["Kernel_Name"]:
                acc[r["Counter_Name"]][0]+=float(r["Counter_Value"]); acc[r["Counter_Name"]][1]+=1
        for k,(v,n) in sorted(acc.items()): print(k, round(v/n,1), n)
PY
tail -3 $O/a.log

#!/bin/bash
# kt_variants_cfg5.sh -- on the GPU box: per-phase transform times of cfg 5 and of the one-of-eight shard proxy of cfg 3 under the default library and the
# build variants frog_amd/lib/variants/libfrog_hip_{ptw2,ptw3,ptw4}.so (scripts/build_variant.sh: thread-per-point transform capped at 2 / 3 / 4 wavefronts per SIMD)
L=frog_amd/lib
cp $L/libfrog_hip.so $L/keep.so
for name in default ptw2 ptw3 ptw4; do
  if [ "$name" = default ]; then cp $L/keep.so $L/libfrog_hip.so; else cp $L/variants/libfrog_hip_$name.so $L/libfrog_hip.so; fi
  python3 bench.py --config 5 --kernel-times 2>/dev/null | python3 -c "
import json, sys
k = json.loads(sys.stdin.read())
print('$name cfg5', round(k['value'], 1), {ph: round(ks['transform']['ms']/ks['transform']['launches'],4) for ph, ks in k['kernels_ms_by_phase'].items()})
"
  python3 bench.py --shard-of 0 8 2>/dev/null | python3 -c "
import json, sys
k = json.loads(sys.stdin.read())
print('$name cfg3 rank0of8', round(k['value'], 1), {ph: round(ks['transform']['ms']/ks['transform']['launches'],4) for ph, ks in k['kernels_ms_by_phase'].items()})
"
done
cp $L/keep.so $L/libfrog_hip.so

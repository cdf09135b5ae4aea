#!/bin/bash
# kt_env.sh "VAR=a" "VAR=b" ... -- per-phase average kernel times (bench.py --kernel-times, 130 steps) under each environment
for envs in "$@"; do
  env $envs python3 bench.py --no-cpu-baseline --steps 130 --kernel-times 2>/dev/null | python3 -c "
import json, sys
k = json.loads(sys.stdin.read())
print('$envs', round(k['value'], 1))
for ph, ks in k['kernels_ms_by_phase'].items():
    print('   ', ph, {n: round(v['ms'] / v['launches'], 4) for n, v in ks.items() if n in ('sweep_deformable', 'sweep_linear', 'scatter', 'lattice', 'transform', 'stats')})
"
done

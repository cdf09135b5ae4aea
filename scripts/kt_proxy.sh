#!/bin/bash
# kt_proxy.sh "VAR=a" "VAR=b" ... -- per-phase kernel times of the one-of-eight shard proxy (bench.py --shard-of 0 8 --steps 130)
for envs in "$@"; do
  env $envs python3 bench.py --shard-of 0 8 --steps 130 2>/dev/null | python3 -c "
import json, sys
k = json.loads(sys.stdin.read())
print('$envs', round(k['value'], 1))
for ph, ks in k['kernels_ms_by_phase'].items():
    print('   ', ph, {n: round(v['ms'] / v['launches'], 4) for n, v in ks.items() if n in ('sweep_deformable', 'sweep_linear', 'scatter', 'lattice', 'transform')})
"
done

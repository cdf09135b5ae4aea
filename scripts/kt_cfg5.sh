#!/bin/bash
# kt_cfg5.sh "VAR=a" ... -- per-phase kernel times of cfg 5 (bench.py --config 5 --kernel-times)
for envs in "$@"; do
  env $envs python3 bench.py --config 5 --kernel-times 2>/dev/null | python3 -c "
import json, sys
k = json.loads(sys.stdin.read())
print('$envs', round(k['value'], 1), k['setup_seconds']['lattice_setups'])
for ph, ks in k['kernels_ms_by_phase'].items():
    print('   ', ph, {n: round(v['ms'] / v['launches'], 4) for n, v in ks.items()})
"
done

#!/bin/bash
# ab_env.sh "VAR=a VAR2=b" "VAR=c" ... -- the driver's 20-step line (three runs) and the 650-step line under each environment
i=0
for envs in "$@"; do
  for r in 1 2 3; do env $envs python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab_${i}_s20_$r.json 2>/dev/null; done
  env $envs python3 bench.py --no-cpu-baseline > gpurun_out/ab_${i}_650.json 2>/dev/null
  python3 - "$envs" $i <<'PY'
import json, sys
envs, i = sys.argv[1], sys.argv[2]
s = [json.load(open(f"gpurun_out/ab_{i}_s20_{r}.json"))["value"] for r in (1, 2, 3)]
a = json.load(open(f"gpurun_out/ab_{i}_650.json"))
print(f"{envs:50s} 20-step {' '.join(f'{v:7.1f}' for v in s)}   650-step {a['value']:7.1f}  E {a['config']['final_E']}")
PY
  i=$((i+1))
done

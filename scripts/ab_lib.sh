#!/bin/bash
# ab_lib.sh NAME ... -- as ab_env.sh, but each run uses frog_amd/lib/variants/libfrog_hip_NAME.so IN PLACE of libfrog_hip.so
# ("default" = the built library): libfrog_host.so is linked against libfrog_hip.so by name, so a variant selected with
# FROG_HIP_LIB would sit beside the default one in the native host's process.  Meant for the GPU box (a scratch copy of the tree).
set -e
L=frog_amd/lib
cp $L/libfrog_hip.so $L/libfrog_hip_default.keep
i=0
for name in "$@"; do
  if [ "$name" = default ]; then cp $L/libfrog_hip_default.keep $L/libfrog_hip.so; else cp $L/variants/libfrog_hip_$name.so $L/libfrog_hip.so; fi
  for r in 1 2 3; do python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab_${i}_s20_$r.json 2>/dev/null; done
  python3 bench.py --no-cpu-baseline --kernel-times > gpurun_out/ab_${i}_650.json 2>/dev/null
  python3 - "$name" $i <<'PY'
import json, sys
name, i = sys.argv[1], sys.argv[2]
s = [json.load(open(f"gpurun_out/ab_{i}_s20_{r}.json"))["value"] for r in (1, 2, 3)]
a = json.load(open(f"gpurun_out/ab_{i}_650.json"))
k = a.get("kernels_ms", {})
per = " ".join(f"{n} {v['total_ms'] / max(v['launches'], 1):.4f}" for n, v in k.items() if isinstance(v, dict) and "total_ms" in v)
print(f"{name:24s} 20-step {' '.join(f'{v:7.1f}' for v in s)}   650-step(kernel-times) {a['value']:7.1f}  E {a['config']['final_E']}  ms/launch: {per}", flush=True)
PY
  i=$((i+1))
done
cp $L/libfrog_hip_default.keep $L/libfrog_hip.so

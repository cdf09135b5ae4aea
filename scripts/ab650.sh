#!/bin/bash
# ab650.sh NAME ... : plain 650-step bench twice per library variant
L=frog_amd/lib
cp $L/libfrog_hip.so $L/libfrog_hip_default.keep
for name in "$@"; do
  if [ "$name" = default ]; then cp $L/libfrog_hip_default.keep $L/libfrog_hip.so; else cp $L/variants/libfrog_hip_$name.so $L/libfrog_hip.so; fi
  for r in 1 2; do python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), {k: round(v,1) for k,v in d['phase_iterations_per_s'].items()})"; done
done
cp $L/libfrog_hip_default.keep $L/libfrog_hip.so

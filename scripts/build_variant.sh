#!/bin/bash
# build_variant.sh NAME [-DFLAG ...]: an experimental build of libfrog_hip.so under frog_amd/lib/variants/
# (select it at run time with FROG_HIP_LIB=variants/libfrog_hip_NAME.so)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$ROOT/frog_amd/lib/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fopenmp -Wall \
  -Wno-unused-result -I"$ROOT/include" "$@" -shared -o "$ROOT/frog_amd/lib/variants/libfrog_hip_$name.so" \
  "$ROOT/frog_amd/csrc/device/frog_hip.hip" "$ROOT/frog_amd/csrc/device/match.hip" "$ROOT/frog_amd/csrc/device/chain.hip"

#!/bin/bash
# bisect_parity.sh -- VERDICT r5 item 1(b): which of round 5's two arithmetic changes moved the product path's raw coefficients
# on cfg 3's whole default schedule (2.0e-4 -> 4.2e-3 on one rim node).  Four cells: B-spline transform f32 / f64
# (FROG_K11_F64=1) x scatter with one rounding per tap (fma) / two (a build with -DFROG_SCATTER_FMA=0 -DFROG_SCATTER_QUADS=0:
# the quad form of phase 2 multiplies-and-adds in one v_fmac_f32_dpp whatever FROG_SCATTER_FMA says -- the first run of this
# script built -DFROG_SCATTER_FMA=0 alone and its "nofma" cells were the fma cells, digit for digit).  Each cell: scripts/parity_reference_order.py, product path vs reference-order mode on the device, 650 iterations.
# Output: gpurun_out/parity_reference_order_<cell>.json.
set -e
cd "$(dirname "$0")/.."
CELLS="${@:-f32_fma f64_fma f32_nofma f64_nofma}"
[ -f frog_amd/lib/variants/libfrog_hip_nofma2.so ] || scripts/build_variant.sh nofma2 -DFROG_SCATTER_FMA=0 -DFROG_SCATTER_QUADS=0
for cell in $CELLS; do
  case $cell in
    f32_fma)   python3 scripts/parity_reference_order.py --tag $cell > gpurun_out/bisect_$cell.log 2>&1 ;;
    f64_fma)   FROG_K11_F64=1 python3 scripts/parity_reference_order.py --tag $cell > gpurun_out/bisect_$cell.log 2>&1 ;;
    f32_nofma) FROG_HIP_LIB=variants/libfrog_hip_nofma2.so python3 scripts/parity_reference_order.py --tag $cell > gpurun_out/bisect_$cell.log 2>&1 ;;
    f64_nofma) FROG_HIP_LIB=variants/libfrog_hip_nofma2.so FROG_K11_F64=1 python3 scripts/parity_reference_order.py --tag $cell > gpurun_out/bisect_$cell.log 2>&1 ;;
  esac
done
python3 - $CELLS <<'PY'
import json, sys
for cell in sys.argv[1:]:
    r = json.load(open(f"gpurun_out/parity_reference_order_{cell}.json"))
    print(cell, "E %.2e chain %.2e (%.2e mm) grids %s" % (r["E"], r["chain"]["rel"], r["chain"]["mm"], r["grids"]))
    for k, d in enumerate(r["lattices"]):
        print("   lattice %d raw %.2e (image %d node %d weight %.2e own support %.2e from %d points) dense %.2e field %.2e weighted %.2e"
              % (k, d["raw"], d["raw_image"], d["raw_node"], d["raw_node_weight"], d["raw_node_support"], d["raw_node_points"],
                 d["dense_field"], d["field"], d["weighted"]))
PY

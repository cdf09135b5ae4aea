#!/bin/bash
# e2e_volume_timeline.sh [N] -- on the GPU box: bin/VolumeTransform as a process, a 256^3 int16 volume (nii.gz) resliced onto a 256^3
# grid through the inverse of a matrix + three-lattice chain, the shell's clock around it.
cd "$(dirname "$0")/.."
ROOT=$PWD
D=/tmp/frog_volume_e2e; rm -rf $D; mkdir -p $D
python3 - $D <<'PY'
import sys, json, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from frog_amd.volume import write_volume
d = sys.argv[1]
n = 256
z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
rng = np.random.default_rng(1)
src = (1000 + 400 * np.sin(x / 16.0) * np.cos(y / 17.0) + 2 * z + rng.normal(0, 12, (n, n, n))).astype(np.int16)   # anatomy-like: smooth + noise
write_volume(d + "/src.nii.gz", src, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
write_volume(d + "/ref.nii.gz", np.zeros((n, n, n), np.uint8), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
out = [{"type": "vtkMatrixToLinearTransform", "matrix": [1.02, 0.01, 0, -3, -0.01, 0.98, 0.02, 2, 0, -0.02, 1.01, 1, 0, 0, 0, 1]}]
for k, cells in enumerate((4, 8, 16)):
    dims = [cells + 3] * 3
    sp = 256.0 / cells
    c = rng.normal(0, 1.5 / (k + 1), (dims[2], dims[1], dims[0], 3))
    out.append({"type": "vtkBSplineTransform", "dimensions": dims, "origin": [-sp] * 3, "spacing": [sp] * 3, "coeffs": c.ravel().tolist()})
open(d + "/t.json", "w").write(json.dumps({"transforms": out}))
PY
cd $D
ls -la src.nii.gz | awk '{print "src.nii.gz", $5, "bytes"}'
for k in $(seq 1 ${1:-3}); do
  t0=$(date +%s.%N)
  env FROG_TIMING=1 $ROOT/bin/VolumeTransform src.nii.gz ref.nii.gz -t t.json -o out.nii.gz > out.txt 2>&1
  t1=$(date +%s.%N)
  python3 -c "import sys; a,b=map(float,sys.argv[1:3]); print('bin/VolumeTransform %.3f s' % (b-a))" $t0 $t1
  grep -i "timing\|computed in\|loaded in\|written in\|error" out.txt | head -12
done
ls -la out.nii.gz | awk '{print "out.nii.gz", $5, "bytes"}'

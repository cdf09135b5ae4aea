#!/bin/bash
# e2e_match_timeline.sh [N] [EXT] -- on the GPU box: `bin/match` then `bin/frog` on the benchmark group's size (100 images x 20 000
# keypoints x 48-D, all 4 950 image pairs), the shell's clock around each process: the whole pipeline a user of the reference runs.
# EXT = csv.gz (default, what the reference's detector writes), csv or bin.
cd "$(dirname "$0")/.."
ROOT=$PWD
D=/tmp/frog_match_e2e; rm -rf $D; mkdir -p $D
EXT=${2:-csv.gz}
python3 - $D $EXT <<'PY'
import sys, time, multiprocessing as mp
sys.path.insert(0, ".")
from frog_amd.match import synthetic_keypoints, write_keypoints
from frog_amd._abi import usable_cpus
d, ext = sys.argv[1], sys.argv[2]
t = time.time()
imgs = synthetic_keypoints(100, 20000, dim=48, seed=1)
print("generated in %.1f s" % (time.time() - t)); t = time.time()
def write(i):
    write_keypoints("%s/points%d.%s" % (d, i, ext), imgs[i])
with mp.get_context("fork").Pool(usable_cpus()) as pool:
    pool.map(write, range(len(imgs)))
with open(d + "/list.txt", "w") as fh:
    for i in range(len(imgs)):
        fh.write("%s/points%d.%s\n" % (d, i, ext))
print("written in %.1f s" % (time.time() - t))
PY
cd $D
du -sh . | cut -f1
for k in $(seq 1 ${1:-2}); do
  t0=$(date +%s.%N)
  env FROG_TIMING=1 $ROOT/bin/match list.txt -o pairs.bin -d 1 > match_out.txt 2>&1
  t1=$(date +%s.%N)
  env FROG_TIMING=1 $ROOT/bin/frog pairs.bin -q 1 -dl 3 > frog_out.txt 2>&1
  t2=$(date +%s.%N)
  python3 -c "import sys; a,b,c=map(float,sys.argv[1:4]); print('bin/match %.3f s, bin/frog %.3f s' % (b-a, c-b))" $t0 $t1 $t2
  grep -o "^[A-Za-z][A-Za-z .]*\.\.\.\|^Found.*\| : [0-9.e-]*s\|Nb Match.*\|\[timing\].*" match_out.txt | tail -25
  ls -la pairs.bin | awk '{print "pairs.bin", $5, "bytes"}'
done

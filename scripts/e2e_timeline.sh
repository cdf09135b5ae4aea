#!/bin/bash
# e2e_timeline.sh [N] -- on the GPU box: bin/frog on the benchmark group (cfg 3, default schedule), FROG_TIMING=1, with the shell's
# clock around the process: what passes before main, inside it by stage, and after it returns.  N runs (default 3).
cd "$(dirname "$0")/.."
D=/tmp/frog_e2e; rm -rf $D; mkdir -p $D
bin/frog --synth $D/pairs.bin 100 20000 10101 1 > /dev/null
cd $D
for k in $(seq 1 ${1:-3}); do
  t0=$(date +%s.%N)
  env FROG_TIMING=1 $FROG_E2E_ENV $OLDPWD/bin/frog pairs.bin -q 1 -dl 3 > out.txt 2>&1
  t1=$(date +%s.%N)
  python3 - $t0 $t1 <<'PY'
import sys, re
t0, t1 = float(sys.argv[1]), float(sys.argv[2])
txt = open("out.txt").read()
a = float(re.search(r"main entered at ([0-9.]+)", txt).group(1)); b = float(re.search(r"main returns at ([0-9.]+)", txt).group(1))
print("wall %.3f s: before main %.3f, main %.3f, after main %.3f" % (t1 - t0, a - t0, b - a, t1 - b))
for line in txt.splitlines():
    if line.startswith("[timing]") and "main " not in line or line.startswith("Iteration loops"): print("   ", line)
PY
done

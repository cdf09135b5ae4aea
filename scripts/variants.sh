set -e
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for w in 0 1; do
  FROG_WIDE_RECORDS=$w timeout -k 10 200 python bench.py --steps 65 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms']; print('wide=$w', round(d['value'],1), 'sweepD', round(k['sweep_deformable']['total_ms']/k['sweep_deformable']['launches'],4), 'sweepL', round(k['sweep_linear']['total_ms']/k['sweep_linear']['launches'],4), d['config']['final_E'], d['setup_seconds'])"
done

set -e
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 200 python bench.py --steps 65 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms']; print(round(d['value'],1), d['phase_iterations_per_s'], 'sweepD', round(k['sweep_deformable']['total_ms']/k['sweep_deformable']['launches'],4), d['config']['final_E'])
for ph,v in d['kernels_ms_by_phase'].items(): print(ph, round(sum(x['ms'] for x in v.values()),2), {a:round(x['ms'],2) for a,x in v.items()})"

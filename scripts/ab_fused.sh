for f in 0 1; do
  FROG_SWEEP_FUSED=$f python3 bench.py --no-cpu-baseline > gpurun_out/r03_e_fused$f.json 2>/dev/null
  FROG_SWEEP_FUSED=$f python3 bench.py --no-cpu-baseline --kernel-times > gpurun_out/r03_e_fused${f}_kt.json 2>/dev/null
  FROG_SWEEP_FUSED=$f python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r03_e_fused${f}_s20.json 2>/dev/null
done
python3 - <<'PY'
import json
for f in (0,1):
    a=json.load(open(f"gpurun_out/r03_e_fused{f}.json")); k=json.load(open(f"gpurun_out/r03_e_fused{f}_kt.json")); s=json.load(open(f"gpurun_out/r03_e_fused{f}_s20.json"))
    print("fused",f,"650:",round(a["value"],1),"20:",round(s["value"],1),"E",a["config"]["final_E"], "phases", {p:round(v,1) for p,v in a["phase_iterations_per_s"].items()})
    for ph,ks in k["kernels_ms_by_phase"].items():
        print("  ",ph,{n:round(v["ms"]/v["launches"],4) for n,v in ks.items()})
PY

import numpy as np
for tag in ("l0","l2"):
    a=np.fromfile(f"gpurun_out/lat_trace_{tag}.bin",dtype=np.uint64).reshape(-1,4)
    n=int((a[:,1]>0).sum()); a=a[:n]
    t0=a[:,0].astype(np.int64); t1=a[:,1].astype(np.int64); base=t0.min()
    w=a[:,3]
    ph=[((w>>(16*k))&0xFFFF)*0.01 for k in range(4)]
    print(tag,"blocks",n,"span us",(t1.max()-base)*0.01,"start spread",(t0.max()-base)*0.01,"block dur mean",((t1-t0)*0.01).mean())
    print("  loads+propose %.1f  accumulate %.1f  mean+centre+stores %.1f  count+ticket %.1f"%tuple(p.mean() for p in ph))

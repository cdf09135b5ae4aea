import numpy as np, sys
for tag in ("l0","l2"):
    a=np.fromfile(f"gpurun_out/scatter_trace_{tag}.bin",dtype=np.uint64).reshape(-1,4)
    n=int((a[:,1]>0).sum())
    a=a[:n]
    t0=a[:,0].astype(np.int64); t1=a[:,1].astype(np.int64)
    base=t0.min(); t0-=base; t1-=base
    dur=(t1-t0)*10e-3  # us (100 MHz)
    w=a[:,3].copy(); a[:,3]=w&0xFFFF
    ld=((w>>16)&0xFFFF)*10e-3; p1=((w>>32)&0xFFFF)*10e-3; p2=((w>>48)&0xFFFF)*10e-3
    print(' per block us: load-wait %.1f phase1 %.1f phase2 %.1f  (of dur %.1f)'%(ld.mean(),p1.mean(),p2.mean(),dur.mean()))
    hw=a[:,2]&0xffffffff; xcc=(a[:,2]>>32)&0xf
    cu=(hw>>8)&0xf; sh=(hw>>12)&1; se=(hw>>13)&7
    print(tag,"blocks",n,"span_us",t1.max()*10e-3,"start spread us", t0.max()*10e-3)
    print(" dur us: mean %.1f p50 %.1f p90 %.1f max %.1f"%(dur.mean(),np.median(dur),np.percentile(dur,90),dur.max()), "points mean", a[:,3].mean())
    # occupancy over time
    T=int(t1.max())+1
    occ=np.zeros(T+1); np.add.at(occ,t0,1); np.add.at(occ,t1,-1); occ=np.cumsum(occ)
    print(" avg concurrent blocks %.0f  max %d"%(occ[:T].mean(), occ.max()))
    bins=np.array_split(occ[:T],10)
    print(" occupancy deciles", [int(b.mean()) for b in bins])
    # starts per decile
    h,_=np.histogram(t0,bins=10,range=(0,T)); print(" starts per decile",h.tolist())
    key=(xcc.astype(np.int64)*64+se*8+sh*4)*16+cu
    u,c=np.unique(key,return_counts=True); print(" distinct CUs",len(u),"blocks/CU min/mean/max",c.min(),c.mean(),c.max())
    # dur vs points
    for lo,hi in ((0,64),(64,192),(192,385)):
        m=(a[:,3]>lo)&(a[:,3]<=hi)
        if m.any(): print("  points (%d,%d]: n %d dur mean %.1f"%(lo,hi,m.sum(),dur[m].mean()))

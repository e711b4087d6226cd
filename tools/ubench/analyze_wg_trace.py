# tools/ubench/analyze_wg_trace.py -- reads gpurun_out/gemm_trace_*.bin (gemm_sweep <reps> trace): per-workgroup wall-clock stamps -> round structure, CU occupancy timeline
import numpy as np
for name,nb in [("p128x128_w32x64_512t",1008),("sb128x128_w32x64_512t",1008),("sb256x256_512t",256)]:
    raw=np.fromfile(f"/root/repo/gpurun_out/gemm_trace_{name}.bin",dtype=np.int64)
    g=raw[:nb*8].reshape(nb,8); pre,post=raw[nb*8],raw[nb*8+1]
    t0=g[:,0].min()
    us=lambda x:(x-pre)/100.0
    st=us(g[:,0]); pro=us(g[:,1]); le=us(g[:,2]); mid=us(g[:,5]); en=us(g[:,3])
    print("==",name," pre-stamp->post-stamp %.1f us"%((post-pre)/100.0))
    print(" first WG start %.1f  last first-round start %.1f ; kernel last WG end %.1f; post stamp at %.1f"%(st.min(), np.sort(st)[min(511,nb-1)], en.max(), us(post)))
    r1=np.argsort(st)[:min(512,nb)]; r2=np.argsort(st)[min(512,nb):]
    for nm,idx in (("round1",r1),("round2",r2)):
        if len(idx)==0: continue
        print(" %s: n=%d start %.1f..%.1f  prologue %.2f  loop %.1f  epilogue %.2f (to LDS-turn %.2f)  end %.1f..%.1f"%(nm,len(idx),st[idx].min(),st[idx].max(),np.median(pro[idx]-st[idx]),np.median(le[idx]-pro[idx]),np.median(en[idx]-le[idx]),np.median(mid[idx]-le[idx]),en[idx].min(),en[idx].max()))
    # per-slot gap between round1 end and round2 start on the same CU: match by hw id
    hw=g[:,4]
    key=((hw>>8)&0xff)|((hw>>32)<<8)
    gaps=[]
    for b in r2:
        cands=[a for a in r1 if key[a]==key[b] and en[a]<=st[b]+0.05]
        if cands:
            a=max(cands,key=lambda a:en[a]); gaps.append(st[b]-en[a])
    if gaps: print(" redispatch gap (round-2 start - latest round-1 end on the same CU): median %.2f us p90 %.2f"%(np.median(gaps),np.percentile(gaps,90)))
print()
for name,nb in [("p128x128_w32x64_512t",1008),("sb128x128_w32x64_512t",1008)]:
    raw=np.fromfile(f"/root/repo/gpurun_out/gemm_trace_{name}.bin",dtype=np.int64)
    g=raw[:nb*8].reshape(nb,8); pre=raw[nb*8]
    st=(g[:,0]-pre)/100.0; en=(g[:,3]-pre)/100.0; le=(g[:,2]-pre)/100.0; pro=(g[:,1]-pre)/100.0
    key=((g[:,4]>>8)&0xff)|((g[:,4]>>32)<<8)
    cus=np.unique(key)
    print(name,"CUs seen:",len(cus))
    ts=np.arange(0,160,10.0)
    for t in ts:
        res=np.array([np.sum((st[key==c]<=t)&(en[key==c]>t)) for c in cus])
        inloop=np.array([np.sum((pro[key==c]<=t)&(le[key==c]>t)) for c in cus])
        print("  t=%5.0f us: CUs with 2 WGs %3d, 1 WG %3d, 0 WGs %3d | WGs in main loop: 2:%3d 1:%3d 0:%3d"%(t,(res==2).sum(),(res==1).sum(),(res==0).sum(),(inloop==2).sum(),(inloop==1).sum(),(inloop==0).sum()))
    # tiles per CU
    cnt=np.array([np.sum(key==c) for c in cus]); print("  tiles per CU: min %d max %d; hist"%(cnt.min(),cnt.max()),np.bincount(cnt))

# tools/ubench/analyze_barrier_trace.py -- reads gpurun_out/gemm_tr2_*.bin (gemm_sweep <reps> trace2): per-wave shader-clock stamps around the K-loop barrier
import numpy as np, sys
IT=24
for name in ["w32x64_512t","w64x64_256t"]:
    raw=np.fromfile(f"/root/repo/gpurun_out/gemm_tr2_{name}.bin",dtype=np.int64)
    n2=1008*16*(IT+1)*2
    tr=raw[:n2].reshape(1008,16,IT+1,2); g=raw[n2:].reshape(-1,8)
    nw = 8 if "512t" in name else 4
    print("==",name)
    # effective clock: loop wall duration (stamps 1->2, 100 MHz) vs shader clocks
    wall=(g[:,2]-g[:,1])*10e-9  # s
    A=tr[:,:nw,:IT,0]; B=tr[:,:nw,:IT,1]
    per_it=(B[:,0,IT-1]-B[:,0,4])/(IT-1-4)   # shader clocks per K-tile iteration, wave 0
    print(" loop wall us: median %.1f  (first-round blocks %.1f, second-round %.1f)"%(np.median(wall)*1e6,np.median(wall[:512])*1e6,np.median(wall[512:])*1e6))
    print(" shader clocks / iteration (wave0): median %.0f  p10 %.0f p90 %.0f"%(np.median(per_it),np.percentile(per_it,10),np.percentile(per_it,90)))
    print(" => effective clock = 64 it * clocks / wall = %.0f MHz"%(np.median(64*per_it/wall)/1e6))
    wait=(B-A)[:,:,4:]; work=(A[:,:,5:]-B[:,:,4:-1])
    print(" barrier wait clocks: mean %.0f median %.0f p90 %.0f ; work between barriers: mean %.0f median %.0f"%(wait.mean(),np.median(wait),np.percentile(wait,90),work.mean(),np.median(work)))
    # per-wave wait
    print(" mean wait by wave:",np.round(wait.mean(axis=(0,2))).astype(int))
    hw=tr[:,:nw,IT,0]
    simd=(hw>>4)&3; cu=(hw>>8)&15; se=(hw>>13)&7; sh=(hw>>12)&1; xcc=(hw>>32)&15
    print(" simd of waves in block 0:",simd[0], "cu",cu[0],"se",se[0],"xcc",xcc[0])
    # find co-resident block of block 0 (same xcc,se,sh,cu) among first 512
    key=(xcc[:,0]*1000+se[:,0]*100+sh[:,0]*20+cu[:,0])
    for b in range(0,3):
        mates=[m for m in np.where(key[:512]==key[b])[0] ]
        print(" block",b,"CU mates (first round):",mates)
        for m in mates:
            print("   blk %d wave0: A-B stamps rel:"%m, [(int(tr[m,0,k,0]-tr[mates[0],0,4,1]),int(tr[m,0,k,1]-tr[mates[0],0,4,1])) for k in range(4,10)])

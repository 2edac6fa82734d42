import numpy as np
lh=np.load('/tmp/kb_sim_lh.npy'); N,R,Wt=lh.shape; K=8; C=16; nch=N//C
for D in (1,2,4,8,16):
    top=np.full((R,Wt,K),-np.inf,np.float32)
    crit=0; meanw=0; items=0; maxq=0
    q=np.zeros((R,Wt),int); pend=[]
    tail=top[:,:,K-1].copy()
    for ch in range(nch):
        blk=lh[ch*C:(ch+1)*C]
        m=(blk>tail[None]).sum(0); q+=m; items+=m.sum()
        pend.append(blk)
        if (ch+1)%D==0 or ch==nch-1:
            r=q.max(1); crit+=r.max(); meanw+=r.mean(); maxq=max(maxq,q.max()); q[:]=0
            allv=np.concatenate([top]+[np.moveaxis(b,0,2) for b in pend],axis=2); top=-np.sort(-allv,axis=2)[:,:,:K]
            tail=top[:,:,K-1].copy(); pend=[]
    print(f"D={D}: critical rounds {crit}, mean-wave rounds {meanw:.0f}, items/lane {items/(R*Wt):.1f}, max queue {maxq}")

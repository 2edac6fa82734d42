import numpy as np
from scipy.ndimage import gaussian_filter
rng = np.random.default_rng(1)
T=64; R=16; Wt=64
vel = np.linspace(5,40,32,endpoint=False); ang=np.linspace(0,1.5,32,endpoint=False)
import sys
sys.path.insert(0,'/root/repo')
from kbmod_amd import fake_data as fd
try:
    vx,vy = fd.kbmod_v1_candidates(32,5.0,40.0,32,0.0,1.5)
except Exception as e:
    print(e); raise
vx=np.asarray(vx); vy=np.asarray(vy)
print(vx[:5],vy[:5], vx[30:36], vy[30:36])
times=np.arange(T)/T
H=R+45; W=Wt+45
sci = rng.normal(0,2,(T,H,W)).astype(np.float32)
psi = np.stack([gaussian_filter(s,1.0,mode='constant') for s in sci])/4.0
phi_c = 0.0795/4  # irrelevant scale
N=len(vx)
lh=np.zeros((N,R,Wt),np.float32)
for c in range(N):
    s=np.zeros((R,Wt),np.float32)
    for t in range(T):
        dx=int(np.floor(vx[c]*times[t]+0.5)); dy=int(np.floor(vy[c]*times[t]+0.5))
        s+=psi[t,dy:dy+R,dx:dx+Wt]
    lh[c]=s
lh/= lh.std()
K=8; C=16
top=np.full((R,Wt,K),-np.inf,np.float32)
nch=N//C
rounds=np.zeros((nch,R),int); items=np.zeros((nch,R),int)
for ch in range(nch):
    tail=np.maximum(top[:,:,K-1],0)  # min_lh = 0 floor? (flag 1024 not default) use -inf
    tail=top[:,:,K-1]
    blk=lh[ch*C:(ch+1)*C]   # C,R,Wt
    m=(blk>tail[None]).sum(0)  # passing screen per lane
    rounds[ch]=m.max(1); items[ch]=m.sum(1)
    # update lists
    allv=np.concatenate([top, np.moveaxis(blk,0,2)],axis=2)
    top=-np.sort(-allv,axis=2)[:,:,:K]
print("mean rounds per wave-chunk", rounds.mean(), "items per wave-chunk", items.mean())
print("critical (max over 16 waves) per chunk mean", rounds.max(1).mean(), " mean of mean", rounds.mean(1).mean())
print("first 8 chunks rounds mean", rounds[:8].mean(), "crit", rounds[:8].max(1).mean())
print("last 48 chunks rounds mean", rounds[16:].mean(), "crit", rounds[16:].max(1).mean())
np.save('/tmp/kb_sim_rounds.npy',rounds); np.save('/tmp/kb_sim_lh.npy',lh)

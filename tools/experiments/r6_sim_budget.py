import numpy as np, bisect
lh=np.load('/tmp/kb_sim_lh.npy'); N,R,Wt=lh.shape; K=8; C=16; nch=N//C
def run(B, first_free=0):
    crit=0; meanw=0; maxq=0; enq=0
    # per lane state
    lists=[[[] for _ in range(Wt)] for _ in range(R)]   # sorted ascending lists of up to K
    queues=[[[] for _ in range(Wt)] for _ in range(R)]
    for ch in range(nch+200):
        rw=np.zeros(R,int)
        for w in range(R):
            for l in range(Wt):
                L=lists[w][l]; tail = L[0] if len(L)==K else -np.inf
                if ch<nch:
                    v=lh[ch*C:(ch+1)*C,w,l]
                    for x in v[v>tail]: queues[w][l].append(x); enq+=1
            need=max(len(q) for q in queues[w])
            bb = B if ch>=first_free else 10**9
            r=min(bb,need); rw[w]=r
            maxq=max(maxq,need)
            for l in range(Wt):
                q=queues[w][l]; L=lists[w][l]
                for x in q[:r]:
                    if len(L)<K: bisect.insort(L,x)
                    elif x>L[0]: L.pop(0); bisect.insort(L,x)
                del q[:r]
        crit+=rw.max(); meanw+=rw.mean()
        if ch>=nch and all(len(q)==0 for w in range(R) for q in queues[w]): break
    print(f"B={B} free={first_free}: critical {crit}, mean-wave {meanw:.0f}, enq/lane {enq/(R*Wt):.1f}, maxq {maxq}")
for B in (1,2,3,4,6,100):
    run(B)
run(2,4); run(3,4); run(2,8)

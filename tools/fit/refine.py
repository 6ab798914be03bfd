import numpy as np, math, sys, random
import picks
from model import *
import os
PROF=os.path.join(os.path.dirname(os.path.abspath(__file__)),"..","..","profiles")
data=load([os.path.join(PROF,"r03_gemm_tune_sd_sdxl_graph.txt"),os.path.join(PROF,"r03_gemm_tune_flux_t5_graph.txt")])
shapes=list(data.items())
cfgs=picks.cfgs
def evaluate(p):
    tot=0; worst=0; n=0
    for (name,M,N,K),row in shapes:
        best=min(row[c] for c in cfgs)
        nkt=K//64
        pk=min((picks.cost_new(c,picks.ntiles(c,[M],1,N),nkt,1,p),c) for c in cfgs)[1]
        r=row[pk]/best; tot+=r; worst=max(worst,r); n+=1
    # Flux in-situ picks that must stay
    pen=0
    want={("qkv",(256,1024),1,9216,3072):(51,1),("mlp0",(256,1024),1,12288,3072):(49,1),("mlp2",(256,1024),1,3072,12288):(51,3),
          ("lin1",(1280,),1,21504,3072):(50,1),("lin2",(1280,),1,3072,15360):(51,3),("proj",(256,1024),1,3072,3072):(47,1)}
    for (nm,g,nb,N,K),(c,S) in want.items():
        o=picks.pick(lambda cc,t,nkt,SS:picks.cost_new(cc,t,nkt,SS,p),list(g),nb,N,K)
        if (o[1],o[2])!=(c,S): pen+=1
    return tot/n+0.02*worst+pen, tot/n, worst, pen
p=np.append(np.load("p4.npy"),[9.0,4.0]); best=evaluate(p); print(best)
random.seed(1)
for it in range(6000):
    q=p.copy()
    k=random.randrange(len(q))
    q[k]*=math.exp(random.gauss(0,0.06))
    if k==1: q[k]=min(q[k],1.0)
    if k>=2 and k<len(q)-2 and (k-2)%3==2: q[k]=min(q[k],1.0)
    e=evaluate(q)
    if e[0]<best[0]-1e-9: p,best=q,e
print(best)
print("L=%.2f beta=%.2f"%(p[0],p[1]))
for c in cfgs: print(c,"ts=%.3f tf=%.2f phi=%.2f"%tuple(p[2+3*picks.idx[c]:5+3*picks.idx[c]]))
np.save("p5.npy",p)

import numpy as np, sys, math
from scipy.optimize import least_squares
from model import *
import os
PROF=os.path.join(os.path.dirname(os.path.abspath(__file__)),"..","..","profiles")
data=load([os.path.join(PROF,"r03_gemm_tune_sd_sdxl_graph.txt"),os.path.join(PROF,"r03_gemm_tune_flux_t5_graph.txt")])
cfgs=[c for c in sorted(TILES) if c in CANDS]          # current candidates only
BPC={c:(2 if c in (4,7,8,9) else 1) for c in cfgs}
idx={c:i for i,c in enumerate(cfgs)}
NP=3
def predict(p,c,M,N,K,S=1):
    L=p[0]; beta=p[1]
    ts,tf,ph=p[2+NP*idx[c]:5+NP*idx[c]]
    t=tiles(c,M,N)*S; slots=256*BPC[c]; x=t/slots; nkt=math.ceil(K/64/S)
    R = 1.0 if x<=1 else (1-beta)*x+beta*math.ceil(x)
    g=ph+(1-ph)*min(1.0,x)
    return L+R*g*(nkt*ts+tf)
shapes=list(data.items())
def resid(p):
    out=[]
    for (name,M,N,K),row in shapes:
        e=[math.log(predict(p,c,M,N,K)/row[c]) for c in cfgs]
        m=sum(e)/len(e)
        out+= [x-m for x in e]+[0.35*x for x in e]
    return out
p0=[6.0,0.5]
for c in cfgs: p0+= [CANDS[c][1],max(CANDS[c][2]-6,0.5),0.7]
lo=[0,0]+[0.05,0.0,0.2]*len(cfgs); hi=[15,1]+[3,40,1.0]*len(cfgs)
r=least_squares(resid,p0,bounds=(lo,hi),loss='soft_l1',f_scale=0.05)
p=r.x
print("L=%.2f beta=%.2f"%(p[0],p[1]))
for c in cfgs: print(c,TILES[c],"ts=%.3f tf=%.2f phi=%.2f"%tuple(p[2+NP*idx[c]:5+NP*idx[c]]))
res=np.array(resid(p)); print("rms %.3f"%res.std())
def regrets(pred):
    out=[]
    for (name,M,N,K),row in shapes:
        best=min((row[c],c) for c in cfgs)
        pick=min((pred(c,M,N,K),c) for c in cfgs)[1]
        out.append((row[pick]/best[0],name,pick,best[1],row[pick],best[0]))
    return sorted(out,reverse=True)
w=regrets(lambda c,M,N,K:predict(p,c,M,N,K))
for x in w[:12]: print("regret %.2f %s pick c%d best c%d  %.1f vs %.1f"%x)
print("mean regret new %.3f"%np.mean([x[0] for x in w]))
w0=regrets(lambda c,M,N,K:cur_model(c,M,N,K))
print("mean regret current %.3f; worst %.2f"%(np.mean([x[0] for x in w0]),w0[0][0]))
np.save("p4.npy",p)

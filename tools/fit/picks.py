import numpy as np, math, sys
from model import *
P=np.load(sys.argv[1]) if len(sys.argv)>1 else None
cfgs=[c for c in sorted(TILES) if c in CANDS]; idx={c:i for i,c in enumerate(cfgs)}
ORDER=[49,50,51,54,46,55,47,7,8,9,4]
BPC={c:CANDS[c][0] for c in cfgs}
MI={49:(4,8),51:(4,6)}
def ntiles(c,groups,nb,N):
    bm,bn=TILES[c]; return sum((m+bm-1)//bm for m in groups)*nb*((N+bn-1)//bn)
def rs_ok(c,S,t): return c in MI and (MI[c][0]%S==0 or MI[c][1]%S==0) and t*S<=256
def cost_old(c,t,nkt,S):
    bpc,ts,tf=CANDS[c]; slots=256*bpc
    rounds=(t*S+slots-1)//slots
    hop=0 if S==1 else (9+(S-2)*1 if rs_ok(c,S,t) else 20+(S-2)*8)
    return rounds*(math.ceil(nkt/S)*ts+tf)+hop
def cost_new(c,t,nkt,S,p=P):
    L,beta=p[0],p[1]; ts,tf,ph=p[2+3*idx[c]:5+3*idx[c]]
    slots=256*BPC[c]; x=t*S/slots
    R=1.0 if x<=1 else (1-beta)*x+beta*math.ceil(x)
    g=ph+(1-ph)*min(1.0,x)
    hop=0 if S==1 else (p[-2]+(S-2)*p[-1] if rs_ok(c,S,t) else 20+(S-2)*8)
    return L+R*g*(math.ceil(nkt/S)*ts+tf)+hop
def pick(cost,groups,nb,N,K):
    best=(1e30,0,1); nkt=K//64
    for c in ORDER:
        t=ntiles(c,groups,nb,N); slots=256*BPC[c]
        for S in (1,2,3,4):
            if S>1 and (BPC[c]!=1 or t*S>slots or nkt//S<16): break
            v=cost(c,t,nkt,S)
            if v<best[0]: best=(v,c,S)
    return best
if __name__=="__main__":
    cases=[]
    for tag,B,txt,img in (("C2",1,256,1024),("C3",1,512,4096),("C5",4,256,4096)):
        T=txt+img
        cases+=[(f"{tag} qkv",[txt,img],B,9216,3072),(f"{tag} proj",[txt,img],B,3072,3072),(f"{tag} mlp0",[txt,img],B,12288,3072),
                (f"{tag} mlp2",[txt,img],B,3072,12288),(f"{tag} lin1",[T],B,21504,3072),(f"{tag} lin2",[T],B,3072,15360),(f"{tag} final",[img],B,64,3072)]
    for name,g,nb,N,K in cases:
        o=pick(cost_old,g,nb,N,K); n=pick(cost_new,g,nb,N,K)
        print(f"{name:10s} old c{o[1]}s{o[2]} ({o[0]:.1f})   new c{n[1]}s{n[2]} ({n[0]:.1f})")

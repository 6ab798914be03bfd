import re, math, os, numpy as np, sys, random
PROF = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'profiles')
from scipy.optimize import least_squares
TILES={4:(64,64),7:(128,128),8:(128,64),9:(64,128),10:(256,128),46:(256,128),47:(128,128),49:(256,256),50:(256,224),51:(256,192),52:(256,128),53:(128,128),54:(256,160),55:(128,256),15:(256,256)}
BPC={c:(2 if c in (4,7,8,9) else 1) for c in TILES}
OLD={49:(1,1.85,6.0),10:(1,1.11,4.8),55:(1,0.98,8.2),7:(2,1.155,4.0),8:(2,0.847,0.30),9:(2,0.672,2.90),4:(2,0.483,1.15)}
rows=[]
shapes={}
for b in (1,2,4,16):
    for l in open(os.path.join(PROF, f"r03_conv_tune_unet_b{b}.txt")):
        m=re.match(r"(\d+)(u?)(s2)?:(\d+)->(\d+)k(\d)(\+res)?\s",l)
        if not m: continue
        H,ups,s2,Cin,Cout,ks=int(m.group(1)),bool(m.group(2)),bool(m.group(3)),int(m.group(4)),int(m.group(5)),int(m.group(6))
        Ho=H*2 if ups else (H//2 if s2 else H)
        M=b*Ho*Ho; K=ks*ks*Cin
        row={int(c):float(v) for c,v in re.findall(r"c(\d+)=([\d.]+)",l.split("BEST")[0])}
        shapes[(b,l.split()[0])]=(M,Cout,K,ups,row)
def tiles(c,M,N):
    bm,bn=TILES[c]; return ((M+bm-1)//bm)*((N+bn-1)//bn)
def old(c,M,N,K):
    bpc,ts,tf=OLD[c]; t=tiles(c,M,N); slots=256*bpc
    return ((t+slots-1)//slots)*((K//64)*ts+tf)
CF=[49,10,55,7,8,9,4,54,51,50]
idx={c:i for i,c in enumerate(CF)}
def new(p,c,M,N,K):
    L,beta=p[0],p[1]; ts,tf,ph=p[2+3*idx[c]:5+3*idx[c]]
    x=tiles(c,M,N)/(256*BPC[c]); R=1.0 if x<=1 else (1-beta)*x+beta*math.ceil(x)
    g=ph+(1-ph)*min(1.0,x)
    return L+R*g*((K//64)*ts+tf)
items=[(k,v) for k,v in shapes.items() if not v[3]]      # the fused-upsample loader is picked separately
def resid(p):
    out=[]
    for k,(M,N,K,ups,row) in items:
        e=[math.log(new(p,c,M,N,K)/row[c]) for c in CF]
        m=sum(e)/len(e); out+=[x-m for x in e]+[0.35*x for x in e]
    return out
p0=[2.0,0.5]
for c in CF: p0+=[OLD.get(c,(1,1.3,10))[1],OLD.get(c,(1,1.3,10))[2],0.7]
lo=[0,0]+[0.05,0,0.2]*len(CF); hi=[15,1]+[4,60,1]*len(CF)
p=least_squares(resid,p0,bounds=(lo,hi),loss='soft_l1',f_scale=0.05).x
def regret(pred,cands):
    out=[]
    for k,(M,N,K,ups,row) in items:
        best=min(row[c] for c in CF)
        pk=min((pred(c,M,N,K),c) for c in cands)[1]
        out.append((row[pk]/best,k,pk,row[pk],best))
    return sorted(out,reverse=True)
def ev(p):
    w=regret(lambda c,M,N,K:new(p,c,M,N,K),CF); return np.mean([x[0] for x in w])+0.02*w[0][0]
best=ev(p); random.seed(0)
for it in range(4000):
    q=p.copy(); k=random.randrange(len(q)); q[k]*=math.exp(random.gauss(0,0.06))
    if k==1 or (k>=2 and (k-2)%3==2): q[k]=min(q[k],1.0)
    e=ev(q)
    if e<best-1e-9: p,best=q,e
print("L=%.2f beta=%.2f"%(p[0],p[1]))
for c in CF: print("    {%d, %d, %.3ff, %.2ff, %.2ff},"%(c,BPC[c],*p[2+3*idx[c]:5+3*idx[c]]))
w=regret(lambda c,M,N,K:new(p,c,M,N,K),CF)
for x in w[:8]: print("regret %.2f %s pick c%d %.1f best %.1f"%x)
print("mean regret new %.3f"%np.mean([x[0] for x in w]))
w0=regret(old,list(OLD)); print("mean regret old(S=1 only) %.3f worst %.2f"%(np.mean([x[0] for x in w0]),w0[0][0]))
# c0 (actual picker incl. split-K) vs best
r0=[v[4][0]/min(v[4][c] for c in CF) for k,v in items]; print("actual old picker (c0, incl split-K) mean ratio to best non-split %.3f"%np.mean(r0))

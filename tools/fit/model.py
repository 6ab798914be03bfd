import re, math, json, sys
TILES={4:(64,64),7:(128,128),8:(128,64),9:(64,128),46:(256,128),47:(128,128),49:(256,256),50:(256,224),51:(256,192),52:(256,128),53:(128,128),54:(256,160),55:(128,256)}
CANDS={49:(1,1.072,22.1),50:(1,1.010,18.7),51:(1,0.875,17.0),54:(1,0.819,14.3),46:(1,0.748,12.1),55:(1,0.787,10.9),47:(1,0.564,5.71),
       7:(2,1.010,5.5),8:(2,0.920,3.8),9:(2,0.672,2.90),4:(2,0.637,0.15)}
def load(paths):
    data={}
    for f in paths:
        for l in open(f):
            m=re.match(r"(\S+)\s+M=(\d+) N=(\d+) K=(\d+):",l)
            if not m: continue
            name,M,N,K=m.group(1),int(m.group(2)),int(m.group(3)),int(m.group(4))
            row={}
            for c,v in re.findall(r"c(\w+)=([\d.]+)",l.split("BEST")[0]):
                if c in('blaslt',): continue
                row[int(c)]=2.0*M*N*K/float(v)/1e6
            data[(name,M,N,K)]=row
    return data
def tiles(cfg,M,N):
    bm,bn=TILES[cfg]; return ((M+bm-1)//bm)*((N+bn-1)//bn)
def cur_model(cfg,M,N,K,S=1):
    bpc,ts,tf=CANDS[cfg]; t=tiles(cfg,M,N); slots=256*bpc; nkt=K//64
    rounds=(t*S+slots-1)//slots
    return rounds*(((nkt+S-1)//S)*ts+tf)
if __name__=="__main__":
    data=load(sys.argv[1:])
    for (name,M,N,K),row in data.items():
        best=min((v,c) for c,v in row.items() if c!=0)
        pred={c:cur_model(c,M,N,K) for c in CANDS}
        pick=min((v,c) for c,v in pred.items())[1]
        print(f"{name:12s} M{M:6d} N{N:5d} K{K:5d} model-pick(S=1) c{pick:2d} pred {pred[pick]:7.1f} actual {row[pick]:7.1f} | best c{best[1]:2d} {best[0]:7.1f}  regret {row[pick]/best[0]:.2f} | " + " ".join(f"{c}:{pred[c]/row[c]:.2f}" for c in CANDS))

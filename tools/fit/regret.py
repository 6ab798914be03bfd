import json,re,sys
f=sys.argv[1]
for l in open(f):
    m=re.match(r"(\S+)\s+M=(\d+) N=(\d+) K=(\d+):",l)
    if not m: continue
    name,M,N,K=m.group(1),int(m.group(2)),int(m.group(3)),int(m.group(4))
    row={}
    for c,v in re.findall(r"c(\w+)=([\d.]+)",l.split("BEST")[0]):
        row[c]=float(v)
    fl=2.0*M*N*K
    t0=fl/row['0']/1e6; 
    best=max((v,c) for c,v in row.items() if c not in('0','blaslt'))
    tb=fl/best[0]/1e6
    flag = "  <<<" if t0>tb*1.08 else ""
    ts=" ".join(f"{c}:{fl/v/1e6:5.1f}" for c,v in row.items() if c!='blaslt')
    print(f"{name:11s} M{M:6d} N{N:5d} K{K:5d} pick {t0:6.1f}us best c{best[1]:>2s} {tb:6.1f}us{flag:5s} | {ts}")

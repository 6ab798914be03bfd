#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of the replayed launch plan, from a rocprofv3 `--kernel-trace --output-format csv`
kernel_trace.csv: for every libfluxhip kernel, the time between the END of its predecessor on the GPU and its own START,
grouped by (predecessor -> kernel).  What a launch boundary costs inside the hipGraph is otherwise only visible as the
difference between a step's wall time and the sum of its kernels.
usage: python tools/gap_stats.py <kernel_trace.csv> [out.txt]"""
import csv, re, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*$", "", n)
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows), key=lambda e: e[0])
pairs = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    if "at::" in n0 or "at::" in n1 or n0.startswith("__amd") or n1.startswith("__amd"):
        continue
    g = s1 - e0
    if g > 200000:          # host-side pause between replays / images: not a launch boundary
        continue
    pairs[(n0[:44], n1[:44])].append(g)
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
tot = sum(sum(v) for v in pairs.values()); cnt = sum(len(v) for v in pairs.values())
out.write(f"# {sys.argv[1].split('/')[-1]}: {cnt} kernel-to-kernel boundaries, {tot / 1e6:.3f} ms idle in total, mean {tot / max(cnt, 1) / 1e3:.2f} us\n")
out.write("# predecessor -> kernel : boundaries, mean us, median us, max us, total ms\n")
for (a, b), v in sorted(pairs.items(), key=lambda kv: -sum(kv[1]))[:40]:
    v.sort()
    out.write(f"{a:44s} -> {b:44s} {len(v):5d} {sum(v) / len(v) / 1e3:7.2f} {v[len(v) // 2] / 1e3:7.2f} {v[-1] / 1e3:8.2f} {sum(v) / 1e6:8.3f}\n")

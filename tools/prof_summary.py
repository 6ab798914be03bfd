#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into a short
table (kernel names truncated, torch init kernels grouped) for profiles/."""
import csv
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
out = []
other = [0, 0.0]
for r in rows:
    name = r["Name"]
    if "at::native" in name or name.startswith("__amd_rocclr"):
        other[0] += int(r["Calls"]); other[1] += float(r["TotalDurationNs"])
        continue
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    out.append((name, int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"])))
tot = sum(o[2] for o in out)
with open(dst, "w") as f:
    f.write(f"# source: {src.split('/')[-1]} (rocprofv3 --kernel-trace --stats); libfluxhip kernels only\n")
    f.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct_of_fluxhip\n")
    for n, c, t, a, mn, mx in sorted(out, key=lambda o: -o[2]):
        f.write(f"\"{n}\",{c},{t/1e6:.3f},{a/1e3:.1f},{mn/1e3:.1f},{mx/1e3:.1f},{100*t/tot:.1f}\n")
    f.write(f"# torch/runtime kernels outside the hot path (weight init, input copies): calls={other[0]} total_ms={other[1]/1e6:.3f}\n")
print(open(dst).read())

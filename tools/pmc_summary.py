#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected in
separate runs as the MI355X guide prescribes).  Units: the counters are in KiB; on gfx950 FETCH_SIZE
reports 1/2 of a wide coalesced read stream, so the read side is doubled (MI355X_MICROARCH.md §HBM);
WRITE_SIZE is uncalibrated and taken as is."""
import csv
import re
import sys
from collections import defaultdict

fetch_csv, write_csv, dst = sys.argv[1:4]


def load(path):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "at::native" in n or n.startswith("__amd_rocclr"):
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n))
        n = re.sub(r"\(.*$", "", n)
        agg[n][0] += 1
        agg[n][1] += float(r["Counter_Value"])
    return agg


F, W = load(fetch_csv), load(write_csv)
with open(dst, "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --profile-only --no-graph\n")
    f.write("# workload: " + (sys.argv[4] if len(sys.argv) > 4 else "flux-schnell B1 T1280") + "\n")
    f.write("# hbm_read_MB = FETCH_SIZE(KiB) * 2 * 1024 / 1e6 (gfx950 x2 correction); hbm_write_MB = WRITE_SIZE(KiB) * 1024 / 1e6\n")
    f.write("kernel,launches,avg_hbm_read_MB,avg_hbm_write_MB,avg_total_MB\n")
    for n in sorted(F, key=lambda k: -F[k][1]):
        c, fs = F[n]
        ws = W.get(n, [c, 0.0])[1] / max(W.get(n, [c, 0.0])[0], 1)
        rd = fs / c * 2 * 1024 / 1e6
        wr = ws * 1024 / 1e6
        f.write(f"\"{n}\",{c},{rd:.2f},{wr:.2f},{rd + wr:.2f}\n")
print(open(dst).read())

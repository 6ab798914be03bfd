#!/usr/bin/env python3
"""Aggregate SQ counters per GEMM kernel from a rocprofv3 --pmc counter_collection.csv."""
import csv, re, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gemm_nt_kernel" not in n and "attn_kernel" not in n:
        continue
    n = re.sub(r"\(.*$", "", re.sub(r"^void ", "", n)) + f" grid={r['Grid_Size']}"
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[n][r["Counter_Name"]] += 1
for n in agg:
    a = {k: v / cnt[n][k] for k, v in agg[n].items()}
    print(n, "launches", max(cnt[n].values()))
    for k, v in sorted(a.items()):
        print(f"   {k:28s} {v:16.0f}")
    if "SQ_WAVE_CYCLES" in a:
        wc = a["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
            if k in a:
                print(f"   {k}/WAVE_CYCLES = {a[k] / wc:.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CYCLES" in a:
        print(f"   MFMA_BUSY/BUSY_CYCLES = {a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['SQ_BUSY_CYCLES']:.3f}")

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
# optional 5th argument (round 6): the plan's algorithmic bytes per launch (`bench.py --profile-only --dump-plan-bytes PATH`) ->
# columns algorithmic_MB and ratio = avg_total_MB / algorithmic_MB for the GEMM kernels of the Flux plan
plan_bytes = None
if len(sys.argv) > 5 and sys.argv[5]:
    import json
    plan_bytes = json.load(open(sys.argv[5]))["labels"]


def algorithmic_mb(kernel, launches, forwards):
    """Match a rocprof kernel row to the plan label(s) with the same tile shape and per-forward launch count."""
    if not plan_bytes or not forwards:
        return None
    t = re.match(r"gemm_nt_kernel<(\d+), (\d+), \d+, \d+, 0,", kernel)
    if not t or launches % forwards:
        return None
    per_fwd = launches // forwards
    hits = [(k, v) for k, v in plan_bytes.items() if v.get("bm") == int(t.group(1)) and v.get("bn") == int(t.group(2))
            and v.get("algorithmic_MB_per_launch")]
    exact = [(k, v) for k, v in hits if v["launches_per_forward"] == per_fwd]
    if len(exact) == 1:
        return exact[0][1]["algorithmic_MB_per_launch"]
    if not exact and hits and sum(v["launches_per_forward"] for _, v in hits) == per_fwd:      # one kernel serving several labels
        return sum(v["launches_per_forward"] * v["algorithmic_MB_per_launch"] for _, v in hits) / per_fwd
    return None


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
    f.write("# algorithmic_MB = every operand of the launch once (A, one weight panel per group, output, residual / gate / bias), from\n")
    f.write("#   bench.py --dump-plan-bytes; ratio = avg_total_MB / algorithmic_MB (empty: not a GEMM of the Flux plan)\n")
    f.write("kernel,launches,avg_hbm_read_MB,avg_hbm_write_MB,avg_total_MB,algorithmic_MB,ratio\n")
    forwards = 0
    if plan_bytes:
        qk = [c for n, (c, _) in F.items() if n.startswith("qk_norm_rope_vt_kernel")]
        per = plan_bytes.get("fluxhip_qk_norm_rope_bf16", {}).get("launches_per_forward", 0)
        forwards = (sum(qk) // per) if (qk and per) else 0
    for n in sorted(F, key=lambda k: -F[k][1]):
        c, fs = F[n]
        ws = W.get(n, [c, 0.0])[1] / max(W.get(n, [c, 0.0])[0], 1)
        rd = fs / c * 2 * 1024 / 1e6
        wr = ws * 1024 / 1e6
        alg = algorithmic_mb(n, c, forwards)
        f.write(f"\"{n}\",{c},{rd:.2f},{wr:.2f},{rd + wr:.2f}," + (f"{alg:.2f},{(rd + wr) / alg:.2f}" if alg else ",") + "\n")
print(open(dst).read())

#!/usr/bin/env python3
"""Static scan of the library's gfx950 assembly for two latency patterns that the memory-bound kernels of this library
turned out to contain (round 5, norm.hip):
  (1) a hidden kernel argument (blockDim / gridDim) fetched through the VECTOR memory path (`global_load_u* v, v, s[0:1]`)
      and waited for before the first real load - one extra memory round trip per wave;
  (2) `if (in_range) x = load(...)` compiled as s_cbranch_execz + global_load + s_waitcnt vmcnt(0): every such load is a
      dependent round trip of its own instead of one of several requests in flight.
usage: python tools/isa_scan.py [file.hip ...]   (default: every translation unit of flux_generator_amd/csrc)"""
import os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "flux_generator_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Xclang -target-feature -Xclang -packed-fp32-ops -S --cuda-device-only".split()
srcs = sys.argv[1:] or [os.path.join(C, f) for f in sorted(os.listdir(C)) if f.endswith(".hip")]
for src in srcs:
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src, "-o", tmp.name], check=True, stderr=subprocess.DEVNULL)
        lines = open(tmp.name).read().split("\n")
    name, body = None, []
    kernels = {}
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            kernels[name] = body
        elif name and ln.startswith("\t") and not ln.startswith("\t."):
            body.append(ln.strip())
    for k, b in kernels.items():
        if not any("s_endpgm" in i for i in b):
            continue
        hidden = [i for i in b if re.match(r"global_load_u(byte|short)\s+v\d+, v\d+, s\[0:1\]", i)]
        cond = 0
        for n, i in enumerate(b):
            if i.startswith(("global_load", "buffer_load")) and "lds" not in i:
                after = b[n + 1:n + 8]
                before = b[max(0, n - 14):n]
                w = next((a for a in after if a.startswith("s_waitcnt") and "vmcnt(0)" in a), None)
                nxt_load = any(a.startswith(("global_load", "buffer_load")) for a in after[:after.index(w)] if w) if w else False
                if w and not nxt_load and any(x.startswith("s_cbranch_exec") for x in before):
                    cond += 1
        if hidden or cond:
            d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print(f"{os.path.basename(src):18s} hidden-arg VMEM loads {len(hidden)}  guarded load+wait {cond:2d}  {d[:110]}")

#!/usr/bin/env python3
"""Time the SDXL / SD UNet conv shapes per forced implicit-GEMM tile configuration (bf16, FLUXHIP_CONV_CFG is read once per
process: the script re-executes itself per configuration).  Launches are replayed from a hipGraph.  TUNE_B = batch."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(os.environ.get("TUNE_B", "16"))
# (H, Cin, Cout, ks, stride, ups, res)
SHAPES = [(64, 320, 320, 3, 1, 0, 1), (64, 640, 320, 3, 1, 0, 0), (64, 960, 320, 3, 1, 0, 0), (64, 320, 320, 3, 2, 0, 0),
          (32, 640, 640, 3, 1, 0, 1), (32, 320, 640, 3, 1, 0, 0), (32, 1280, 640, 3, 1, 0, 0), (32, 1920, 640, 3, 1, 0, 0),
          (32, 960, 640, 3, 1, 0, 0), (32, 640, 640, 3, 2, 0, 0), (32, 640, 640, 3, 1, 1, 0),
          (16, 1280, 1280, 3, 1, 0, 1), (16, 640, 1280, 3, 1, 0, 0), (16, 2560, 1280, 3, 1, 0, 0), (16, 1920, 1280, 3, 1, 0, 0),
          (16, 1280, 1280, 3, 1, 1, 0), (64, 640, 320, 1, 1, 0, 0), (16, 2560, 1280, 1, 1, 0, 0), (32, 1920, 640, 1, 1, 0, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from flux_generator_amd import ops
    torch.manual_seed(0)
    out = {}
    for (H, Cin, Cout, ks, st, ups, res) in SHAPES:
        x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
        w = (torch.randn(Cout, ks, ks, Cin, device="cuda") * (ks * ks * Cin) ** -0.5).to(torch.bfloat16)
        if ks == 1:
            w = w.view(Cout, Cin)
        b = torch.randn(Cout, device="cuda").to(torch.bfloat16)
        Ho = H * 2 if ups else H // st
        r = torch.randn(B, Ho, Ho, Cout, device="cuda").to(torch.bfloat16) if res else None
        key = f"{H}{'u' if ups else ''}{'s2' if st == 2 else ''}:{Cin}->{Cout}k{ks}{'+res' if res else ''}"
        try:
            kw = dict(stride=st, ups=bool(ups), res=r)
            for _ in range(3):
                y = ops.conv2d(x, w, b, **kw)
            n_it = 10
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n_it):
                    ops.conv2d(x, w, b, out=y, **kw)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n_it * 1e3
            out[key] = (round(us, 1), float(y.float().abs().mean()))
        except Exception as ex:
            out[key] = ("ERR", 0)
    print("RESULT " + json.dumps(out))
else:
    cfgs = sys.argv[1:] or ["0"]
    rows = {}
    for c in cfgs:
        env = dict(os.environ); env["FLUXHIP_CONV_CFG"] = c
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        rows[c] = json.loads(line[0][7:]) if line else {"fail": (r.stderr[-300:], 0)}
    keys = list(next(iter(rows.values())).keys())
    for k in keys:
        vals = {c: rows[c].get(k, ["-"])[0] for c in cfgs}
        best = min((v, c) for c, v in vals.items() if isinstance(v, (int, float)) and c != "0")
        print(f"{k:24s} " + " ".join(f"c{c}={vals[c]}" for c in cfgs) + f"  BEST c{best[1]}={best[0]}")
    ref = rows[cfgs[0]]
    for c in cfgs[1:]:
        bad = [k for k in keys if k in rows[c] and isinstance(rows[c][k][0], float) and abs(rows[c][k][1] - ref[k][1]) > 0.02 * abs(ref[k][1])]
        if bad: print("MISMATCH", c, bad)

#!/usr/bin/env python3
"""Time the fp32-faithful (bf16x3) implicit-GEMM conv on the VAE decoder shapes per forced tile | split << 8
(FLUXHIP_CONV_X3_CFG is read once per process: the script re-executes itself per configuration).  Launches are replayed
from a hipGraph.  TUNE_B = batch, TUNE_SCALE = 1 (512^2 image) or 2 (1024^2)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(os.environ.get("TUNE_B", "1")); SC = int(os.environ.get("TUNE_SCALE", "1"))
SHAPES = [(64 * SC, 512, 512), (128 * SC, 512, 512), (256 * SC, 512, 256), (256 * SC, 256, 256), (512 * SC, 256, 128), (512 * SC, 128, 128)]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from flux_generator_amd import ops
    torch.manual_seed(0)
    out = {}
    for (H, Cin, Cout) in SHAPES:
        x = ops.split_f32(torch.randn(B, H, H, Cin, device="cuda"))
        w = ops.split_f32(torch.randn(Cout, 3, 3, Cin, device="cuda") * (9 * Cin) ** -0.5)
        b = torch.randn(Cout, device="cuda")
        key = f"{H}:{Cin}->{Cout}"
        try:
            for _ in range(2):
                y = ops.conv2d_x3(x, w, b)
            n_it = 6
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n_it):
                    ops.conv2d_x3(x, w, b, out=y)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            out[key] = (round(e0.elapsed_time(e1) / n_it * 1e3, 1), float(ops.join_f32(y).abs().mean()))
        except Exception as ex:
            out[key] = ("ERR", 0)
    print("RESULT " + json.dumps(out))
else:
    cfgs = sys.argv[1:] or ["0"]
    rows = {}
    for c in cfgs:
        code = int(c.split("s")[0]) | (int(c.split("s")[1]) << 8) if "s" in c else int(c)
        env = dict(os.environ); env["FLUXHIP_CONV_X3_CFG"] = str(code)
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        rows[c] = json.loads(line[0][7:]) if line else {}
    keys = list(next(iter(rows.values())).keys())
    for k in keys:
        vals = {c: rows[c].get(k, ["-"])[0] for c in cfgs}
        ok = [(v, c) for c, v in vals.items() if isinstance(v, (int, float)) and c != "0"]
        best = min(ok) if ok else ("-", "-")
        print(f"B{B} {k:16s} " + " ".join(f"c{c}={vals[c]}" for c in cfgs) + f"  BEST c{best[1]}={best[0]}")
    ref = rows[cfgs[0]]
    for c in cfgs[1:]:
        bad = [k for k in keys if k in rows[c] and isinstance(rows[c][k][0], float) and k in ref and isinstance(ref[k][0], float) and abs(rows[c][k][1] - ref[k][1]) > 0.02 * abs(ref[k][1])]
        if bad: print("MISMATCH", c, bad)

#!/usr/bin/env python3
"""Time the VAE decoder's conv shapes per forced tile configuration (FLUXHIP_CONV_CFG is read once per
process, so this script re-executes itself per configuration).  TUNE_X3=1: the fp32-faithful split-bf16 conv
(FLUXHIP_CONV_X3_CFG; reported TFLOP/s count the 3 MFMA passes)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(512, 128, 128, False), (512, 256, 128, False), (64, 512, 512, False), (64, 512, 512, True), (128, 512, 512, False), (128, 512, 512, True), (256, 512, 256, False),
          (256, 256, 256, False), (256, 256, 256, True), (512, 256, 128, False), (512, 128, 128, False)]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from flux_generator_amd import ops
    torch.manual_seed(0)
    out = {}
    for (H, Cin, Cout, ups) in SHAPES:
        X3 = bool(os.environ.get("TUNE_X3"))
        x = torch.randn(1, H, H, Cin, device="cuda")
        w = torch.randn(Cout, 3, 3, Cin, device="cuda") * (9 * Cin) ** -0.5
        b = torch.randn(Cout, device="cuda")
        if X3:
            x, w = ops.split_f32(x), ops.split_f32(w)
            conv = ops.conv2d_x3
        else:
            x, w, b = x.to(torch.bfloat16), w.to(torch.bfloat16), b.to(torch.bfloat16)
            conv = ops.conv2d
        try:
            for _ in range(3):
                y = conv(x, w, b, ups=ups)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                conv(x, w, b, ups=ups, out=y)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            Ho = H * 2 if ups else H
            out[f"{H}{'u' if ups else ''}:{Cin}->{Cout}"] = (round((3 if X3 else 1) * 2.0 * Ho * Ho * Cout * Cin * 9 / ms / 1e9), float(y.float().abs().mean()))
            if X3 and ups:       # the sub-pixel form of the same op (4/9 of the MFMA work)
                w4 = ops.split_f32(ops.subpixel_weights(ops.join_f32(w)))
                for _ in range(3):
                    y = ops.conv_up2x_x3(x, w4, b)
                e0.record()
                for _ in range(10):
                    ops.conv_up2x_x3(x, w4, b, out=y)
                e1.record(); torch.cuda.synchronize()
                ms2 = e0.elapsed_time(e1) / 10
                # quoted on the 9-tap flops of the op it replaces, so the row compares directly with the `u` row above
                out[f"{H}u-subpixel:{Cin}->{Cout}"] = (round(3 * 2.0 * Ho * Ho * Cout * Cin * 9 / ms2 / 1e9), float(ops.join_f32(y).abs().mean()))
        except Exception as ex:
            out[f"{H}:{Cin}->{Cout}"] = ("ERR", 0)
    print("RESULT " + json.dumps(out))
else:
    cfgs = sys.argv[1:] or ["0"]
    rows = {}
    for c in cfgs:
        env = dict(os.environ); env["FLUXHIP_CONV_CFG"] = c; env["FLUXHIP_CONV_X3_CFG"] = c
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        rows[c] = json.loads(line[0][7:]) if line else {"fail": (r.stderr[-300:], 0)}
    keys = list(next(iter(rows.values())).keys())
    for k in keys:
        print(f"{k:16s} " + " ".join(f"c{c}={rows[c].get(k, ['-'])[0]}" for c in cfgs))
    ref = rows[cfgs[0]]
    for c in cfgs[1:]:
        bad = [k for k in keys if k in rows[c] and isinstance(rows[c][k][0], int) and abs(rows[c][k][1] - ref[k][1]) > 0.02 * abs(ref[k][1])]
        if bad: print("MISMATCH", c, bad)

#!/usr/bin/env python3
"""A/B of the persistent tile loop on multi-round GEMM shapes (GPU box): FLUXHIP_PERSIST is read when the library loads, so this
script re-executes itself per setting.  Shapes: linear1 at batch 1 (2 rounds) and the Flux block GEMMs at M = 4608 (Flux-dev
1024 x 1024) and M = 17408 (batch 4 at 1024 x 1024), incl. fp8."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(1280, 21504, 3072), (4608, 9216, 3072), (4608, 12288, 3072), (4608, 21504, 3072), (4608, 3072, 12288), (4608, 3072, 15360),
          (17408, 9216, 3072), (17408, 21504, 3072), (17408, 3072, 15360)]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from flux_generator_amd import ops
    torch.manual_seed(0)
    out = {}
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16) for _ in range(4)]
        b = torch.randn(N, device="cuda").to(torch.bfloat16)
        y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        for w in ws:
            ops.linear(x, w, b, out=y, epi=ops.EPI_GELU_TANH)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(12):
            ops.linear(x, ws[it & 3], b, out=y, epi=ops.EPI_GELU_TANH)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 12
        ref = torch.nn.functional.gelu(x[:64].float() @ ws[3].float().T + b.float(), approximate="tanh")
        err = float((y[:64].float() - ref).norm() / ref.norm())
        out[f"{M}x{N}x{K}"] = (round(ms * 1e3, 1), round(2.0 * M * N * K / ms / 1e9), round(err, 4))
    print("RESULT " + json.dumps(out))
else:
    rows = {}
    order = ["1", "0", "1", "0"]
    for i, v in enumerate(order):
        env = dict(os.environ); env["FLUXHIP_PERSIST"] = v
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        rows[f"{v}#{i}"] = json.loads(line[0][7:]) if line else {"fail": (r.stderr[-400:], 0, 0)}
    for k in next(iter(rows.values())).keys():
        print(f"{k:20s} " + "  ".join(f"persist={c.split('#')[0]}: {rows[c].get(k, ['-'])[0]} us {rows[c].get(k, ['-', '-'])[1]} TF (err {rows[c].get(k, ['-', '-', '-'])[2]})" for c in rows))

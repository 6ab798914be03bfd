#!/usr/bin/env python3
"""Time the head_dim-128 attention entry point on the Flux shapes (GPU box).  FLUXHIP_ATTN selects the kernel variant
(0 auto, 2 = single wave set even on small grids, 3 = two wave sets even on large grids); it is read once per process, so this script re-executes itself."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(1, 24, 1280), (1, 24, 4352), (1, 24, 4608), (4, 24, 1280), (4, 24, 4352), (1, 24, 200)]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from flux_generator_amd import ops
    torch.manual_seed(0)
    out = {}
    for B, H, T in SHAPES:
        Tp = (T + 63) // 64 * 64
        q = torch.randn(B, H, T, 128, device="cuda").to(torch.bfloat16)
        k = torch.randn(B, H, T, 128, device="cuda").to(torch.bfloat16)
        v = torch.randn(B, H, T, 128, device="cuda").to(torch.bfloat16)
        vt = torch.zeros(B, H, 128, Tp, dtype=torch.bfloat16, device="cuda"); vt[..., :T] = v.transpose(-1, -2)
        vt = vt[..., ops.vt_key_permutation(Tp, "cuda")].contiguous()      # key-permuted V^T layout (include/fluxhip.h)
        o = torch.empty(B, T, H * 128, dtype=torch.bfloat16, device="cuda")
        for _ in range(3):
            ops.attention_d128(q, k, vt, o, H * 128, B, H, T, Tp, 128 ** -0.5)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attention_d128(q, k, vt, o, H * 128, B, H, T, Tp, 128 ** -0.5)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float()).transpose(1, 2).reshape(B, T, -1)
        err = float((o.float() - ref).norm() / ref.norm())
        out[f"B{B} T{T}"] = (round(ms * 1e3, 1), round(4.0 * B * H * T * T * 128 / ms / 1e9), round(err, 5))
    print("RESULT " + json.dumps(out))
else:
    for v in sys.argv[1:] or ["0"]:
        env = dict(os.environ); env["FLUXHIP_ATTN"] = v
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        print("variant", v, json.loads(line[0][7:]) if line else r.stderr[-400:])

#!/usr/bin/env python3
"""Time the head_dim-128 attention entry point on the Flux shapes (GPU box) for several kernel variants
(fluxhip_attention_set_variant: 0 auto, 2 / 3 = attn_kernel with one / two wave sets, 4 / 5 / 6 = the 64-queries-per-wave kernel
with 256 / 128 queries per workgroup / chosen by grid size, + 0x100 = its online-rescale fallback path).
Each variant runs 20 launches from a captured hipGraph (kernel + boundary) and is checked against torch SDPA in float32."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flux_generator_amd import ops, _lib

lib = _lib.load()
SHAPES = [(1, 24, 1280), (1, 24, 4352), (1, 24, 4608), (4, 24, 1280), (4, 24, 4352), (1, 24, 200)]
variants = [int(v, 0) for v in sys.argv[1:]] or [0]
torch.manual_seed(0)
for B, H, T in SHAPES:
    Tp = (T + 63) // 64 * 64
    q = torch.randn(B, H, T, 128, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, H, T, 128, device="cuda").to(torch.bfloat16)
    v = torch.randn(B, H, T, 128, device="cuda").to(torch.bfloat16)
    vt = torch.zeros(B, H, 128, Tp, dtype=torch.bfloat16, device="cuda"); vt[..., :T] = v.transpose(-1, -2)
    vt = vt[..., ops.vt_key_permutation(Tp, "cuda")].contiguous()      # key-permuted V^T layout (include/fluxhip.h)
    o = torch.empty(B, T, H * 128, dtype=torch.bfloat16, device="cuda")
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float()).transpose(1, 2).reshape(B, T, -1)
    row = {}
    for var in variants:
        lib.fluxhip_attention_set_variant(var)
        run = lambda: ops.attention_d128(q, k, vt, o, H * 128, B, H, T, Tp, 128 ** -0.5)      # noqa: E731
        o.zero_()
        run()
        err = float((o.float() - ref).norm() / ref.norm())
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                run()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        row[hex(var)] = (round(ms * 1e3, 1), round(4.0 * B * H * T * T * 128 / ms / 1e9), round(err, 5))
    print(f"B{B} T{T}: " + "  ".join(f"{k_}: {v_[0]} us {v_[1]} TF err {v_[2]}" for k_, v_ in row.items()), flush=True)
lib.fluxhip_attention_set_variant(0)

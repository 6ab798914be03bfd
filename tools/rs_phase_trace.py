#!/usr/bin/env python3
"""Where a split-K launch of the Flux N = 3072 projections spends its time (GPU box): the phase-stamped 256 x 192 ping-pong
tile (cfg 56) with S = 3 on linear2 / mlp2 / attn.proj shapes — setup, main loop, the three segments of the reduce-scatter
hand-off (write-through stores + drain | arrival, poll, acquire | peers' partials loaded and added) and the epilogue, per wave,
in microseconds at the wave's own measured clock.  usage: rs_phase_trace.py [S] [mode]   (mode 1 = chain hand-off)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flux_generator_amd import ops, _lib

lib = _lib.load()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib.fluxhip_gemm_set_splitk_mode(mode)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for name, (M, N, K) in {"linear2": (1280, 3072, 15360), "mlp2": (1280, 3072, 12288), "attn.proj": (1280, 3072, 3072)}.items():
    x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(6)]
    b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    trace = torch.zeros(4096 * 8 * 16, dtype=torch.int64, device=dev)
    code = 56 | (S << 8)
    for i in range(5):
        ops.linear(x, ws[i], b, out=out, epi=ops.EPI_GATE_RES, res=res, tile_cfg=code)
    lib.fluxhip_gemm_set_trace(trace.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.linear(x, ws[5], b, out=out, epi=ops.EPI_GATE_RES, res=res, tile_cfg=code)
    e1.record()
    torch.cuda.synchronize()
    lib.fluxhip_gemm_set_trace(None)
    t = trace.view(-1, 16).cpu().double()
    t = t[t[:, 8] > 0]
    ghz = float((t[:, 8] / t[:, 9]).mean()) * 0.1
    us = lambda c: float(c.mean()) / ghz / 1e3     # noqa: E731
    setup, epi, whole = t[:, 10], t[:, 11], t[:, 8]
    rs = t[:, 0] + t[:, 1] + t[:, 2]
    print(f"{name} {M}x{N}x{K} cfg56 S={S} mode={mode}: launch {e0.elapsed_time(e1) * 1e3:.1f} us; {len(t)} waves at {ghz:.2f} GHz: whole wave "
          f"{us(whole):.1f} us = setup {us(setup):.1f} + main loop {us(whole - setup - epi):.1f} + [stores+drain {us(t[:, 0]):.1f} | "
          f"arrive/poll/acquire {us(t[:, 1]):.1f} (max {float(t[:, 1].max()) / ghz / 1e3:.1f}) | load+add {us(t[:, 2]):.1f}] + rest of epilogue {us(epi - rs):.1f}")
lib.fluxhip_gemm_set_splitk_mode(0)

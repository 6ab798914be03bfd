#!/usr/bin/env python3
"""Per-phase cycle breakdown of the pipelined GEMM main loop (phase-timed tile configs, GPU box only).
usage: gemm_phase_trace.py [48] [M N K]   (48 = the phase-stamped 256x256 PIPE-5 tile)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flux_generator_amd import ops, _lib

lib = _lib.load()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 48
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (1280, 21504, 3072)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
ws = [(torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(6)]
b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
trace = torch.zeros(4096 * 8 * 16, dtype=torch.int64, device=dev)
for i in range(5):
    ops.linear(x, ws[i], b, out=out, tile_cfg=cfg)
lib.fluxhip_gemm_set_trace(trace.data_ptr())
ops.linear(x, ws[5], b, out=out, tile_cfg=cfg)
torch.cuda.synchronize()
lib.fluxhip_gemm_set_trace(None)
t = trace.view(-1, 16).cpu().double()
t = t[t[:, 7] > 0]
per = t[:, :4] / t[:, 7:8]
names = ["cluster 1 (32 MFMA + next reads + W pieces)", "LDS-DMA wait", "barrier", "cluster 2 (32 MFMA + next reads + A pieces) + loop turn"]
mean = per.mean(0)
print(f"cfg {cfg}  M={M} N={N} K={K}: {len(t)} waves, mean cycles per K-step = {float(mean.sum()):.0f} (+ stamp overhead)")
for n, v, lo, hi in zip(names, mean, per.min(0).values, per.max(0).values):
    print(f"  {n:58s} {float(v):7.0f}   [{float(lo):.0f} .. {float(hi):.0f}]")
tot, rt, pro, epi = (float(t[:, i].mean()) for i in (8, 9, 10, 11))
print(f"  whole wave {tot:.0f} cycles = {rt / 100:.1f} us -> shader clock {tot / rt * 0.1:.2f} GHz; setup {pro:.0f}, "
      f"main loop {tot - pro - epi:.0f}, epilogue {epi:.0f} cycles")
print(f"  end-of-loop barrier {float(t[:, 14].mean()):.0f};", end="")
print(f"  epilogue: phase A (incl. barrier) {float(t[:, 12].mean()):.0f}, phase B issue {float(t[:, 13].mean()):.0f}, store drain {float((t[:, 11] - t[:, 12] - t[:, 13]).mean()):.0f} cycles")

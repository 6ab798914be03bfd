#!/usr/bin/env python3
"""Where a NON-split launch of the 256 x 192 ping-pong tile spends its time (GPU box): the phase-stamped generic kernel (cfg 56,
S = 1) on the Flux K = 3072 shapes - address setup, main loop, barrier after the loop (wave skew), epilogue phase A (bias, bf16,
LDS writes) and phase B (LDS reads, fused math, 128-byte stores) - per wave, in microseconds at the wave's own measured clock, with
the spread over the launch's waves (min / mean / max of the whole-wave time: the launch ends with its slowest wave).
usage: gemm_phase_trace2.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flux_generator_amd import ops, _lib

lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
CASES = {"qkv 1280x9216x3072 bias": (1280, 9216, 3072, ops.EPI_BIAS), "mlp0 1280x12288x3072 gelu": (1280, 12288, 3072, ops.EPI_GELU_TANH),
         "linear1-sized 1280x21504x3072 bias (2 rounds of this tile: 560 tiles)": (1280, 21504, 3072, ops.EPI_BIAS),
         "K=12288 1280x9216x12288 bias": (1280, 9216, 12288, ops.EPI_BIAS)}
if os.environ.get("TRACE_SHAPES"):      # "name:M:N:K:epi,..." e.g. the SDXL UNet's short-K, multi-round launches
    CASES = {t.split(":")[0]: tuple(int(v) for v in t.split(":")[1:]) for t in os.environ["TRACE_SHAPES"].split(",")}
for name, (M, N, K, epi) in CASES.items():
    x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(4)]
    b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    trace = torch.zeros(4096 * 8 * 16, dtype=torch.int64, device=dev)
    for i in range(3):
        ops.linear(x, ws[i], b, out=out, epi=epi, tile_cfg=56)
    lib.fluxhip_gemm_set_trace(trace.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.linear(x, ws[3], b, out=out, epi=epi, tile_cfg=56)
    e1.record()
    torch.cuda.synchronize()
    lib.fluxhip_gemm_set_trace(None)
    t = trace.view(-1, 16).cpu().double()
    t = t[t[:, 8] > 0]
    ghz = float((t[:, 8] / t[:, 9]).mean()) * 0.1
    us = lambda c: float(c.mean()) / ghz / 1e3     # noqa: E731
    whole, setup, epi_all, epa, epb, skew = t[:, 8], t[:, 10], t[:, 11], t[:, 12], t[:, 13], t[:, 14]
    wus = t[:, 9] / 100.0                           # whole wave by the 100 MHz real-time clock
    nk = K // 64
    main = whole - setup - epi_all
    print(f"{name}: launch (events) {e0.elapsed_time(e1) * 1e3:.1f} us; {len(t)} waves at {ghz:.2f} GHz; whole wave min / mean / max "
          f"{float(wus.min()):.1f} / {float(wus.mean()):.1f} / {float(wus.max()):.1f} us = setup {us(setup):.1f} + main loop {us(main):.1f} "
          f"({us(main) / nk:.3f} us per K-step incl. the cold start) + post-loop barrier {us(skew):.1f} + epilogue A {us(epa - skew):.1f} + "
          f"B {us(epb):.1f} + store drain {us(epi_all - epa - epb):.1f}", flush=True)
    # per-phase stamps of the ping-pong loop (shader cycles per K-step): waves 0-3 of a block = group 0 (stages A), 4-7 = group 1
    wave = torch.arange(len(t)) % 8
    g0, g1 = t[wave < 4], t[wave >= 4]
    cyc = lambda c: float(c.mean()) / nk       # noqa: E731
    print(f"    group 0 per K-step [cycles]: MFMAs issued {cyc(g0[:, 1]):.0f} | wait A(kt+1) {cyc(g0[:, 2]):.0f} | B1 wait {cyc(g0[:, 3]):.0f} | "
          f"reads + A pieces issued, B2, fragments landed {cyc(g0[:, 0]):.0f}   (sum {cyc(g0[:, 0] + g0[:, 1] + g0[:, 2] + g0[:, 3]):.0f})")
    print(f"    group 1 per K-step [cycles]: reads + W pieces issued, W(kt+1) + fragments landed {cyc(g1[:, 4]):.0f} | B1 wait {cyc(g1[:, 5]):.0f} | "
          f"MFMAs issued {cyc(g1[:, 6]):.0f} | B2 wait {cyc(g1[:, 7]):.0f}   (sum {cyc(g1[:, 4] + g1[:, 5] + g1[:, 6] + g1[:, 7]):.0f}); "
          f"2 x MI x NJ x 16 = {2 * 4 * 6 * 16} MFMA cycles per group and K-step", flush=True)

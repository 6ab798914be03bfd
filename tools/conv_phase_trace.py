#!/usr/bin/env python3
"""Where an fp32-faithful (bf16x3) implicit-GEMM conv launch of the VAE decoder spends its time (GPU box): the phase-stamped twins
of the two ping-pong conv tiles (cfg 49 = 256 x 256, cfg 52 = 256 x 128; gemm.hip swaps them in while a trace buffer is set).
Per wave: setup, main loop, epilogue; per K-step of the ping-pong loop the same eight stamps as tools/gemm_phase_trace2.py.
usage: FLUXHIP_CONV_X3_CFG=<49|52|cfg + (S << 8)> python tools/conv_phase_trace.py      (the picker's choice when unset)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flux_generator_amd import ops, _lib

lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
forced = int(os.environ.get("FLUXHIP_CONV_X3_CFG", "0"))
# the 3x3 layers of the Flux decoder at 512 x 512 (B, H, W, Cin, Cout)
CASES = [(1, 64, 64, 512, 512), (1, 128, 128, 512, 512), (1, 256, 256, 512, 256), (1, 256, 256, 256, 256), (1, 512, 512, 256, 128),
         (1, 512, 512, 128, 128)]
if os.environ.get("TRACE_B"):
    CASES = [(int(os.environ["TRACE_B"]),) + c[1:] for c in CASES]
for (B, H, W, Cin, Cout) in CASES:
    x = ops.split_f32(torch.randn(B, H, W, Cin, generator=g, device=dev))
    w = ops.split_f32(torch.randn(Cout, 3, 3, Cin, generator=g, device=dev) * (9 * Cin) ** -0.5)
    b = torch.randn(Cout, generator=g, device=dev)
    out = torch.empty(2, B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    for _ in range(2):
        ops.conv2d_x3(x, w, b, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv2d_x3(x, w, b, out=out); e1.record(); torch.cuda.synchronize()
    plain_us = e0.elapsed_time(e1) * 1e3
    trace = torch.zeros(16384 * 8 * 16, dtype=torch.int64, device=dev)
    lib.fluxhip_gemm_set_trace(trace.data_ptr())
    e0.record(); ops.conv2d_x3(x, w, b, out=out); e1.record(); torch.cuda.synchronize()
    lib.fluxhip_gemm_set_trace(None)
    t = trace.view(-1, 16).cpu().double()
    t = t[t[:, 8] > 0]
    M, K = B * H * W, 9 * Cin
    fl = 3 * 2.0 * M * Cout * K
    if len(t) == 0:
        print(f"B{B} {H}x{W} {Cin}->{Cout}: launch {plain_us:.1f} us = {fl / plain_us / 1e6:.0f} TFLOP/s of MFMA work; no stamps (not a ping-pong conv tile)", flush=True)
        continue
    nblk = int(torch.nonzero(trace.view(-1, 16)[:, 8].cpu() > 0).max()) // 8 + 1     # launched blocks (chain split-K: only a tile's last block has stamps)
    ghz = float((t[:, 8] / t[:, 9]).mean()) * 0.1
    us = lambda c: float(c.mean()) / ghz / 1e3     # noqa: E731
    whole, setup, epi = t[:, 8], t[:, 10], t[:, 11]
    main = whole - setup - epi
    wave = torch.arange(len(t)) % 8
    g0, g1 = t[wave < 4], t[wave >= 4]
    steps_total = 3 * (K // 64)
    # tile and split of this launch: the (tile, S) whose block count matches and whose MFMA phase matches the stamps
    best = None
    for (bm, bn) in ((256, 256), (256, 128)):
        tiles = ((M + bm - 1) // bm) * ((Cout + bn - 1) // bn)
        if nblk % tiles:
            continue
        S_ = nblk // tiles
        per = float(g0[:, 1].mean()) / (steps_total / S_) / (2 * (bm // 64) * (bn // 32) * 16)     # measured / ideal MFMA issue time
        if best is None or abs(per - 1.07) < abs(best[3] - 1.07):
            best = (bm, bn, S_, per)
    bm, bn, S, _ = best
    nk = steps_total / S
    cyc = lambda c: float(c.mean()) / nk       # noqa: E731
    mfma = 2 * (bm // 64) * (bn // 32) * 16
    print(f"B{B} {H}x{W} {Cin}->{Cout}: tile {bm}x{bn} S={S} ({nblk} blocks, {nk:.0f} K-steps each); launch {plain_us:.1f} us (stamped {e0.elapsed_time(e1) * 1e3:.1f}) = "
          f"{fl / plain_us / 1e6:.0f} TFLOP/s of MFMA work; {ghz:.2f} GHz; whole wave {us(whole):.1f} us = setup {us(setup):.1f} + main loop {us(main):.1f} "
          f"({us(main) / nk * 1e3:.0f} ns = {float(main.mean()) / nk:.0f} cycles per K-step) + epilogue {us(epi):.1f}", flush=True)
    print(f"    group 0 per K-step [cycles]: MFMAs issued {cyc(g0[:, 1]):.0f} | wait A {cyc(g0[:, 2]):.0f} | B1 wait {cyc(g0[:, 3]):.0f} | "
          f"reads + A pieces issued, B2, fragments landed {cyc(g0[:, 0]):.0f}   (sum {cyc(g0[:, 0] + g0[:, 1] + g0[:, 2] + g0[:, 3]):.0f})")
    print(f"    group 1 per K-step [cycles]: reads + W pieces issued, W landed {cyc(g1[:, 4]):.0f} | B1 wait {cyc(g1[:, 5]):.0f} | "
          f"MFMAs issued {cyc(g1[:, 6]):.0f} | B2 wait {cyc(g1[:, 7]):.0f}   (sum {cyc(g1[:, 4] + g1[:, 5] + g1[:, 6] + g1[:, 7]):.0f}); "
          f"{mfma} MFMA cycles per group and K-step", flush=True)

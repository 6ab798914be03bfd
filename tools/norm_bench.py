#!/usr/bin/env python3
"""Time the two memory-bound passes of a Flux block — fluxhip_ln_modulate_bf16 and fluxhip_qk_norm_rope_bf16 — at the
C2 (B=1, T=1280) and C5 (B=4, T=4352) shapes on the GPU box.  50 launches are captured in one hipGraph (a Python ctypes
call costs more than these kernels run), so the number is per launch INCLUDING the ~1.5 us dependent-kernel boundary;
algorithmic bytes / that time is printed next to it (HBM roofline 6.3 TB/s achievable)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flux_generator_amd import _lib

lib = _lib.load()
dev = torch.device("cuda")
_lib.bind_device("cuda")
BF = torch.bfloat16
D, H, N = 3072, 24, 50


def timed(fn):
    st = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(st)
    with torch.cuda.stream(side):
        fn(side.cuda_stream)
    st.wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(N):
            fn(s)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (4 * N) * 1e3     # us


for B, S, L in [(1, 256, 1024), (4, 256, 4096), (1, 512, 4096)]:
    T = S + L
    Tpad = (T + 63) // 64 * 64
    x = torch.randn(B, T, D, device=dev).to(BF)
    xm = torch.empty_like(x)
    mods = (torch.randn(B, 4 * D, device=dev) * 0.3).to(BF)
    mp = mods.data_ptr()

    def ln(s):
        assert lib.fluxhip_ln_modulate_bf16(x.data_ptr(), xm.data_ptr(), B, T, D, S, T * D, T * D, mp, mp + 2 * D, mp + 4 * D,
                                            mp + 6 * D, 4 * D, 1e-6, s) == 0

    us = timed(ln)
    by = 2 * B * T * D * 2
    print(f"B{B} T{T} ln_modulate      {us:7.2f} us/launch  {by / us / 1e6:6.2f} TB/s ({by / 1e6:.1f} MB)")
    qkv = torch.randn(B, T, 3 * D, device=dev).to(BF)
    w = [torch.ones(128, device=dev).to(BF) for _ in range(4)]
    rope = torch.randn(B, T, 64, 2, device=dev).to(BF)
    Q = torch.empty(B, H, T, 128, dtype=BF, device=dev)
    K = torch.empty_like(Q)
    Vt = torch.zeros(B, H, 128, Tpad, dtype=BF, device=dev)

    def qk(s):
        assert lib.fluxhip_qk_norm_rope_bf16(qkv.data_ptr(), 3 * D, B, T, S, H, w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(),
                                             w[3].data_ptr(), rope.data_ptr(), T * 128, Q.data_ptr(), K.data_ptr(), Vt.data_ptr(),
                                             Tpad, 1e-5, s) == 0

    us = timed(qk)
    by = 6 * B * T * D * 2
    print(f"B{B} T{T} qk_norm_rope_vt  {us:7.2f} us/launch  {by / us / 1e6:6.2f} TB/s ({by / 1e6:.1f} MB)")

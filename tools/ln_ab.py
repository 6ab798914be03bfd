#!/usr/bin/env python3
"""Build-to-build A/B of fluxhip_layernorm_affine_bf16 (raw ctypes, 50 dependent launches per hipGraph).
usage: python tools/ln_ab.py name=path/to/libfluxhip.so [name=path ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
libs = {}
for a in sys.argv[1:]:
    n, p = a.split("=")
    L = C.CDLL(os.path.abspath(p))
    L.fluxhip_layernorm_affine_bf16.restype = C.c_int
    L.fluxhip_layernorm_affine_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    libs[n] = L
BF = torch.bfloat16
for rows, D in ((4096, 1280), (16384, 640), (1024, 1280), (512, 1280), (2464, 2048), (512, 4096), (8192, 320)):
    x = torch.randn(rows, D, device="cuda").to(BF); y = torch.empty_like(x)
    g_, b_ = torch.randn(D, device="cuda").to(BF), torch.randn(D, device="cuda").to(BF)
    line = f"rows {rows:6d} D {D:5d}:"
    for rnd in range(2):
        for n, L in libs.items():
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                def run():
                    for i in range(25):
                        L.fluxhip_layernorm_affine_bf16(x.data_ptr(), y.data_ptr(), rows, D, g_.data_ptr(), b_.data_ptr(), 1e-5, torch.cuda.current_stream().cuda_stream)
                        L.fluxhip_layernorm_affine_bf16(y.data_ptr(), x.data_ptr(), rows, D, g_.data_ptr(), b_.data_ptr(), 1e-5, torch.cuda.current_stream().cuda_stream)
                run(); torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=s):
                    run()
                gr.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            if rnd == 1:
                us = e0.elapsed_time(e1) * 1e3 / 50
                line += f"  {n} {us:6.2f} us ({4.0 * rows * D / us / 1e6:.2f} TB/s)"
    print(line)

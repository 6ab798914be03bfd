#!/usr/bin/env python3
"""Build-to-build A/B of the GEMM entry point on the Flux batch-1 shapes (GPU box): loads the given libfluxhip builds side by side
through raw ctypes (only fluxhip_gemm_bf16 / fluxhip_set_workspace, whose signatures have not changed since round 1), rotates
the weights, interleaves the builds, reports the median of 5 rounds.  A flag inside one binary is not a baseline: code that is
merely PRESENT changes register allocation and layout for every path that shares the kernel.

usage: python tools/lib_ab.py name=path/to/libfluxhip.so [name=path ...]"""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flux_generator_amd._lib import GemmDesc
from flux_generator_amd.ops import make_gemm_desc, EPI_BIAS, EPI_GATE_RES, EPI_GELU_TANH, EPI_SPLIT_GELU

libs = {}
ws = torch.zeros(96 << 20, dtype=torch.uint8, device="cuda")
for a in sys.argv[1:]:
    n, p = a.split("=")
    L = C.CDLL(os.path.abspath(p))
    L.fluxhip_gemm_bf16.restype = C.c_int
    L.fluxhip_gemm_bf16.argtypes = [C.POINTER(GemmDesc), C.c_void_p]
    L.fluxhip_set_workspace.restype = C.c_int
    L.fluxhip_set_workspace.argtypes = [C.c_void_p, C.c_int64]
    assert L.fluxhip_set_workspace(ws.data_ptr(), ws.numel()) == 0
    libs[n] = L
BF = torch.bfloat16
M, D = 1280, 3072
torch.manual_seed(0)
SHAPES = {"qkv 9216x3072 bias": (9216, 3072, EPI_BIAS), "mlp0 12288x3072 gelu": (12288, 3072, EPI_GELU_TANH),
          "linear1 21504x3072 split-gelu": (21504, 3072, EPI_SPLIT_GELU), "attn.proj 3072x3072 gate-res": (3072, 3072, EPI_GATE_RES),
          "mlp2 3072x12288 gate-res": (3072, 12288, EPI_GATE_RES), "linear2 3072x15360 gate-res": (3072, 15360, EPI_GATE_RES)}
stream = torch.cuda.current_stream().cuda_stream
for name, (N, K, epi) in SHAPES.items():
    x = torch.randn(M, K, device="cuda").to(BF)
    w = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(BF) for _ in range(4)]
    b = torch.randn(N, device="cuda").to(BF)
    res = torch.randn(M, 3072, device="cuda").to(BF)
    gate = torch.randn(3072, device="cuda").to(BF)
    if epi == EPI_SPLIT_GELU:
        c1 = torch.empty(M, 9216, dtype=BF, device="cuda"); c2 = torch.empty(M, D + 12288, dtype=BF, device="cuda")
        mk = lambda wi: make_gemm_desc([dict(A=x.data_ptr(), W=wi.data_ptr(), bias=b.data_ptr(), C=c1.data_ptr(), M=M)], 1, N, K, K, 9216, epi,
                                       n_split=9216, C2=c2.data_ptr(), ldc2=D + 12288, c2_coloff=D)      # noqa: E731
    elif epi == EPI_GATE_RES:
        out = res.clone()
        mk = lambda wi: make_gemm_desc([dict(A=x.data_ptr(), W=wi.data_ptr(), bias=b.data_ptr(), C=out.data_ptr(), res=out.data_ptr(),
                                             gate=gate.data_ptr(), M=M)], 1, N, K, K, N, epi)                                         # noqa: E731
    else:
        out = torch.empty(M, N, dtype=BF, device="cuda")
        mk = lambda wi: make_gemm_desc([dict(A=x.data_ptr(), W=wi.data_ptr(), bias=b.data_ptr(), C=out.data_ptr(), M=M)], 1, N, K, K, N, epi)  # noqa: E731
    descs = [mk(wi) for wi in w]
    times = {n: [] for n in libs}
    for rnd in range(6):
        for n, L in libs.items():
            for d in descs:
                assert L.fluxhip_gemm_bf16(C.byref(d), stream) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(12):
                assert L.fluxhip_gemm_bf16(C.byref(descs[it & 3]), stream) == 0
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[n].append(e0.elapsed_time(e1) / 12 * 1e3)
    fl = 2.0 * M * N * K
    print(f"{name:32s} " + "  ".join(f"{n}: {statistics.median(t):6.1f} us {fl / statistics.median(t) / 1e6:5.0f} TF" for n, t in times.items()), flush=True)

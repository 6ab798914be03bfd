#!/usr/bin/env python3
"""Time every compiled GEMM tile configuration on the Flux GEMM shapes (run on the GPU box).
Random bf16 operands (never zeros: DVFS inflates zero-data numbers), HIP events on the launch stream."""
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flux_generator_amd import ops, _lib

dev = torch.device("cuda:0")
BF = torch.float16 if os.environ.get("TUNE_F16") else torch.bfloat16      # TUNE_F16=1: the float16-storage kernels (SD / SDXL UNet)
lib = _lib.load()
ncfg = 0
import ctypes
bm, bn, th = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
while lib.fluxhip_gemm_tile_shape(ncfg + 1, bm, bn, th) == 0:
    ncfg += 1

M = int(os.environ.get("TUNE_M", "1280"))
shapes = {  # name: (M, N, K, epi)
    "qkv(2grp)": (M, 9216, 3072, 0),
    "proj(2grp)": (M, 3072, 3072, 2),
    "mlp0(2grp)": (M, 12288, 3072, 1),
    "mlp2(2grp)": (M, 3072, 12288, 2),
    "linear1": (M, 21504, 3072, 0),
    "linear2": (M, 3072, 15360, 2),
}
if os.environ.get("TUNE_SHAPES"):      # "name:M:N:K:epi,..." e.g. the SDXL UNet transformer GEMMs at batch 16
    shapes = {t.split(":")[0]: tuple(int(v) for v in t.split(":")[1:]) for t in os.environ["TUNE_SHAPES"].split(",")}
only = sys.argv[1:]  # optional list of cfg ids
def _code(c):   # "41s2" = tile cfg 41 with split-K 2
    c = str(c)
    return int(c.split("s")[0]) | (int(c.split("s")[1]) << 8) if "s" in c else int(c)
cfgs = [_code(c) for c in only] if only else list(range(1, ncfg + 1))
res = {}
for name, (m, n, k, epi) in shapes.items():
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(m, k, generator=g, device=dev).to(BF)
    nrot = 1 if os.environ.get('TUNE_WARM') else max(2, int(700e6 // (n * k * 2)) + 1)      # rotate weights so they stream from HBM (cold), as in situ
    ws = [(torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(BF) for _ in range(nrot)]
    w = ws[0]
    b = torch.randn(n, generator=g, device=dev).to(BF)
    r = torch.randn(m, n, generator=g, device=dev).to(BF)
    gate = torch.randn(n, generator=g, device=dev).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    ref = None
    row = {}
    FP8 = bool(os.environ.get("TUNE_FP8"))       # fp8 kernels: e4m3 operands quantised once up front
    if FP8:
        xq, xs = ops.quantize_rows_fp8(x)
        wqs = [ops.quantize_rows_fp8(w_) for w_ in ws]
        _lin = ops.linear
        ops_linear = lambda x_, w_, b_, **kw_: ops.linear_fp8(xq, xs, *wqs[[id(t) for t in ws].index(id(w_))], b_, **kw_)   # noqa: E731
    else:
        ops_linear = ops.linear
    for c in cfgs:
        try:
            kw = dict(epi=epi, out=out, tile_cfg=c)
            if c == 0:
                kw.pop("tile_cfg")          # 0 = whatever the picker chooses
            if epi == 2:
                kw.update(res=r, gate=gate)
            ops_linear(x, w, b, **kw)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.float().clone()
            else:
                err = float((out.float() - ref).abs().max())
                assert err < 0.1, (name, c, err)
            for _ in range(3):
                ops_linear(x, w, b, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_it = 20
            if os.environ.get("TUNE_GRAPH"):     # short kernels: replay the launches from a hipGraph (no host launch path in the timing)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for it in range(n_it):
                        ops_linear(x, ws[it % nrot], b, **kw)
                gr.replay()
                torch.cuda.synchronize()
                e0.record(); gr.replay(); e1.record()
            else:
                e0.record()
                for it in range(n_it):
                    ops_linear(x, ws[it % nrot], b, **kw)
                e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n_it
            row[c] = round(2.0 * m * n * k / (ms * 1e-3) / 1e12, 1)
        except Exception as ex:  # noqa
            row[c] = f"ERR {ex}"
    if os.environ.get("TUNE_BLASLT"):        # yardstick only: the vendor library on the same operands
        for _ in range(3):
            torch.nn.functional.linear(x, w, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(20):
            torch.nn.functional.linear(x, ws[it % nrot], b)
        e1.record()
        torch.cuda.synchronize()
        row["blaslt"] = round(2.0 * m * n * k / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12, 1)
    res[name] = row
    best = max([(v, c) for c, v in row.items() if isinstance(v, float) and c != "blaslt"] or [(0.0, 0)])
    print(f"{name:12s} M={m} N={n} K={k}: " + " ".join(f"c{c if isinstance(c, str) or c < 256 else str(c & 255) + 's' + str(c >> 8)}={v}" for c, v in row.items()) + f"  BEST c{best[1]}={best[0]}", flush=True)
print(json.dumps(res))

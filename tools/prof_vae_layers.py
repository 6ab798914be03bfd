#!/usr/bin/env python3
"""Per-call timing of one Flux VAE decode (HIP events around every ops.* call, eager)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
os.environ.setdefault("FLUX_ALLOW_RANDOM_INIT", "1")
from flux_generator_amd import ops
from flux_generator_amd.flux.utils import load_ae

log = []
def wrap(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        shp = tuple(a[0].shape)
        if name.endswith("_x3") and len(shp) == 5:
            shp = shp[1:]                                   # split tensors: [2, B, H, W, C]
        extra = tuple(a[1].shape) if name in ("conv2d", "linear", "conv2d_x3", "conv_up2x_x3", "linear_x3") else ()
        if name in ("conv2d_x3", "linear_x3"):
            extra = extra[1:]                               # [2, Cout, kh, kw, Cin]
        if name == "conv_up2x_x3":
            extra = (extra[2], 3, 3, extra[-1])             # [2, 4, Cout, 2, 2, Cin]: report the 3x3 conv it replaces
        log.append((name, shp, extra, k.get("ups", False), e0, e1))
        return r
    setattr(ops, name, w)
for n in ("conv2d", "groupnorm_silu", "linear", "softmax_rows", "conv2d_out_image", "unpack_latents", "conv2d_x3", "conv_up2x_x3",
          "groupnorm_silu_x3", "linear_x3", "gemm_x3", "softmax_rows_x3", "conv2d_out_image_x3", "unpack_latents_x3"):
    if hasattr(ops, n):
        wrap(n)
ae = load_ae("flux-schnell", device="cuda")
x = torch.randn(1, 1024, 64, device="cuda").to(torch.bfloat16)
for it in range(3):
    log.clear()
    ae.decode_packed(x, (64, 64))
    torch.cuda.synchronize()
tot = 0.0
agg = collections.OrderedDict()
for name, shp, extra, ups, e0, e1 in log:
    ms = e0.elapsed_time(e1); tot += ms
    key = (name, shp, extra, ups)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += ms
for (name, shp, extra, ups), (n, ms) in agg.items():
    fl = ""
    if name in ("conv2d", "conv2d_x3", "conv_up2x_x3"):
        ups = ups or name == "conv_up2x_x3"
        B, H, W, Cin = shp; Cout = extra[0]; ks = 1 if len(extra) == 2 else extra[1]
        Ho, Wo = (H * 2, W * 2) if ups else (H, W)
        gf = 2.0 * B * Ho * Wo * Cout * Cin * ks * ks / 1e9
        fl = f"{gf * n / ms:8.0f} TF/s-ish(GF/ms)"
    print(f"{name:18s} x{n:2d} in={shp} w={extra} ups={ups}: {ms:7.3f} ms {fl}")
print(f"total {tot:.3f} ms")

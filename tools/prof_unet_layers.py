#!/usr/bin/env python3
"""Per-call timing of one SDXL UNet step at batch 16, 64x64 latents (HIP events around every ops.* call and the raw
attention entry point, eager).  Aggregated by (op, shapes); GEMM-shaped ops get GFLOP/ms (= TFLOP/s)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
os.environ.setdefault("FLUX_ALLOW_RANDOM_INIT", "1")
from flux_generator_amd import ops, _lib
from flux_generator_amd.stable_diffusion import StableDiffusionXL
from flux_generator_amd.stable_diffusion import unet as unet_mod

log = []
def ev():
    return torch.cuda.Event(enable_timing=True)

def wrap(mod, name, describe):
    fn = getattr(mod, name)
    def w(*a, **k):
        e0, e1 = ev(), ev()
        e0.record(); r = fn(*a, **k); e1.record()
        log.append((name,) + describe(a, k) + (e0, e1))
        return r
    setattr(mod, name, w)

def d_linear(a, k):
    x, w = a[0], a[1]
    M = x.numel() // x.shape[-1]
    return (f"M{M} N{w.shape[0]} K{w.shape[1]} epi{k.get('epi', 0)}", 2.0 * M * w.shape[0] * w.shape[1])
def d_conv(a, k):
    x, w = a[0], a[1]
    B, H, W, Cin = x.shape
    ks = w.shape[1] if w.dim() == 4 else 1
    s = k.get("stride", 1)
    Ho, Wo = (H * 2, W * 2) if k.get("ups") else (H // s, W // s)
    return (f"B{B} {H}x{W} Cin{w.shape[-1]} Cout{w.shape[0]} k{ks} s{s} ups{int(bool(k.get('ups')))}", 2.0 * B * Ho * Wo * w.shape[0] * w.shape[-1] * ks * ks)
def d_gemm(a, k):
    d = a[0]
    try:
        g0 = d.groups[0]
        return (f"batched M{g0['M']} N{d.N} K{d.K} nb{d.nbatch}", 2.0 * g0['M'] * d.N * d.K * d.nbatch)
    except Exception:
        return ("desc", 0.0)
def d_shape(a, k):
    return (str(tuple(a[0].shape)), 0.0)

wrap(ops, "linear", d_linear)
wrap(ops, "conv2d", d_conv)
for n in ("groupnorm_silu", "layernorm_affine", "concat_channels", "small_linear", "sincos_embed"):
    wrap(ops, n, d_shape)

# the batched V^T GEMM goes through ops.gemm(make_gemm_desc(...)): record the descriptor arguments
_mk = unet_mod.make_gemm_desc
_last = {}
def mk(groups, nbatch, N, K, lda, ldc, *a, **k):
    _last["d"] = (groups[0]["M"], nbatch, N, K)
    _last["fresh"] = True
    return _mk(groups, nbatch, N, K, lda, ldc, *a, **k)
unet_mod.make_gemm_desc = mk
_g = ops.gemm
def gemm(desc, *a, **k):
    if not _last.pop("fresh", False):
        return _g(desc, *a, **k)
    e0, e1 = ev(), ev()
    e0.record(); _g(desc, *a, **k); e1.record()
    M, nb, N, K = _last["d"]
    log.append(("gemm(V^T)", f"M{M} N{N} K{K} nb{nb}", 2.0 * M * N * K * nb, e0, e1))
ops.gemm = gemm

lib = _lib.load()
_att = lib.fluxhip_attention_strided_bf16
class LibProxy:
    def __getattr__(self, n):
        if n not in ("fluxhip_attention_strided_bf16", "fluxhip_attention_strided_vt_bf16"):
            return getattr(lib, n)
        fn, sh = getattr(lib, n), int(n.endswith("_vt_bf16"))
        def f(*a):
            # (q, qs0, qs1, qs2, k, ks0, ks1, ks2, vt, [vt_bs,] o, ldo, B, H, hd, N, Tk, Tkpad, scale, stream)
            B, H, hd, N, Tk = a[11 + sh], a[12 + sh], a[13 + sh], a[14 + sh], a[15 + sh]
            e0, e1 = ev(), ev()
            e0.record(); rc = fn(*a); e1.record()
            log.append(("attention", f"B{B} H{H} N{N} Tk{Tk}", 4.0 * B * H * N * Tk * hd, e0, e1))
            return rc
        return f
_load = _lib.load
unet_mod._lib.load = lambda: LibProxy()

B = int(os.environ.get("SDXL_BATCH", "16"))
dev = torch.device("cuda:0")
pipe = StableDiffusionXL("stabilityai/sdxl-turbo", float16=True)
g = torch.Generator(device=dev).manual_seed(0)
x_T = pipe.sampler.sample_prior((B, 64, 64, 4), dtype=pipe.dtype, key=g, device=dev)
cond = torch.randn(B, 77, 2048, generator=g, device=dev).to(pipe.dtype)
pooled = torch.randn(B, 1280, generator=g, device=dev).to(pipe.dtype)
tt = (pooled, torch.tensor([[512, 512, 0, 0, 512, 512.0]] * B, device=dev))
ts = torch.full((B,), 999.0, device=dev)
for it in range(3):
    log.clear()
    e0, e1 = ev(), ev()
    e0.record(); pipe.unet(x_T.to(pipe.dtype), ts, cond, text_time=tt); e1.record()
    torch.cuda.synchronize()
tot = 0.0
agg = collections.OrderedDict()
for name, desc, fl, a, b in log:
    ms = a.elapsed_time(b); tot += ms
    r = agg.setdefault((name, desc), [0, 0.0, fl]); r[0] += 1; r[1] += ms
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (name, desc), (n, ms, fl) in rows:
    tf = f"{fl * n / ms / 1e9:7.0f} TFLOP/s" if fl else ""
    print(f"{ms:8.3f} ms  x{n:3d} {1e3 * ms / n:8.1f} us  {name:16s} {desc:48s} {tf}")
print(f"sum of op events {tot:.3f} ms; eager wall (events) {e0.elapsed_time(e1):.3f} ms")

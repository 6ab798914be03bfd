"""Shared building blocks of the two VAE decoders (flux/autoencoder.py and stable_diffusion/.../vae.py of the
reference are the same architecture family with different parameter names, GroupNorm eps and scaling).

Two arithmetic modes, selected per call:

  "fp32"  the reference's arithmetic.  Both reference decoders run in float32: flux/utils.py:137-143 loads
          ae.safetensors in its checkpoint dtype and never casts, stable_diffusion/__init__.py:25 calls
          load_autoencoder(model, False).  Here: the fp32-faithful split-bf16 ("bf16x3") kernels of libfluxhip
          (include/fluxhip.h) — tensors are (hi, lo) bf16 plane pairs [2, ...], products are three MFMA passes
          with fp32 accumulation, GroupNorm / softmax / the final conv are float32 arithmetic.
  "bf16"  bf16 storage / fp32 accumulate: a third of the MFMA work and half the bytes, ~1e-2 image error.

Parameters are held once, in float32 (the checkpoint's dtype); the operand layouts the kernels read (bf16 copy,
split planes, 64-channel padding of conv_in) are derived on first use and rebuilt if the master tensor changes.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional, Tuple, Union

import torch

from . import ops
from .ops import EPI_BIAS, EPI_GATE_RES, make_gemm_desc

BF16 = torch.bfloat16
F32 = torch.float32


class ParamStore:
    """float32 master parameters + derived operand layouts."""

    def __init__(self, shapes: Dict[str, Tuple[int, ...]], device: torch.device, pad64: Iterable[str] = ()):
        self.device = device
        self.master = {k: torch.empty(*shp, dtype=F32, device=device) for k, shp in shapes.items()}
        self._pad64 = set(pad64)         # weights whose last (input-channel) dim is zero-padded to a multiple of 64
        self._derived: Dict[Tuple[str, str], Tuple[int, torch.Tensor]] = {}
        self.epoch = 0                   # bumped by load() / init_random() / broadcast(): part of the pipelines' decode-graph keys

    def __contains__(self, name: str) -> bool:
        return name in self.master

    def init_random(self, seed: int) -> None:
        """MLX defaults: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for conv / linear weight and bias, (1, 0) for norms."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        self.epoch += 1
        for name, t in self.master.items():
            wt = self.master[f"{name.rsplit('.', 1)[0]}.weight"]
            if wt.dim() == 1:
                t.fill_(1.0 if name.endswith(".weight") else 0.0)
                continue
            k = 1.0 / math.sqrt(wt[0].numel())
            t.copy_((torch.rand(t.shape, generator=g, device=self.device) * 2 - 1) * k)

    def load(self, weights: Union[Dict[str, torch.Tensor], Iterable[Tuple[str, torch.Tensor]]], strict: bool,
             skip_prefixes: Tuple[str, ...]) -> None:
        items = weights.items() if isinstance(weights, dict) else weights
        self.epoch += 1           # derived layouts are rebuilt lazily: hipGraphs captured over the old ones are stale
        seen = set()
        for k, w in items:
            if k not in self.master:
                if k.startswith(skip_prefixes) or not strict:
                    continue
                raise ValueError(f"Unexpected parameter {k}")
            dst = self.master[k]
            if tuple(dst.shape) != tuple(w.shape):
                raise ValueError(f"Shape mismatch for {k}: expected {tuple(dst.shape)}, got {tuple(w.shape)}")
            dst.copy_(w.to(device=self.device, dtype=F32))
            seen.add(k)
        if strict and set(self.master) - seen:
            raise ValueError(f"Missing parameters: {sorted(set(self.master) - seen)[:5]} ...")

    def get(self, name: str, kind: str) -> torch.Tensor:
        """kind: "f32" master, "bf16" copy, "x3" split (hi, lo) planes, "x3up" split planes of the sub-pixel form of
        a 3x3 conv that follows a nearest x2 upsample (ops.subpixel_weights)."""
        t = self.master[name]
        if kind == "f32":
            return t
        ent = self._derived.get((name, kind))
        if ent is not None and ent[0] == t._version:
            return ent[1]
        src = t
        if name in self._pad64 and t.shape[-1] % 64:
            src = torch.nn.functional.pad(t, (0, 64 - t.shape[-1] % 64)).contiguous()
        if kind == "x3up":
            src = ops.subpixel_weights(src)
        d = src.to(BF16) if kind == "bf16" else ops.split_f32(src)
        if ent is not None and ent[1].shape == d.shape:
            # the master changed (in-place edit, load, init_random): rebuild INTO the existing operand tensor — captured
            # hipGraphs (and the launch plans of the callers) hold its address, so it is never freed or replaced
            ent[1].copy_(d)
            d = ent[1]
        self._derived[(name, kind)] = (t._version, d)
        return d

    def broadcast(self, src: int = 0) -> None:
        """Every master parameter from rank `src` over RCCL (multi-GPU: only rank `src` read the checkpoint), derived
        operand layouts marked stale (rebuilt in place on next use)."""
        from . import parallel
        parallel.broadcast_tensors(list(self.master.values()), src)
        self.epoch += 1
        self._derived = {k: (-1, d) for k, (_, d) in self._derived.items()}


# Queries per block of the single-head VAE attention: the float32 logits buffer is ATTN_QUERY_BLOCK x N (64 MiB at
# 512x512, 256 MiB at 1024x1024 where N x N would be 1 GiB per image), the softmax sees whole rows, so no online rescaling.
ATTN_QUERY_BLOCK = 4096

# ---------------------------------------------------------------------------------------------- blocks
# `names` maps the role of a sub-module to the reference's attribute name in that decoder:
#   Flux AE (flux/autoencoder.py):   shortcut "nin_shortcut", attention norm "norm", q/k/v/out "q","k","v","proj_out"
#   SD VAE  (.../vae.py, unet.py):   shortcut "conv_shortcut", "group_norm", "query_proj","key_proj","value_proj","out_proj"

def resnet(P: ParamStore, fp32: bool, p: str, x: torch.Tensor, shortcut: str, groups: int, eps: float) -> torch.Tensor:
    """ResnetBlock (flux/autoencoder.py:83-98) / ResnetBlock2D without temb (unet.py:152-170 via vae.py:56-63):
    GN -> SiLU -> conv3x3 -> GN -> SiLU -> conv3x3, + (1x1 shortcut of) x folded into the last conv's epilogue."""
    if fp32:
        W3, F = (lambda n: P.get(n, "x3")), (lambda n: P.get(n, "f32"))     # noqa: E731
        h = ops.groupnorm_silu_x3(x, F(f"{p}.norm1.weight"), F(f"{p}.norm1.bias"), groups, eps, True)
        # every 3x3 conv output of the decoders feeds a GroupNorm next: its epilogue leaves the partial sums (gn_stats)
        h = ops.conv2d_x3(h, W3(f"{p}.conv1.weight"), F(f"{p}.conv1.bias"), gn_stats=True)
        h = ops.groupnorm_silu_x3(h, F(f"{p}.norm2.weight"), F(f"{p}.norm2.bias"), groups, eps, True)
        if f"{p}.{shortcut}.weight" in P:
            x = ops.conv2d_x3(x, W3(f"{p}.{shortcut}.weight"), F(f"{p}.{shortcut}.bias"))
        return ops.conv2d_x3(h, W3(f"{p}.conv2.weight"), F(f"{p}.conv2.bias"), res=x, gn_stats=True)
    W = lambda n: P.get(n, "bf16")     # noqa: E731
    h = ops.groupnorm_silu(x, W(f"{p}.norm1.weight"), W(f"{p}.norm1.bias"), groups, eps, True)
    h = ops.conv2d(h, W(f"{p}.conv1.weight"), W(f"{p}.conv1.bias"))
    h = ops.groupnorm_silu(h, W(f"{p}.norm2.weight"), W(f"{p}.norm2.bias"), groups, eps, True)
    if f"{p}.{shortcut}.weight" in P:
        x = ops.conv2d(x, W(f"{p}.{shortcut}.weight"), W(f"{p}.{shortcut}.bias"))
    return ops.conv2d(h, W(f"{p}.conv2.weight"), W(f"{p}.conv2.bias"), res=x)


def attention(P: ParamStore, fp32: bool, p: str, x: torch.Tensor, norm: str, q: str, k: str, v: str, o: str,
              groups: int, eps: float) -> torch.Tensor:
    """Single-head attention over the H*W pixels (flux/autoencoder.py:42-52; vae.py:25-42): GN, q / k / v Linears,
    softmax(q k^T / sqrt(C)) v, out projection + residual.  V is produced transposed (V^T = Wv y^T with a row bias)
    so the PV product needs no transpose pass; its contraction runs over Np keys (zero padded)."""
    if fp32:
        W3, F = (lambda n: P.get(n, "x3")), (lambda n: P.get(n, "f32"))     # noqa: E731
        _, B, H, Wd, C = x.shape
        N = H * Wd
        dev = x.device
        y = ops.groupnorm_silu_x3(x, F(f"{p}.{norm}.weight"), F(f"{p}.{norm}.bias"), groups, eps, False).view(2, B, N, C)
        qq = ops.linear_x3(y, W3(f"{p}.{q}.weight"), F(f"{p}.{q}.bias"))
        kk = ops.linear_x3(y, W3(f"{p}.{k}.weight"), F(f"{p}.{k}.bias"))
        xr = x.view(2, B, N, C)
        out = torch.empty_like(xr)
        Np = (N + 63) // 64 * 64
        vt = torch.zeros(2, C, Np, dtype=BF16, device=dev)
        QB = min(N, ATTN_QUERY_BLOCK)            # logits exist for one block of queries at a time: QB x N, never N x N
        s = torch.empty(QB, Np, dtype=F32, device=dev)
        pm = torch.zeros(2, QB, Np, dtype=BF16, device=dev)       # padded key columns stay zero
        ob = torch.empty(2, N, C, dtype=BF16, device=dev)
        for b in range(B):
            ops.gemm_x3(W3(f"{p}.{v}.weight"), y[:, b], vt, C, N, C, C, Np, bias=F(f"{p}.{v}.bias"), row_bias=True)
            for r0 in range(0, N, QB):
                m = min(QB, N - r0)
                ops.gemm_x3(qq[:, b, r0:r0 + m], kk[:, b], s, m, N, C, C, Np, out_f32=True)
                ops.softmax_rows_x3(s[:m], C ** -0.5, pm[:, :m], cols=N)
                ops.gemm_x3(pm[:, :m], vt, ob[:, r0:r0 + m], m, C, Np, Np, C)
            ops.gemm_x3(ob, W3(f"{p}.{o}.weight"), out[:, b], N, C, C, C, C, bias=F(f"{p}.{o}.bias"), res=xr[:, b])
        return out.view(2, B, H, Wd, C)
    W = lambda n: P.get(n, "bf16")     # noqa: E731
    B, H, Wd, C = x.shape
    N = H * Wd
    y = ops.groupnorm_silu(x, W(f"{p}.{norm}.weight"), W(f"{p}.{norm}.bias"), groups, eps, False)
    qq = ops.linear(y.view(B, N, C), W(f"{p}.{q}.weight"), W(f"{p}.{q}.bias"))
    kk = ops.linear(y.view(B, N, C), W(f"{p}.{k}.weight"), W(f"{p}.{k}.bias"))
    out = torch.empty_like(x)
    Np = (N + 63) // 64 * 64
    vt = torch.zeros(C, Np, dtype=BF16, device=x.device)
    QB = min(N, ATTN_QUERY_BLOCK)
    s = torch.empty(QB, Np, dtype=F32, device=x.device)
    pm = torch.zeros(QB, Np, dtype=BF16, device=x.device)
    ob = torch.empty(N, C, dtype=BF16, device=x.device)
    for b in range(B):
        yb = y[b].view(N, C)
        ops.gemm(make_gemm_desc([dict(A=W(f"{p}.{v}.weight").data_ptr(), W=yb.data_ptr(), bias=W(f"{p}.{v}.bias").data_ptr(),
                                      C=vt.data_ptr(), M=C)], 1, N, C, C, Np, EPI_BIAS, row_bias=True))
        for r0 in range(0, N, QB):
            m = min(QB, N - r0)
            ops.gemm(make_gemm_desc([dict(A=qq[b, r0:r0 + m].data_ptr(), W=kk[b].data_ptr(), C=s.data_ptr(), M=m)], 1, N, C, C, Np,
                                    EPI_BIAS, out_f32=True))
            ops.softmax_rows(s[:m], C ** -0.5, out=pm[:m], cols=N)
            ops.gemm(make_gemm_desc([dict(A=pm.data_ptr(), W=vt.data_ptr(), C=ob[r0:r0 + m].data_ptr(), M=m)], 1, C, Np, Np, C))
        ops.linear(ob, W(f"{p}.{o}.weight"), W(f"{p}.{o}.bias"), epi=EPI_GATE_RES, out=out[b].view(N, C),
                   res=x[b].view(N, C))
    return out


def conv(P: ParamStore, fp32: bool, p: str, x: torch.Tensor, ups: bool = False) -> torch.Tensor:
    if fp32 and ups:     # Upsample + conv as four 2x2 convs of the low-res input (4/9 of the MFMA work)
        return ops.conv_up2x_x3(x, P.get(f"{p}.weight", "x3up"), P.get(f"{p}.bias", "f32"), gn_stats=True)
    if fp32:
        return ops.conv2d_x3(x, P.get(f"{p}.weight", "x3"), P.get(f"{p}.bias", "f32"), ups=ups, gn_stats=True)
    return ops.conv2d(x, P.get(f"{p}.weight", "bf16"), P.get(f"{p}.bias", "bf16"), ups=ups)


def norm_out_conv_out(P: ParamStore, fp32: bool, norm: str, conv_out: str, x: torch.Tensor, groups: int, eps: float,
                      clip01: bool) -> torch.Tensor:
    """GN -> SiLU -> conv_out (C -> 3) -> float32 image [, clip((y + 1), 0, 2) / 2]."""
    if fp32:
        F = lambda n: P.get(n, "f32")     # noqa: E731
        h = ops.groupnorm_silu_x3(x, F(f"{norm}.weight"), F(f"{norm}.bias"), groups, eps, True)
        return ops.conv2d_out_image_x3(h, F(f"{conv_out}.weight"), F(f"{conv_out}.bias"), clip01)
    W = lambda n: P.get(n, "bf16")     # noqa: E731
    h = ops.groupnorm_silu(x, W(f"{norm}.weight"), W(f"{norm}.bias"), groups, eps, True)
    return ops.conv2d_out_image(h, W(f"{conv_out}.weight"), W(f"{conv_out}.bias"), clip01)


def check_precision(precision: Optional[str]) -> None:
    if precision not in (None, "fp32", "bf16"):
        raise ValueError("precision must be 'fp32' (the reference's arithmetic) or 'bf16'")

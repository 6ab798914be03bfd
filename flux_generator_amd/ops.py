"""Torch-tensor front end of the libfluxhip C ABI.

torch is used only for device memory and the current HIP stream; every computation below is a
hand-written gfx950 kernel launched through ctypes.  All tensors must be CUDA(HIP) tensors.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import Fp8Mx, Fp8Scales, GemmDesc, GemmX3Desc

EPI_BIAS, EPI_GELU_TANH, EPI_GATE_RES, EPI_SPLIT_GELU, EPI_SILU, EPI_GEGLU, EPI_QUICK_GELU, EPI_GELU_ERF, EPI_GEGLU_PAIR = 0, 1, 2, 3, 4, 5, 6, 7, 8
BF16 = torch.bfloat16


class FluxHipError(RuntimeError):
    pass


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise FluxHipError("libfluxhip ops need device tensors (no CPU fallback exists)")
    return t.data_ptr()


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise FluxHipError(f"{what} failed with code {rc} ({'bad argument' if rc == -1 else 'launch failure'})")


def _bf16c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != BF16 or not t.is_contiguous():
        raise FluxHipError(f"{name} must be a contiguous bfloat16 tensor")
    return t


F16 = torch.float16


def _e16(*named, strided: bool = False) -> bool:
    """Operands of an operator that exists for both 16-bit storage types (bfloat16, and IEEE float16 for the
    stable_diffusion/ models with float16=True): all given (tensor, name) pairs must be contiguous and of ONE of the two
    types.  Returns True for float16 (-> the `_f16` entry point of include/fluxhip.h).  strided=True: the operator addresses
    its operands through explicit strides (views into fused projections), so only the element type is checked."""
    dt = None
    for i, (t, name) in enumerate(named):
        if t is None:
            continue
        if t.dtype not in (BF16, F16) or (i < 2 and not strided and not t.is_contiguous()):      # (the first two are the streamed operands)
            raise FluxHipError(f"{name} must be a contiguous bfloat16 or float16 tensor")
        if dt is None:
            dt = t.dtype
        elif t.dtype != dt:
            raise FluxHipError(f"{name} is {t.dtype} but the other operands are {dt}: no mixed 16-bit types")
    return dt == F16


def _fn(name: str, f16: bool):
    """The bf16 entry point `name`, or its float16 twin."""
    if f16:
        name = {"fluxhip_sincos_embed_f32": "fluxhip_sincos_embed_f32_f16",
                "fluxhip_pixel_linear_x3": "fluxhip_pixel_linear_x3_f16in"}.get(name, name.replace("_bf16", "_f16"))
    return getattr(_lib.load(), name), name


# ------------------------------------------------------------------------------------------------
def gemm(desc: GemmDesc, f16: bool = False) -> None:
    """f16: the descriptor's 16-bit operands are IEEE float16 (fluxhip_gemm_f16) instead of bfloat16."""
    fn, name = _fn("fluxhip_gemm_bf16", f16)
    _check(fn(desc, _stream()), name)


def make_gemm_desc(groups: Sequence[dict], nbatch: int, N: int, K: int, lda: int, ldc: int, epi: int = EPI_BIAS,
                   n_split: int = 0, C2: Optional[int] = None, ldc2: int = 0, c2_bstride: int = 0,
                   c2_coloff: int = 0, row_bias: bool = False, alpha: float = 1.0, tile_cfg: int = 0, out_f32: bool = False,
                   ld_add: int = 0) -> GemmDesc:
    """groups: dicts of raw device addresses: A, W, bias, C, res, gate (ints or None) + a_bstride,
    c_bstride, gate_bstride, M; optionally add (+ add_bstride) with the desc's ld_add: a matrix addend."""
    d = GemmDesc()
    d.ngroups, d.nbatch, d.N, d.K, d.lda, d.ldc, d.epi = len(groups), nbatch, N, K, lda, ldc, epi
    d.row_bias, d.n_split, d.ldc2, d.C2, d.c2_bstride, d.c2_coloff = int(row_bias), n_split, ldc2, C2, c2_bstride, c2_coloff
    d.tile_cfg, d.alpha, d.out_f32 = tile_cfg, alpha, int(out_f32)
    for i, g in enumerate(groups):
        t = d.g[i]
        t.A, t.W, t.bias, t.C = g["A"], g["W"], g.get("bias"), g["C"]
        t.res, t.gate = g.get("res"), g.get("gate")
        t.a_bstride, t.c_bstride, t.gate_bstride = g.get("a_bstride", 0), g.get("c_bstride", 0), g.get("gate_bstride", 0)
        t.w_bstride = g.get("w_bstride", 0)
        t.M = g["M"]
        t.add, t.add_bstride = g.get("add"), g.get("add_bstride", 0)
    d.ld_add = ld_add
    return d


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, epi: int = EPI_BIAS,
           out: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
           gate: Optional[torch.Tensor] = None, tile_cfg: int = 0) -> torch.Tensor:
    """y = epi(x @ w.T + b) for x [..., K], w [N, K]; gate [N] (broadcast over rows) optional."""
    f16 = _e16((x, "x"), (w, "w"), (b, "bias"), (res, "res"), (gate, "gate"), (out, "out"))
    K = x.shape[-1]
    N = w.shape[0]
    M = x.numel() // K
    Nout = N // 2 if epi == EPI_GEGLU_PAIR else N      # the pair epilogue multiplies value and gate columns: N / 2 outputs
    if out is None:
        out = torch.empty(*x.shape[:-1], Nout, dtype=x.dtype, device=x.device)
    g = dict(A=_p(x), W=_p(w), bias=_p(b), C=_p(out), res=_p(res), gate=_p(gate), M=M)
    gemm(make_gemm_desc([g], 1, N, K, K, Nout, epi, tile_cfg=tile_cfg), f16)
    return out


def interleave_geglu(value: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    """[value rows | gate rows] of the two GEGLU Linears (weights [N, K] or biases [N]) interleaved in blocks of 16 rows: the
    operand layout of EPI_GEGLU_PAIR (include/fluxhip.h)."""
    n = value.shape[0]
    if n % 16 or value.shape != gate.shape:
        raise FluxHipError("GEGLU halves must have the same shape and a multiple of 16 rows")
    v = value.reshape(n // 16, 16, *value.shape[1:])
    g = gate.reshape(n // 16, 16, *gate.shape[1:])
    return torch.stack([v, g], dim=1).reshape(2 * n, *value.shape[1:]).contiguous()


def small_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], out: Optional[torch.Tensor] = None,
                 silu_in: bool = False, accum: bool = False) -> torch.Tensor:
    f16 = _e16((x, "x"), (w, "w"), (b, "bias"), (out, "out"))
    B, K = x.shape
    N = w.shape[0]
    if out is None:
        if accum:
            raise FluxHipError("accum needs an existing output")
        out = torch.empty(B, N, dtype=x.dtype, device=x.device)
    fn, name = _fn("fluxhip_small_linear_bf16", f16)
    _check(fn(_p(x), _p(w), _p(b), _p(out), B, N, K, int(silu_in), int(accum), _stream()), name)
    return out


def ln_modulate(x: torch.Tensor, out: torch.Tensor, B: int, Tr: int, D: int, S: int, x_bstride: int, out_bstride: int,
                shift_txt, scale_txt, shift_img, scale_img, mod_bstride: int, eps: float = 1e-6) -> None:
    """shift/scale arguments are raw device addresses (views into the per-step modulation table)."""
    _check(_lib.load().fluxhip_ln_modulate_bf16(_p(x) if isinstance(x, torch.Tensor) else x,
                                                 _p(out) if isinstance(out, torch.Tensor) else out,
                                                 B, Tr, D, S, x_bstride, out_bstride, shift_txt, scale_txt, shift_img,
                                                 scale_img, mod_bstride, eps, _stream()), "fluxhip_ln_modulate_bf16")


def qk_norm_rope(qkv: torch.Tensor, ld: int, B: int, T: int, S: int, H: int, qw_txt, kw_txt, qw_img, kw_img,
                 rope: torch.Tensor, rope_bstride: int, Q: torch.Tensor, K: torch.Tensor, Vt: torch.Tensor, Tpad: int,
                 eps: float = 1e-5) -> None:
    _check(_lib.load().fluxhip_qk_norm_rope_bf16(_p(qkv), ld, B, T, S, H, _p(qw_txt), _p(kw_txt), _p(qw_img),
                                                  _p(kw_img), _p(rope), rope_bstride, _p(Q), _p(K), _p(Vt), Tpad, eps,
                                                  _stream()), "fluxhip_qk_norm_rope_bf16")


def vt_key_permutation(tpad: int, device=None) -> torch.Tensor:
    """Index map of the key-permuted V^T layout (include/fluxhip.h, fluxhip_qk_norm_rope_bf16): stored column p holds
    key perm[p]; inside every aligned group of 16 keys the stored order is [0-3, 8-11, 4-7, 12-15].  For callers that
    build V^T themselves (tests, tools): vt_stored = vt_natural[..., perm]."""
    p = torch.arange(tpad, device=device)
    pg = (p & 15) >> 2
    g = ((pg & 1) << 1) | (pg >> 1)
    return (p & ~15) + g * 4 + (p & 3)


def attention_d128(Q: torch.Tensor, K: torch.Tensor, Vt: torch.Tensor, O, ldo: int, B: int, H: int, T: int, Tpad: int,
                   scale: float) -> None:
    _check(_lib.load().fluxhip_attention_d128_bf16(_p(Q), _p(K), _p(Vt), _p(O) if isinstance(O, torch.Tensor) else O,
                                                    ldo, B, H, T, Tpad, scale, _stream()), "fluxhip_attention_d128_bf16")


def timestep_embedding(t: torch.Tensor, dim: int = 256, max_period: float = 10000.0, time_factor: float = 1000.0):
    _bf16c(t, "t")
    out = torch.empty(t.shape[0], dim, dtype=BF16, device=t.device)
    _check(_lib.load().fluxhip_timestep_embedding_bf16(_p(t), _p(out), t.shape[0], dim, time_factor, max_period,
                                                        _stream()), "fluxhip_timestep_embedding_bf16")
    return out


def rope_table(ids: torch.Tensor, axes_dim: Sequence[int], theta: float) -> torch.Tensor:
    """ids int32 [..., 3] -> bf16 [..., sum(axes)/2, 2] (cos, sin)."""
    if ids.dtype != torch.int32 or not ids.is_contiguous() or ids.shape[-1] != 3:
        raise FluxHipError("ids must be contiguous int32 [..., 3]")
    ntok = ids.numel() // 3
    out = torch.empty(*ids.shape[:-1], sum(axes_dim) // 2, 2, dtype=BF16, device=ids.device)
    _check(_lib.load().fluxhip_rope_table_bf16(_p(ids), _p(out), ntok, 3, axes_dim[0], axes_dim[1], axes_dim[2],
                                                float(theta), _stream()), "fluxhip_rope_table_bf16")
    return out


def silu(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """silu(x) elementwise, rounded to x's 16-bit type (fluxhip_silu_bf16 / _f16)."""
    f16 = _e16((x, "x"), (out, "out"))
    if out is None:
        out = torch.empty_like(x)
    fn, name = _fn("fluxhip_silu_bf16", f16)
    _check(fn(_p(x), _p(out), x.numel(), _stream()), name)
    return out


def euler_step(x: torch.Tensor, pred: torch.Tensor, dt: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _bf16c(x, "x"); _bf16c(pred, "pred")
    if out is None:
        out = torch.empty_like(x)
    _check(_lib.load().fluxhip_euler_step_bf16(_p(x), _p(pred), _p(out), x.numel(), float(dt), _stream()),
           "fluxhip_euler_step_bf16")
    return out


def pack_latents(x: torch.Tensor) -> torch.Tensor:
    _bf16c(x, "x")
    B, h, w, c = x.shape
    out = torch.empty(B, (h // 2) * (w // 2), c * 4, dtype=BF16, device=x.device)
    _check(_lib.load().fluxhip_pack_latents_bf16(_p(x), _p(out), B, h, w, c, _stream()), "fluxhip_pack_latents_bf16")
    return out


def unpack_latents(x: torch.Tensor, h: int, w: int, scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    _bf16c(x, "x")
    B = x.shape[0]
    c = x.shape[-1] // 4
    out = torch.empty(B, h, w, c, dtype=BF16, device=x.device)
    _check(_lib.load().fluxhip_unpack_latents_bf16(_p(x), _p(out), B, h, w, c, float(scale), float(shift), _stream()),
           "fluxhip_unpack_latents_bf16")
    return out


_zero16 = {}


def _zeros16(device) -> torch.Tensor:
    z = _zero16.get(device)
    if z is None:
        z = torch.zeros(64, dtype=BF16, device=device)
        _zero16[device] = z
    return z


def conv2d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], stride: int = 1, pad: int = 1, ups: bool = False,
           res: Optional[torch.Tensor] = None, epi: int = EPI_BIAS, out: Optional[torch.Tensor] = None,
           addvec: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NHWC conv, w [Cout,kh,kw,Cin] (or [Cout,Cin] for 1x1). res => out = res + conv(x)."""
    f16 = _e16((x, "x"), (w, "w"), (b, "bias"), (res, "res"), (addvec, "addvec"), (out, "out"))
    B, Hs, Ws, Cin = x.shape
    Cout = w.shape[0]
    ks = 1 if w.dim() == 2 else w.shape[1]
    if ks == 1:
        pad = 0
    Hl, Wl = (Hs * 2, Ws * 2) if ups else (Hs, Ws)
    Ho, Wo = (Hl + 2 * pad - ks) // stride + 1, (Wl + 2 * pad - ks) // stride + 1
    if out is None:
        out = torch.empty(B, Ho, Wo, Cout, dtype=x.dtype, device=x.device)
    if res is not None:
        epi = EPI_GATE_RES
    lib = _lib.load()
    if Cin % 64 == 0 and Cout % 4 == 0:
        fn, name = _fn("fluxhip_conv2d_bf16", f16)
        _check(fn(_p(x), _p(w), _p(b), _p(res), _p(addvec), _p(out), B, Hs, Ws, Cin, Cout, ks, stride, pad,
                  int(ups), epi, _p(_zeros16(x.device)), _stream()), name)
    elif f16:
        raise FluxHipError("float16 convs need Cin % 64 == 0 and Cout % 4 == 0 (pad the channels)")
    else:
        if ks != 3 or stride != 1 or pad != 1 or ups or res is not None or addvec is not None:
            raise FluxHipError("small-channel conv path only supports 3x3/s1/p1")
        _check(lib.fluxhip_conv2d_small(_p(x), _p(w), _p(b), _p(out), B, Hs, Ws, Cin, Cout, 0, 0, _stream()),
               "fluxhip_conv2d_small")
    return out


def conv2d_out_image(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], clip01: bool) -> torch.Tensor:
    """Final decoder conv (Cout=3) -> float32 NHWC image, optionally clip(y+1,0,2)*0.5."""
    _bf16c(x, "x"); _bf16c(w, "w")
    B, H, W_, Cin = x.shape
    Cout = w.shape[0]
    out = torch.empty(B, H, W_, Cout, dtype=torch.float32, device=x.device)
    _check(_lib.load().fluxhip_conv2d_small(_p(x), _p(w), _p(b), _p(out), B, H, W_, Cin, Cout, 1, int(clip01),
                                             _stream()), "fluxhip_conv2d_small")
    return out


_gn_ws = {}
_gn_ws_retired = []


def _grow_gn_ws(old: Optional[torch.Tensor], need_bytes: int, device) -> torch.Tensor:
    """The GroupNorm scratch buffer grows GEOMETRICALLY (at least x2): a captured hipGraph may still launch kernels that
    point at an outgrown buffer, so those are kept — at most log2(max need) of them, together smaller than the live one.
    The total is therefore bounded by twice the largest request ever made, however many shapes a server sees."""
    if old is not None:
        _gn_ws_retired.append(old)
    n = max(need_bytes // 4, 1 << 18, 2 * (old.numel() if old is not None else 0))
    return torch.empty(n, dtype=torch.float32, device=device)


def groupnorm_silu(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32, eps: float = 1e-6,
                   silu: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    f16 = _e16((x, "x"), (gamma, "gamma"), (beta, "beta"), (out, "out"))
    B, H, W_, Cc = x.shape
    if out is None:
        out = torch.empty_like(x)
    ws = _gn_ws.get(x.device)
    need = (B * ((H * W_ + 31) // 32) * Cc + B * groups) * 2 * 4
    if ws is None or ws.numel() * 4 < need:
        ws = _grow_gn_ws(ws, need, x.device)
        _gn_ws[x.device] = ws
    fn, name = _fn("fluxhip_groupnorm_silu_bf16", f16)
    _check(fn(_p(x), _p(gamma), _p(beta), _p(out), B, H * W_, Cc, groups, eps, int(silu), _p(ws), ws.numel() * 4, _stream()),
           name)
    return out


def softmax_rows(s: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None,
                 cols: Optional[int] = None) -> torch.Tensor:
    """s float32 [..., ld] -> bf16 softmax(scale * s[..., :cols]) written to out[..., :cols]."""
    if s.dtype != torch.float32 or not s.is_contiguous():
        raise FluxHipError("s must be contiguous float32")
    ld = s.shape[-1]
    cols = ld if cols is None else cols
    rows = s.numel() // ld
    if out is None:
        out = torch.zeros(s.shape, dtype=BF16, device=s.device)
    _check(_lib.load().fluxhip_softmax_rows_f32(_p(s), _p(out), rows, cols, ld, float(scale), _stream()),
           "fluxhip_softmax_rows_f32")
    return out


# ------------------------------------------------------------------------------------------------ fp8 path
FP8_MAX = 448.0      # largest finite OCP e4m3fn value


def quantize_rows_fp8(x: torch.Tensor, out: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None):
    """x [rows, K] (bf16 or float32, rows contiguous) -> (q uint8 e4m3fn [rows, K], scale float32 [rows]) with
    scale = max|row| / 448 and q = rne(x / scale): per-token scales for activations, per-output-channel for weights."""
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype not in (BF16, torch.float32):
        raise FluxHipError("quantize_rows_fp8 takes a 2-D bf16 / float32 tensor with contiguous rows")
    rows, K = x.shape
    if out is None:
        out = torch.empty(rows, K, dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty(rows, dtype=torch.float32, device=x.device)
    fn = _lib.load().fluxhip_quantize_rows_fp8 if x.dtype == BF16 else _lib.load().fluxhip_quantize_rows_fp8_f32
    _check(fn(_p(x), _p(out), _p(scale), rows, K, x.stride(0), _stream()), "fluxhip_quantize_rows_fp8")
    return out, scale


def make_fp8_scales(a_scales, w_scales, a_scale_bstride: int = 0) -> Fp8Scales:
    """a_scales / w_scales: raw device addresses per group."""
    sc = Fp8Scales()
    for i, (a, w) in enumerate(zip(a_scales, w_scales)):
        sc.a_scale[i], sc.w_scale[i] = a, w
    sc.a_scale_bstride = a_scale_bstride
    return sc


def gemm_fp8(desc: GemmDesc, scales: Fp8Scales) -> None:
    _check(_lib.load().fluxhip_gemm_fp8(desc, scales, _stream()), "fluxhip_gemm_fp8")


def linear_fp8(xq: torch.Tensor, x_scale: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor,
               b: Optional[torch.Tensor] = None, epi: int = EPI_BIAS, out: Optional[torch.Tensor] = None,
               res: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None, tile_cfg: int = 0) -> torch.Tensor:
    """y = epi((xq * x_scale[:, None]) @ (wq * w_scale[:, None]).T + b) on the fp8 matrix cores; y bf16."""
    M, K = xq.shape
    N = wq.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=xq.device)
    g = dict(A=_p(xq), W=_p(wq), bias=_p(b), C=_p(out), res=_p(res), gate=_p(gate), M=M)
    gemm_fp8(make_gemm_desc([g], 1, N, K, K, N, epi, tile_cfg=tile_cfg), make_fp8_scales([_p(x_scale)], [_p(w_scale)]))
    return out


# block-scaled ("MX") fp8: e4m3 elements + one E8M0 scale per 32 consecutive columns, scale bytes tiled for the GEMM's K loop
# (include/fluxhip.h, fluxhip_fp8_mx)
def mx_scale_buffer(rows: int, K: int, device) -> torch.Tensor:
    """Scale buffer for an [rows, K] e4m3 matrix: uint8 [(K / 128) * rows * 4]; rows % 64 == 0, K % 128 == 0."""
    if rows % 64 or K % 128:
        raise FluxHipError("block-scale buffers cover whole 64-row groups and 128-element K-steps")
    return torch.full(((K // 128) * rows * 4,), 127, dtype=torch.uint8, device=device)


def mx_scale_index(rows_idx: torch.Tensor, kb: torch.Tensor, kstride: int) -> torch.Tensor:
    """Byte offset of the scale of (scale-buffer row, 32-column block kb) in the tiled layout."""
    return ((((kb >> 2) * kstride + (rows_idx >> 6) * 64 + (kb & 3) * 16 + (rows_idx & 15)) << 2) + ((rows_idx >> 4) & 3))


def quantize_mx_fp8(x: torch.Tensor, out: Optional[torch.Tensor] = None, mx: Optional[torch.Tensor] = None, col0: int = 0,
                    row0: int = 0, kstride: Optional[int] = None):
    """x bf16 [rows, K] -> (q uint8 e4m3 [rows, ld_out] written at columns [col0, col0 + K), tiled E8M0 scale bytes)."""
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype != BF16:
        raise FluxHipError("quantize_mx_fp8 takes a 2-D bf16 tensor with contiguous rows")
    rows, K = x.shape
    if out is None:
        out = torch.empty(rows, col0 + K, dtype=torch.uint8, device=x.device)
    if kstride is None:
        kstride = row0 + rows
    if mx is None:
        mx = mx_scale_buffer(kstride, out.shape[1], x.device)
    _check(_lib.load().fluxhip_quantize_mx_fp8(_p(x), _p(out), _p(mx), rows, K, x.stride(0), out.stride(0), col0, row0, kstride,
                                               _stream()), "fluxhip_quantize_mx_fp8")
    return out, mx


def make_fp8_mx(a_mx=None, a_row0=(0, 0), a_bstride=0, a_kstride=0, c8=None, c8_bstride=0, ldc8=0, c8_coloff=0, c_mx=None,
                c_row0=(0, 0), c_bstride=0, c_kstride=0) -> Fp8Mx:
    """Raw device addresses: consumer form (a_mx ...) or producer form (c8 per group, c_mx ...)."""
    m = Fp8Mx()
    m.a_mx, m.a_mx_bstride, m.a_mx_kstride = a_mx, a_bstride, a_kstride
    m.c_mx, m.c_mx_bstride, m.c_mx_kstride = c_mx, c_bstride, c_kstride
    m.c8_bstride, m.ldc8, m.c8_coloff = c8_bstride, ldc8, c8_coloff
    for i in range(2):
        m.a_mx_row0[i] = a_row0[min(i, len(a_row0) - 1)]
        m.c_mx_row0[i] = c_row0[min(i, len(c_row0) - 1)]
    for i, c in enumerate(c8 or ()):
        m.c8[i] = c
    return m


def gemm_fp8_mx(desc: GemmDesc, scales: Fp8Scales, mx: Fp8Mx) -> None:
    _check(_lib.load().fluxhip_gemm_fp8_mx(desc, scales, mx, _stream()), "fluxhip_gemm_fp8_mx")


def linear_fp8_mxa(xq: torch.Tensor, x_mx: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor,
                   b: Optional[torch.Tensor] = None, epi: int = EPI_BIAS, out: Optional[torch.Tensor] = None,
                   res: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None, tile_cfg: int = 0) -> torch.Tensor:
    """y = epi(dequant_mx(xq, x_mx) @ (wq * w_scale[:, None]).T + b): block-scaled activation operand; y bf16."""
    M, K = xq.shape
    N = wq.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=xq.device)
    g = dict(A=_p(xq), W=_p(wq), bias=_p(b), C=_p(out), res=_p(res), gate=_p(gate), M=M)
    gemm_fp8_mx(make_gemm_desc([g], 1, N, K, xq.stride(0), N, epi, tile_cfg=tile_cfg), make_fp8_scales([None], [_p(w_scale)]),
                make_fp8_mx(a_mx=_p(x_mx), a_kstride=M))
    return out


def linear_fp8_gelu_mxc(xq: torch.Tensor, x_scale: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor,
                        b: Optional[torch.Tensor] = None, tile_cfg: int = 0):
    """(q, mx) = quantize_mx(gelu_tanh((xq * x_scale) @ (wq * w_scale).T + b)): the quantisation runs in the GEMM's epilogue."""
    M, K = xq.shape
    N = wq.shape[0]
    q = torch.empty(M, N, dtype=torch.uint8, device=xq.device)
    mx = mx_scale_buffer(M, N, xq.device)
    dummy = torch.empty(8, dtype=BF16, device=xq.device)      # C is not written by the all-GELU form
    g = dict(A=_p(xq), W=_p(wq), bias=_p(b), C=_p(dummy), M=M)
    gemm_fp8_mx(make_gemm_desc([g], 1, N, K, K, N, EPI_GELU_TANH, tile_cfg=tile_cfg), make_fp8_scales([_p(x_scale)], [_p(w_scale)]),
                make_fp8_mx(c8=[_p(q)], ldc8=N, c_mx=_p(mx), c_kstride=M))
    return q, mx


# ------------------------------------------------------------------------------------------------ fp32-faithful VAE path
# A "split tensor" is a bf16 tensor of shape [2, ...]: plane 0 = hi = bf16(x), plane 1 = lo = bf16(x - hi), so that
# x = hi + lo to 2^-17 relative (include/fluxhip.h, "bf16x3").  Views that keep dim 0 (t[:, b]) stay split tensors.
F32 = torch.float32


def _split_ok(t: torch.Tensor, name: str) -> None:
    if t.dtype != BF16 or t.dim() < 2 or t.shape[0] != 2 or not t[0].is_contiguous() or not t.is_cuda:
        raise FluxHipError(f"{name} must be a split tensor: bf16 [2, ...] with contiguous planes on the device")


def _f32c(t: Optional[torch.Tensor], name: str) -> Optional[torch.Tensor]:
    if t is not None and (t.dtype != F32 or not t.is_contiguous()):
        raise FluxHipError(f"{name} must be a contiguous float32 tensor")
    return t


def split_f32(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float32 [...] -> split tensor [2, ...]."""
    _f32c(x, "x")
    if out is None:
        out = torch.empty(2, *x.shape, dtype=BF16, device=x.device)
    _check(_lib.load().fluxhip_split_f32(_p(x), _p(out[0]), _p(out[1]), x.numel(), _stream()), "fluxhip_split_f32")
    return out


def join_f32(t: torch.Tensor) -> torch.Tensor:
    """split tensor [2, ...] -> float32 [...] (hi + lo)."""
    _split_ok(t, "t")
    out = torch.empty(t.shape[1:], dtype=F32, device=t.device)
    _check(_lib.load().fluxhip_join_f32(_p(t[0]), _p(t[1]), _p(out), out.numel(), _stream()), "fluxhip_join_f32")
    return out


def gemm_x3(A: torch.Tensor, W: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, lda: int, ldc: int,
            bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, row_bias: bool = False,
            alpha: float = 1.0, out_f32: bool = False, tile_cfg: int = 0) -> None:
    """C = alpha * A W^T + bias [+ res] in the fp32-faithful mode.  A [2][M][lda], W [2][N][K] split; C split
    [2][M][ldc] or float32 [M][ldc] (out_f32); bias float32."""
    _split_ok(A, "A"); _split_ok(W, "W")
    d = GemmX3Desc()
    d.A, d.W, d.bias, d.res = _p(A[0]), _p(W[0]), _p(_f32c(bias, "bias")), (_p(res[0]) if res is not None else None)
    d.a_lo, d.w_lo = A.stride(0), W.stride(0)
    if out_f32:
        _f32c(C, "C")
        d.C, d.c_lo = _p(C), 0
    else:
        _split_ok(C, "C")
        d.C, d.c_lo = _p(C[0]), C.stride(0)
    if res is not None:
        _split_ok(res, "res")
        d.res_lo = res.stride(0)
    d.M, d.nbatch, d.N, d.K, d.lda, d.ldc = M, 1, N, K, lda, ldc
    d.epi, d.row_bias, d.out_f32, d.tile_cfg, d.alpha = (EPI_GATE_RES if res is not None else EPI_BIAS), int(row_bias), \
        int(out_f32), tile_cfg, alpha
    _check(_lib.load().fluxhip_gemm_x3(d, _stream()), "fluxhip_gemm_x3")


def linear_x3(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], res: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear in float32 on split tensors: x [2, ..., K], w [2, N, K], b float32 [N]; res adds a split residual."""
    K, N = x.shape[-1], w.shape[1]
    M = x[0].numel() // K
    if out is None:
        out = torch.empty(2, *x.shape[1:-1], N, dtype=BF16, device=x.device)
    gemm_x3(x, w, out, M, N, K, K, N, bias=b, res=res)
    return out


def _gn_scratch(B: int, hw: int, parities: int, C: int, device) -> torch.Tensor:
    """float32 scratch for the GroupNorm partials a conv epilogue emits: worst case 32-row wave tiles."""
    return torch.empty((B * parities * ((hw + 31) // 32) * C + B * 64) * 2, dtype=torch.float32, device=device)


def conv2d_x3(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], stride: int = 1, pad: int = 1, ups: bool = False,
              res: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, gn_stats: bool = False) -> torch.Tensor:
    """NHWC conv in float32 on split tensors: x [2,B,H,W,Cin], w [2,Cout,kh,kw,Cin] (or [2,Cout,Cin]), b float32.
    gn_stats: the epilogue also emits the GroupNorm partial sums of the output; they ride on the returned tensor
    (`out._gn = (scratch, nchunks)`) and groupnorm_silu_x3 picks them up instead of reading the tensor a second time."""
    _split_ok(x, "x"); _split_ok(w, "w")
    _, B, Hs, Ws, Cin = x.shape
    Cout = w.shape[1]
    ks = 1 if w.dim() == 3 else w.shape[2]
    if ks == 1:
        pad = 0
    Hl, Wl = (Hs * 2, Ws * 2) if ups else (Hs, Ws)
    Ho, Wo = (Hl + 2 * pad - ks) // stride + 1, (Wl + 2 * pad - ks) // stride + 1
    if out is None:
        out = torch.empty(2, B, Ho, Wo, Cout, dtype=BF16, device=x.device)
    if res is not None:
        _split_ok(res, "res")
    if hasattr(out, "_gn"):
        del out._gn                      # a reused output tensor must not keep the statistics of its previous contents
    gws = _gn_scratch(B, Ho * Wo, 1, Cout, x.device) if gn_stats else None
    nck = ctypes.c_int(0)
    _check(_lib.load().fluxhip_conv2d_x3(_p(x[0]), x.stride(0), _p(w[0]), w.stride(0), _p(_f32c(b, "bias")),
                                         _p(res[0]) if res is not None else None, res.stride(0) if res is not None else 0,
                                         _p(out[0]), out.stride(0), B, Hs, Ws, Cin, Cout, ks, stride, pad, int(ups),
                                         _p(gws) if gn_stats else None, gws.numel() * 4 if gn_stats else 0,
                                         ctypes.byref(nck) if gn_stats else None,
                                         _p(_zeros16(x.device)), _stream()), "fluxhip_conv2d_x3")
    if nck.value > 0:
        out._gn = (gws, nck.value, out._version)      # valid for exactly these contents of `out`
    return out


def subpixel_weights(w: torch.Tensor) -> torch.Tensor:
    """float32 3x3 weights [Cout,3,3,Cin] -> [4,Cout,2,2,Cin]: the four 2x2 kernels that (nearest-upsample x2 ->
    3x3 conv, pad 1) applies to the low-res input, one per output-pixel parity 2*dy + dx.  Taps reading the same
    source pixel are summed in float32 (dy = 0: rows {w0, w1+w2}; dy = 1: rows {w0+w1, w2}; same along x)."""
    if w.dtype != torch.float32 or w.dim() != 4 or w.shape[1:3] != (3, 3):
        raise ValueError("subpixel_weights: float32 [Cout,3,3,Cin] expected")
    rows = (torch.stack([w[:, 0], w[:, 1] + w[:, 2]], 1), torch.stack([w[:, 0] + w[:, 1], w[:, 2]], 1))   # [Cout,2,3,Cin]
    par = []
    for dy in (0, 1):
        r = rows[dy]
        par.append(torch.stack([r[:, :, 0], r[:, :, 1] + r[:, :, 2]], 2))
        par.append(torch.stack([r[:, :, 0] + r[:, :, 1], r[:, :, 2]], 2))
    return torch.stack(par, 0).contiguous()


def conv_up2x_x3(x: torch.Tensor, w4: torch.Tensor, b: Optional[torch.Tensor], out: Optional[torch.Tensor] = None,
                 gn_stats: bool = False) -> torch.Tensor:
    """Upsample (nearest x2) + 3x3 conv in float32 as four 2x2 convs of the low-res split tensor x [2,B,H,W,Cin];
    w4 = split_f32(subpixel_weights(w)) = [2,4,Cout,2,2,Cin]."""
    _split_ok(x, "x"); _split_ok(w4, "w4")
    _, B, Hs, Ws, Cin = x.shape
    if w4.dim() != 6 or w4.shape[1] != 4 or w4.shape[3:5] != (2, 2) or w4.shape[5] != Cin:
        raise ValueError("conv_up2x_x3: w4 must be [2,4,Cout,2,2,Cin]")
    Cout = w4.shape[2]
    if out is None:
        out = torch.empty(2, B, Hs * 2, Ws * 2, Cout, dtype=BF16, device=x.device)
    if hasattr(out, "_gn"):
        del out._gn
    gws = _gn_scratch(B, Hs * Ws, 4, Cout, x.device) if gn_stats else None
    nck = ctypes.c_int(0)
    _check(_lib.load().fluxhip_conv_up2x_x3(_p(x[0]), x.stride(0), _p(w4[0]), w4.stride(0), _p(_f32c(b, "bias")),
                                            _p(out[0]), out.stride(0), B, Hs, Ws, Cin, Cout,
                                            _p(gws) if gn_stats else None, gws.numel() * 4 if gn_stats else 0,
                                            ctypes.byref(nck) if gn_stats else None,
                                            _p(_zeros16(x.device)), _stream()), "fluxhip_conv_up2x_x3")
    if nck.value > 0:
        out._gn = (gws, nck.value, out._version)      # valid for exactly these contents of `out`
    return out


def groupnorm_silu_x3(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32, eps: float = 1e-6,
                      silu: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _split_ok(x, "x")
    _, B, H, W_, Cc = x.shape
    if out is None:
        out = torch.empty_like(x)
    st = getattr(x, "_gn", None)
    if st is not None and st[2] != x._version:
        st = None      # x was modified in place after the conv that produced it: its epilogue statistics are stale -> full pass
    if st is not None and (Cc // groups) % 4 == 0:     # partial sums (per channel quad) left by the epilogue of the conv that produced x: finalize + apply only
        gws, nck, _ = st
        _check(_lib.load().fluxhip_groupnorm_apply_x3(_p(x[0]), x.stride(0), _p(_f32c(gamma, "gamma")), _p(_f32c(beta, "beta")),
                                                      _p(out[0]), out.stride(0), B, H * W_, Cc, groups, eps, int(silu),
                                                      _p(gws), gws.numel() * 4, nck, _stream()), "fluxhip_groupnorm_apply_x3")
        return out
    ws = _gn_ws.get(x.device)
    need = (B * ((H * W_ + 31) // 32) * Cc + B * groups) * 2 * 4
    if ws is None or ws.numel() * 4 < need:
        ws = _grow_gn_ws(ws, need, x.device)
        _gn_ws[x.device] = ws
    _check(_lib.load().fluxhip_groupnorm_silu_x3(_p(x[0]), x.stride(0), _p(_f32c(gamma, "gamma")), _p(_f32c(beta, "beta")),
                                                 _p(out[0]), out.stride(0), B, H * W_, Cc, groups, eps, int(silu), _p(ws),
                                                 ws.numel() * 4, _stream()), "fluxhip_groupnorm_silu_x3")
    return out


def softmax_rows_x3(s: torch.Tensor, scale: float, out: torch.Tensor, cols: Optional[int] = None) -> torch.Tensor:
    """float32 logits [rows, ld] -> split probabilities out [2, rows, ld] (columns >= cols untouched)."""
    _f32c(s, "s"); _split_ok(out, "out")
    ld = s.shape[-1]
    cols = ld if cols is None else cols
    _check(_lib.load().fluxhip_softmax_rows_x3(_p(s), _p(out[0]), out.stride(0), s.numel() // ld, cols, ld, float(scale),
                                               _stream()), "fluxhip_softmax_rows_x3")
    return out


def conv2d_out_image_x3(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], clip01: bool) -> torch.Tensor:
    """Final decoder conv (Cout <= 4) of a split tensor with float32 weights -> float32 NHWC image."""
    _split_ok(x, "x"); _f32c(w, "w")
    _, B, H, W_, Cin = x.shape
    Cout = w.shape[0]
    out = torch.empty(B, H, W_, Cout, dtype=F32, device=x.device)
    _check(_lib.load().fluxhip_conv2d_small_x3(_p(x[0]), x.stride(0), _p(w), _p(_f32c(b, "bias")), _p(out), B, H, W_, Cin,
                                               Cout, int(clip01), _stream()), "fluxhip_conv2d_small_x3")
    return out


def unpack_latents_x3(x: torch.Tensor, h: int, w: int, scale: float, shift: float, cpad: int) -> torch.Tensor:
    """packed bf16 latents [B,L,4C] -> split [2,B,h,w,cpad] = unpack(x) / scale + shift in float32, zero-padded channels."""
    _bf16c(x, "x")
    B, c = x.shape[0], x.shape[-1] // 4
    out = torch.empty(2, B, h, w, cpad, dtype=BF16, device=x.device)
    _check(_lib.load().fluxhip_unpack_latents_x3(_p(x), _p(out[0]), out.stride(0), B, h, w, c, cpad, float(scale),
                                                 float(shift), _stream()), "fluxhip_unpack_latents_x3")
    return out


def pixel_linear_x3(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], pad_to: int, in_div: float) -> torch.Tensor:
    """x bf16 or float16 (the latents of a float16=True pipeline: x / in_div is then rounded to float16 like the reference's
    array division before the float32 Linear); output = split tensor."""
    f16 = _e16((x, "x")); _f32c(w, "w")
    Cin, Cout = x.shape[-1], w.shape[0]
    out = torch.empty(2, *x.shape[:-1], pad_to, dtype=BF16, device=x.device)
    fn, name = _fn("fluxhip_pixel_linear_x3", f16)
    _check(fn(_p(x), _p(w), _p(_f32c(b, "bias")), _p(out[0]), out.stride(0), x.numel() // Cin, Cin, Cout, pad_to,
              float(in_div), _stream()), name)
    return out


# ------------------------------------------------------------------------------------------------ UNet path
def attention_strided(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, B: int, H: int, hd: int,
                      Tq: int, Tk: int, Tkpad: int, q_strides, k_strides, ldo: int, scale: float) -> None:
    """q/k addressed as base + b*bs + h*hs + t*rs (element strides); vt [B][H*hd][Tkpad]."""
    f16 = _e16((q, "q"), (k, "k"), (vt, "vt"), (out, "out"), strided=True)     # one 16-bit type for all four: no silent reinterpretation
    fn, name = _fn("fluxhip_attention_strided_bf16", f16)
    _check(fn(_p(q), *q_strides, _p(k), *k_strides, _p(vt), _p(out), ldo, B, H, hd, Tq, Tk, Tkpad, float(scale), _stream()), name)


def attention_strided_vt(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, B: int, H: int, hd: int,
                         Tq: int, Tk: int, Tkpad: int, q_strides, k_strides, vt_bstride: int, ldo: int, scale: float,
                         k_off: int = 0, vt_off: int = 0) -> None:
    """attention_strided with an explicit V^T batch stride; k_off / vt_off = element offsets into k / vt (the UNet's cross-
    attention layers read their slice of ONE projected text image, stable_diffusion/unet.py `text_kv`)."""
    f16 = _e16((q, "q"), (k, "k"), (vt, "vt"), (out, "out"), strided=True)
    fn, name = _fn("fluxhip_attention_strided_vt_bf16", f16)
    _check(fn(_p(q), *q_strides, _p(k) + k_off * 2, *k_strides, _p(vt) + vt_off * 2, vt_bstride, _p(out), ldo, B, H, hd, Tq, Tk,
              Tkpad, float(scale), _stream()), name)


def layernorm_affine(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    f16 = _e16((x, "x"), (gamma, "gamma"), (beta, "beta"), (out, "out"))
    D = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    fn, name = _fn("fluxhip_layernorm_affine_bf16", f16)
    _check(fn(_p(x), _p(out), x.numel() // D, D, _p(gamma), _p(beta), eps, _stream()), name)
    return out


def concat_channels(a: torch.Tensor, b: Optional[torch.Tensor], pad_to: int = 0) -> torch.Tensor:
    """cat([a, b], -1) on [..., C] tensors; b=None zero-pads a to pad_to channels."""
    _e16((a, "a"), (b, "b"))                      # a 16-bit copy: one entry point serves both storage types
    Ca = a.shape[-1]
    Cb = b.shape[-1] if b is not None else pad_to - Ca
    out = torch.empty(*a.shape[:-1], Ca + Cb, dtype=a.dtype, device=a.device)
    _check(_lib.load().fluxhip_concat_channels_bf16(_p(a), _p(b), _p(out), a.numel() // Ca, Ca, Cb, _stream()),
           "fluxhip_concat_channels_bf16")
    return out


def axpbypcz(x: torch.Tensor, y: torch.Tensor, z: Optional[torch.Tensor], ca: float, cb: float, cc: float = 0.0,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if x.dtype == torch.float32:            # float32 latents (float16=False pipelines)
        return axpbypcz_f32(x, y, z, ca, cb, cc, out=out)
    f16 = _e16((x, "x"), (y, "y"), (z, "z"), (out, "out"))
    if out is None:
        out = torch.empty_like(x)
    fn, name = _fn("fluxhip_axpbypcz_bf16", f16)
    _check(fn(_p(x), _p(y), _p(z), _p(out), x.numel(), float(ca), float(cb), float(cc), _stream()), name)
    return out


def axpbypcz_dev(x: torch.Tensor, y: torch.Tensor, z: Optional[torch.Tensor], coef: torch.Tensor,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = coef[0]*x + coef[1]*y + coef[2]*z with coef a float32[3] DEVICE tensor (graph-replayable sampler step)."""
    if x.dtype == torch.float32:
        return axpbypcz_f32(x, y, z, coef=coef, out=out)
    f16 = _e16((x, "x"), (y, "y"), (z, "z"), (out, "out"))
    if coef.dtype != torch.float32 or coef.numel() < 3 or not coef.is_cuda:
        raise FluxHipError("coef must be a float32[3] device tensor")
    if out is None:
        out = torch.empty_like(x)
    fn, name = _fn("fluxhip_axpbypcz_dev_bf16", f16)
    _check(fn(_p(x), _p(y), _p(z), _p(out), x.numel(), _p(coef), _stream()), name)
    return out


def pixel_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], pad_to: int, in_div: float) -> torch.Tensor:
    _bf16c(x, "x"); _bf16c(w, "w")
    Cin, Cout = x.shape[-1], w.shape[0]
    out = torch.empty(*x.shape[:-1], pad_to, dtype=BF16, device=x.device)
    _check(_lib.load().fluxhip_pixel_linear_bf16(_p(x), _p(w), _p(b), _p(out), x.numel() // Cin, Cin, Cout, pad_to,
                                                 float(in_div), _stream()), "fluxhip_pixel_linear_bf16")
    return out


def sincos_embed(x: torch.Tensor, sig: torch.Tensor, dtype=BF16) -> torch.Tensor:
    """x float32 [n], sig float32 [half] -> `dtype` (bf16 / float16) [n, 2*half] = [cos | sin]."""
    if x.dtype != torch.float32 or sig.dtype != torch.float32:
        raise FluxHipError("sincos_embed takes float32 inputs")
    n, half = x.numel(), sig.numel()
    out = torch.empty(n, 2 * half, dtype=dtype, device=x.device)
    fn, name = _fn("fluxhip_sincos_embed_f32", dtype == F16)
    _check(fn(_p(x.contiguous()), _p(sig.contiguous()), _p(out), n, half, _stream()), name)
    return out


# ------------------------------------------------------------------------------------------------ text encoders
def attention_masked(q, k, vt, out, B: int, H: int, Tq: int, Tk: int, Tkpad: int, q_strides, k_strides, ldo: int,
                     scale: float, bias: Optional[torch.Tensor] = None, causal: bool = False) -> None:
    f16 = _e16((q, "q"), (k, "k"), (vt, "vt"), (out, "out"), (bias, "bias"), strided=True)
    fn, name = _fn("fluxhip_attention_masked_bf16", f16)
    _check(fn(_p(q), *q_strides, _p(k), *k_strides, _p(vt), _p(out), ldo, B, H, Tq, Tk, Tkpad, float(scale), _p(bias),
              int(causal), _stream()), name)


def rmsnorm(x: torch.Tensor, gamma: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _bf16c(x, "x")
    D = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    _check(_lib.load().fluxhip_rmsnorm_bf16(_p(x), _p(out), x.numel() // D, D, _p(gamma), float(eps), _stream()),
           "fluxhip_rmsnorm_bf16")
    return out


def embedding(idx: torch.Tensor, table: torch.Tensor, pos: Optional[torch.Tensor] = None) -> torch.Tensor:
    """idx int32 [..., T] -> bf16 [..., T, D]; pos [>=T, D] is added per position when given."""
    if idx.dtype != torch.int32 or not idx.is_contiguous():
        raise FluxHipError("idx must be contiguous int32")
    f16 = _e16((table, "table"), (pos, "pos"))
    D = table.shape[1]
    out = torch.empty(*idx.shape, D, dtype=table.dtype, device=table.device)
    T = idx.shape[-1] if pos is not None else 0
    fn, name = _fn("fluxhip_embedding_bf16", f16)
    _check(fn(_p(idx), _p(table), _p(pos), _p(out), idx.numel(), D, T, table.shape[0], _stream()), name)
    return out


# ------------------------------------------------------------------------------------------------ float32 arithmetic (float16=False) of the
# stable_diffusion/ UNet and text towers: split tensors [2, ...] (hi, lo bf16 planes), include/fluxhip.h "ABI 9"
ACT_SILU, ACT_GELU_ERF, ACT_QUICK_GELU, ACT_GEGLU = 0, 1, 2, 3


def _split_rows_ok(t: torch.Tensor, name: str) -> None:
    """A split tensor whose planes may be row-strided VIEWS (last dim contiguous)."""
    if t.dtype != BF16 or t.dim() < 2 or t.shape[0] != 2 or t.stride(-1) != 1 or not t.is_cuda:
        raise FluxHipError(f"{name} must be a split tensor: bf16 [2, ...] planes with a contiguous last dimension")


def layernorm_x3(x: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor], eps: float = 1e-5,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.LayerNorm over the last dim of a split tensor [2, ..., D] in float32 (gamma / beta float32)."""
    _split_ok(x, "x")
    D = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    _check(_lib.load().fluxhip_layernorm_x3(_p(x[0]), x.stride(0), _p(_f32c(gamma, "gamma")), _p(_f32c(beta, "beta")), _p(out[0]),
                                            out.stride(0), x[0].numel() // D, D, float(eps), _stream()), "fluxhip_layernorm_x3")
    return out


def act_x3(a: torch.Tensor, mode: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Elementwise activation of a split tensor [2, ..., C] in float32; ACT_GEGLU: a is [2, ..., 2C] = [value | gate] and the
    result [2, ..., C] = value * gelu_erf(gate)."""
    _split_ok(a, "a")
    ld = a.shape[-1]
    cols = ld // 2 if mode == ACT_GEGLU else ld
    rows = a[0].numel() // ld
    if out is None:
        out = torch.empty(2, *a.shape[1:-1], cols, dtype=BF16, device=a.device)
    _check(_lib.load().fluxhip_act_x3(_p(a[0]), a.stride(0), ld, _p(out[0]), out.stride(0), cols, rows, cols, int(mode),
                                      cols if mode == ACT_GEGLU else 0, _stream()), "fluxhip_act_x3")
    return out


def addvec_x3(x: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """x [2, B, ..., C] += v [2, B, C] (broadcast over the pixels of image b), in place."""
    _split_ok(x, "x"); _split_ok(v, "v")
    B, C = x.shape[1], x.shape[-1]
    _check(_lib.load().fluxhip_addvec_x3(_p(x[0]), x.stride(0), _p(v[0]), v.stride(0), B, x[0].numel() // (B * C), C, _stream()),
           "fluxhip_addvec_x3")
    return x


def sincos_embed_x3(x: torch.Tensor, sig: torch.Tensor) -> torch.Tensor:
    """x float32 [n], sig float32 [half] -> split [2, n, 2 half] = [cos | sin] in float32."""
    if x.dtype != torch.float32 or sig.dtype != torch.float32:
        raise FluxHipError("sincos_embed_x3 takes float32 inputs")
    n, half = x.numel(), sig.numel()
    out = torch.empty(2, n, 2 * half, dtype=BF16, device=x.device)
    _check(_lib.load().fluxhip_sincos_embed_x3(_p(x.contiguous()), _p(sig.contiguous()), _p(out[0]), out.stride(0), n, half,
                                               _stream()), "fluxhip_sincos_embed_x3")
    return out


def axpbypcz_f32(x: torch.Tensor, y: torch.Tensor, z: Optional[torch.Tensor], ca: float = 0.0, cb: float = 0.0, cc: float = 0.0,
                 coef: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = ca x + cb y + cc z on contiguous float32 tensors; coef: float32[3] device tensor overriding (ca, cb, cc)."""
    for t, n in ((x, "x"), (y, "y"), (z, "z"), (out, "out")):
        _f32c(t, n)
    if coef is not None and (coef.dtype != torch.float32 or coef.numel() < 3 or not coef.is_cuda):
        raise FluxHipError("coef must be a float32[3] device tensor")
    if out is None:
        out = torch.empty_like(x)
    _check(_lib.load().fluxhip_axpbypcz_f32(_p(x), _p(y), _p(z), _p(out), x.numel(), float(ca), float(cb), float(cc), _p(coef),
                                            _stream()), "fluxhip_axpbypcz_f32")
    return out


def softmax_rows_masked_x3(s: torch.Tensor, scale: float, out: torch.Tensor, cols: Optional[int] = None, causal_T: int = 0):
    """s float32 [..., ld] -> split probabilities out [2, ..., ld]: softmax(scale * s[..., :cols]); causal_T > 0: row r of the
    flattened [..., causal_T, ld] logits sees columns [0, r % causal_T]; every other column is written as zero."""
    _f32c(s, "s"); _split_ok(out, "out")
    ld = s.shape[-1]
    _check(_lib.load().fluxhip_softmax_rows_masked_x3(_p(s), _p(out[0]), out.stride(0), s.numel() // ld, ld if cols is None else cols,
                                                      ld, float(scale), int(causal_T), _stream()), "fluxhip_softmax_rows_masked_x3")
    return out


def embedding_x3(idx: torch.Tensor, table: torch.Tensor, pos: Optional[torch.Tensor] = None) -> torch.Tensor:
    """idx int32 [..., T], float32 table [V, D] (+ float32 pos [T, D]) -> split [2, ..., T, D]."""
    _f32c(table, "table"); _f32c(pos, "pos")
    if idx.dtype != torch.int32 or not idx.is_contiguous():
        raise FluxHipError("idx must be contiguous int32")
    V, D = table.shape
    out = torch.empty(2, *idx.shape, D, dtype=BF16, device=table.device)
    _check(_lib.load().fluxhip_embedding_x3(_p(idx), _p(table), _p(pos), _p(out[0]), out.stride(0), idx.numel(), D,
                                            idx.shape[-1] if pos is not None else 1, V, _stream()), "fluxhip_embedding_x3")
    return out


def pixel_linear_x3_f32in(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], pad_to: int, in_div: float) -> torch.Tensor:
    """pixel_linear_x3 on float32 latents (float16=False pipelines)."""
    _f32c(x, "x"); _f32c(w, "w")
    Cin, Cout = x.shape[-1], w.shape[0]
    out = torch.empty(2, *x.shape[:-1], pad_to, dtype=BF16, device=x.device)
    _check(_lib.load().fluxhip_pixel_linear_x3_f32in(_p(x), _p(w), _p(_f32c(b, "bias")), _p(out[0]), out.stride(0), x.numel() // Cin,
                                                     Cin, Cout, pad_to, float(in_div), _stream()), "fluxhip_pixel_linear_x3_f32in")
    return out


def gemm_x3_batched(A: torch.Tensor, W: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, lda: int, ldc: int, nbatch: int,
                    a_bstride: int, w_bstride: int, c_bstride: int, bias: Optional[torch.Tensor] = None, row_bias: bool = False,
                    alpha: float = 1.0, out_f32: bool = False) -> None:
    """`gemm_x3` over `nbatch` problems that differ by element strides of A / W / C (0 = shared): the per-head products of the
    float32 multi-head attention - Q_h K_h^T with a_bstride = w_bstride = head_dim (the heads are column blocks of the [T, C]
    projections), P_h V_h^T with c_bstride = head_dim.  A / W / C may be row-strided views of split tensors."""
    _split_rows_ok(A, "A"); _split_rows_ok(W, "W")
    d = GemmX3Desc()
    d.A, d.W, d.bias, d.res = _p(A[0]), _p(W[0]), _p(_f32c(bias, "bias")), None
    d.a_lo, d.w_lo = A.stride(0), W.stride(0)
    if out_f32:
        if C.dtype != F32 or C.stride(-1) != 1:
            raise FluxHipError("C must be float32 with a contiguous last dimension")
        d.C, d.c_lo = _p(C), 0
    else:
        _split_rows_ok(C, "C")
        d.C, d.c_lo = _p(C[0]), C.stride(0)
    d.M, d.nbatch, d.N, d.K, d.lda, d.ldc = M, nbatch, N, K, lda, ldc
    d.a_bstride, d.w_bstride, d.c_bstride = a_bstride, w_bstride, c_bstride
    d.epi, d.row_bias, d.out_f32, d.tile_cfg, d.alpha = EPI_BIAS, int(row_bias), int(out_f32), 0, alpha
    _check(_lib.load().fluxhip_gemm_x3(d, _stream()), "fluxhip_gemm_x3")


def attention_x3(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, H: int, Tk: int, scale: float, causal: bool = False,
                 max_logit_bytes: int = 1 << 30) -> torch.Tensor:
    """Multi-head attention (head_dim 64) in float32 arithmetic on split tensors, ALL images and heads per launch:
    q [2,B,N,C] and k [2,B,Tkp,C] token-major projections, vt [2,B,C,Tkpad] = V transposed (zero beyond Tkp); returns
    o [2,B,N,C].  Q and K are copied head-major once ([B,H,T,64]: a permutation, no arithmetic) so that (image, head) is ONE
    batch index with uniform strides; then logits Q_h K_h^T (float32 [B*H, N, Tkpad]), float32 softmax over the first Tk keys
    (causal: row t sees keys [0, t]) with split probabilities, P_h V_h^T (the V^T image of (b, h) sits at (b H + h) * 64 * Tkpad
    already), and the heads copied back side by side.  Images are processed in chunks whose logits stay under max_logit_bytes."""
    _split_ok(q, "q"); _split_ok(k, "k"); _split_ok(vt, "vt")
    _, B, N, C = q.shape
    Tkp, Tkpad = k.shape[2], vt.shape[3]
    if C != 64 * H or k.shape[3] != C or vt.shape[2] != C or Tkpad % 64 or Tkp > Tkpad or Tkp % 4:
        raise FluxHipError("attention_x3: head_dim 64, k [2,B,Tkp,C], vt [2,B,C,Tkpad] with Tkpad % 64 == 0 and Tkp % 4 == 0")
    dev = q.device
    o = torch.empty(2, B, N, C, dtype=BF16, device=dev)
    nb = max(1, min(B, max_logit_bytes // max(1, H * N * Tkpad * 4)))
    s = torch.empty(nb * H, N, Tkpad, dtype=F32, device=dev)
    pm = torch.empty(2, nb * H, N, Tkpad, dtype=BF16, device=dev)
    for b0 in range(0, B, nb):
        n = min(nb, B - b0)
        q_hm = q[:, b0:b0 + n].reshape(2, n, N, H, 64).permute(0, 1, 3, 2, 4).contiguous()        # [2,n,H,N,64]
        k_hm = k[:, b0:b0 + n].reshape(2, n, Tkp, H, 64).permute(0, 1, 3, 2, 4).contiguous()      # [2,n,H,Tkp,64]
        o_hm = torch.empty(2, n, H, N, 64, dtype=BF16, device=dev)
        gemm_x3_batched(q_hm, k_hm, s, N, Tkp, 64, 64, Tkpad, n * H, N * 64, Tkp * 64, N * Tkpad, out_f32=True)
        softmax_rows_masked_x3(s[: n * H], scale, pm[:, : n * H], cols=Tk, causal_T=N if causal else 0)
        gemm_x3_batched(pm, vt[:, b0:b0 + n], o_hm, N, 64, Tkpad, Tkpad, 64, n * H, N * Tkpad, 64 * Tkpad, N * 64)
        o[:, b0:b0 + n].copy_(o_hm.permute(0, 1, 3, 2, 4).reshape(2, n, N, C))
    return o

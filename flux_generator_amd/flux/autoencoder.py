"""Flux 16-channel VAE decoder on MI355X (mirror of the reference's flux/autoencoder.py decode path).

NHWC bf16 activations; every conv is the libfluxhip implicit-GEMM kernel (MFMA, K = 9*Cin), with
the nearest-2x upsample folded into the conv's loader, the residual add folded into the second
conv's epilogue, GroupNorm+SiLU as a two-launch HBM-bound pair, and the single-head 512-wide
AttnBlock as QK^T / softmax / PV GEMMs.  Encoder / DiagonalGaussian (training, img2img) are out
of the hot-path scope.

Precision note: the reference runs this decoder in fp32 (checkpoint dtype).  Here weights and
activations are bf16 with fp32 accumulation and fp32 GroupNorm/softmax statistics; the tolerance of
that deliberate change is stated in tests/test_vae_gpu.py.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Iterable, List, Tuple, Union

import torch

from .. import _lib, ops
from ..ops import EPI_BIAS, EPI_GATE_RES, FluxHipError, make_gemm_desc

BF16 = torch.bfloat16


@dataclass
class AutoEncoderParams:
    """flux/autoencoder.py:11-21."""
    resolution: int
    in_channels: int
    ch: int
    out_ch: int
    ch_mult: List[int]
    num_res_blocks: int
    z_channels: int
    scale_factor: float
    shift_factor: float


def decoder_weight_shapes(A: AutoEncoderParams) -> Dict[str, Tuple[int, ...]]:
    """Parameter tree of Decoder (flux/autoencoder.py:212-269) in the sanitized layout
    (conv weights [O,kh,kw,I], 1x1 convs squeezed to Linear [O,I]; flux/autoencoder.py:336-345)."""
    s: Dict[str, Tuple[int, ...]] = {}

    def conv(name, o, i, k=3):
        s[f"{name}.weight"] = (o, k, k, i)
        s[f"{name}.bias"] = (o,)

    def res(name, i, o):
        s[f"{name}.norm1.weight"] = (i,)
        s[f"{name}.norm1.bias"] = (i,)
        conv(f"{name}.conv1", o, i)
        s[f"{name}.norm2.weight"] = (o,)
        s[f"{name}.norm2.bias"] = (o,)
        conv(f"{name}.conv2", o, o)
        if i != o:
            s[f"{name}.nin_shortcut.weight"] = (o, i)
            s[f"{name}.nin_shortcut.bias"] = (o,)

    nres = len(A.ch_mult)
    block_in = A.ch * A.ch_mult[nres - 1]
    conv("decoder.conv_in", block_in, A.z_channels)
    res("decoder.mid.block_1", block_in, block_in)
    s["decoder.mid.attn_1.norm.weight"] = (block_in,)
    s["decoder.mid.attn_1.norm.bias"] = (block_in,)
    for n in ("q", "k", "v", "proj_out"):
        s[f"decoder.mid.attn_1.{n}.weight"] = (block_in, block_in)
        s[f"decoder.mid.attn_1.{n}.bias"] = (block_in,)
    res("decoder.mid.block_2", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = A.ch * A.ch_mult[lvl]
        for i in range(A.num_res_blocks + 1):
            res(f"decoder.up.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", block_in, block_in)
    s["decoder.norm_out.weight"] = (block_in,)
    s["decoder.norm_out.bias"] = (block_in,)
    conv("decoder.conv_out", A.out_ch, block_in)
    return s


class AutoEncoder:
    def __init__(self, params: AutoEncoderParams, device: Union[str, torch.device] = "cuda"):
        self.params = params
        self.scale_factor = params.scale_factor
        self.shift_factor = params.shift_factor
        if torch.device(device).type != "cuda":
            raise FluxHipError("AutoEncoder needs a HIP device: there is no CPU fallback for the decode path")
        self.device = _lib.bind_device(device)
        _lib.load()
        self._params = {k: torch.empty(*shp, dtype=BF16, device=self.device)
                        for k, shp in decoder_weight_shapes(params).items()}

    def parameters(self) -> Dict[str, torch.Tensor]:
        return self._params

    def init_random(self, seed: int = 0) -> "AutoEncoder":
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, t in self._params.items():
            base = name.rsplit(".", 1)[0]
            wt = self._params[f"{base}.weight"]
            if wt.dim() == 1:       # GroupNorm affine
                t.fill_(1.0 if name.endswith(".weight") else 0.0)
                continue
            k = 1.0 / math.sqrt(wt[0].numel())
            t.copy_(((torch.rand(t.shape, generator=g, device=self.device) * 2 - 1) * k).to(BF16))
        return self

    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Checkpoint layout mapping (flux/autoencoder.py:336-345): conv [O,I,kh,kw] -> [O,kh,kw,I],
        1x1 convs squeezed to [O,I]."""
        new = {}
        for k, w in weights.items():
            if w.ndim == 4:
                w = w.permute(0, 2, 3, 1).contiguous()
                if w.shape[1:3] == (1, 1):
                    w = w.squeeze(1).squeeze(1)
            new[k] = w
        return new

    def load_weights(self, weights: Union[Dict[str, torch.Tensor], Iterable[Tuple[str, torch.Tensor]]],
                     strict: bool = True) -> "AutoEncoder":
        items = weights.items() if isinstance(weights, dict) else weights
        seen = set()
        for k, w in items:
            if k not in self._params:
                if k.startswith("encoder.") or not strict:
                    continue        # the encoder is not part of the decode hot path
                raise ValueError(f"Unexpected parameter {k}")
            dst = self._params[k]
            if tuple(dst.shape) != tuple(w.shape):
                raise ValueError(f"Shape mismatch for {k}: expected {tuple(dst.shape)}, got {tuple(w.shape)}")
            dst.copy_(w.to(device=self.device, dtype=BF16))
            seen.add(k)
        if strict and set(self._params) - seen:
            raise ValueError(f"Missing parameters: {sorted(set(self._params) - seen)[:5]} ...")
        return self

    # ------------------------------------------------------------------ blocks
    def _resnet(self, p: str, x: torch.Tensor) -> torch.Tensor:
        """ResnetBlock.__call__ (flux/autoencoder.py:83-98)."""
        W = self._params
        h = ops.groupnorm_silu(x, W[f"{p}.norm1.weight"], W[f"{p}.norm1.bias"], 32, 1e-6, True)
        h = ops.conv2d(h, W[f"{p}.conv1.weight"], W[f"{p}.conv1.bias"])
        h = ops.groupnorm_silu(h, W[f"{p}.norm2.weight"], W[f"{p}.norm2.bias"], 32, 1e-6, True)
        if f"{p}.nin_shortcut.weight" in W:
            x = ops.conv2d(x, W[f"{p}.nin_shortcut.weight"], W[f"{p}.nin_shortcut.bias"])
        return ops.conv2d(h, W[f"{p}.conv2.weight"], W[f"{p}.conv2.bias"], res=x)

    def _attn(self, p: str, x: torch.Tensor) -> torch.Tensor:
        """AttnBlock.__call__ (flux/autoencoder.py:42-52): one 512-wide head over H*W tokens."""
        W = self._params
        B, H, Wd, C = x.shape
        N = H * Wd
        y = ops.groupnorm_silu(x, W[f"{p}.norm.weight"], W[f"{p}.norm.bias"], 32, 1e-6, False)
        q = ops.linear(y.view(B, N, C), W[f"{p}.q.weight"], W[f"{p}.q.bias"])
        k = ops.linear(y.view(B, N, C), W[f"{p}.k.weight"], W[f"{p}.k.bias"])
        out = torch.empty_like(x)
        Np = (N + 63) // 64 * 64      # the PV contraction runs over Np keys (zero padded)
        vt = torch.zeros(C, Np, dtype=BF16, device=x.device)
        s = torch.empty(N, Np, dtype=torch.float32, device=x.device)
        pm = torch.zeros(N, Np, dtype=BF16, device=x.device)
        o = torch.empty(N, C, dtype=BF16, device=x.device)
        for b in range(B):
            yb, qb, kb = y[b].view(N, C), q[b], k[b]
            # V^T[C,N] = Wv y^T + bv (row bias): the PV product then needs no transpose
            ops.gemm(make_gemm_desc([dict(A=W[f"{p}.v.weight"].data_ptr(), W=yb.data_ptr(), bias=W[f"{p}.v.bias"].data_ptr(),
                                          C=vt.data_ptr(), M=C)], 1, N, C, C, Np, EPI_BIAS, row_bias=True))
            ops.gemm(make_gemm_desc([dict(A=qb.data_ptr(), W=kb.data_ptr(), C=s.data_ptr(), M=N)], 1, N, C, C, Np,
                                    EPI_BIAS, out_f32=True))
            ops.softmax_rows(s, C ** -0.5, out=pm, cols=N)
            ops.gemm(make_gemm_desc([dict(A=pm.data_ptr(), W=vt.data_ptr(), C=o.data_ptr(), M=N)], 1, C, Np, Np, C))
            ops.linear(o, W[f"{p}.proj_out.weight"], W[f"{p}.proj_out.bias"], epi=EPI_GATE_RES,
                       out=out[b].view(N, C), res=x[b].view(N, C))
        return out

    def _conv_in(self, z: torch.Tensor) -> torch.Tensor:
        """decoder.conv_in (16 -> 512 channels).  16 input channels are below the implicit-GEMM loader's
        64-channel K-step, so z and the weight are zero-padded to 64 channels (exact: the extra products are 0)
        and the layer runs on the MFMA path like every other conv."""
        W = self._params
        w = W["decoder.conv_in.weight"]
        cin = w.shape[-1]
        if cin % 64 == 0:
            return ops.conv2d(z, w, W["decoder.conv_in.bias"])
        pad = 64 - cin % 64
        key = (w.data_ptr(), w._version)
        if getattr(self, "_conv_in_key", None) != key:
            self._conv_in_w = torch.nn.functional.pad(w, (0, pad)).contiguous()
            self._conv_in_key = key
        return ops.conv2d(torch.nn.functional.pad(z, (0, pad)), self._conv_in_w, W["decoder.conv_in.bias"])

    def _decoder(self, z: torch.Tensor, clip01: bool) -> torch.Tensor:
        """Decoder.__call__ (flux/autoencoder.py:271-297). z NHWC bf16 -> float32 NHWC image."""
        A, W = self.params, self._params
        nres = len(A.ch_mult)
        h = self._conv_in(z)
        h = self._resnet("decoder.mid.block_1", h)
        h = self._attn("decoder.mid.attn_1", h)
        h = self._resnet("decoder.mid.block_2", h)
        for lvl in reversed(range(nres)):
            for i in range(A.num_res_blocks + 1):
                h = self._resnet(f"decoder.up.{lvl}.block.{i}", h)
            if lvl != 0:   # Upsample: nearest x2 fused into the conv loader (flux/autoencoder.py:120-123)
                h = ops.conv2d(h, W[f"decoder.up.{lvl}.upsample.conv.weight"], W[f"decoder.up.{lvl}.upsample.conv.bias"],
                               ups=True)
        h = ops.groupnorm_silu(h, W["decoder.norm_out.weight"], W["decoder.norm_out.bias"], 32, 1e-6, True)
        return ops.conv2d_out_image(h, W["decoder.conv_out.weight"], W["decoder.conv_out.bias"], clip01)

    # ------------------------------------------------------------------ public surface
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """AutoEncoder.decode (flux/autoencoder.py:352-354): z [B,h,w,16] -> [B,8h,8w,3] float32."""
        z = z.to(BF16).contiguous()
        B, h, w, c = z.shape
        # z / scale_factor + shift_factor rides on the unpack kernel (pack then unpack = identity permutation)
        zz = ops.unpack_latents(ops.pack_latents(z), h, w, self.scale_factor, self.shift_factor)
        return self._decoder(zz, clip01=False)

    def decode_packed(self, x: torch.Tensor, latent_size: Tuple[int, int]) -> torch.Tensor:
        """FluxPipeline.decode fused (flux/flux.py:157-162): packed latents [B,L,64] ->
        clip(decode(unpack(x)) + 1, 0, 2) * 0.5, float32 NHWC in [0,1]."""
        h, w = latent_size
        z = ops.unpack_latents(x.to(BF16).contiguous(), h, w, self.scale_factor, self.shift_factor)
        return self._decoder(z, clip01=True)

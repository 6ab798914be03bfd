"""Flux 16-channel VAE decoder on MI355X (mirror of the reference's flux/autoencoder.py decode path).

NHWC bf16 activations; every conv is the libfluxhip implicit-GEMM kernel (MFMA, K = 9*Cin), with
the nearest-2x upsample folded into the conv's loader, the residual add folded into the second
conv's epilogue, GroupNorm+SiLU as a two-launch HBM-bound pair, and the single-head 512-wide
AttnBlock as QK^T / softmax / PV GEMMs.  Encoder / DiagonalGaussian (training, img2img) are out
of the hot-path scope.

Precision: the reference runs this decoder in fp32 (checkpoint dtype, flux/utils.py:137-143).  The default
here is the fp32-faithful split-bf16 path (3 MFMA passes per product, fp32 accumulation / norms / softmax);
a bf16-storage mode remains as an opt-in.  Tolerances of both are stated in tests/test_vae_gpu.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Tuple, Union

import torch

from .. import _lib, ops
from .. import vae_common as V
from ..ops import FluxHipError

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
    """precision = "fp32" (default) is the reference's arithmetic, run on the fp32-faithful split-bf16 kernels;
    "bf16" is the bf16-storage opt-in (see vae_common.py)."""

    def __init__(self, params: AutoEncoderParams, device: Union[str, torch.device] = "cuda", precision: str = "fp32"):
        V.check_precision(precision)
        self.params = params
        self.scale_factor = params.scale_factor
        self.shift_factor = params.shift_factor
        self.precision = precision
        if torch.device(device).type != "cuda":
            raise FluxHipError("AutoEncoder needs a HIP device: there is no CPU fallback for the decode path")
        self.device = _lib.bind_device(device)
        _lib.load()
        # conv_in's 16 input channels are zero-padded to the implicit-GEMM loader's 64-channel K-step (exact)
        self._store = V.ParamStore(decoder_weight_shapes(params), self.device, pad64=("decoder.conv_in.weight",))

    def parameters(self) -> Dict[str, torch.Tensor]:
        """float32 master parameters (the checkpoint's dtype)."""
        return self._store.master

    def init_random(self, seed: int = 0) -> "AutoEncoder":
        self._store.init_random(seed)
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
        self._store.load(weights, strict, skip_prefixes=("encoder.",))     # the encoder is not on the decode path
        return self

    # ------------------------------------------------------------------ blocks (x: bf16 NHWC, or split [2,B,H,W,C])
    def _resnet(self, p: str, x: torch.Tensor, fp32: bool = False) -> torch.Tensor:
        """ResnetBlock.__call__ (flux/autoencoder.py:83-98)."""
        return V.resnet(self._store, fp32, p, x, "nin_shortcut", 32, 1e-6)

    def _attn(self, p: str, x: torch.Tensor, fp32: bool = False) -> torch.Tensor:
        """AttnBlock.__call__ (flux/autoencoder.py:42-52): one 512-wide head over H*W tokens."""
        return V.attention(self._store, fp32, p, x, "norm", "q", "k", "v", "proj_out", 32, 1e-6)

    def _decoder(self, z: torch.Tensor, clip01: bool, fp32: bool) -> torch.Tensor:
        """Decoder.__call__ (flux/autoencoder.py:271-297). z (64 channels, zero padded) -> float32 NHWC image."""
        A, S = self.params, self._store
        nres = len(A.ch_mult)
        h = V.conv(S, fp32, "decoder.conv_in", z)
        h = self._resnet("decoder.mid.block_1", h, fp32)
        h = self._attn("decoder.mid.attn_1", h, fp32)
        h = self._resnet("decoder.mid.block_2", h, fp32)
        for lvl in reversed(range(nres)):
            for i in range(A.num_res_blocks + 1):
                h = self._resnet(f"decoder.up.{lvl}.block.{i}", h, fp32)
            if lvl != 0:   # Upsample: nearest x2 fused into the conv loader (flux/autoencoder.py:120-123)
                h = V.conv(S, fp32, f"decoder.up.{lvl}.upsample.conv", h, ups=True)
        return V.norm_out_conv_out(S, fp32, "decoder.norm_out", "decoder.conv_out", h, 32, 1e-6, clip01)

    # ------------------------------------------------------------------ public surface
    def _cpad(self) -> int:
        return (self.params.z_channels + 63) // 64 * 64

    def decode(self, z: torch.Tensor, precision: Optional[str] = None) -> torch.Tensor:
        """AutoEncoder.decode (flux/autoencoder.py:352-354): z [B,h,w,16] -> [B,8h,8w,3] float32."""
        z = z.to(BF16).contiguous()
        _, h, w, _ = z.shape
        return self.decode_packed(ops.pack_latents(z), (h, w), precision, clip01=False)

    def decode_packed(self, x: torch.Tensor, latent_size: Tuple[int, int], precision: Optional[str] = None,
                      clip01: bool = True) -> torch.Tensor:
        """FluxPipeline.decode fused (flux/flux.py:157-162): packed latents [B,L,64] ->
        clip(decode(unpack(x)) + 1, 0, 2) * 0.5, float32 NHWC in [0,1].  z / scale_factor + shift_factor
        (flux/autoencoder.py:353) rides on the unpack kernel."""
        V.check_precision(precision)
        h, w = latent_size
        x = x.to(BF16).contiguous()
        if (precision or self.precision) == "fp32":
            z = ops.unpack_latents_x3(x, h, w, self.scale_factor, self.shift_factor, self._cpad())
            return self._decoder(z, clip01, True)
        z = ops.unpack_latents(x, h, w, self.scale_factor, self.shift_factor)
        if z.shape[-1] % 64:
            z = ops.concat_channels(z, None, pad_to=self._cpad())
        return self._decoder(z, clip01, False)

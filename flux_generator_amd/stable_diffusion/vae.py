"""SD / SDXL VAE decoder on MI355X (mirror of the reference's stable_diffusion/stable_diffusion/vae.py
decode path: Autoencoder.decode :256-258, Decoder.__call__ :209-223, Attention :25-42).
Encoder / quant_proj (image2image) are out of the hot-path scope.  Parameter names = MLX module tree
(what model_io.map_vae_weights produces).  bf16 storage / fp32 accumulate (reference: fp32)."""
from __future__ import annotations

import math
from typing import Dict, Tuple, Union

import torch

from .. import _lib, ops
from ..ops import EPI_BIAS, EPI_GATE_RES, FluxHipError, make_gemm_desc
from .config import AutoencoderConfig

BF16 = torch.bfloat16


def vae_decoder_weight_shapes(cfg: AutoencoderConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}

    def conv(n, o, i):
        s[f"{n}.weight"] = (o, 3, 3, i); s[f"{n}.bias"] = (o,)

    def norm(n, c):
        s[f"{n}.weight"] = (c,); s[f"{n}.bias"] = (c,)

    def lin(n, o, i):
        s[f"{n}.weight"] = (o, i); s[f"{n}.bias"] = (o,)

    def resnet(n, i, o):
        norm(f"{n}.norm1", i); conv(f"{n}.conv1", o, i); norm(f"{n}.norm2", o); conv(f"{n}.conv2", o, o)
        if i != o:
            lin(f"{n}.conv_shortcut", o, i)

    boc = list(cfg.block_out_channels)
    lin("post_quant_proj", cfg.latent_channels_in, cfg.latent_channels_in)
    conv("decoder.conv_in", boc[-1], cfg.latent_channels_in)
    resnet("decoder.mid_blocks.0", boc[-1], boc[-1])
    norm("decoder.mid_blocks.1.group_norm", boc[-1])
    for k in ("query_proj", "key_proj", "value_proj", "out_proj"):
        lin(f"decoder.mid_blocks.1.{k}", boc[-1], boc[-1])
    resnet("decoder.mid_blocks.2", boc[-1], boc[-1])
    ch = list(reversed(boc))
    ch = [ch[0]] + ch
    for i, (ic, oc) in enumerate(zip(ch, ch[1:])):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ic if j == 0 else oc, oc)
        if i < len(boc) - 1:
            conv(f"decoder.up_blocks.{i}.upsample", oc, oc)
    norm("decoder.conv_norm_out", boc[0]); conv("decoder.conv_out", cfg.out_channels, boc[0])
    return s


class Autoencoder:
    def __init__(self, config: AutoencoderConfig, device: Union[str, torch.device] = "cuda"):
        self.config = config
        self.latent_channels = config.latent_channels_in
        self.scaling_factor = config.scaling_factor
        if torch.device(device).type != "cuda":
            raise FluxHipError("Autoencoder needs a HIP device: there is no CPU fallback for the decode path")
        self.device = _lib.bind_device(device)
        _lib.load()
        self._params = {k: torch.empty(*shp, dtype=BF16, device=self.device)
                        for k, shp in vae_decoder_weight_shapes(config).items()}
        self._conv_in_w = None

    def parameters(self):
        return self._params

    def init_random(self, seed: int = 0) -> "Autoencoder":
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, t in self._params.items():
            base = name.rsplit(".", 1)[0]
            wt = self._params[f"{base}.weight"]
            if wt.dim() == 1:
                t.fill_(1.0 if name.endswith(".weight") else 0.0)
                continue
            k = 1.0 / math.sqrt(wt[0].numel())
            t.copy_(((torch.rand(t.shape, generator=g, device=self.device) * 2 - 1) * k).to(BF16))
        return self.finalize()

    def load_weights(self, weights, strict: bool = True) -> "Autoencoder":
        items = weights.items() if isinstance(weights, dict) else weights
        seen = set()
        for k, w in items:
            if k not in self._params:
                if k.startswith("encoder.") or k.startswith("quant_proj") or not strict:
                    continue
                raise ValueError(f"Unexpected parameter {k}")
            dst = self._params[k]
            if tuple(dst.shape) != tuple(w.shape):
                raise ValueError(f"Shape mismatch for {k}: expected {tuple(dst.shape)}, got {tuple(w.shape)}")
            dst.copy_(w.to(device=self.device, dtype=BF16))
            seen.add(k)
        if strict and set(self._params) - seen:
            raise ValueError(f"Missing parameters: {sorted(set(self._params) - seen)[:5]} ...")
        return self.finalize()

    def finalize(self) -> "Autoencoder":
        w = self._params["decoder.conv_in.weight"]        # input channels zero-padded to 64: MFMA implicit-GEMM path
        cin = w.shape[-1]
        wp = torch.zeros(*w.shape[:-1], (cin + 63) // 64 * 64, dtype=BF16, device=self.device)
        wp[..., :cin] = w
        self._conv_in_w = wp
        return self

    # ------------------------------------------------------------------ blocks
    def _resnet(self, p: str, x: torch.Tensor) -> torch.Tensor:
        """ResnetBlock2D without time embedding (unet.py:152-170 as used by vae.py:56-63)."""
        W, G = self._params, self.config.norm_num_groups
        h = ops.groupnorm_silu(x, W[f"{p}.norm1.weight"], W[f"{p}.norm1.bias"], G, 1e-5, True)
        h = ops.conv2d(h, W[f"{p}.conv1.weight"], W[f"{p}.conv1.bias"])
        h = ops.groupnorm_silu(h, W[f"{p}.norm2.weight"], W[f"{p}.norm2.bias"], G, 1e-5, True)
        if f"{p}.conv_shortcut.weight" in W:
            x = ops.conv2d(x, W[f"{p}.conv_shortcut.weight"], W[f"{p}.conv_shortcut.bias"])
        return ops.conv2d(h, W[f"{p}.conv2.weight"], W[f"{p}.conv2.bias"], res=x)

    def _attention(self, p: str, x: torch.Tensor) -> torch.Tensor:
        """Attention.__call__ (vae.py:25-42): single head, softmax((q/sqrt(C)) k^T) v, fp32 logits."""
        W = self._params
        B, H, Wd, C = x.shape
        N = H * Wd
        Np = (N + 63) // 64 * 64
        y = ops.groupnorm_silu(x, W[f"{p}.group_norm.weight"], W[f"{p}.group_norm.bias"], self.config.norm_num_groups,
                               1e-5, False)
        q = ops.linear(y.view(B, N, C), W[f"{p}.query_proj.weight"], W[f"{p}.query_proj.bias"])
        k = ops.linear(y.view(B, N, C), W[f"{p}.key_proj.weight"], W[f"{p}.key_proj.bias"])
        out = torch.empty_like(x)
        vt = torch.zeros(C, Np, dtype=BF16, device=x.device)
        s = torch.empty(N, Np, dtype=torch.float32, device=x.device)
        pm = torch.zeros(N, Np, dtype=BF16, device=x.device)
        o = torch.empty(N, C, dtype=BF16, device=x.device)
        for b in range(B):
            yb = y[b].view(N, C)
            ops.gemm(make_gemm_desc([dict(A=W[f"{p}.value_proj.weight"].data_ptr(), W=yb.data_ptr(),
                                          bias=W[f"{p}.value_proj.bias"].data_ptr(), C=vt.data_ptr(), M=C)],
                                    1, N, C, C, Np, EPI_BIAS, row_bias=True))
            ops.gemm(make_gemm_desc([dict(A=q[b].data_ptr(), W=k[b].data_ptr(), C=s.data_ptr(), M=N)], 1, N, C, C, Np,
                                    EPI_BIAS, out_f32=True))
            ops.softmax_rows(s, C ** -0.5, out=pm, cols=N)
            ops.gemm(make_gemm_desc([dict(A=pm.data_ptr(), W=vt.data_ptr(), C=o.data_ptr(), M=N)], 1, C, Np, Np, C))
            ops.linear(o, W[f"{p}.out_proj.weight"], W[f"{p}.out_proj.bias"], epi=EPI_GATE_RES,
                       out=out[b].view(N, C), res=x[b].view(N, C))
        return out

    def _decode(self, z: torch.Tensor, clip01: bool) -> torch.Tensor:
        cfg, W = self.config, self._params
        z = z.to(BF16).contiguous()
        cin = cfg.latent_channels_in
        # z / scaling_factor -> post_quant_proj (vae.py:256-258), output zero-padded to 64 channels
        x = ops.pixel_linear(z, W["post_quant_proj.weight"], W["post_quant_proj.bias"], (cin + 63) // 64 * 64,
                             self.scaling_factor)
        x = ops.conv2d(x, self._conv_in_w, W["decoder.conv_in.bias"])
        x = self._resnet("decoder.mid_blocks.0", x)
        x = self._attention("decoder.mid_blocks.1", x)
        x = self._resnet("decoder.mid_blocks.2", x)
        n = len(cfg.block_out_channels)
        for i in range(n):
            for j in range(cfg.layers_per_block + 1):
                x = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", x)
            if i < n - 1:
                x = ops.conv2d(x, W[f"decoder.up_blocks.{i}.upsample.weight"], W[f"decoder.up_blocks.{i}.upsample.bias"],
                               ups=True)
        x = ops.groupnorm_silu(x, W["decoder.conv_norm_out.weight"], W["decoder.conv_norm_out.bias"],
                               cfg.norm_num_groups, 1e-5, True)
        return ops.conv2d_out_image(x, W["decoder.conv_out.weight"], W["decoder.conv_out.bias"], clip01)

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """Autoencoder.decode (vae.py:256-258): [B,h,w,4] -> [B,8h,8w,3] float32 (unclipped)."""
        return self._decode(z, clip01=False)

    def decode_image(self, z: torch.Tensor) -> torch.Tensor:
        """StableDiffusion.decode fused (__init__.py:166-169): clip(decode(z)/2 + 0.5, 0, 1)."""
        return self._decode(z, clip01=True)

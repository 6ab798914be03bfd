"""SD / SDXL VAE decoder on MI355X (mirror of the reference's stable_diffusion/stable_diffusion/vae.py
decode path: Autoencoder.decode :256-258, Decoder.__call__ :209-223, Attention :25-42).
Encoder / quant_proj (image2image) are out of the hot-path scope.  Parameter names = MLX module tree
(what model_io.map_vae_weights produces).  Default arithmetic: the reference's float32, on the fp32-faithful
split-bf16 kernels (vae_common.py); bf16 storage is an opt-in."""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import torch

from .. import _lib, ops
from .. import vae_common as V
from ..ops import FluxHipError
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
    """precision = "fp32" (default): the reference builds this model with float16=False
    (stable_diffusion/__init__.py:25), i.e. decodes in float32 — here the fp32-faithful split-bf16 kernels;
    "bf16" is the bf16-storage opt-in (see vae_common.py)."""

    def __init__(self, config: AutoencoderConfig, device: Union[str, torch.device] = "cuda", precision: str = "fp32"):
        V.check_precision(precision)
        self.config = config
        self.latent_channels = config.latent_channels_in
        self.scaling_factor = config.scaling_factor
        self.precision = precision
        if torch.device(device).type != "cuda":
            raise FluxHipError("Autoencoder needs a HIP device: there is no CPU fallback for the decode path")
        self.device = _lib.bind_device(device)
        _lib.load()
        self._store = V.ParamStore(vae_decoder_weight_shapes(config), self.device, pad64=("decoder.conv_in.weight",))

    def parameters(self):
        """float32 master parameters."""
        return self._store.master

    def init_random(self, seed: int = 0) -> "Autoencoder":
        self._store.init_random(seed)
        return self

    def load_weights(self, weights, strict: bool = True) -> "Autoencoder":
        self._store.load(weights, strict, skip_prefixes=("encoder.", "quant_proj"))    # image2image only
        return self

    # ------------------------------------------------------------------ blocks
    def _resnet(self, p: str, x: torch.Tensor, fp32: bool = False) -> torch.Tensor:
        """ResnetBlock2D without time embedding (unet.py:152-170 as used by vae.py:56-63)."""
        return V.resnet(self._store, fp32, p, x, "conv_shortcut", self.config.norm_num_groups, 1e-5)

    def _attention(self, p: str, x: torch.Tensor, fp32: bool = False) -> torch.Tensor:
        """Attention.__call__ (vae.py:25-42): single head, softmax((q/sqrt(C)) k^T) v, fp32 logits."""
        return V.attention(self._store, fp32, p, x, "group_norm", "query_proj", "key_proj", "value_proj", "out_proj",
                           self.config.norm_num_groups, 1e-5)

    def _decode(self, z: torch.Tensor, clip01: bool, precision: Optional[str] = None) -> torch.Tensor:
        V.check_precision(precision)
        fp32 = (precision or self.precision) == "fp32"
        cfg, S = self.config, self._store
        # float16 latents (float16=True pipelines) stay float16 into the fp32-faithful path: the reference promotes them to the
        # VAE's float32 after the division by the scaling factor (vae.py:256-258); the bf16-storage opt-in takes bf16
        # float32 latents (float16=False pipelines) go in as they are: the whole decode is then the reference's float32
        z = (z if (fp32 and z.dtype in (torch.float16, torch.float32)) else z.to(BF16)).contiguous()
        cin = cfg.latent_channels_in
        cpad = (cin + 63) // 64 * 64
        # z / scaling_factor -> post_quant_proj (vae.py:256-258), output zero-padded to 64 channels
        if fp32 and z.dtype == torch.float32:
            x = ops.pixel_linear_x3_f32in(z, S.get("post_quant_proj.weight", "f32"), S.get("post_quant_proj.bias", "f32"), cpad,
                                          self.scaling_factor)
        elif fp32:
            x = ops.pixel_linear_x3(z, S.get("post_quant_proj.weight", "f32"), S.get("post_quant_proj.bias", "f32"), cpad,
                                    self.scaling_factor)
        else:
            x = ops.pixel_linear(z, S.get("post_quant_proj.weight", "bf16"), S.get("post_quant_proj.bias", "bf16"), cpad,
                                 self.scaling_factor)
        x = V.conv(S, fp32, "decoder.conv_in", x)
        x = self._resnet("decoder.mid_blocks.0", x, fp32)
        x = self._attention("decoder.mid_blocks.1", x, fp32)
        x = self._resnet("decoder.mid_blocks.2", x, fp32)
        n = len(cfg.block_out_channels)
        for i in range(n):
            for j in range(cfg.layers_per_block + 1):
                x = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, fp32)
            if i < n - 1:
                x = V.conv(S, fp32, f"decoder.up_blocks.{i}.upsample", x, ups=True)
        return V.norm_out_conv_out(S, fp32, "decoder.conv_norm_out", "decoder.conv_out", x, cfg.norm_num_groups, 1e-5,
                                   clip01)

    def decode(self, z: torch.Tensor, precision: Optional[str] = None) -> torch.Tensor:
        """Autoencoder.decode (vae.py:256-258): [B,h,w,4] -> [B,8h,8w,3] float32 (unclipped)."""
        return self._decode(z, False, precision)

    def decode_image(self, z: torch.Tensor, precision: Optional[str] = None) -> torch.Tensor:
        """StableDiffusion.decode fused (__init__.py:166-169): clip(decode(z)/2 + 0.5, 0, 1)."""
        return self._decode(z, True, precision)

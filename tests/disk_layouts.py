"""Test helpers: write checkpoints in the ON-DISK key / layout conventions of the files the reference downloads
(the inverse of the reference's sanitize / map_* functions), from weights given in module-tree names.

  Flux transformer   flux1-*.safetensors: optional "model.diffusion_model." prefix, RMSNorm "*.scale",
                     Sequential indices "img_mlp.0/2", "adaLN_modulation.1"             (flux/model.py:85-97)
  Flux AE            ae.safetensors: conv weights [O,I,kh,kw], 1x1 convs as [O,I,1,1]   (flux/autoencoder.py:336-345)
  SD UNet / VAE      diffusers names: downsamplers.0.conv, upsamplers.0.conv, mid_block.{resnets,attentions}.N,
                     to_q/to_k/to_v/to_out.0, ff.net.0.proj (GEGLU halves concatenated), ff.net.2,
                     conv_shortcut [O,I,1,1], quant_conv / post_quant_conv [C,C,1,1]     (.../model_io.py:49-164)
These are pure renames / permutes of tensors the TEST creates; nothing here is reference data.
"""
import re

import torch


def flux_to_disk(W, prefix=""):
    out = {}
    for k, v in W.items():
        k = k.replace(".layers.", ".")                               # nn.Sequential index without "layers"
        if k.endswith("_norm.weight"):
            k = k[: -len(".weight")] + ".scale"                      # RMSNorm scale
        out[prefix + k] = v.contiguous()
    return out


def ae_to_disk(W, with_encoder_keys=True):
    out = {}
    for k, v in W.items():
        if v.dim() == 4:
            v = v.permute(0, 3, 1, 2)                                # [O,kh,kw,I] -> [O,I,kh,kw]
        elif v.dim() == 2:
            v = v[:, :, None, None]                                  # Linear (1x1 conv) -> [O,I,1,1]
        out[k] = v.contiguous()
    if with_encoder_keys:                                            # the checkpoint also holds the (unused) encoder
        out["encoder.conv_in.weight"] = torch.zeros(8, 3, 3, 3)
        out["encoder.conv_in.bias"] = torch.zeros(8)
    return out


_ATTN = (("query_proj", "to_q"), ("key_proj", "to_k"), ("value_proj", "to_v"), ("out_proj", "to_out.0"))


def unet_to_disk(W, conv_proj=False):
    """conv_proj: proj_in / proj_out stored as 1x1 convs [C,C,1,1] (SD 2.x) instead of Linear (SDXL)."""
    out = {}
    geglu = {}
    for k, v in W.items():
        k = k.replace(".downsample.", ".downsamplers.0.conv.").replace(".upsample.", ".upsamplers.0.conv.")
        k = k.replace("mid_blocks.0.", "mid_block.resnets.0.").replace("mid_blocks.1.", "mid_block.attentions.0.")
        k = k.replace("mid_blocks.2.", "mid_block.resnets.1.")
        for a, b in _ATTN:
            k = k.replace(f".{a}.", f".{b}.")
        m = re.match(r"(.*)\.linear([12])\.(weight|bias)$", k)
        if m and ".transformer_blocks." in k:
            geglu.setdefault((m.group(1), m.group(3)), {})[m.group(2)] = v
            continue
        if ".transformer_blocks." in k:
            k = k.replace(".linear3.", ".ff.net.2.")
        if "conv_shortcut.weight" in k:
            v = v[:, :, None, None]
        elif v.dim() == 2 and conv_proj and (".proj_in." in k or ".proj_out." in k):
            v = v[:, :, None, None]
        elif v.dim() == 4:
            v = v.permute(0, 3, 1, 2)
        out[k] = v.contiguous()
    for (base, kind), halves in geglu.items():                      # ff.net.0.proj = [linear1 (value) ; linear2 (gate)]
        out[f"{base}.ff.net.0.proj.{kind}"] = torch.cat([halves["1"], halves["2"]], dim=0).contiguous()
    return out


def vae_to_disk(W, with_encoder_keys=True):
    out = {}
    for k, v in W.items():
        k = k.replace(".upsample.", ".upsamplers.0.conv.")
        k = k.replace("mid_blocks.0.", "mid_block.resnets.0.").replace("mid_blocks.1.", "mid_block.attentions.0.")
        k = k.replace("mid_blocks.2.", "mid_block.resnets.1.")
        for a, b in _ATTN:
            k = k.replace(f".{a}.", f".{b}.")
        if k.startswith("post_quant_proj."):
            k = k.replace("post_quant_proj.", "post_quant_conv.")
            if v.dim() == 2:
                v = v[:, :, None, None]
        elif "conv_shortcut.weight" in k:
            v = v[:, :, None, None]
        elif v.dim() == 4:
            v = v.permute(0, 3, 1, 2)
        out[k] = v.contiguous()
    if with_encoder_keys:
        out["encoder.conv_in.weight"] = torch.zeros(8, 3, 3, 3)
        out["encoder.conv_in.bias"] = torch.zeros(8)
        out["quant_conv.weight"] = torch.zeros(8, 8, 1, 1)
        out["quant_conv.bias"] = torch.zeros(8)
    return out

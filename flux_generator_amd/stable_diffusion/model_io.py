"""Model zoo / loaders of the stable_diffusion path (mirror of the reference's
stable_diffusion/stable_diffusion/model_io.py).

The reference downloads config.json + safetensors from the HF hub at run time (model_io.py:185-330).
There is no network here, so the two supported models' hyper-parameters (the values of their HF
config.json files) are built in, weights are read from a local directory given by SD_WEIGHTS_DIR
(same relative file names as the hub) and fall back to random init (with a warning) otherwise.
The checkpoint key mappings are the reference's (model_io.py:49-164)."""
from __future__ import annotations

import os
import warnings
from typing import Optional

import torch

from ..flux.text import HashTokenizer
from ..flux.utils import random_init_or_raise
from ..flux.tokenizers import CLIPTokenizer
from .clip import CLIPTextModel, CLIPTextModelConfig, map_clip_text_encoder_weights
from .config import AutoencoderConfig, DiffusionConfig, UNetConfig
from .unet import UNetModel
from .vae import Autoencoder

_DEFAULT_MODEL = "stabilityai/stable-diffusion-2-1-base"

_MODELS = {
    "stabilityai/sdxl-turbo": dict(
        unet="unet/diffusion_pytorch_model.safetensors", vae="vae/diffusion_pytorch_model.safetensors",
        unet_config=UNetConfig(
            in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=(2, 2, 2),
            transformer_layers_per_block=(1, 2, 10), num_attention_heads=(5, 10, 20), cross_attention_dim=(2048,) * 3,
            norm_num_groups=32, down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
            up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),   # HF order reversed (:212)
            addition_embed_type="text_time", addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816),
        vae_config=AutoencoderConfig(scaling_factor=0.13025),
        text_dims=(768, 1280), pooled_dim=1280),
    "stabilityai/stable-diffusion-2-1-base": dict(
        unet="unet/diffusion_pytorch_model.safetensors", vae="vae/diffusion_pytorch_model.safetensors",
        unet_config=UNetConfig(
            in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=(2, 2, 2, 2),
            transformer_layers_per_block=(1, 1, 1, 1), num_attention_heads=(5, 10, 20, 20),
            cross_attention_dim=(1024,) * 4, norm_num_groups=32,
            down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
            # HF lists ["UpBlock2D", "CrossAttn...", ...] deepest-first; the reference reverses it (model_io.py:212)
            up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D")),
        vae_config=AutoencoderConfig(scaling_factor=0.18215),
        text_dims=(1024,), pooled_dim=1024),
}


def _check_key(key: str, part: str):
    if key not in _MODELS:
        raise ValueError(f"[{part}] '{key}' model not found, choose one of {{{','.join(_MODELS.keys())}}}")


def map_unet_weights(key, value):
    """model_io.py:49-95 (HF diffusers names -> module tree; GEGLU split; conv [O,I,kh,kw] -> [O,kh,kw,I])."""
    if "downsamplers" in key:
        key = key.replace("downsamplers.0.conv", "downsample")
    if "upsamplers" in key:
        key = key.replace("upsamplers.0.conv", "upsample")
    if "mid_block.resnets.0" in key:
        key = key.replace("mid_block.resnets.0", "mid_blocks.0")
    if "mid_block.attentions.0" in key:
        key = key.replace("mid_block.attentions.0", "mid_blocks.1")
    if "mid_block.resnets.1" in key:
        key = key.replace("mid_block.resnets.1", "mid_blocks.2")
    if "to_k" in key:
        key = key.replace("to_k", "key_proj")
    if "to_out.0" in key:
        key = key.replace("to_out.0", "out_proj")
    if "to_q" in key:
        key = key.replace("to_q", "query_proj")
    if "to_v" in key:
        key = key.replace("to_v", "value_proj")
    if "ff.net.2" in key:
        key = key.replace("ff.net.2", "linear3")
    if "ff.net.0" in key:
        k1 = key.replace("ff.net.0.proj", "linear1")
        k2 = key.replace("ff.net.0.proj", "linear2")
        v1, v2 = torch.chunk(value, 2, dim=0)
        return [(k1, v1), (k2, v2)]
    if "conv_shortcut.weight" in key:
        value = value.squeeze()
    if value.dim() == 4 and ("proj_in" in key or "proj_out" in key):
        value = value.squeeze()
    if value.dim() == 4:
        value = value.permute(0, 2, 3, 1).contiguous()
    return [(key, value)]


def map_vae_weights(key, value):
    """model_io.py:126-164."""
    if "downsamplers" in key:
        key = key.replace("downsamplers.0.conv", "downsample")
    if "upsamplers" in key:
        key = key.replace("upsamplers.0.conv", "upsample")
    for a, b in (("to_k", "key_proj"), ("to_out.0", "out_proj"), ("to_q", "query_proj"), ("to_v", "value_proj")):
        if a in key:
            key = key.replace(a, b)
    if "mid_block.resnets.0" in key:
        key = key.replace("mid_block.resnets.0", "mid_blocks.0")
    if "mid_block.attentions.0" in key:
        key = key.replace("mid_block.attentions.0", "mid_blocks.1")
    if "mid_block.resnets.1" in key:
        key = key.replace("mid_block.resnets.1", "mid_blocks.2")
    if "quant_conv" in key:
        key = key.replace("quant_conv", "quant_proj")
        value = value.squeeze()
    if "conv_shortcut.weight" in key:
        value = value.squeeze()
    if value.dim() == 4:
        value = value.permute(0, 2, 3, 1).contiguous()
    return [(key, value)]


def _weights_file(key: str, rel: str) -> Optional[str]:
    root = os.getenv("SD_WEIGHTS_DIR")
    if root:
        p = os.path.join(root, key, rel)
        if os.path.exists(p):
            return p
    return None


def _load_mapped(mapper, path):
    from safetensors.torch import load_file
    out = {}
    for k, v in load_file(path).items():
        for kk, vv in mapper(k, v):
            out[kk] = vv
    return out


def load_unet(key: str = _DEFAULT_MODEL, float16: bool = False, device="cuda", seed: int = 0, dtype=None) -> UNetModel:
    """dtype (an addition): the model's arithmetic when it is not implied by float16 - torch.float32 = the reference's
    float32 (what float16=False means there, model_io.py:171-174), torch.bfloat16 = the bf16-storage opt-in."""
    _check_key(key, "load_unet")
    model = UNetModel(_MODELS[key]["unet_config"], device=device, dtype=dtype or (torch.float16 if float16 else torch.bfloat16))
    path = _weights_file(key, _MODELS[key]["unet"])
    if path:
        model.load_weights(_load_mapped(map_unet_weights, path))
    else:
        random_init_or_raise(f"{key} UNet", "put the hub files under SD_WEIGHTS_DIR")
        model.init_random(seed)
    return model


def load_autoencoder(key: str = _DEFAULT_MODEL, float16: bool = False, device="cuda", seed: int = 1) -> Autoencoder:
    _check_key(key, "load_autoencoder")
    model = Autoencoder(_MODELS[key]["vae_config"], device=device)
    path = _weights_file(key, _MODELS[key]["vae"])
    if path:
        model.load_weights(_load_mapped(map_vae_weights, path))   # strict: encoder.* / quant_proj keys are skipped explicitly
    else:
        random_init_or_raise(f"{key} VAE", "put the hub files under SD_WEIGHTS_DIR")
        model.init_random(seed)
    return model


def load_diffusion_config(key: str = _DEFAULT_MODEL) -> DiffusionConfig:
    _check_key(key, "load_diffusion_config")
    return DiffusionConfig(beta_schedule="scaled_linear", beta_start=0.00085, beta_end=0.012, num_train_steps=1000)


# --- text side (stable_diffusion/.../model_io.py:229-265,313-330) -------------------------------------
# The values of the hub's text_encoder*/config.json files, used when SD_WEIGHTS_DIR has no config.json.
_TEXT_CONFIGS = {
    ("stabilityai/sdxl-turbo", "text_encoder"): dict(num_layers=12, model_dims=768, num_heads=12, max_length=77,
                                                    vocab_size=49408, hidden_act="quick_gelu", projection_dim=None),
    ("stabilityai/sdxl-turbo", "text_encoder_2"): dict(num_layers=32, model_dims=1280, num_heads=20, max_length=77,
                                                      vocab_size=49408, hidden_act="gelu", projection_dim=1280),
    ("stabilityai/stable-diffusion-2-1-base", "text_encoder"): dict(num_layers=23, model_dims=1024, num_heads=16,
                                                                    max_length=77, vocab_size=49408, hidden_act="gelu",
                                                                    projection_dim=None),
}


def load_text_encoder(key: str = _DEFAULT_MODEL, float16: bool = False, model_key: str = "text_encoder",
                      config_key: Optional[str] = None, device="cuda", seed: int = 21, dtype=None) -> CLIPTextModel:
    """The real CLIP text transformer on libfluxhip (flux/clip.py): config from <SD_WEIGHTS_DIR>/<key>/<model_key>/
    config.json when present (else the built-in hub values), weights from .../model.safetensors through
    map_clip_text_encoder_weights; random init (with a warning) when no checkpoint is configured."""
    import json
    _check_key(key, "load_text_encoder")
    cfg_path = _weights_file(key, f"{model_key}/config.json")
    if cfg_path:
        with open(cfg_path) as f:
            config = CLIPTextModelConfig.from_dict(json.load(f))
    else:
        config = CLIPTextModelConfig(**_TEXT_CONFIGS[(key, model_key)])
    model = CLIPTextModel(config, device=device, dtype=dtype or (torch.float16 if float16 else torch.bfloat16))
    path = _weights_file(key, f"{model_key}/model.safetensors")
    if path:
        model.load_weights(_load_mapped(map_clip_text_encoder_weights, path))
    else:
        random_init_or_raise(f"{key} {model_key}", "put the hub files under SD_WEIGHTS_DIR")
        model.init_random(seed + (1 if model_key.endswith("_2") else 0))
    return model


class _HashTok(HashTokenizer):
    """Stand-in used only when no vocab.json / merges.txt is available offline (says so when substituted)."""

    def tokenize(self, text, prepend_bos=True, append_eos=True):
        return HashTokenizer.tokenize(self, text)


def load_tokenizer(key: str = _DEFAULT_MODEL, vocab_key: str = "tokenizer_vocab", merges_key: str = "tokenizer_merges"):
    """The CLIP BPE tokenizer from <SD_WEIGHTS_DIR>/<key>/tokenizer[_2]/{vocab.json,merges.txt}
    (model_io.py:313-330: merges lines 1 .. 49152-256-2)."""
    import json
    _check_key(key, "load_tokenizer")
    sub = "tokenizer_2" if "tokenizer_2" in vocab_key else "tokenizer"
    vf, mf = _weights_file(key, f"{sub}/vocab.json"), _weights_file(key, f"{sub}/merges.txt")
    if vf and mf:
        with open(vf, encoding="utf-8") as f:
            vocab = json.load(f)
        with open(mf, encoding="utf-8") as f:
            merges = f.read().strip().split("\n")[1: 49152 - 256 - 2 + 1]
        ranks = {tuple(m.split()): i for i, m in enumerate(merges)}
        return CLIPTokenizer(ranks, vocab, max_length=77)
    warnings.warn(f"{key}: no {sub}/vocab.json + merges.txt under SD_WEIGHTS_DIR; substituting a deterministic hash "
                  "tokenizer (token ids are NOT CLIP's)")
    return _HashTok(max_length=77, vocab=49408, pad_with_eos=True)

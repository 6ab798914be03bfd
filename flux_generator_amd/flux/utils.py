"""Model zoo / loaders (mirror of the reference's flux/utils.py:30-210).

The hyper-parameters are the reference's; weight files are safetensors given by the same
environment variables (FLUX_DEV / FLUX_SCHNELL / AE).  There is no network in this environment,
so when no checkpoint path is configured the models are random-initialised with the reference
framework's default init (and say so) instead of calling hf_hub_download.
"""
from __future__ import annotations

import os
import warnings
from dataclasses import dataclass
from typing import Optional

import torch

from .autoencoder import AutoEncoder, AutoEncoderParams
from .model import Flux, FluxParams
from .text import SyntheticCLIP, SyntheticT5, HashTokenizer


@dataclass
class ModelSpec:
    params: FluxParams
    ae_params: AutoEncoderParams
    ckpt_path: Optional[str]
    ae_path: Optional[str]
    repo_id: Optional[str]
    repo_flow: Optional[str]
    repo_ae: Optional[str]


def _flux_params(guidance_embed: bool) -> FluxParams:
    return FluxParams(in_channels=64, vec_in_dim=768, context_in_dim=4096, hidden_size=3072, mlp_ratio=4.0,
                      num_heads=24, depth=19, depth_single_blocks=38, axes_dim=[16, 56, 56], theta=10_000,
                      qkv_bias=True, guidance_embed=guidance_embed)


def _ae_params() -> AutoEncoderParams:
    return AutoEncoderParams(resolution=256, in_channels=3, ch=128, out_ch=3, ch_mult=[1, 2, 4, 4],
                             num_res_blocks=2, z_channels=16, scale_factor=0.3611, shift_factor=0.1159)


# flux/utils.py:30-95
configs = {
    "flux-dev": ModelSpec(repo_id="black-forest-labs/FLUX.1-dev", repo_flow="flux1-dev.safetensors",
                          repo_ae="ae.safetensors", ckpt_path=os.getenv("FLUX_DEV"), params=_flux_params(True),
                          ae_path=os.getenv("AE"), ae_params=_ae_params()),
    "flux-schnell": ModelSpec(repo_id="black-forest-labs/FLUX.1-schnell", repo_flow="flux1-schnell.safetensors",
                              repo_ae="ae.safetensors", ckpt_path=os.getenv("FLUX_SCHNELL"),
                              params=_flux_params(False), ae_path=os.getenv("AE"), ae_params=_ae_params()),
}


def _load_safetensors(path: str):
    from safetensors.torch import load_file
    return load_file(path)


def load_flow_model(name: str, hf_download: bool = True, device="cuda", seed: int = 0) -> Flux:
    """flux/utils.py:98-121."""
    spec = configs[name]
    model = Flux(spec.params, device=device)
    if spec.ckpt_path is not None:
        model.load_weights(model.sanitize(_load_safetensors(spec.ckpt_path)))
    else:
        warnings.warn(f"{name}: no checkpoint configured (set FLUX_SCHNELL / FLUX_DEV); using random-init weights")
        model.init_random(seed)
    return model


def load_ae(name: str, hf_download: bool = True, device="cuda", seed: int = 1) -> AutoEncoder:
    """flux/utils.py:124-147."""
    spec = configs[name]
    ae = AutoEncoder(spec.ae_params, device=device)
    if spec.ae_path is not None:
        ae.load_weights(ae.sanitize(_load_safetensors(spec.ae_path)), strict=False)
    else:
        warnings.warn(f"{name}: no AE checkpoint configured (set AE); using random-init weights")
        ae.init_random(seed)
    return ae


# The text encoders and tokenizers are the next row of the scope table;
# until they land, conditioning tensors of the right shape/dtype come from these stand-ins.
def load_clip(name: str, device="cuda"):
    return SyntheticCLIP(dim=configs[name].params.vec_in_dim, device=device)


def load_t5(name: str, device="cuda"):
    return SyntheticT5(dim=configs[name].params.context_in_dim, device=device)


def load_clip_tokenizer(name: str):
    return HashTokenizer(max_length=77, vocab=49408)


def load_t5_tokenizer(name: str, pad: bool = True):
    return HashTokenizer(max_length=256 if "schnell" in name else 512, vocab=32128)

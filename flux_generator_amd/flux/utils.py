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


from .autoencoder import AutoEncoder, AutoEncoderParams
from .model import Flux, FluxParams
from .clip import CLIP_L, CLIPTextModel, CLIPTextModelConfig
from .t5 import T5_XXL, T5Config, T5Encoder
from .text import HashTokenizer
from .tokenizers import CLIPTokenizer, T5Tokenizer


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


def _hub_file(repo_id: Optional[str], filename: Optional[str], hf_download: bool) -> Optional[str]:
    """The reference falls back to hf_hub_download(repo_id, filename) when no path is configured (flux/utils.py:102-110,
    128-136).  Here: a file already in the local Hugging Face cache is always used; a network download is attempted only
    with FLUX_HUB_DOWNLOAD=1 (this build's environments have no egress, and a connection attempt per model load is not a
    sensible default there).  None -> the caller random-initialises and says so."""
    if not (hf_download and repo_id and filename):
        return None
    try:
        from huggingface_hub import hf_hub_download, try_to_load_from_cache
        hit = try_to_load_from_cache(repo_id, filename)
        if isinstance(hit, str) and os.path.exists(hit):
            return hit
        if os.environ.get("FLUX_HUB_DOWNLOAD") == "1":
            return hf_hub_download(repo_id, filename)
    except Exception as e:      # no network, gated repo without a token, ...
        warnings.warn(f"hub download of {repo_id}/{filename} failed: {type(e).__name__}: {e}")
    return None


def random_init_or_raise(what: str, how: str) -> None:
    """No checkpoint for `what`.  The reference would call hf_hub_download here and fail without network
    (flux/utils.py:102-110); a server that silently answers with noise images from random weights is worse than an error, so
    random initialisation needs an explicit FLUX_ALLOW_RANDOM_INIT=1 (tests, bench.py and the tools set it: this build's
    environments have no checkpoints).  `how` names the configuration that would supply real weights."""
    if os.environ.get("FLUX_ALLOW_RANDOM_INIT") != "1":
        raise FileNotFoundError(f"{what}: no weights found ({how}). Set FLUX_ALLOW_RANDOM_INIT=1 to run on random-initialised "
                                "weights (benchmarks / tests only: the images are noise).")
    warnings.warn(f"{what}: no weights found ({how}); FLUX_ALLOW_RANDOM_INIT=1 -> random-init weights")


def _receives_weights(from_rank0: bool) -> bool:
    """True on the ranks that skip the disk read and take the weights from rank 0 over RCCL."""
    from .. import parallel
    return bool(from_rank0) and parallel.active() and parallel.world()[0] != 0


def load_flow_model(name: str, hf_download: bool = True, device="cuda", seed: int = 0, from_rank0: bool = False) -> Flux:
    """flux/utils.py:98-121.  from_rank0 (multi-GPU): only rank 0 reads the checkpoint (or draws the
    random init); the other ranks allocate and receive the 23.8 GB over RCCL/xGMI (`Flux.broadcast_weights`)."""
    spec = configs[name]
    model = Flux(spec.params, device=device)
    if _receives_weights(from_rank0):
        model.broadcast_weights(0)
        return model
    model = _load_flow_local(model, name, spec, seed, hf_download)
    if from_rank0:
        model.broadcast_weights(0)
    return model


def _load_flow_local(model: Flux, name: str, spec: ModelSpec, seed: int, hf_download: bool = True) -> Flux:
    path = spec.ckpt_path or _hub_file(spec.repo_id, spec.repo_flow, hf_download)
    if path is not None:
        model.load_weights(model.sanitize(_load_safetensors(path)))
    else:
        random_init_or_raise(f"{name} flow model", "set FLUX_SCHNELL / FLUX_DEV, or FLUX_HUB_DOWNLOAD=1 with network access, or "
                             "pre-populate the Hugging Face cache")
        model.init_random(seed)
    return model


def load_ae(name: str, hf_download: bool = True, device="cuda", seed: int = 1, from_rank0: bool = False) -> AutoEncoder:
    """flux/utils.py:124-147.  from_rank0: as for load_flow_model."""
    spec = configs[name]
    ae = AutoEncoder(spec.ae_params, device=device)
    if _receives_weights(from_rank0):
        ae._store.broadcast(0)
        return ae
    ae = _load_ae_local(ae, name, spec, seed, hf_download)
    if from_rank0:
        ae._store.broadcast(0)
    return ae


def _load_ae_local(ae: AutoEncoder, name: str, spec: ModelSpec, seed: int, hf_download: bool = True) -> AutoEncoder:
    path = spec.ae_path or _hub_file(spec.repo_id, spec.repo_ae, hf_download)
    if path is not None:
        ae.load_weights(ae.sanitize(_load_safetensors(path)))   # strict: encoder.* keys are skipped explicitly
    else:
        random_init_or_raise(f"{name} autoencoder", "set AE, or FLUX_HUB_DOWNLOAD=1 with network access, or pre-populate the "
                             "Hugging Face cache")
        ae.init_random(seed)
    return ae


# ---- text side.  FLUX_TEXT_DIR may point at a local copy of the hub layout
# (text_encoder/{config.json,model.safetensors}, text_encoder_2/{config.json,model*.safetensors,
# model.safetensors.index.json}, tokenizer/{vocab.json,merges.txt}, tokenizer_2/spiece.model); without it the
# real architectures are random-initialised and the tokenizers fall back to a deterministic hash tokenizer.
def _text_dir():
    d = os.getenv("FLUX_TEXT_DIR")
    return d if d and os.path.isdir(d) else None


def load_clip(name: str, device="cuda", seed: int = 2) -> CLIPTextModel:
    """flux/utils.py:150-166."""
    import json
    d = _text_dir()
    if d and os.path.exists(os.path.join(d, "text_encoder", "model.safetensors")):
        with open(os.path.join(d, "text_encoder", "config.json")) as f:
            clip = CLIPTextModel(CLIPTextModelConfig.from_dict(json.load(f)), device=device)
        return clip.load_weights(clip.sanitize(_load_safetensors(os.path.join(d, "text_encoder", "model.safetensors"))))
    random_init_or_raise("CLIP text encoder", "point FLUX_TEXT_DIR at a copy of the hub layout (text_encoder/)")
    return CLIPTextModel(CLIPTextModelConfig(**CLIP_L), device=device).init_random(seed)


def load_t5(name: str, device="cuda", seed: int = 3) -> T5Encoder:
    """flux/utils.py:169-194 (sharded safetensors via model.safetensors.index.json)."""
    import json
    d = _text_dir()
    idx = os.path.join(d, "text_encoder_2", "model.safetensors.index.json") if d else None
    if idx and os.path.exists(idx):
        with open(os.path.join(d, "text_encoder_2", "config.json")) as f:
            t5 = T5Encoder(T5Config.from_dict(json.load(f)), device=device)
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
        weights = {}
        for w in files:
            weights.update(_load_safetensors(os.path.join(d, "text_encoder_2", w)))
        return t5.load_weights(t5.sanitize(weights))    # strict: decoder.* / lm_head.* keys are skipped explicitly
    random_init_or_raise("T5-XXL encoder", "point FLUX_TEXT_DIR at a copy of the hub layout (text_encoder_2/)")
    return T5Encoder(T5Config(**T5_XXL), device=device).init_random(seed)


def load_clip_tokenizer(name: str):
    """flux/utils.py:197-208."""
    import json
    d = _text_dir()
    vf = os.path.join(d, "tokenizer", "vocab.json") if d else None
    if vf and os.path.exists(vf):
        with open(vf, encoding="utf-8") as f:
            vocab = json.load(f)
        with open(os.path.join(d, "tokenizer", "merges.txt"), encoding="utf-8") as f:
            merges = f.read().strip().split("\n")[1: 49152 - 256 - 2 + 1]
        ranks = {tuple(m.split()): i for i, m in enumerate(merges)}
        return CLIPTokenizer(ranks, vocab, max_length=77)
    warnings.warn("CLIP tokenizer: no tokenizer/vocab.json under FLUX_TEXT_DIR; substituting a deterministic hash "
                  "tokenizer (token ids are NOT CLIP's)")
    return HashTokenizer(max_length=77, vocab=49408, pad_with_eos=True)


def load_t5_tokenizer(name: str, pad: bool = True):
    """flux/utils.py:208-210."""
    d = _text_dir()
    mf = os.path.join(d, "tokenizer_2", "spiece.model") if d else None
    n = 256 if "schnell" in name else 512
    if mf and os.path.exists(mf):
        return T5Tokenizer(mf, n)
    warnings.warn("T5 tokenizer: no tokenizer_2/spiece.model under FLUX_TEXT_DIR; substituting a deterministic hash "
                  "tokenizer (token ids are NOT T5's)")
    return HashTokenizer(max_length=n, vocab=32128)

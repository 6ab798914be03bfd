"""CPU oracle for the text encoders in front of the Flux hot path (T5 encoder, CLIP text model).
TEST INFRASTRUCTURE ONLY (same rules as oracle/flux_oracle.py; the reference's MLX cannot run here and
its tests hold no numeric golden for these models => parity unpinned by the reference).

Plain PyTorch-CPU restatement of flux/t5.py and flux/clip.py.  Unlike the rest of the oracle these two
models have an INDEPENDENT implementation available offline (``transformers`` T5EncoderModel /
CLIPTextModel with random tiny configs): tests/test_text_oracle.py pins this restatement against them.
Weights: dict keyed by the reference's sanitized parameter names (flux/t5.py:232-241, flux/clip.py:96-125).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .flux_oracle import linear, rms_norm

Tensor = torch.Tensor


# ------------------------------------------------------------------ T5 encoder (flux/t5.py)
@dataclass
class T5Config:
    vocab_size: int = 32128
    num_layers: int = 24
    num_heads: int = 64
    relative_attention_num_buckets: int = 32
    d_kv: int = 64
    d_model: int = 4096
    feed_forward_proj: str = "gated-gelu"
    tie_word_embeddings: bool = False
    d_ff: int = 10240
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6


def relative_position_bucket(rpos: Tensor, bidirectional: bool, num_buckets: int, max_distance: int) -> Tensor:
    """RelativePositionBias._relative_position_bucket (flux/t5.py:78-97)."""
    num_buckets = num_buckets // 2 if bidirectional else num_buckets
    max_exact = num_buckets // 2
    abspos = rpos.abs()
    is_small = abspos < max_exact
    scale = (num_buckets - max_exact) / math.log(max_distance / max_exact)
    big = (torch.log(abspos.float() / max_exact) * scale).to(torch.int16).to(torch.int64)
    big = torch.minimum(max_exact + big, torch.tensor(num_buckets - 1))
    buckets = torch.where(is_small, abspos, big)
    if bidirectional:
        buckets = buckets + (rpos > 0).long() * num_buckets
    else:
        buckets = buckets * (rpos < 0).long()
    return buckets


def t5_position_bias(cfg: T5Config, emb: Tensor, q_len: int, k_len: int) -> Tensor:
    """RelativePositionBias.__call__ (flux/t5.py:99-116): -> [H, q_len, k_len]."""
    ctx = torch.arange(q_len)[:, None]
    mem = torch.arange(k_len)[None, :]
    b = relative_position_bucket(mem - ctx, True, cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance)
    return emb[b].permute(2, 0, 1)


def t5_encoder(cfg: T5Config, W: Dict[str, Tensor], tokens: Tensor) -> Tensor:
    """T5Encoder.__call__ (flux/t5.py:243-244) -> TransformerEncoder (:218-224) -> layers (:200-207)."""
    x = W["wte.weight"][tokens.long()]
    B, T, _ = x.shape
    H = cfg.num_heads
    bias = t5_position_bias(cfg, W["encoder.relative_attention_bias.embeddings.weight"], T, T).to(x.dtype)
    for i in range(cfg.num_layers):
        p = f"encoder.layers.{i}"
        y = rms_norm(x, W[f"{p}.ln1.weight"], cfg.layer_norm_epsilon)
        q = linear(y, W[f"{p}.attention.query_proj.weight"]).reshape(B, T, H, -1).transpose(1, 2)
        k = linear(y, W[f"{p}.attention.key_proj.weight"]).reshape(B, T, H, -1).transpose(1, 2)
        v = linear(y, W[f"{p}.attention.value_proj.weight"]).reshape(B, T, H, -1).transpose(1, 2)
        # mx.fast.scaled_dot_product_attention(q, k, v, scale=1.0, mask=pos_bias): fp32 softmax
        s = torch.matmul(q.float(), k.float().transpose(-1, -2)) + bias.float()[None]
        o = torch.matmul(torch.softmax(s, dim=-1), v.float()).to(x.dtype).transpose(1, 2).reshape(B, T, -1)
        x = x + linear(o, W[f"{p}.attention.out_proj.weight"])
        y = rms_norm(x, W[f"{p}.ln2.weight"], cfg.layer_norm_epsilon)
        # DenseActivation (flux/t5.py:183-189): gated, act = nn.gelu (EXACT erf GELU, not HF's gelu_new)
        h = F.gelu(linear(y, W[f"{p}.dense.wi_0.weight"])) * linear(y, W[f"{p}.dense.wi_1.weight"])
        x = x + linear(h, W[f"{p}.dense.wo.weight"])
    return rms_norm(x, W["encoder.ln.weight"], cfg.layer_norm_epsilon)


def t5_weight_shapes(cfg: T5Config) -> Dict[str, Tuple[int, ...]]:
    inner = cfg.d_kv * cfg.num_heads
    s = {"wte.weight": (cfg.vocab_size, cfg.d_model),
         "encoder.relative_attention_bias.embeddings.weight": (cfg.relative_attention_num_buckets, cfg.num_heads),
         "encoder.ln.weight": (cfg.d_model,)}
    for i in range(cfg.num_layers):
        p = f"encoder.layers.{i}"
        s[f"{p}.ln1.weight"] = (cfg.d_model,)
        s[f"{p}.ln2.weight"] = (cfg.d_model,)
        for n in ("query_proj", "key_proj", "value_proj"):
            s[f"{p}.attention.{n}.weight"] = (inner, cfg.d_model)
        s[f"{p}.attention.out_proj.weight"] = (cfg.d_model, inner)
        s[f"{p}.dense.wi_0.weight"] = (cfg.d_ff, cfg.d_model)
        s[f"{p}.dense.wi_1.weight"] = (cfg.d_ff, cfg.d_model)
        s[f"{p}.dense.wo.weight"] = (cfg.d_model, cfg.d_ff)
    return s


# ------------------------------------------------------------------ CLIP text model (flux/clip.py)
@dataclass
class CLIPTextModelConfig:
    num_layers: int = 12
    model_dims: int = 768
    num_heads: int = 12
    max_length: int = 77
    vocab_size: int = 49408
    hidden_act: str = "quick_gelu"
    projection_dim: Optional[int] = None      # stable_diffusion/stable_diffusion/clip.py:76-79 (text_projection, no bias)


def quick_gelu(x: Tensor) -> Tensor:
    """nn.gelu_fast_approx = x * sigmoid(1.702 x)."""
    return x * torch.sigmoid(1.702 * x)


@dataclass
class CLIPOutput:
    pooled_output: Tensor
    last_hidden_state: Tensor
    hidden_states: List[Tensor]


def clip_text_model(cfg: CLIPTextModelConfig, W: Dict[str, Tensor], tokens: Tensor) -> CLIPOutput:
    """CLIPTextModel.__call__ (flux/clip.py:127-155) with CLIPEncoderLayer (:60-73)."""
    B, N = tokens.shape
    eos = tokens.argmax(-1)
    x = W["token_embedding.weight"][tokens.long()] + W["position_embedding.weight"][:N]
    H = cfg.num_heads
    idx = torch.arange(N)
    mask = (idx[:, None] < idx[None]).float() * -1e9
    act = quick_gelu if cfg.hidden_act == "quick_gelu" else F.gelu
    hs = []
    for i in range(cfg.num_layers):
        p = f"layers.{i}"
        y = F.layer_norm(x.float(), (cfg.model_dims,), W[f"{p}.layer_norm1.weight"].float(), W[f"{p}.layer_norm1.bias"].float(), 1e-5).to(x.dtype)
        q = linear(y, W[f"{p}.attention.query_proj.weight"], W[f"{p}.attention.query_proj.bias"]).reshape(B, N, H, -1).transpose(1, 2)
        k = linear(y, W[f"{p}.attention.key_proj.weight"], W[f"{p}.attention.key_proj.bias"]).reshape(B, N, H, -1).transpose(1, 2)
        v = linear(y, W[f"{p}.attention.value_proj.weight"], W[f"{p}.attention.value_proj.bias"]).reshape(B, N, H, -1).transpose(1, 2)
        s = torch.matmul(q.float() * math.sqrt(1 / q.shape[-1]), k.float().transpose(-1, -2)) + mask
        o = torch.matmul(torch.softmax(s, dim=-1), v.float()).to(x.dtype).transpose(1, 2).reshape(B, N, -1)
        x = linear(o, W[f"{p}.attention.out_proj.weight"], W[f"{p}.attention.out_proj.bias"]) + x
        y = F.layer_norm(x.float(), (cfg.model_dims,), W[f"{p}.layer_norm2.weight"].float(), W[f"{p}.layer_norm2.bias"].float(), 1e-5).to(x.dtype)
        y = linear(act(linear(y, W[f"{p}.linear1.weight"], W[f"{p}.linear1.bias"])), W[f"{p}.linear2.weight"], W[f"{p}.linear2.bias"])
        x = y + x
        hs.append(x)
    x = F.layer_norm(x.float(), (cfg.model_dims,), W["final_layer_norm.weight"].float(), W["final_layer_norm.bias"].float(), 1e-5).to(x.dtype)
    pooled = x[torch.arange(B), eos]
    if cfg.projection_dim is not None:      # stable_diffusion/stable_diffusion/clip.py:107-108
        pooled = linear(pooled, W["text_projection.weight"])
    return CLIPOutput(pooled_output=pooled, last_hidden_state=x, hidden_states=hs)


def clip_weight_shapes(cfg: CLIPTextModelConfig) -> Dict[str, Tuple[int, ...]]:
    D = cfg.model_dims
    s = {"token_embedding.weight": (cfg.vocab_size, D), "position_embedding.weight": (cfg.max_length, D),
         "final_layer_norm.weight": (D,), "final_layer_norm.bias": (D,)}
    for i in range(cfg.num_layers):
        p = f"layers.{i}"
        for n in ("layer_norm1", "layer_norm2"):
            s[f"{p}.{n}.weight"] = (D,)
            s[f"{p}.{n}.bias"] = (D,)
        for n in ("query_proj", "key_proj", "value_proj", "out_proj"):
            s[f"{p}.attention.{n}.weight"] = (D, D)
            s[f"{p}.attention.{n}.bias"] = (D,)
        s[f"{p}.linear1.weight"] = (4 * D, D); s[f"{p}.linear1.bias"] = (4 * D,)
        s[f"{p}.linear2.weight"] = (D, 4 * D); s[f"{p}.linear2.bias"] = (D,)
    if cfg.projection_dim is not None:
        s["text_projection.weight"] = (cfg.projection_dim, D)
    return s

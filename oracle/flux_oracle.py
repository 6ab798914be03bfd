"""CPU oracle for the Flux denoise + decode hot path.  TEST INFRASTRUCTURE ONLY.

This is a plain PyTorch-CPU restatement of the reference's algorithm (voipnuggets/flux-generator,
MLX).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product path (``flux_generator_amd``) never does and fails loudly without the HIP
library.

PARITY UNPINNED: the reference's arithmetic lives in third-party MLX (``mlx>=0.11``,
requirements.txt:2 — unpinned, not vendored, not importable in this image) and none of the
reference's own tests holds a numeric golden for this path.  The restatement
is therefore pinned only by (1) hand-derived known answers (tests/test_oracle_kat.py), (2)
agreement with independent torch.nn.functional implementations of each op, and (3) golden vectors
generated *by this oracle* (tests/golden/, script tests/golden/make_golden.py).  The MLX facts it assumes, what
pins each of them here and what an MLX run could still contradict are listed in oracle/UNVERIFIED.md.

Every function cites the reference lines it follows (paths relative to the reference tree).
All functions are dtype-generic: run them in float32 for the "exact" answer, or in bfloat16 to
reproduce MLX's op-boundary rounding (each MLX op returns an array of the promoted input dtype).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# configuration (flux/model.py:20-32, flux/autoencoder.py:11-21, flux/utils.py:30-95)
# --------------------------------------------------------------------------------------------
@dataclass
class FluxParams:
    in_channels: int = 64
    vec_in_dim: int = 768
    context_in_dim: int = 4096
    hidden_size: int = 3072
    mlp_ratio: float = 4.0
    num_heads: int = 24
    depth: int = 19
    depth_single_blocks: int = 38
    axes_dim: List[int] = field(default_factory=lambda: [16, 56, 56])
    theta: int = 10_000
    qkv_bias: bool = True
    guidance_embed: bool = False


@dataclass
class AutoEncoderParams:
    resolution: int = 256
    in_channels: int = 3
    ch: int = 128
    out_ch: int = 3
    ch_mult: List[int] = field(default_factory=lambda: [1, 2, 4, 4])
    num_res_blocks: int = 2
    z_channels: int = 16
    scale_factor: float = 0.3611
    shift_factor: float = 0.1159


# --------------------------------------------------------------------------------------------
# embeddings / rope (flux/layers.py:12-75)
# --------------------------------------------------------------------------------------------
def rope(pos: Tensor, dim: int, theta: float) -> Tensor:
    """flux/layers.py:12-21 — pos [...], returns [..., dim/2, 2, 2] fp32 rotation matrices."""
    scale = torch.arange(0, dim, 2, dtype=torch.float32) / dim
    omega = 1.0 / (theta ** scale)
    x = pos[..., None].to(torch.float32) * omega
    cosx, sinx = torch.cos(x), torch.sin(x)
    pe = torch.stack([cosx, -sinx, sinx, cosx], dim=-1)
    return pe.reshape(*pe.shape[:-1], 2, 2)


def embed_nd(ids: Tensor, axes_dim: List[int], theta: float) -> Tensor:
    """EmbedND.__call__ (flux/layers.py:67-75): ids [B,T,n_axes] -> pe [B,1,T,sum(axes)/2,2,2]."""
    pe = torch.cat([rope(ids[..., i], axes_dim[i], theta) for i in range(ids.shape[-1])], dim=-3)
    return pe[:, None]


def apply_rope(x: Tensor, pe: Tensor) -> Tensor:
    """_apply_rope (flux/layers.py:29-33): pairs are consecutive elements; a*b + c*d."""
    s = x.shape
    x = x.reshape(*s[:-1], -1, 1, 2)
    x = x[..., 0] * pe[..., 0] + x[..., 1] * pe[..., 1]
    return x.reshape(s)


def timestep_embedding(t: Tensor, dim: int, max_period: int = 10000, time_factor: float = 1000.0) -> Tensor:
    """flux/layers.py:46-57. ``time_factor * t`` stays in t.dtype (python scalar x array), the
    product with the fp32 freqs promotes to fp32, the result is cast back to t.dtype."""
    half = dim // 2
    freqs = torch.arange(0, half, dtype=torch.float32) / half
    freqs = torch.exp(freqs * (-math.log(max_period)))
    x = (time_factor * t)[:, None].to(torch.float32) * freqs[None]
    x = torch.cat([torch.cos(x), torch.sin(x)], dim=-1)
    return x.to(t.dtype)


# --------------------------------------------------------------------------------------------
# primitive ops with MLX semantics
# --------------------------------------------------------------------------------------------
def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """nn.Linear: y = x W^T + b, W [out,in]."""
    return F.linear(x, w, b)


def layer_norm(x: Tensor, eps: float = 1e-6) -> Tensor:
    """nn.LayerNorm(affine=False, eps=1e-6) (flux/layers.py:156): fp32 statistics, biased var."""
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(-1, keepdim=True)
    return ((xf - mu) * torch.rsqrt(var + eps)).to(x.dtype)


def rms_norm(x: Tensor, w: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.RMSNorm(dims) default eps 1e-5 (flux/layers.py:91-92)."""
    xf = x.float()
    y = xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + eps) * w.float()
    return y.to(x.dtype)


def sdpa(q: Tensor, k: Tensor, v: Tensor, scale: float) -> Tensor:
    """mx.fast.scaled_dot_product_attention (no mask): fp32 softmax, output in input dtype."""
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, v.float()).to(q.dtype)


def gelu_tanh(x: Tensor) -> Tensor:
    """nn.GELU(approx="tanh") (flux/layers.py:164,177,259)."""
    return F.gelu(x, approximate="tanh")


def silu(x: Tensor) -> Tensor:
    return F.silu(x)


# --------------------------------------------------------------------------------------------
# Flux MMDiT (flux/layers.py:78-302, flux/model.py:99-136)
# Weights: dict keyed by the reference's *sanitized* parameter names (flux/model.py:85-97).
# --------------------------------------------------------------------------------------------
def mlp_embedder(W: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """MLPEmbedder (flux/layers.py:78-85)."""
    h = linear(x, W[f"{prefix}.in_layer.weight"], W[f"{prefix}.in_layer.bias"])
    return linear(silu(h), W[f"{prefix}.out_layer.weight"], W[f"{prefix}.out_layer.bias"])


def modulation(W: Dict[str, Tensor], prefix: str, vec: Tensor, multiplier: int) -> List[Tensor]:
    """Modulation (flux/layers.py:129-143): lin(silu(vec)) split into shift/scale/gate triples."""
    x = linear(silu(vec), W[f"{prefix}.lin.weight"], W[f"{prefix}.lin.bias"])
    return list(torch.chunk(x[:, None, :], multiplier, dim=-1))


def attention(q: Tensor, k: Tensor, v: Tensor, pe: Tensor) -> Tensor:
    """_attention (flux/layers.py:36-43)."""
    B, H, L, D = q.shape
    q, k = apply_rope(q, pe), apply_rope(k, pe)
    x = sdpa(q, k, v, scale=D ** (-0.5))
    return x.transpose(1, 2).reshape(B, L, -1)


def _split_heads(x: Tensor, H: int) -> Tensor:
    B, L, _ = x.shape
    return x.reshape(B, L, H, -1).transpose(1, 2)


def double_stream_block(W, prefix: str, H: int, img: Tensor, txt: Tensor, vec: Tensor, pe: Tensor):
    """DoubleStreamBlock.__call__ (flux/layers.py:181-231)."""
    S = txt.shape[1]
    i1s, i1c, i1g, i2s, i2c, i2g = modulation(W, f"{prefix}.img_mod", vec, 6)
    t1s, t1c, t1g, t2s, t2c, t2g = modulation(W, f"{prefix}.txt_mod", vec, 6)

    def qkv(stream, x, shift, scale):
        xm = (1 + scale) * layer_norm(x) + shift
        o = linear(xm, W[f"{prefix}.{stream}_attn.qkv.weight"], W.get(f"{prefix}.{stream}_attn.qkv.bias"))
        q, k, v = torch.chunk(o, 3, dim=-1)
        q, k, v = _split_heads(q, H), _split_heads(k, H), _split_heads(v, H)
        q = rms_norm(q, W[f"{prefix}.{stream}_attn.norm.query_norm.weight"])
        k = rms_norm(k, W[f"{prefix}.{stream}_attn.norm.key_norm.weight"])
        return q, k, v

    iq, ik, iv = qkv("img", img, i1s, i1c)
    tq, tk, tv = qkv("txt", txt, t1s, t1c)
    q = torch.cat([tq, iq], dim=2)
    k = torch.cat([tk, ik], dim=2)
    v = torch.cat([tv, iv], dim=2)
    attn = attention(q, k, v, pe)
    txt_attn, img_attn = attn[:, :S], attn[:, S:]

    def tail(stream, x, a, g1, s2, c2, g2):
        x = x + g1 * linear(a, W[f"{prefix}.{stream}_attn.proj.weight"], W[f"{prefix}.{stream}_attn.proj.bias"])
        h = (1 + c2) * layer_norm(x) + s2
        h = linear(h, W[f"{prefix}.{stream}_mlp.layers.0.weight"], W[f"{prefix}.{stream}_mlp.layers.0.bias"])
        h = gelu_tanh(h)
        h = linear(h, W[f"{prefix}.{stream}_mlp.layers.2.weight"], W[f"{prefix}.{stream}_mlp.layers.2.bias"])
        return x + g2 * h

    img = tail("img", img, img_attn, i1g, i2s, i2c, i2g)
    txt = tail("txt", txt, txt_attn, t1g, t2s, t2c, t2g)
    return img, txt


def single_stream_block(W, prefix: str, H: int, x: Tensor, vec: Tensor, pe: Tensor) -> Tensor:
    """SingleStreamBlock.__call__ (flux/layers.py:262-284)."""
    hidden = x.shape[-1]
    shift, scale, gate = modulation(W, f"{prefix}.modulation", vec, 3)
    x_mod = (1 + scale) * layer_norm(x) + shift
    o = linear(x_mod, W[f"{prefix}.linear1.weight"], W[f"{prefix}.linear1.bias"])
    q, k, v, mlp = torch.split(o, [hidden, hidden, hidden, o.shape[-1] - 3 * hidden], dim=-1)
    q, k, v = _split_heads(q, H), _split_heads(k, H), _split_heads(v, H)
    q = rms_norm(q, W[f"{prefix}.norm.query_norm.weight"])
    k = rms_norm(k, W[f"{prefix}.norm.key_norm.weight"])
    y = attention(q, k, v, pe)
    y = linear(torch.cat([y, gelu_tanh(mlp)], dim=2), W[f"{prefix}.linear2.weight"], W[f"{prefix}.linear2.bias"])
    return x + gate * y


def last_layer(W, x: Tensor, vec: Tensor) -> Tensor:
    """LastLayer.__call__ (flux/layers.py:298-302)."""
    m = linear(silu(vec), W["final_layer.adaLN_modulation.layers.1.weight"],
               W["final_layer.adaLN_modulation.layers.1.bias"])
    shift, scale = torch.chunk(m, 2, dim=1)
    x = (1 + scale[:, None, :]) * layer_norm(x) + shift[:, None, :]
    return linear(x, W["final_layer.linear.weight"], W["final_layer.linear.bias"])


def flux_forward(P: FluxParams, W: Dict[str, Tensor], img: Tensor, img_ids: Tensor, txt: Tensor,
                 txt_ids: Tensor, timesteps: Tensor, y: Tensor, guidance: Optional[Tensor] = None,
                 table_dtype=torch.bfloat16) -> Tensor:
    """Flux.__call__ (flux/model.py:99-136).

    ``table_dtype`` is the reference pipeline's dtype (bf16, flux/flux.py:24): the timestep /
    guidance vectors are created in it (flux/flux.py:101-102), so ``1000*t`` and the sinusoidal
    table are rounded to it (flux/layers.py:54-57), and so is the RoPE table (flux/model.py:124).
    These roundings are part of the function being computed, not arithmetic noise, so they are
    applied even when the rest of the oracle runs in fp32."""
    if img.ndim != 3 or txt.ndim != 3:
        raise ValueError("Input img and txt tensors must have 3 dimensions.")
    dt = img.dtype
    img = linear(img, W["img_in.weight"], W["img_in.bias"])
    vec = mlp_embedder(W, "time_in", timestep_embedding(timesteps.to(table_dtype), 256).to(dt))
    if P.guidance_embed:
        if guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")
        vec = vec + mlp_embedder(W, "guidance_in", timestep_embedding(guidance.to(table_dtype), 256).to(dt))
    vec = vec + mlp_embedder(W, "vector_in", y)
    txt = linear(txt, W["txt_in.weight"], W["txt_in.bias"])

    ids = torch.cat([txt_ids, img_ids], dim=1)
    pe = embed_nd(ids, P.axes_dim, P.theta).to(table_dtype).to(dt)

    for i in range(P.depth):
        img, txt = double_stream_block(W, f"double_blocks.{i}", P.num_heads, img, txt, vec, pe)
    x = torch.cat([txt, img], dim=1)
    for i in range(P.depth_single_blocks):
        x = single_stream_block(W, f"single_blocks.{i}", P.num_heads, x, vec, pe)
    x = x[:, txt.shape[1]:, ...]
    return last_layer(W, x, vec)


def flux_weight_shapes(P: FluxParams) -> Dict[str, Tuple[int, ...]]:
    """Parameter tree of Flux (flux/model.py:35-83, flux/layers.py) under sanitized names."""
    Hd, mlp = P.hidden_size, int(P.hidden_size * P.mlp_ratio)
    hd = Hd // P.num_heads
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(name, out_d, in_d, bias=True):
        s[f"{name}.weight"] = (out_d, in_d)
        if bias:
            s[f"{name}.bias"] = (out_d,)

    lin("img_in", Hd, P.in_channels)
    lin("txt_in", Hd, P.context_in_dim)
    for e, d in (("time_in", 256), ("vector_in", P.vec_in_dim)) + ((("guidance_in", 256),) if P.guidance_embed else ()):
        lin(f"{e}.in_layer", Hd, d)
        lin(f"{e}.out_layer", Hd, Hd)
    for i in range(P.depth):
        p = f"double_blocks.{i}"
        for st in ("img", "txt"):
            lin(f"{p}.{st}_mod.lin", 6 * Hd, Hd)
            lin(f"{p}.{st}_attn.qkv", 3 * Hd, Hd, bias=P.qkv_bias)
            s[f"{p}.{st}_attn.norm.query_norm.weight"] = (hd,)
            s[f"{p}.{st}_attn.norm.key_norm.weight"] = (hd,)
            lin(f"{p}.{st}_attn.proj", Hd, Hd)
            lin(f"{p}.{st}_mlp.layers.0", mlp, Hd)
            lin(f"{p}.{st}_mlp.layers.2", Hd, mlp)
    for i in range(P.depth_single_blocks):
        p = f"single_blocks.{i}"
        lin(f"{p}.modulation.lin", 3 * Hd, Hd)
        lin(f"{p}.linear1", 3 * Hd + mlp, Hd)
        lin(f"{p}.linear2", Hd, Hd + mlp)
        s[f"{p}.norm.query_norm.weight"] = (hd,)
        s[f"{p}.norm.key_norm.weight"] = (hd,)
    lin("final_layer.adaLN_modulation.layers.1", 2 * Hd, Hd)
    lin("final_layer.linear", P.in_channels, Hd)
    return s


def init_weights(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, dtype=torch.float32,
                 norm_jitter: float = 0.0) -> Dict[str, Tensor]:
    """Random init like MLX's defaults: Linear/Conv W,b ~ U(-1/sqrt(fan_in), +),
    norm scales = 1 (+ optional jitter so tests exercise the affine), norm biases = 0."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, Tensor] = {}
    pending_bias: Dict[str, float] = {}
    for name, shp in shapes.items():
        if name.endswith(".weight") and len(shp) >= 2:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            k = 1.0 / math.sqrt(fan_in)
            W[name] = ((torch.rand(shp, generator=g) * 2 - 1) * k).to(dtype)
            pending_bias[name[:-7]] = k
        elif name.endswith(".bias") and name[:-5] in pending_bias:
            k = pending_bias[name[:-5]]
            W[name] = ((torch.rand(shp, generator=g) * 2 - 1) * k).to(dtype)
        elif name.endswith(".weight"):   # norm scale
            W[name] = (1.0 + norm_jitter * (torch.rand(shp, generator=g) * 2 - 1)).to(dtype)
        else:                              # norm bias
            W[name] = (norm_jitter * (torch.rand(shp, generator=g) * 2 - 1)).to(dtype)
    return W


# --------------------------------------------------------------------------------------------
# sampler (flux/sampler.py)
# --------------------------------------------------------------------------------------------
def time_shift(x: float, t: float, base_shift: float = 0.5, max_shift: float = 1.15) -> float:
    """FluxSampler._time_shift (flux/sampler.py:15-20), elementwise; t = 0 -> 0 (1/0 = inf)."""
    x1, x2 = 256, 4096
    exp_mu = math.exp((x - x1) * (max_shift - base_shift) / (x2 - x1) + base_shift)
    if t == 0:
        return 0.0
    return exp_mu / (exp_mu + (1 / t - 1))


def timesteps(name: str, num_steps: int, image_sequence_length: int, start: float = 1.0,
              stop: float = 0.0) -> List[float]:
    """FluxSampler.timesteps (flux/sampler.py:22-31): linspace in fp32; shifted unless schnell."""
    t = torch.linspace(start, stop, num_steps + 1, dtype=torch.float32).tolist()
    if "schnell" not in name:
        t = [float(torch.tensor(time_shift(image_sequence_length, v), dtype=torch.float32)) for v in t]
    return t


def euler_step(pred: Tensor, x_t: Tensor, t: float, t_prev: float, table_dtype=torch.bfloat16) -> Tensor:
    """FluxSampler.step (flux/sampler.py:56-57). MLX converts the python scalar (t_prev - t) to the
    array dtype before multiplying (weak scalar typing): with the pipeline's bf16 latents dt is
    rounded to bf16. That rounding is part of the function, so it is kept in an fp32 oracle run."""
    return x_t + torch.tensor(t_prev - t, dtype=table_dtype).to(pred.dtype) * pred


# --------------------------------------------------------------------------------------------
# pipeline glue (flux/flux.py)
# --------------------------------------------------------------------------------------------
def prepare_latent_images(x: Tensor) -> Tuple[Tensor, Tensor]:
    """FluxPipeline._prepare_latent_images (flux/flux.py:53-71): x [b,h,w,c] NHWC."""
    b, h, w, c = x.shape
    x = x.reshape(b, h // 2, 2, w // 2, 2, c)
    x = x.permute(0, 1, 3, 5, 2, 4).reshape(b, h * w // 4, c * 4)
    i = torch.zeros((h // 2, w // 2), dtype=torch.int32)
    j, k = torch.meshgrid(torch.arange(h // 2, dtype=torch.int32), torch.arange(w // 2, dtype=torch.int32),
                          indexing="ij")
    x_ids = torch.stack([i, j, k], dim=-1).reshape(1, h * w // 4, 3).repeat(b, 1, 1)
    return x, x_ids


def unpack_latents(x: Tensor, latent_size: Tuple[int, int]) -> Tensor:
    """First half of FluxPipeline.decode (flux/flux.py:158-160)."""
    h, w = latent_size
    x = x.reshape(len(x), h // 2, w // 2, -1, 2, 2)
    return x.permute(0, 1, 4, 2, 5, 3).reshape(len(x), h, w, -1)


def denoising_loop(P: FluxParams, W, name: str, x_t: Tensor, x_ids: Tensor, txt: Tensor, txt_ids: Tensor,
                   vec: Tensor, num_steps: int, guidance: float = 4.0, dtype=torch.bfloat16):
    """FluxPipeline._denoising_loop (flux/flux.py:87-126). ``scalar(x)`` = full((B,), x, dtype):
    the timestep and guidance are rounded to the pipeline dtype (bf16 in the reference)."""
    B = len(x_t)
    ts = timesteps(name, num_steps, x_t.shape[1])
    g = torch.full((B,), guidance, dtype=dtype).to(x_t.dtype)
    out = []
    for i in range(num_steps):
        t, t_prev = ts[i], ts[i + 1]
        tt = torch.full((B,), t, dtype=dtype).to(x_t.dtype)
        pred = flux_forward(P, W, x_t, x_ids, txt, txt_ids, tt, vec, g)
        x_t = euler_step(pred, x_t, t, t_prev)
        out.append(x_t)
    return out


# --------------------------------------------------------------------------------------------
# AutoEncoder decoder (flux/autoencoder.py). Activations NHWC, conv weights [O,kh,kw,I]
# (the layout AutoEncoder.sanitize produces, flux/autoencoder.py:336-345).
# --------------------------------------------------------------------------------------------
def group_norm(x: Tensor, w: Tensor, b: Tensor, groups: int = 32, eps: float = 1e-6) -> Tensor:
    """nn.GroupNorm(32, C, eps=1e-6, affine, pytorch_compatible=True) on NHWC."""
    y = F.group_norm(x.permute(0, 3, 1, 2).float(), groups, w.float(), b.float(), eps)
    return y.permute(0, 2, 3, 1).to(x.dtype)


def conv2d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 1, padding: int = 1) -> Tensor:
    """nn.Conv2d on NHWC with weight [O,kh,kw,I]."""
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, stride=stride, padding=padding)
    return y.permute(0, 2, 3, 1)


def upsample_nearest2(x: Tensor) -> Tensor:
    """upsample_nearest(x, (2, 2)) (flux/autoencoder.py:121)."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def resnet_block(W, p: str, x: Tensor) -> Tensor:
    """ResnetBlock.__call__ (flux/autoencoder.py:83-98)."""
    h = silu(group_norm(x, W[f"{p}.norm1.weight"], W[f"{p}.norm1.bias"]))
    h = conv2d(h, W[f"{p}.conv1.weight"], W[f"{p}.conv1.bias"])
    h = silu(group_norm(h, W[f"{p}.norm2.weight"], W[f"{p}.norm2.bias"]))
    h = conv2d(h, W[f"{p}.conv2.weight"], W[f"{p}.conv2.bias"])
    if f"{p}.nin_shortcut.weight" in W:
        x = linear(x, W[f"{p}.nin_shortcut.weight"], W[f"{p}.nin_shortcut.bias"])
    return x + h


def attn_block(W, p: str, x: Tensor) -> Tensor:
    """AttnBlock.__call__ (flux/autoencoder.py:42-52): single head over H*W tokens."""
    B, H, Wd, C = x.shape
    y = x.reshape(B, 1, -1, C)
    y = group_norm(y.reshape(B, H, Wd, C), W[f"{p}.norm.weight"], W[f"{p}.norm.bias"]).reshape(B, 1, -1, C)
    q = linear(y, W[f"{p}.q.weight"], W[f"{p}.q.bias"])
    k = linear(y, W[f"{p}.k.weight"], W[f"{p}.k.bias"])
    v = linear(y, W[f"{p}.v.weight"], W[f"{p}.v.bias"])
    y = sdpa(q, k, v, scale=C ** (-0.5))
    y = linear(y, W[f"{p}.proj_out.weight"], W[f"{p}.proj_out.bias"])
    return x + y.reshape(B, H, Wd, C)


def decoder_forward(A: AutoEncoderParams, W, z: Tensor) -> Tensor:
    """Decoder.__call__ (flux/autoencoder.py:271-297)."""
    nres = len(A.ch_mult)
    h = conv2d(z, W["decoder.conv_in.weight"], W["decoder.conv_in.bias"])
    h = resnet_block(W, "decoder.mid.block_1", h)
    h = attn_block(W, "decoder.mid.attn_1", h)
    h = resnet_block(W, "decoder.mid.block_2", h)
    for lvl in reversed(range(nres)):
        for i in range(A.num_res_blocks + 1):
            h = resnet_block(W, f"decoder.up.{lvl}.block.{i}", h)
        if lvl != 0:
            h = upsample_nearest2(h)
            h = conv2d(h, W[f"decoder.up.{lvl}.upsample.conv.weight"], W[f"decoder.up.{lvl}.upsample.conv.bias"])
    h = silu(group_norm(h, W["decoder.norm_out.weight"], W["decoder.norm_out.bias"]))
    return conv2d(h, W["decoder.conv_out.weight"], W["decoder.conv_out.bias"])


def ae_decode(A: AutoEncoderParams, W, z: Tensor) -> Tensor:
    """AutoEncoder.decode (flux/autoencoder.py:352-354)."""
    z = z / A.scale_factor + A.shift_factor
    return decoder_forward(A, W, z)


def pipeline_decode(A: AutoEncoderParams, W, x: Tensor, latent_size: Tuple[int, int]) -> Tensor:
    """FluxPipeline.decode (flux/flux.py:157-162): unpack, VAE decode, clip(x+1,0,2)*0.5.
    The AE weights are fp32 in the reference checkpoint, so bf16 latents promote to fp32."""
    z = unpack_latents(x, latent_size)
    wdtype = W["decoder.conv_in.weight"].dtype
    y = ae_decode(A, W, z.to(wdtype))
    return torch.clip(y + 1, 0, 2) * 0.5


def decoder_weight_shapes(A: AutoEncoderParams) -> Dict[str, Tuple[int, ...]]:
    """Parameter tree of Decoder (flux/autoencoder.py:212-269), sanitized layouts."""
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

"""CPU oracle for the stable_diffusion/ hot path (UNet, Euler samplers, CFG, SD VAE decoder).
TEST INFRASTRUCTURE ONLY — same rules and same "PARITY UNPINNED" status as oracle/flux_oracle.py:
the reference's arithmetic lives in MLX (not importable here) and its tests hold no numeric golden.

Plain PyTorch-CPU restatement of stable_diffusion/stable_diffusion/{unet,sampler,vae,__init__}.py;
every function cites the reference lines it follows.  Weights are a dict keyed by the MLX module
tree's flattened names (what model_io.map_unet_weights / map_vae_weights produce).
Activations NHWC, conv weights [O,kh,kw,I], 1x1 convs as Linear [O,I] (model_io.py:84-93).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .flux_oracle import conv2d, group_norm, linear, silu, upsample_nearest2

Tensor = torch.Tensor


# ------------------------------------------------------------------ configs (config.py:8-65)
@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    conv_in_kernel: int = 3
    conv_out_kernel: int = 3
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: Tuple[int, ...] = (2, 2, 2, 2)
    transformer_layers_per_block: Tuple[int, ...] = (1, 1, 1, 1)
    num_attention_heads: Tuple[int, ...] = (5, 10, 20, 20)
    cross_attention_dim: Tuple[int, ...] = (1024,) * 4
    norm_num_groups: int = 32
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                         "DownBlock2D")
    up_block_types: Tuple[str, ...] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    addition_embed_type: Optional[str] = None
    addition_time_embed_dim: Optional[int] = None
    projection_class_embeddings_input_dim: Optional[int] = None


@dataclass
class AutoencoderConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels_out: int = 8
    latent_channels_in: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


@dataclass
class DiffusionConfig:
    beta_schedule: str = "scaled_linear"
    beta_start: float = 0.00085
    beta_end: float = 0.012
    num_train_steps: int = 1000


# ------------------------------------------------------------------ samplers (sampler.py)
def _linspace(a, b, num):
    """sampler.py:8-10."""
    x = torch.arange(0, num, dtype=torch.float32) / (num - 1)
    return (b - a) * x + a


def _interp(y: Tensor, x_new: Tensor) -> Tensor:
    """sampler.py:13-23: linear interpolation of y at fractional index x_new."""
    x_low = x_new.to(torch.int32).long()
    x_high = torch.clamp(x_low + 1, max=len(y) - 1)
    y_low, y_high = y[x_low], y[x_high]
    delta = x_new - x_low
    return y_low * (1 - delta) + delta * y_high


class EulerSampler:
    """SimpleEulerSampler (sampler.py:26-85)."""

    def __init__(self, config: DiffusionConfig):
        if config.beta_schedule == "linear":
            betas = _linspace(config.beta_start, config.beta_end, config.num_train_steps)
        elif config.beta_schedule == "scaled_linear":
            betas = _linspace(config.beta_start ** 0.5, config.beta_end ** 0.5, config.num_train_steps).square()
        else:
            raise NotImplementedError(f"{config.beta_schedule} is not implemented.")
        alphas_cumprod = torch.cumprod(1 - betas, dim=0)
        self._sigmas = torch.cat([torch.zeros(1), ((1 - alphas_cumprod) / alphas_cumprod).sqrt()])

    @property
    def max_time(self):
        return len(self._sigmas) - 1

    def prior_scale(self) -> float:
        s = self._sigmas[-1]
        return float(s * torch.rsqrt(s.square() + 1))

    def sigmas(self, t) -> Tensor:
        return _interp(self._sigmas, torch.as_tensor(t, dtype=torch.float32))

    def timesteps(self, num_steps: int, start_time=None):
        start_time = start_time or (len(self._sigmas) - 1)
        assert 0 < start_time <= (len(self._sigmas) - 1)
        steps = _linspace(start_time, 0, num_steps + 1)
        return list(zip(steps.tolist(), steps[1:].tolist()))

    def step(self, eps_pred: Tensor, x_t: Tensor, t, t_prev, noise: Optional[Tensor] = None) -> Tensor:
        sigma, sigma_prev = self.sigmas(t).to(eps_pred.dtype), self.sigmas(t_prev).to(eps_pred.dtype)
        dt = sigma_prev - sigma
        x = (sigma.square() + 1).sqrt() * x_t + eps_pred * dt
        return x * torch.rsqrt(sigma_prev.square() + 1)


class EulerAncestralSampler(EulerSampler):
    """SimpleEulerAncestralSampler.step (sampler.py:88-105). The fresh N(0,1) draw is an input here
    (MLX's RNG stream cannot be reproduced; parity is defined on identical noise)."""

    def coefficients(self, t, t_prev, dtype=torch.float32):
        """sigma and sigma_prev are cast to the eps dtype FIRST (sampler.py:90-91) and everything derived from them
        (sigma^2, sigma_up, sigma_down) is then evaluated in that dtype (sampler.py:93-96): irrelevant in float32,
        visible in float16 (the reference's UNet dtype), where e.g. sigma^2 = 213.6 rounds to an 11-bit mantissa."""
        sigma, sigma_prev = self.sigmas(t).to(dtype), self.sigmas(t_prev).to(dtype)
        sigma2, sigma_prev2 = sigma.square(), sigma_prev.square()
        sigma_up = (sigma_prev2 * (sigma2 - sigma_prev2) / sigma2).sqrt()
        sigma_down = (sigma_prev2 - sigma_up ** 2).sqrt()
        return sigma, sigma_prev, sigma_up, sigma_down

    def step(self, eps_pred, x_t, t, t_prev, noise=None):
        sigma, sigma_prev, sigma_up, sigma_down = self.coefficients(t, t_prev, eps_pred.dtype)
        dt = sigma_down - sigma
        x = (sigma.square() + 1).sqrt() * x_t + eps_pred * dt
        x = x + noise.to(x.dtype) * sigma_up
        return x * torch.rsqrt(sigma_prev.square() + 1)


# ------------------------------------------------------------------ UNet pieces (unet.py)
def sinusoidal_encoding(x: Tensor, dims: int) -> Tensor:
    """nn.SinusoidalPositionalEncoding(dims, min_freq=exp(-ln1e4 + 2 ln1e4/dims), max_freq=1, scale=1,
    cos_first=True, full_turns=False) as configured at unet.py:283-292 (MLX semantics):
    sigma_i = exp(lerp(ln max, ln min, i/(dims/2-1))), y = [cos(x sigma) | sin(x sigma)]."""
    half = dims // 2
    min_freq = math.exp(-math.log(10000) + 2 * math.log(10000) / dims)
    one_zero = 1 - torch.arange(0, half, dtype=torch.float32) / (half - 1)
    lmin, lmax = math.log(min_freq), math.log(1.0)
    sig = torch.exp(one_zero * (lmax - lmin) + lmin)
    y = x.to(torch.float32)[..., None] * sig
    return torch.cat([torch.cos(y), torch.sin(y)], dim=-1)


def timestep_embedding(W, p: str, x: Tensor) -> Tensor:
    """TimestepEmbedding (unet.py:20-32)."""
    return linear(silu(linear(x, W[f"{p}.linear_1.weight"], W[f"{p}.linear_1.bias"])), W[f"{p}.linear_2.weight"],
                  W[f"{p}.linear_2.bias"])


def layer_norm_affine(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm(dims) (affine, eps 1e-5) as used by TransformerBlock (unet.py:45,50,57)."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps).to(x.dtype)


def mha(W, p: str, H: int, q_in: Tensor, kv_in: Tensor) -> Tensor:
    """nn.MultiHeadAttention without q/k/v bias, out_proj with bias (unet.py:46-54), no mask."""
    q = linear(q_in, W[f"{p}.query_proj.weight"])
    k = linear(kv_in, W[f"{p}.key_proj.weight"])
    v = linear(kv_in, W[f"{p}.value_proj.weight"])
    B, L, D = q.shape
    S = k.shape[1]
    q = q.reshape(B, L, H, -1).transpose(1, 2)
    k = k.reshape(B, S, H, -1).transpose(1, 2)
    v = v.reshape(B, S, H, -1).transpose(1, 2)
    scale = math.sqrt(1 / q.shape[-1])
    s = torch.matmul(q.float() * scale, k.float().transpose(-1, -2))
    o = torch.matmul(torch.softmax(s, dim=-1), v.float()).to(q_in.dtype)
    o = o.transpose(1, 2).reshape(B, L, D)
    return linear(o, W[f"{p}.out_proj.weight"], W[f"{p}.out_proj.bias"])


def transformer_block(W, p: str, H: int, x: Tensor, memory: Tensor) -> Tensor:
    """TransformerBlock.__call__ (unet.py:61-81): self-attn, cross-attn, GEGLU FFN (exact erf GELU)."""
    y = layer_norm_affine(x, W[f"{p}.norm1.weight"], W[f"{p}.norm1.bias"])
    x = x + mha(W, f"{p}.attn1", H, y, y)
    y = layer_norm_affine(x, W[f"{p}.norm2.weight"], W[f"{p}.norm2.bias"])
    x = x + mha(W, f"{p}.attn2", H, y, memory)
    y = layer_norm_affine(x, W[f"{p}.norm3.weight"], W[f"{p}.norm3.bias"])
    ya = linear(y, W[f"{p}.linear1.weight"], W[f"{p}.linear1.bias"])
    yb = linear(y, W[f"{p}.linear2.weight"], W[f"{p}.linear2.bias"])
    y = linear(ya * F.gelu(yb), W[f"{p}.linear3.weight"], W[f"{p}.linear3.bias"])
    return x + y


def transformer_2d(W, p: str, H: int, n_layers: int, x: Tensor, encoder_x: Tensor, groups: int = 32) -> Tensor:
    """Transformer2D.__call__ (unet.py:108-124)."""
    B, Hh, Ww, C = x.shape
    y = group_norm(x, W[f"{p}.norm.weight"], W[f"{p}.norm.bias"], groups, 1e-5).reshape(B, -1, C)
    y = linear(y, W[f"{p}.proj_in.weight"], W[f"{p}.proj_in.bias"])
    for i in range(n_layers):
        y = transformer_block(W, f"{p}.transformer_blocks.{i}", H, y, encoder_x)
    y = linear(y, W[f"{p}.proj_out.weight"], W[f"{p}.proj_out.bias"])
    return y.reshape(B, Hh, Ww, C) + x


def resnet_block_2d(W, p: str, x: Tensor, temb: Optional[Tensor] = None, groups: int = 32) -> Tensor:
    """ResnetBlock2D.__call__ (unet.py:152-170); GroupNorm eps default 1e-5."""
    if temb is not None:
        temb = linear(silu(temb), W[f"{p}.time_emb_proj.weight"], W[f"{p}.time_emb_proj.bias"])
    y = silu(group_norm(x, W[f"{p}.norm1.weight"], W[f"{p}.norm1.bias"], groups, 1e-5))
    y = conv2d(y, W[f"{p}.conv1.weight"], W[f"{p}.conv1.bias"])
    if temb is not None:
        y = y + temb[:, None, None, :]
    y = silu(group_norm(y, W[f"{p}.norm2.weight"], W[f"{p}.norm2.bias"], groups, 1e-5))
    y = conv2d(y, W[f"{p}.conv2.weight"], W[f"{p}.conv2.bias"])
    if f"{p}.conv_shortcut.weight" in W:
        x = linear(x, W[f"{p}.conv_shortcut.weight"], W[f"{p}.conv_shortcut.bias"])
    return y + x


def _block_plan(cfg: UNetConfig):
    """Static structure of UNetModel.__init__ (unet.py:318-389): per block the resnet in/out channels,
    whether it has attention / down / up sampling. Returns (down, up) lists in EXECUTION order."""
    boc = list(cfg.block_out_channels)
    n = len(boc)
    down = []
    chans = [boc[0]] + boc
    for i, (ic, oc) in enumerate(zip(chans, chans[1:])):
        L = cfg.layers_per_block[i]
        down.append(dict(idx=i, resnets=[(ic if j == 0 else oc, oc) for j in range(L)],
                         attn="CrossAttn" in cfg.down_block_types[i], heads=cfg.num_attention_heads[i],
                         tlayers=cfg.transformer_layers_per_block[i], enc=cfg.cross_attention_dim[i],
                         down=i < n - 1, up=False, out=oc))
    chans = [boc[0]] + boc + [boc[-1]]
    up = []
    for i, (ic, oc, pc) in reversed(list(enumerate(zip(chans, chans[1:], chans[2:])))):
        L = cfg.layers_per_block[i] + 1
        in_list = [pc] + [oc] * (L - 1)
        res_list = [oc] * (L - 1) + [ic]
        up.append(dict(idx=i, resnets=[(a + b, oc) for a, b in zip(in_list, res_list)],
                       attn="CrossAttn" in cfg.up_block_types[i], heads=cfg.num_attention_heads[i],
                       tlayers=cfg.transformer_layers_per_block[i], enc=cfg.cross_attention_dim[i],
                       down=False, up=i > 0, out=oc))
    return down, up


def unet_block(W, p: str, blk: dict, x, encoder_x, temb, residuals=None, groups=32):
    """UNetBlock2D.__call__ (unet.py:232-267)."""
    outs = []
    for j in range(len(blk["resnets"])):
        if residuals is not None:
            x = torch.cat([x, residuals.pop()], dim=-1)
        x = resnet_block_2d(W, f"{p}.resnets.{j}", x, temb, groups)
        if blk["attn"]:
            x = transformer_2d(W, f"{p}.attentions.{j}", blk["heads"], blk["tlayers"], x, encoder_x, groups)
        outs.append(x)
    if blk["down"]:
        x = conv2d(x, W[f"{p}.downsample.weight"], W[f"{p}.downsample.bias"], stride=2, padding=1)
        outs.append(x)
    if blk["up"]:
        x = conv2d(upsample_nearest2(x), W[f"{p}.upsample.weight"], W[f"{p}.upsample.bias"])
        outs.append(x)
    return x, outs


def unet_forward(cfg: UNetConfig, W, x: Tensor, timestep: Tensor, encoder_x: Tensor, text_time=None) -> Tensor:
    """UNetModel.__call__ (unet.py:403-460)."""
    boc = cfg.block_out_channels
    g = cfg.norm_num_groups
    temb = sinusoidal_encoding(timestep, boc[0]).to(x.dtype)
    temb = timestep_embedding(W, "time_embedding", temb)
    if text_time is not None:
        text_emb, time_ids = text_time
        emb = sinusoidal_encoding(time_ids, cfg.addition_time_embed_dim).flatten(1).to(x.dtype)
        emb = torch.cat([text_emb, emb], dim=-1)
        temb = temb + timestep_embedding(W, "add_embedding", emb)
    x = conv2d(x, W["conv_in.weight"], W["conv_in.bias"], padding=(cfg.conv_in_kernel - 1) // 2)
    down, up = _block_plan(cfg)
    residuals = [x]
    for blk in down:
        x, res = unet_block(W, f"down_blocks.{blk['idx']}", blk, x, encoder_x, temb, None, g)
        residuals.extend(res)
    H = cfg.num_attention_heads[-1]
    x = resnet_block_2d(W, "mid_blocks.0", x, temb, g)
    x = transformer_2d(W, "mid_blocks.1", H, cfg.transformer_layers_per_block[-1], x, encoder_x, g)
    x = resnet_block_2d(W, "mid_blocks.2", x, temb, g)
    for k, blk in enumerate(up):
        x, _ = unet_block(W, f"up_blocks.{k}", blk, x, encoder_x, temb, residuals, g)
    x = silu(group_norm(x, W["conv_norm_out.weight"], W["conv_norm_out.bias"], g, 1e-5))
    return conv2d(x, W["conv_out.weight"], W["conv_out.bias"], padding=(cfg.conv_out_kernel - 1) // 2)


def unet_weight_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    boc = cfg.block_out_channels
    tdim = boc[0] * 4

    def lin(n, o, i, bias=True):
        s[f"{n}.weight"] = (o, i)
        if bias:
            s[f"{n}.bias"] = (o,)

    def conv(n, o, i, k=3):
        s[f"{n}.weight"] = (o, k, k, i)
        s[f"{n}.bias"] = (o,)

    def norm(n, c):
        s[f"{n}.weight"] = (c,)
        s[f"{n}.bias"] = (c,)

    def resnet(n, i, o):
        norm(f"{n}.norm1", i); conv(f"{n}.conv1", o, i); lin(f"{n}.time_emb_proj", o, tdim)
        norm(f"{n}.norm2", o); conv(f"{n}.conv2", o, o)
        if i != o:
            lin(f"{n}.conv_shortcut", o, i)

    def t2d(n, c, enc, layers):
        norm(f"{n}.norm", c); lin(f"{n}.proj_in", c, c); lin(f"{n}.proj_out", c, c)
        for l in range(layers):
            b = f"{n}.transformer_blocks.{l}"
            for a, kd in (("attn1", c), ("attn2", enc)):
                lin(f"{b}.{a}.query_proj", c, c, False); lin(f"{b}.{a}.key_proj", c, kd, False)
                lin(f"{b}.{a}.value_proj", c, kd, False); lin(f"{b}.{a}.out_proj", c, c)
            for k in (1, 2, 3):
                norm(f"{b}.norm{k}", c)
            lin(f"{b}.linear1", 4 * c, c); lin(f"{b}.linear2", 4 * c, c); lin(f"{b}.linear3", c, 4 * c)

    conv("conv_in", boc[0], cfg.in_channels, cfg.conv_in_kernel)
    lin("time_embedding.linear_1", tdim, boc[0]); lin("time_embedding.linear_2", tdim, tdim)
    if cfg.addition_embed_type == "text_time":
        lin("add_embedding.linear_1", tdim, cfg.projection_class_embeddings_input_dim)
        lin("add_embedding.linear_2", tdim, tdim)
    down, up = _block_plan(cfg)
    for name, blocks, key in (("down_blocks", down, lambda k, b: b["idx"]), ("up_blocks", up, lambda k, b: k)):
        for k, b in enumerate(blocks):
            p = f"{name}.{key(k, b)}"
            for j, (i, o) in enumerate(b["resnets"]):
                resnet(f"{p}.resnets.{j}", i, o)
                if b["attn"]:
                    t2d(f"{p}.attentions.{j}", o, b["enc"], b["tlayers"])
            if b["down"]:
                conv(f"{p}.downsample", b["out"], b["out"])
            if b["up"]:
                conv(f"{p}.upsample", b["out"], b["out"])
    resnet("mid_blocks.0", boc[-1], boc[-1])
    t2d("mid_blocks.1", boc[-1], cfg.cross_attention_dim[-1], cfg.transformer_layers_per_block[-1])
    resnet("mid_blocks.2", boc[-1], boc[-1])
    norm("conv_norm_out", boc[0]); conv("conv_out", cfg.out_channels, boc[0], cfg.conv_out_kernel)
    return s


def denoising_step(cfg, W, sampler, x_t, t, t_prev, conditioning, cfg_weight=7.5, text_time=None, noise=None):
    """StableDiffusion._denoising_step (__init__.py:67-82): CFG batch doubling (text first, negative second)."""
    x_unet = torch.cat([x_t] * 2, dim=0) if cfg_weight > 1 else x_t
    t_unet = torch.full((len(x_unet),), float(t), dtype=torch.float32)
    eps = unet_forward(cfg, W, x_unet, t_unet, conditioning, text_time)
    if cfg_weight > 1:
        eps_text, eps_neg = eps.chunk(2)
        eps = eps_neg + cfg_weight * (eps_text - eps_neg)
    return sampler.step(eps, x_t, t, t_prev, noise)


# ------------------------------------------------------------------ SD VAE decoder (vae.py)
def vae_attention(W, p: str, x: Tensor, groups: int = 32) -> Tensor:
    """Attention.__call__ (vae.py:25-42): single head, explicit softmax((q*scale) k^T) v."""
    B, H, Wd, C = x.shape
    y = group_norm(x, W[f"{p}.group_norm.weight"], W[f"{p}.group_norm.bias"], groups, 1e-5)
    q = linear(y, W[f"{p}.query_proj.weight"], W[f"{p}.query_proj.bias"]).reshape(B, H * Wd, C)
    k = linear(y, W[f"{p}.key_proj.weight"], W[f"{p}.key_proj.bias"]).reshape(B, H * Wd, C)
    v = linear(y, W[f"{p}.value_proj.weight"], W[f"{p}.value_proj.bias"]).reshape(B, H * Wd, C)
    s = torch.matmul(q.float() * (1 / math.sqrt(C)), k.float().transpose(1, 2))
    y = torch.matmul(torch.softmax(s, dim=-1), v.float()).to(x.dtype).reshape(B, H, Wd, C)
    return x + linear(y, W[f"{p}.out_proj.weight"], W[f"{p}.out_proj.bias"])


def vae_decode(cfg: AutoencoderConfig, W, z: Tensor) -> Tensor:
    """Autoencoder.decode (vae.py:256-258) + Decoder.__call__ (vae.py:209-223)."""
    g = cfg.norm_num_groups
    z = z / cfg.scaling_factor
    x = linear(z, W["post_quant_proj.weight"], W["post_quant_proj.bias"])
    x = conv2d(x, W["decoder.conv_in.weight"], W["decoder.conv_in.bias"])
    x = resnet_block_2d(W, "decoder.mid_blocks.0", x, None, g)
    x = vae_attention(W, "decoder.mid_blocks.1", x, g)
    x = resnet_block_2d(W, "decoder.mid_blocks.2", x, None, g)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            x = resnet_block_2d(W, f"decoder.up_blocks.{i}.resnets.{j}", x, None, g)
        if i < n - 1:
            x = conv2d(upsample_nearest2(x), W[f"decoder.up_blocks.{i}.upsample.weight"],
                       W[f"decoder.up_blocks.{i}.upsample.bias"])
    x = silu(group_norm(x, W["decoder.conv_norm_out.weight"], W["decoder.conv_norm_out.bias"], g, 1e-5))
    return conv2d(x, W["decoder.conv_out.weight"], W["decoder.conv_out.bias"])


def sd_decode(cfg: AutoencoderConfig, W, x_t: Tensor) -> Tensor:
    """StableDiffusion.decode (__init__.py:166-169): clip(x/2 + 0.5, 0, 1)."""
    return torch.clip(vae_decode(cfg, W, x_t) / 2 + 0.5, 0, 1)


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

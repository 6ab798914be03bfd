"""UNetModel in the reference's float32 arithmetic — what `StableDiffusion(model)` / `StableDiffusionXL(model)` run with
their DEFAULT float16=False (stable_diffusion/stable_diffusion/__init__.py:19-25; unet.py:403-460 evaluated on float32
arrays).  Same launch structure as the 16-bit path of unet.py, on the float32-faithful kernels the VAE decoders use:

  * a float32 tensor is a split tensor [2, ...] (hi + lo bf16 planes); every Linear / conv is `fluxhip_gemm_x3` /
    `fluxhip_conv2d_x3` (three MFMA passes over the planes, float32 accumulation, float32 bias / residual);
  * GroupNorm + SiLU, LayerNorm, the GEGLU product, SiLU of the time embedding, the per-image time-embedding add and the
    sinusoidal embeddings are float32 kernels on split tensors (include/fluxhip.h, "ABI 9");
  * multi-head attention (`ops.attention_x3`): Q and K copied head-major, V projected transposed, then ONE batched
    float32-faithful GEMM for the logits Q_h K_h^T of every (image, head) with float32 output [B*H, N, Tkpad], a float32
    softmax with split probabilities, and one for P_h V_h^T — the per-(image, head) products of nn.MultiHeadAttention
    (unet.py:46-54), never a 16-bit logit or probability.

This is the arithmetic path, not a tuned one: 3x the MFMA work of the 16-bit kernels, unfused norms / activations, no captured-graph specialisation beyond what the pipeline does.  tests/test_sd_f32_gpu.py compares it with the
float32 oracle (oracle/sd_oracle.py): tiny UNet <= 1e-4, full-width blocks <= 1e-3."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops
from ..ops import ACT_GEGLU, ACT_SILU

BF16 = torch.bfloat16
F32 = torch.float32


def build_operands(params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Split (hi, lo) operands of every weight matrix / filter, plus the fused layouts: the
    GEGLU pair [linear1; linear2] as ONE weight (value rows, then gate rows) with its concatenated bias, conv_in padded to a
    64-channel K-step.  Biases and norm parameters stay float32 (the kernels read them as such)."""
    X: Dict[str, torch.Tensor] = {}
    for k, w in params.items():
        if k.endswith(".weight") and w.dim() >= 2:
            if w.dim() == 2 and w.shape[1] % 64:          # a Linear whose input width is not a whole 64-element K-step
                w = torch.nn.functional.pad(w, (0, 64 - w.shape[1] % 64))      # (add_embedding.linear_1 of small configs): zero columns
            X[k] = ops.split_f32(w.contiguous())
    for k in list(params):
        if k.endswith(".linear2.weight") and k[: -len("2.weight")] + "1.weight" in params:
            b = k[: -len(".linear2.weight")]
            X[f"{b}.geglu.w"] = ops.split_f32(torch.cat([params[f"{b}.linear1.weight"], params[k]], dim=0).contiguous())
            X[f"{b}.geglu.b"] = torch.cat([params[f"{b}.linear1.bias"], params[f"{b}.linear2.bias"]], dim=0).contiguous()
    w = params["conv_in.weight"]
    cin = w.shape[-1]
    if cin % 64:
        wp = torch.zeros(*w.shape[:-1], (cin + 63) // 64 * 64, dtype=F32, device=w.device)
        wp[..., :cin] = w
        X["conv_in.weight"] = ops.split_f32(wp)
    return X


def _cat_channels(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """mx.concatenate([x, res], axis=-1) on split tensors: a 16-bit copy per plane."""
    out = torch.empty(2, *a.shape[1:-1], a.shape[-1] + b.shape[-1], dtype=BF16, device=a.device)
    for pl in (0, 1):
        out[pl].copy_(ops.concat_channels(a[pl], b[pl]))
    return out


def _pad_channels(x: torch.Tensor, to: int) -> torch.Tensor:
    out = torch.empty(2, *x.shape[1:-1], to, dtype=BF16, device=x.device)
    for pl in (0, 1):
        out[pl].copy_(ops.concat_channels(x[pl], None, pad_to=to))
    return out


class UNetF32:
    """Forward of a UNetModel whose parameters are float32 (`model.dtype == torch.float32`)."""

    def __init__(self, model):
        self.m = model

    # ------------------------------------------------------------------ blocks
    def resnet(self, p: str, x: torch.Tensor, temb_act: torch.Tensor) -> torch.Tensor:
        """ResnetBlock2D.__call__ (unet.py:152-170)."""
        W, X, G = self.m._params, self.m._x3, self.m.config.norm_num_groups
        _, B, H, Wd, C = x.shape
        tproj = ops.linear_x3(temb_act, X[f"{p}.time_emb_proj.weight"], W[f"{p}.time_emb_proj.bias"])      # [2, B, Cout]
        h = ops.groupnorm_silu_x3(x, W[f"{p}.norm1.weight"], W[f"{p}.norm1.bias"], G, 1e-5, True)
        h = ops.conv2d_x3(h, X[f"{p}.conv1.weight"], W[f"{p}.conv1.bias"])
        ops.addvec_x3(h, tproj)                                                   # y + time_emb_proj(silu(temb))[:, None, None, :]
        h = ops.groupnorm_silu_x3(h, W[f"{p}.norm2.weight"], W[f"{p}.norm2.bias"], G, 1e-5, True)
        if f"{p}.conv_shortcut.weight" in W:
            x = ops.linear_x3(x.view(2, B, H * Wd, C), X[f"{p}.conv_shortcut.weight"], W[f"{p}.conv_shortcut.bias"]).view(2, B, H, Wd, -1)
        return ops.conv2d_x3(h, X[f"{p}.conv2.weight"], W[f"{p}.conv2.bias"], res=x)

    def mha(self, p: str, H: int, y: torch.Tensor, n: torch.Tensor, mem: Optional[torch.Tensor], Tk: int) -> torch.Tensor:
        """x + nn.MultiHeadAttention(n, kv, kv) (unet.py:46-54,64-71), head_dim 64.  y: residual [2,B,N,C]; n: the normed
        queries; mem: None (self-attention) or the zero-padded encoder states [2,B,Tkp,enc]; Tk: number of real keys."""
        W, X = self.m._params, self.m._x3
        _, B, N, C = n.shape
        dev = n.device
        src = n if mem is None else mem
        Tkp, Kd = src.shape[2], src.shape[3]
        Tkpad = (Tkp + 63) // 64 * 64
        q = ops.linear_x3(n, X[f"{p}.query_proj.weight"], None)                              # token-major [2,B,N,C]
        k = ops.linear_x3(src, X[f"{p}.key_proj.weight"], None)                              # [2,B,Tkp,C]
        # V^T[b] = Wv src[b]^T for every image in one launch: A = Wv shared, "W" operand = the rows of image b
        vt = torch.zeros(2, B, C, Tkpad, dtype=BF16, device=dev)                            # padded key columns stay zero
        ops.gemm_x3_batched(X[f"{p}.value_proj.weight"], src, vt, C, Tkp, Kd, Kd, Tkpad, B, 0, Tkp * Kd, C * Tkpad)
        o = ops.attention_x3(q, k, vt, H, Tk, 64 ** -0.5)
        return ops.linear_x3(o, X[f"{p}.out_proj.weight"], W[f"{p}.out_proj.bias"], res=y)

    def transformer(self, p: str, H: int, layers: int, x: torch.Tensor, mem: torch.Tensor, Tk: int) -> torch.Tensor:
        """Transformer2D.__call__ (unet.py:108-124) with TransformerBlock (unet.py:61-81)."""
        W, X, G = self.m._params, self.m._x3, self.m.config.norm_num_groups
        _, B, Hh, Ww, C = x.shape
        N = Hh * Ww
        y = ops.groupnorm_silu_x3(x, W[f"{p}.norm.weight"], W[f"{p}.norm.bias"], G, 1e-5, False).view(2, B, N, C)
        y = ops.linear_x3(y, X[f"{p}.proj_in.weight"], W[f"{p}.proj_in.bias"])
        for l in range(layers):
            b = f"{p}.transformer_blocks.{l}"
            n = ops.layernorm_x3(y, W[f"{b}.norm1.weight"], W[f"{b}.norm1.bias"])
            y = self.mha(f"{b}.attn1", H, y, n, None, N)
            n = ops.layernorm_x3(y, W[f"{b}.norm2.weight"], W[f"{b}.norm2.bias"])
            y = self.mha(f"{b}.attn2", H, y, n, mem, Tk)
            n = ops.layernorm_x3(y, W[f"{b}.norm3.weight"], W[f"{b}.norm3.bias"])
            ag = ops.linear_x3(n, X[f"{b}.geglu.w"], X[f"{b}.geglu.b"])                       # [.., 8C] = linear1(n) | linear2(n)
            g = ops.act_x3(ag, ACT_GEGLU)                                                     # linear1(n) * gelu(linear2(n))
            y = ops.linear_x3(g, X[f"{b}.linear3.weight"], W[f"{b}.linear3.bias"], res=y)
        out = ops.linear_x3(y, X[f"{p}.proj_out.weight"], W[f"{p}.proj_out.bias"], res=x.view(2, B, N, C))
        return out.view(2, B, Hh, Ww, C)

    def block(self, blk: dict, x, mem, Tk, temb_act, residuals):
        """UNetBlock2D.__call__ (unet.py:232-267)."""
        W, X = self.m._params, self.m._x3
        p = blk["name"]
        outs = []
        for j in range(len(blk["resnets"])):
            if residuals is not None:
                x = _cat_channels(x, residuals.pop())
            x = self.resnet(f"{p}.resnets.{j}", x, temb_act)
            if blk["attn"]:
                x = self.transformer(f"{p}.attentions.{j}", blk["heads"], blk["tlayers"], x, mem, Tk)
            outs.append(x)
        if blk["down"]:
            x = ops.conv2d_x3(x, X[f"{p}.downsample.weight"], W[f"{p}.downsample.bias"], stride=2, pad=1)
            outs.append(x)
        if blk["up"]:
            x = ops.conv2d_x3(x, X[f"{p}.upsample.weight"], W[f"{p}.upsample.bias"], ups=True)
            outs.append(x)
        return x, outs

    # ------------------------------------------------------------------ forward
    def __call__(self, x: torch.Tensor, timestep: torch.Tensor, encoder_x: torch.Tensor, text_time=None) -> torch.Tensor:
        """UNetModel.__call__ (unet.py:403-460): float32 in, float32 out."""
        m = self.m
        cfg, W, X = m.config, m._params, m._x3
        dev = m.device
        B = x.shape[0]
        x = ops.split_f32(x.to(device=dev, dtype=F32).contiguous())
        temb = ops.sincos_embed_x3(timestep.to(device=dev, dtype=F32).reshape(B), m._sig_t)
        h1 = ops.linear_x3(temb, X["time_embedding.linear_1.weight"], W["time_embedding.linear_1.bias"])
        temb = ops.linear_x3(ops.act_x3(h1, ACT_SILU), X["time_embedding.linear_2.weight"], W["time_embedding.linear_2.bias"])
        if text_time is not None:
            text_emb, time_ids = text_time
            e = ops.sincos_embed_x3(time_ids.to(device=dev, dtype=F32).reshape(-1), m._sig_add).view(2, B, -1)
            e = _cat_channels(ops.split_f32(text_emb.to(device=dev, dtype=F32).contiguous()), e)
            kp = X["add_embedding.linear_1.weight"].shape[-1]
            if e.shape[-1] != kp:
                e = _pad_channels(e, kp)                       # zero inputs for the zero weight columns of build_operands
            h1 = ops.linear_x3(e, X["add_embedding.linear_1.weight"], W["add_embedding.linear_1.bias"])
            temb = ops.linear_x3(ops.act_x3(h1, ACT_SILU), X["add_embedding.linear_2.weight"], W["add_embedding.linear_2.bias"],
                                 res=temb)
        temb_act = ops.act_x3(temb, ACT_SILU)                  # nn.silu(temb), shared by every ResnetBlock2D (unet.py:158)
        S = encoder_x.shape[1]
        Sp = (S + 7) // 8 * 8                                  # keys padded to the GEMM's N granularity (masked by the softmax)
        mem32 = torch.zeros(B, Sp, encoder_x.shape[2], dtype=F32, device=dev)
        mem32[:, :S].copy_(encoder_x)
        mem = ops.split_f32(mem32)

        cin = cfg.in_channels
        if cin % 64:
            x = _pad_channels(x, (cin + 63) // 64 * 64)
        x = ops.conv2d_x3(x, X["conv_in.weight"], W["conv_in.bias"])
        residuals = [x]
        for blk in m.down:
            x, res = self.block(blk, x, mem, S, temb_act, None)
            residuals.extend(res)
        H = cfg.num_attention_heads[-1]
        x = self.resnet("mid_blocks.0", x, temb_act)
        x = self.transformer("mid_blocks.1", H, cfg.transformer_layers_per_block[-1], x, mem, S)
        x = self.resnet("mid_blocks.2", x, temb_act)
        for blk in m.up:
            x, _ = self.block(blk, x, mem, S, temb_act, residuals)
        x = ops.groupnorm_silu_x3(x, W["conv_norm_out.weight"], W["conv_norm_out.bias"], cfg.norm_num_groups, 1e-5, True)
        return ops.join_f32(ops.conv2d_x3(x, X["conv_out.weight"], W["conv_out.bias"]))

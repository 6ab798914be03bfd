"""SD / SDXL UNet on MI355X — host-side mirror of the reference's
stable_diffusion/stable_diffusion/unet.py (UNetModel.__call__, :403-460).

Same constructor argument (UNetConfig) and call signature; parameter names are the MLX module tree's
flattened names, i.e. exactly what model_io.map_unet_weights produces.  The body runs on libfluxhip:
  * ResnetBlock2D: GroupNorm+SiLU kernels, implicit-GEMM 3x3 convs (MFMA); the time-embedding add,
    the 1x1 shortcut and the residual add are conv epilogues / a second GEMM;
  * Transformer2D: q/k projections fused into one GEMM, V projected directly TRANSPOSED
    (V^T = Wv y^T) so the head_dim-64 flash attention needs no transpose pass; out-projection +
    residual, GEGLU (linear1(y) * gelu(linear2(y))) and linear3 + residual are GEMM epilogues;
  * skip concats are one copy kernel; down/upsampling are strided / upsample-fused conv loaders.
Precision: the reference runs this model in float16 (float16=True, flux_app.py:77-79) or float32 (its default).  dtype
torch.float16: IEEE-half storage on the f16 matrix cores; torch.float32: float32 arithmetic on the float32-faithful
split-bf16 kernels (unet_f32.py); torch.bfloat16: bf16 storage, an explicit opt-in of the pipelines.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Tuple, Union

import torch

from .. import _lib, ops
from ..ops import EPI_GATE_RES, EPI_GEGLU, EPI_GEGLU_PAIR, FluxHipError, interleave_geglu, make_gemm_desc
from .config import UNetConfig

BF16 = torch.bfloat16


def block_plan(cfg: UNetConfig):
    """Static structure built by UNetModel.__init__ (unet.py:318-389), in execution order."""
    boc = list(cfg.block_out_channels)
    n = len(boc)
    down, up = [], []
    chans = [boc[0]] + boc
    for i, (ic, oc) in enumerate(zip(chans, chans[1:])):
        L = cfg.layers_per_block[i]
        down.append(dict(name=f"down_blocks.{i}", resnets=[(ic if j == 0 else oc, oc) for j in range(L)],
                         attn="CrossAttn" in cfg.down_block_types[i], heads=cfg.num_attention_heads[i],
                         tlayers=cfg.transformer_layers_per_block[i], enc=cfg.cross_attention_dim[i],
                         down=i < n - 1, up=False, out=oc))
    chans = [boc[0]] + boc + [boc[-1]]
    for k, (i, (ic, oc, pc)) in enumerate(reversed(list(enumerate(zip(chans, chans[1:], chans[2:]))))):
        L = cfg.layers_per_block[i] + 1
        in_list = [pc] + [oc] * (L - 1)
        res_list = [oc] * (L - 1) + [ic]
        up.append(dict(name=f"up_blocks.{k}", resnets=[(a + b, oc) for a, b in zip(in_list, res_list)],
                       attn="CrossAttn" in cfg.up_block_types[i], heads=cfg.num_attention_heads[i],
                       tlayers=cfg.transformer_layers_per_block[i], enc=cfg.cross_attention_dim[i],
                       down=False, up=i > 0, out=oc))
    return down, up


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
    down, up = block_plan(cfg)
    for b in down + up:
        for j, (i, o) in enumerate(b["resnets"]):
            resnet(f"{b['name']}.resnets.{j}", i, o)
            if b["attn"]:
                t2d(f"{b['name']}.attentions.{j}", o, b["enc"], b["tlayers"])
        if b["down"]:
            conv(f"{b['name']}.downsample", b["out"], b["out"])
        if b["up"]:
            conv(f"{b['name']}.upsample", b["out"], b["out"])
    resnet("mid_blocks.0", boc[-1], boc[-1])
    t2d("mid_blocks.1", boc[-1], cfg.cross_attention_dim[-1], cfg.transformer_layers_per_block[-1])
    resnet("mid_blocks.2", boc[-1], boc[-1])
    norm("conv_norm_out", boc[0]); conv("conv_out", cfg.out_channels, boc[0], cfg.conv_out_kernel)
    return s


def sinusoidal_sigmas(dims: int) -> torch.Tensor:
    """The frequency table of nn.SinusoidalPositionalEncoding as configured at unet.py:283-292."""
    half = dims // 2
    min_freq = math.exp(-math.log(10000) + 2 * math.log(10000) / dims)
    one_zero = 1 - torch.arange(0, half, dtype=torch.float32) / (half - 1)
    lmin = math.log(min_freq)
    return torch.exp(one_zero * (0.0 - lmin) + lmin)


def small_linear_any(x, w, b, silu_in=False, out=None, accum=False):
    """fluxhip_small_linear handles <= 16 rows per launch; larger batches are chunked."""
    B = x.shape[0]
    if out is None:
        out = torch.empty(B, w.shape[0], dtype=x.dtype, device=x.device)
    for lo in range(0, B, 16):
        ops.small_linear(x[lo:lo + 16], w, b, out=out[lo:lo + 16], silu_in=silu_in, accum=accum)
    return out


class _TembAct:
    """silu(time embedding) of one forward, shared by every ResnetBlock2D."""
    def __init__(self, temb: torch.Tensor):
        self.act = ops.silu(temb)


class UNetModel:
    def __init__(self, config: UNetConfig, device: Union[str, torch.device] = "cuda", dtype: torch.dtype = BF16):
        """dtype: the 16-bit storage type of weights and activations — torch.float16 is the reference's arithmetic under
        float16=True (stable_diffusion/__init__.py:20-27; v_mfma_f32_16x16x32_f16, fp32 accumulate / norms / softmax),
        torch.bfloat16 the range-safe default of this path."""
        if dtype not in (BF16, torch.float16, torch.float32):
            raise ValueError("UNetModel dtype must be torch.bfloat16, torch.float16 or torch.float32")
        self.dtype = dtype
        # torch.float32: the reference's arithmetic under its DEFAULT float16=False - float32 master parameters, the forward on
        # the float32-faithful split-bf16 kernels (unet_f32.py)
        self.x3 = dtype == torch.float32
        self._x3: Dict[str, torch.Tensor] = {}
        self.config = config
        if torch.device(device).type != "cuda":
            raise FluxHipError("UNetModel needs a HIP device: there is no CPU fallback for the denoise path")
        self.device = _lib.bind_device(device)
        for i, c in enumerate(config.block_out_channels):
            if c % 64 or c // config.num_attention_heads[i] != 64:
                raise ValueError("libfluxhip UNet path needs channels % 64 == 0 and attention head_dim 64")
        _lib.load()
        self._params = {k: torch.empty(*shp, dtype=self.dtype, device=self.device)
                        for k, shp in unet_weight_shapes(config).items()}
        self._fused: Dict[str, torch.Tensor] = {}
        self.down, self.up = block_plan(config)
        self._sig_t = sinusoidal_sigmas(config.block_out_channels[0]).to(self.device)
        self._sig_add = (sinusoidal_sigmas(config.addition_time_embed_dim).to(self.device)
                         if config.addition_embed_type == "text_time" else None)

    # ------------------------------------------------------------------ parameters
    def parameters(self):
        return self._params

    def init_random(self, seed: int = 0) -> "UNetModel":
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, t in self._params.items():
            base = name.rsplit(".", 1)[0]
            wt = self._params[f"{base}.weight"]
            if wt.dim() == 1:
                t.fill_(1.0 if name.endswith(".weight") else 0.0)
                continue
            k = 1.0 / math.sqrt(wt[0].numel())
            t.copy_(((torch.rand(t.shape, generator=g, device=self.device) * 2 - 1) * k).to(self.dtype))
        return self.finalize()

    def load_weights(self, weights, strict: bool = True) -> "UNetModel":
        items = weights.items() if isinstance(weights, dict) else weights
        seen = set()
        for k, w in items:
            if k not in self._params:
                if strict:
                    raise ValueError(f"Unexpected parameter {k}")
                continue
            dst = self._params[k]
            if tuple(dst.shape) != tuple(w.shape):
                raise ValueError(f"Shape mismatch for {k}: expected {tuple(dst.shape)}, got {tuple(w.shape)}")
            dst.copy_(w.to(device=self.device, dtype=self.dtype))
            seen.add(k)
        if strict and set(self._params) - seen:
            raise ValueError(f"Missing parameters: {sorted(set(self._params) - seen)[:5]} ...")
        return self.finalize()

    def finalize(self) -> "UNetModel":
        """Derived weight layouts: [q;k] projection fused, conv_in input channels zero-padded to 64 (one K-step of
        the implicit-GEMM loader, so the layer runs on the MFMA path; the extra products are exact zeros)."""
        P, F = self._params, {}
        if self.x3:
            from .unet_f32 import UNetF32, build_operands
            self._x3 = build_operands(P)
            self._f32 = UNetF32(self)
            self._fused = {}
            return self
        for k in list(P):
            if k.endswith(".attn1.query_proj.weight"):
                b = k[: -len(".query_proj.weight")]
                F[f"{b}.qk"] = torch.cat([P[k], P[f"{b}.key_proj.weight"]], dim=0).contiguous()
        # GEGLU (unet.py:74-78: linear1(y) * gelu(linear2(y))): both Linears as ONE launch with the pair epilogue - value and
        # gate rows interleaved in blocks of 16, the product formed in the epilogue (include/fluxhip.h, FLUXHIP_EPI_GEGLU_PAIR)
        fuse_geglu = os.environ.get("FLUXHIP_UNET_GEGLU", "pair") != "split"      # "split": the two-launch form (A/B timing)
        for k in list(P):
            if fuse_geglu and k.endswith(".linear2.weight") and k[: -len("2.weight")] + "1.weight" in P:
                b = k[: -len(".linear2.weight")]
                F[f"{b}.geglu.w"] = interleave_geglu(P[f"{b}.linear1.weight"], P[k])
                F[f"{b}.geglu.b"] = interleave_geglu(P[f"{b}.linear1.bias"], P[f"{b}.linear2.bias"])
        # cross-attention: K and V^T depend on the text only -> all layers of one width are projected together
        # (text_kv); group = (channels, encoder width), layer slot = row offset inside the concatenated weights
        groups: Dict[Tuple[int, int], list] = {}
        for k in P:
            if k.endswith(".attn2.key_proj.weight"):
                groups.setdefault(tuple(P[k].shape), []).append(k[: -len(".key_proj.weight")])
        self._kv_slot: Dict[str, Tuple[Tuple[int, int], int]] = {}
        for (C, enc), names in groups.items():
            F[f"kv.{C}.{enc}.k"] = torch.cat([P[f"{n}.key_proj.weight"] for n in names], dim=0).contiguous()
            F[f"kv.{C}.{enc}.v"] = torch.cat([P[f"{n}.value_proj.weight"] for n in names], dim=0).contiguous()
            for i, n in enumerate(names):
                self._kv_slot[n] = ((C, enc), i * C)
        w = P["conv_in.weight"]
        cin = w.shape[-1]
        if cin % 64:
            wp = torch.zeros(*w.shape[:-1], (cin + 63) // 64 * 64, dtype=self.dtype, device=self.device)
            wp[..., :cin] = w
            F["conv_in.weight"] = wp
        else:
            F["conv_in.weight"] = w
        self._fused = F
        return self

    # ------------------------------------------------------------------ blocks
    def _resnet(self, p: str, x: torch.Tensor, temb: torch.Tensor) -> torch.Tensor:
        """ResnetBlock2D.__call__ (unet.py:152-170)."""
        W, G = self._params, self.config.norm_num_groups
        # time_emb_proj(silu(temb)) (unet.py:158-159).  At batch > 4 the 22 projections go through the MFMA GEMM on silu(temb)
        # computed ONCE per forward (_TembAct): the GEMV kernel re-evaluates the silu per output-row pair and spends 16 FMAs per
        # weight element - 50 us per resnet at batch 16 against ~10 for the GEMM
        if isinstance(temb, _TembAct):
            tproj = ops.linear(temb.act, W[f"{p}.time_emb_proj.weight"], W[f"{p}.time_emb_proj.bias"])
        else:
            tproj = small_linear_any(temb, W[f"{p}.time_emb_proj.weight"], W[f"{p}.time_emb_proj.bias"], silu_in=True)
        h = ops.groupnorm_silu(x, W[f"{p}.norm1.weight"], W[f"{p}.norm1.bias"], G, 1e-5, True)
        h = ops.conv2d(h, W[f"{p}.conv1.weight"], W[f"{p}.conv1.bias"], addvec=tproj)
        h = ops.groupnorm_silu(h, W[f"{p}.norm2.weight"], W[f"{p}.norm2.bias"], G, 1e-5, True)
        if f"{p}.conv_shortcut.weight" in W:
            x = ops.conv2d(x, W[f"{p}.conv_shortcut.weight"], W[f"{p}.conv_shortcut.bias"])
        return ops.conv2d(h, W[f"{p}.conv2.weight"], W[f"{p}.conv2.bias"], res=x)

    def text_kv(self, mem: torch.Tensor, out: Optional[dict] = None) -> dict:
        """key_proj / value_proj of the (zero-padded) encoder states `mem` [B,Tkp,enc] for EVERY cross-attention layer
        (unet.py:46-54 applies them per layer; they depend on the text only): per (channels, enc) group one GEMM against
        the concatenated key weights -> K [B,Tkp,sum C] and one batched GEMM V^T[b] = Wv_all mem[b]^T -> [B,sum C,Tkpad]
        (padded keys stay zero).  A pipeline computes this once per job and passes it to every step (`text_kv=`);
        `out` = a dict from an earlier call to overwrite in place (static buffers of a captured step graph)."""
        if self.x3:        # the float32 forward projects K / V^T per layer from the encoder states it is given
            return out if out is not None else {}
        B, Tkp, enc = mem.shape
        Tkpad = (Tkp + 63) // 64 * 64
        kv = out if out is not None else {"Tkp": Tkp, "Tkpad": Tkpad}
        for key in {g for g, _ in self._kv_slot.values()}:
            C, e = key
            if e != enc:
                raise ValueError(f"encoder width {enc} does not match the cross-attention projections ({e})")
            Wk, Wv = self._fused[f"kv.{C}.{e}.k"], self._fused[f"kv.{C}.{e}.v"]
            LC = Wk.shape[0]
            if out is None:
                kv[key] = (torch.empty(B, Tkp, LC, dtype=self.dtype, device=mem.device),
                           torch.zeros(B, LC, Tkpad, dtype=self.dtype, device=mem.device))
            k_all, vt_all = kv[key]
            ops.linear(mem, Wk, out=k_all)
            ops.gemm(make_gemm_desc([dict(A=Wv.data_ptr(), W=mem.data_ptr(), C=vt_all.data_ptr(), a_bstride=0,
                                          w_bstride=Tkp * enc, c_bstride=LC * Tkpad, M=LC)], B, Tkp, enc, enc, Tkpad),
                     self.dtype == torch.float16)
        return kv

    def _mha(self, p: str, H: int, y: torch.Tensor, n: torch.Tensor, kv: Optional[dict], Tk: int) -> torch.Tensor:
        """x + nn.MultiHeadAttention(n, kv, kv) (unet.py:46-54,64-71). y: residual [B,N,C]; n: normed
        queries; kv: None for self-attention, else the projected encoder states of every layer (text_kv)."""
        W = self._params
        B, N, C = n.shape
        dev = n.device
        lib = _lib.load()
        f16 = self.dtype == torch.float16
        o = torch.empty(B, N, C, dtype=self.dtype, device=dev)
        if kv is None:
            Tkpad = (N + 63) // 64 * 64
            vt = torch.zeros(B, C, Tkpad, dtype=self.dtype, device=dev) if Tkpad != N else torch.empty(B, C, N, dtype=self.dtype, device=dev)
            # V^T[b] = Wv n[b]^T : A = Wv (shared), "W" operand = the rows of batch b
            # (round 6: this launch forked onto a side stream beside the [q;k] projection - both read the same LayerNorm output and
            #  nothing of each other - measured 34.9 ms per UNet step against 34.4 in line, profiles/r06_negative_experiments.json)
            ops.gemm(make_gemm_desc([dict(A=W[f"{p}.value_proj.weight"].data_ptr(), W=n.data_ptr(), C=vt.data_ptr(),
                                          a_bstride=0, w_bstride=N * C, c_bstride=C * Tkpad, M=C)],
                                    B, N, C, C, Tkpad), f16)
            qk = ops.linear(n, self._fused[f"{p}.qk"])                       # [B,N,2C]
            ops.attention_strided(qk, qk[..., C:], vt, o, B, H, 64, N, Tk, Tkpad, (N * 2 * C, 64, 2 * C), (N * 2 * C, 64, 2 * C), C,
                                  64 ** -0.5)
        else:
            q = ops.linear(n, W[f"{p}.query_proj.weight"])                    # [B,N,C]
            key, off = self._kv_slot[p]
            k_all, vt_all = kv[key]
            LC, Tkp, Tkpad = k_all.shape[2], kv["Tkp"], kv["Tkpad"]
            ops.attention_strided_vt(q, k_all, vt_all, o, B, H, 64, N, Tk, Tkpad, (N * C, 64, C), (Tkp * LC, 64, LC), LC * Tkpad, C,
                                     64 ** -0.5, k_off=off, vt_off=off * Tkpad)
        return ops.linear(o, W[f"{p}.out_proj.weight"], W[f"{p}.out_proj.bias"], epi=EPI_GATE_RES, res=y)

    def _transformer(self, p: str, H: int, layers: int, x: torch.Tensor, mem: dict, Tk: int) -> torch.Tensor:
        """Transformer2D.__call__ (unet.py:108-124) with TransformerBlock (unet.py:61-81)."""
        W, G = self._params, self.config.norm_num_groups
        B, Hh, Ww, C = x.shape
        y = ops.groupnorm_silu(x, W[f"{p}.norm.weight"], W[f"{p}.norm.bias"], G, 1e-5, False).view(B, Hh * Ww, C)
        y = ops.linear(y, W[f"{p}.proj_in.weight"], W[f"{p}.proj_in.bias"])
        for l in range(layers):
            b = f"{p}.transformer_blocks.{l}"
            n = ops.layernorm_affine(y, W[f"{b}.norm1.weight"], W[f"{b}.norm1.bias"])
            y = self._mha(f"{b}.attn1", H, y, n, None, Hh * Ww)
            n = ops.layernorm_affine(y, W[f"{b}.norm2.weight"], W[f"{b}.norm2.bias"])
            y = self._mha(f"{b}.attn2", H, y, n, mem, Tk)
            n = ops.layernorm_affine(y, W[f"{b}.norm3.weight"], W[f"{b}.norm3.bias"])
            if f"{b}.geglu.w" in self._fused:
                g = ops.linear(n, self._fused[f"{b}.geglu.w"], self._fused[f"{b}.geglu.b"], epi=EPI_GEGLU_PAIR)
            else:
                a = ops.linear(n, W[f"{b}.linear1.weight"], W[f"{b}.linear1.bias"])
                g = ops.linear(n, W[f"{b}.linear2.weight"], W[f"{b}.linear2.bias"], epi=EPI_GEGLU, res=a)
            y = ops.linear(g, W[f"{b}.linear3.weight"], W[f"{b}.linear3.bias"], epi=EPI_GATE_RES, res=y)
        out = ops.linear(y, W[f"{p}.proj_out.weight"], W[f"{p}.proj_out.bias"], epi=EPI_GATE_RES,
                         res=x.view(B, Hh * Ww, C))
        return out.view(B, Hh, Ww, C)

    def _block(self, blk: dict, x, mem, Tk, temb, residuals: Optional[list]):
        """UNetBlock2D.__call__ (unet.py:232-267)."""
        W = self._params
        p = blk["name"]
        outs = []
        for j in range(len(blk["resnets"])):
            if residuals is not None:
                x = ops.concat_channels(x, residuals.pop())
            x = self._resnet(f"{p}.resnets.{j}", x, temb)
            if blk["attn"]:
                x = self._transformer(f"{p}.attentions.{j}", blk["heads"], blk["tlayers"], x, mem, Tk)
            outs.append(x)
        if blk["down"]:
            x = ops.conv2d(x, W[f"{p}.downsample.weight"], W[f"{p}.downsample.bias"], stride=2, pad=1)
            outs.append(x)
        if blk["up"]:
            x = ops.conv2d(x, W[f"{p}.upsample.weight"], W[f"{p}.upsample.bias"], ups=True)
            outs.append(x)
        return x, outs

    # ------------------------------------------------------------------ forward
    def pad_encoder_states(self, encoder_x: torch.Tensor) -> torch.Tensor:
        """encoder states zero-padded to a multiple of 8 tokens (GEMM N granularity); padded keys are masked."""
        if self.x3:
            return encoder_x
        B, S, e = encoder_x.shape
        mem = torch.zeros(B, (S + 7) // 8 * 8, e, dtype=self.dtype, device=self.device)
        mem[:, :S].copy_(encoder_x)
        return mem

    def __call__(self, x: torch.Tensor, timestep: torch.Tensor, encoder_x: torch.Tensor, attn_mask=None,
                 encoder_attn_mask=None, text_time=None, text_kv: Optional[dict] = None) -> torch.Tensor:
        """UNetModel.__call__ (unet.py:403-460). x [B,h,w,4] NHWC, timestep [B], encoder_x [B,S,enc]."""
        if attn_mask is not None or encoder_attn_mask is not None:
            raise NotImplementedError("masks are always None on the reference's path (unet.py:403-411)")
        if self.x3:
            return self._f32(x, timestep, encoder_x, text_time)
        cfg, W = self.config, self._params
        x = x.to(self.dtype).contiguous()
        B = x.shape[0]
        temb = ops.sincos_embed(timestep.to(device=self.device, dtype=torch.float32).reshape(B), self._sig_t, self.dtype)
        h1 = small_linear_any(temb, W["time_embedding.linear_1.weight"], W["time_embedding.linear_1.bias"])
        temb = small_linear_any(h1, W["time_embedding.linear_2.weight"], W["time_embedding.linear_2.bias"], silu_in=True)
        if text_time is not None:
            text_emb, time_ids = text_time
            e = ops.sincos_embed(time_ids.to(device=self.device, dtype=torch.float32).reshape(-1), self._sig_add, self.dtype)
            e = ops.concat_channels(text_emb.to(self.dtype).contiguous(), e.view(B, -1))
            h1 = small_linear_any(e, W["add_embedding.linear_1.weight"], W["add_embedding.linear_1.bias"])
            small_linear_any(h1, W["add_embedding.linear_2.weight"], W["add_embedding.linear_2.bias"], silu_in=True,
                             out=temb, accum=True)
        if B > 4 and os.environ.get("FLUXHIP_UNET_TEMB") != "gemv":      # ("gemv": A/B timing)
            temb = _TembAct(temb)
        # the cross-attention K / V^T of every layer: given by the caller (once per job) or projected here
        S = encoder_x.shape[1]
        mem = text_kv if text_kv is not None else self.text_kv(self.pad_encoder_states(encoder_x))

        cin = cfg.in_channels
        if cin % 64:
            x = ops.concat_channels(x, None, pad_to=(cin + 63) // 64 * 64)
        x = ops.conv2d(x, self._fused["conv_in.weight"], W["conv_in.bias"])
        residuals = [x]
        for blk in self.down:
            x, res = self._block(blk, x, mem, S, temb, None)
            residuals.extend(res)
        H = cfg.num_attention_heads[-1]
        x = self._resnet("mid_blocks.0", x, temb)
        x = self._transformer("mid_blocks.1", H, cfg.transformer_layers_per_block[-1], x, mem, S)
        x = self._resnet("mid_blocks.2", x, temb)
        for blk in self.up:
            x, _ = self._block(blk, x, mem, S, temb, residuals)
        x = ops.groupnorm_silu(x, W["conv_norm_out.weight"], W["conv_norm_out.bias"], cfg.norm_num_groups, 1e-5, True)
        return ops.conv2d(x, W["conv_out.weight"], W["conv_out.bias"])

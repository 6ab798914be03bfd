"""CLIP text model on MI355X — host-side mirror of the reference's flux/clip.py (CLIPTextModelConfig,
CLIPOutput, CLIPTextModel.sanitize/__call__) and of stable_diffusion/stable_diffusion/clip.py (the same
transformer with an optional bias-free text_projection on the pooled row and exact-erf "gelu" towers).  LayerNorm kernel -> fused [q;k] GEMM (+bias) + V^T GEMM
(row bias) -> causal head_dim-64 flash attention -> out_proj + residual epilogue -> LayerNorm ->
linear1 with the quick_gelu epilogue -> linear2 + residual; pooled output = row at argmax(token id)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Union

import torch

from .. import _lib, ops
from ..ops import EPI_GATE_RES, EPI_GELU_ERF, EPI_QUICK_GELU, FluxHipError, make_gemm_desc

_ACT_EPI = {"quick_gelu": EPI_QUICK_GELU, "gelu": EPI_GELU_ERF}     # stable_diffusion/.../clip.py:11

BF16 = torch.bfloat16


@dataclass
class CLIPTextModelConfig:
    """flux/clip.py:12-30."""
    num_layers: int = 23
    model_dims: int = 1024
    num_heads: int = 16
    max_length: int = 77
    vocab_size: int = 49408
    hidden_act: str = "quick_gelu"
    projection_dim: Optional[int] = None     # stable_diffusion/.../config.py: text_projection of "...WithProjection" towers

    @classmethod
    def from_dict(cls, config):
        """HF config.json -> config.  The projection exists only for CLIPTextModelWithProjection checkpoints
        (stable_diffusion/.../model_io.py:246-257)."""
        with_projection = "WithProjection" in (config.get("architectures") or [""])[0]
        return cls(num_layers=config["num_hidden_layers"], model_dims=config["hidden_size"],
                   num_heads=config["num_attention_heads"], max_length=config["max_position_embeddings"],
                   vocab_size=config["vocab_size"], hidden_act=config.get("hidden_act", "quick_gelu"),
                   projection_dim=config["projection_dim"] if with_projection else None)


# openai/clip-vit-large-patch14 text tower as shipped in FLUX.1 text_encoder/config.json
CLIP_L = dict(num_layers=12, model_dims=768, num_heads=12, max_length=77, vocab_size=49408, hidden_act="quick_gelu")


@dataclass
class CLIPOutput:
    pooled_output: Optional[torch.Tensor] = None
    last_hidden_state: Optional[torch.Tensor] = None
    hidden_states: Optional[List[torch.Tensor]] = None


class CLIPTextModel:
    def __init__(self, config: CLIPTextModelConfig, device: Union[str, torch.device] = "cuda", dtype: torch.dtype = BF16):
        """dtype: 16-bit storage type — torch.bfloat16 (the Flux pipeline's dtype, flux/flux.py:24) or torch.float16 (the
        stable_diffusion/ towers under float16=True, stable_diffusion/.../model_io.py:171-174)."""
        if dtype not in (BF16, torch.float16, torch.float32):
            raise ValueError("CLIPTextModel dtype must be torch.bfloat16, torch.float16 or torch.float32")
        self.dtype = dtype
        # torch.float32: the stable_diffusion/ towers under the reference's default float16=False (model_io.py:171-174):
        # float32 master parameters, the forward on the float32-faithful split-bf16 kernels (`_call_f32`)
        self.x3 = dtype == torch.float32
        self._x3: Dict[str, torch.Tensor] = {}
        if config.model_dims // config.num_heads != 64:
            raise ValueError("libfluxhip CLIP attention is built for head_dim 64")
        if config.hidden_act not in _ACT_EPI:
            raise ValueError(f"hidden_act must be one of {sorted(_ACT_EPI)}")
        self.config = config
        if torch.device(device).type != "cuda":
            raise FluxHipError("CLIPTextModel needs a HIP device")
        self.device = _lib.bind_device(device)
        _lib.load()
        D = config.model_dims
        shp = {"token_embedding.weight": (config.vocab_size, D), "position_embedding.weight": (config.max_length, D),
               "final_layer_norm.weight": (D,), "final_layer_norm.bias": (D,)}
        for i in range(config.num_layers):
            p = f"layers.{i}"
            for n in ("layer_norm1", "layer_norm2"):
                shp[f"{p}.{n}.weight"] = (D,); shp[f"{p}.{n}.bias"] = (D,)
            for n in ("query_proj", "key_proj", "value_proj", "out_proj"):
                shp[f"{p}.attention.{n}.weight"] = (D, D); shp[f"{p}.attention.{n}.bias"] = (D,)
            shp[f"{p}.linear1.weight"] = (4 * D, D); shp[f"{p}.linear1.bias"] = (4 * D,)
            shp[f"{p}.linear2.weight"] = (D, 4 * D); shp[f"{p}.linear2.bias"] = (D,)
        if config.projection_dim is not None:
            shp["text_projection.weight"] = (config.projection_dim, D)
        self._params = {k: torch.empty(*v, dtype=self.dtype, device=self.device) for k, v in shp.items()}
        self._qk: Dict[int, tuple] = {}
        self.fp8 = False
        self._w8: Dict[int, tuple] = {}

    def enable_fp8(self, enabled: bool = True) -> "CLIPTextModel":
        """`--quantize` on the CLIP tower (txt2image.py:79-82): the reference's predicate (in_dim % 512 == 0) selects the
        layers' second MLP Linear only (in_dim = 4 x width; the width-768 / 1280 inputs of every other Linear do not pass it),
        and so does this: e4m3 weights per output channel, per-token e4m3 inputs, fp8 MFMA (bf16 towers only)."""
        if enabled and self.dtype != BF16:
            raise ValueError("fp8 Linears are built for the bf16 towers")
        if enabled and not self._w8 and (4 * self.config.model_dims) % 512 == 0:
            for i in range(self.config.num_layers):
                self._w8[i] = ops.quantize_rows_fp8(self._params[f"layers.{i}.linear2.weight"])
        self.fp8 = bool(enabled)
        return self

    def parameters(self):
        return self._params

    def sanitize(self, weights):
        """flux/clip.py:96-125."""
        out = {}
        for key, w in weights.items():
            for pre in ("text_model.", "embeddings.", "encoder."):
                if key.startswith(pre):
                    key = key[len(pre):]
            for a, b in (("self_attn.", "attention."), ("q_proj.", "query_proj."), ("k_proj.", "key_proj."),
                         ("v_proj.", "value_proj."), ("mlp.fc1", "linear1"), ("mlp.fc2", "linear2")):
                if a in key:
                    key = key.replace(a, b)
            out[key] = w
        return out

    def init_random(self, seed: int = 0) -> "CLIPTextModel":
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, t in self._params.items():
            if "layer_norm" in name:
                t.fill_(1.0 if name.endswith(".weight") else 0.0)
            elif "embedding" in name:
                t.copy_((torch.randn(t.shape, generator=g, device=self.device) * 0.5).to(self.dtype))
            else:
                base = name.rsplit(".", 1)[0]
                k = 1.0 / math.sqrt(self._params[f"{base}.weight"].shape[1])
                t.copy_(((torch.rand(t.shape, generator=g, device=self.device) * 2 - 1) * k).to(self.dtype))
        return self.finalize()

    def load_weights(self, weights, strict: bool = True) -> "CLIPTextModel":
        items = weights.items() if isinstance(weights, dict) else weights
        seen = set()
        for k, w in items:
            if k not in self._params:
                if strict and "position_ids" not in k:
                    raise ValueError(f"Unexpected parameter {k}")
                continue
            if tuple(self._params[k].shape) != tuple(w.shape):
                raise ValueError(f"Shape mismatch for {k}")
            self._params[k].copy_(w.to(device=self.device, dtype=self.dtype))
            seen.add(k)
        if strict and set(self._params) - seen:
            raise ValueError(f"Missing parameters: {sorted(set(self._params) - seen)[:5]} ...")
        return self.finalize()

    def finalize(self) -> "CLIPTextModel":
        P = self._params
        self._qk = {}
        if self.x3:
            self._x3 = {k: ops.split_f32(w.contiguous()) for k, w in P.items()
                        if k.endswith(".weight") and w.dim() == 2 and "embedding" not in k}
            T, D = P["position_embedding.weight"].shape
            pos = torch.zeros((T + 7) // 8 * 8, D, dtype=torch.float32, device=self.device)     # rows padded with the sequence (see _call_f32)
            pos[:T] = P["position_embedding.weight"]
            self._x3["pos"] = pos
            return self
        was8, self._w8 = getattr(self, "fp8", False), {}
        for i in range(self.config.num_layers):
            a = f"layers.{i}.attention"
            self._qk[i] = (torch.cat([P[f"{a}.query_proj.weight"], P[f"{a}.key_proj.weight"]], 0).contiguous(),
                           torch.cat([P[f"{a}.query_proj.bias"], P[f"{a}.key_proj.bias"]], 0).contiguous())
        if was8:
            self.enable_fp8(True)
        return self

    def _call_f32(self, x: torch.Tensor) -> CLIPOutput:
        """`__call__` in float32 arithmetic on split tensors (hi + lo bf16 planes; include/fluxhip.h "ABI 9"): every Linear is
        fluxhip_gemm_x3, LayerNorm / activation / softmax are float32 kernels, the per-head attention products are H-batched
        float32-faithful GEMMs with a float32 causal softmax in between (clip.py:127-155).  The sequence is zero-padded to a
        multiple of 8 tokens (GEMM granularity): under the causal mask no real row sees a padding row, and the padding rows are
        dropped from every output.  Outputs are float32."""
        c, P, X = self.config, self._params, self._x3
        tokens = x.to(dtype=torch.int32)
        B, N = tokens.shape
        eos = tokens.argmax(-1)
        D, H = c.model_dims, c.num_heads
        Np = (N + 7) // 8 * 8
        Tpad = (Np + 63) // 64 * 64
        tok = torch.zeros(B, Np, dtype=torch.int32)
        tok[:, :N] = tokens
        tok = tok.to(self.device).contiguous()
        dev = self.device
        h = ops.embedding_x3(tok, P["token_embedding.weight"], X["pos"][:Np].contiguous())            # [2,B,Np,D]
        vt = torch.zeros(2, B, D, Tpad, dtype=BF16, device=dev)
        act = ops.ACT_QUICK_GELU if c.hidden_act == "quick_gelu" else ops.ACT_GELU_ERF
        hs = []
        for i in range(c.num_layers):
            p = f"layers.{i}"
            a = f"{p}.attention"
            y = ops.layernorm_x3(h, P[f"{p}.layer_norm1.weight"], P[f"{p}.layer_norm1.bias"])
            q = ops.linear_x3(y, X[f"{a}.query_proj.weight"], P[f"{a}.query_proj.bias"])
            k = ops.linear_x3(y, X[f"{a}.key_proj.weight"], P[f"{a}.key_proj.bias"])
            ops.gemm_x3_batched(X[f"{a}.value_proj.weight"], y, vt, D, Np, D, D, Tpad, B, 0, Np * D, D * Tpad,
                                bias=P[f"{a}.value_proj.bias"], row_bias=True)
            o = ops.attention_x3(q, k, vt, H, Np, 64 ** -0.5, causal=True)
            h = ops.linear_x3(o, X[f"{a}.out_proj.weight"], P[f"{a}.out_proj.bias"], res=h)
            y = ops.layernorm_x3(h, P[f"{p}.layer_norm2.weight"], P[f"{p}.layer_norm2.bias"])
            y = ops.act_x3(ops.linear_x3(y, X[f"{p}.linear1.weight"], P[f"{p}.linear1.bias"]), act)
            h = ops.linear_x3(y, X[f"{p}.linear2.weight"], P[f"{p}.linear2.bias"], res=h)
            hs.append(ops.join_f32(h)[:, :N].contiguous())
        last = ops.layernorm_x3(h, P["final_layer_norm.weight"], P["final_layer_norm.bias"])
        rows = (torch.arange(B, dtype=torch.int32) * Np + eos.cpu().to(torch.int32)).to(dev).contiguous()
        pooled = torch.stack([ops.embedding(rows, last[pl].view(B * Np, D)) for pl in (0, 1)])      # [2,B,D]: a gather per plane
        if c.projection_dim is not None:
            pooled = ops.linear_x3(pooled, X["text_projection.weight"], None)
        return CLIPOutput(pooled_output=ops.join_f32(pooled), last_hidden_state=ops.join_f32(last)[:, :N].contiguous(),
                          hidden_states=hs)

    def __call__(self, x: torch.Tensor) -> CLIPOutput:
        """CLIPTextModel.__call__ (flux/clip.py:127-155): tokens [B,N]."""
        if self.x3:
            return self._call_f32(x)
        c, P = self.config, self._params
        tokens = x.to(dtype=torch.int32)
        B, N = tokens.shape
        eos = tokens.argmax(-1)                                   # EOS has the highest id (flux/clip.py:130)
        tok = tokens.to(self.device).contiguous()
        D, H = c.model_dims, c.num_heads
        Np = (N + 7) // 8 * 8                                    # rows padded for the V^T GEMM's N granularity
        h = ops.embedding(tok, P["token_embedding.weight"], P["position_embedding.weight"])      # [B,N,D]
        Tpad = (N + 63) // 64 * 64
        vt = torch.zeros(B, D, Tpad, dtype=self.dtype, device=self.device)
        ypad = torch.zeros(B, Np, D, dtype=self.dtype, device=self.device)
        o = torch.empty(B, N, D, dtype=self.dtype, device=self.device)
        hs = []
        for i in range(c.num_layers):
            p = f"layers.{i}"
            y = ops.layernorm_affine(h, P[f"{p}.layer_norm1.weight"], P[f"{p}.layer_norm1.bias"])
            qkw, qkb = self._qk[i]
            qk = ops.linear(y, qkw, qkb)                                                     # [B,N,2D]
            ypad[:, :N].copy_(y)
            ops.gemm(make_gemm_desc([dict(A=P[f"{p}.attention.value_proj.weight"].data_ptr(), W=ypad.data_ptr(),
                                          bias=P[f"{p}.attention.value_proj.bias"].data_ptr(), C=vt.data_ptr(), a_bstride=0,
                                          w_bstride=Np * D, c_bstride=D * Tpad, M=D)], B, Np, D, D, Tpad, row_bias=True),
                     self.dtype == torch.float16)
            st = (N * 2 * D, 64, 2 * D)
            ops.attention_masked(qk, qk[..., D:], vt, o, B, H, N, N, Tpad, st, st, D, 64 ** -0.5, causal=True)
            h = ops.linear(o, P[f"{p}.attention.out_proj.weight"], P[f"{p}.attention.out_proj.bias"], epi=EPI_GATE_RES, res=h)
            y = ops.layernorm_affine(h, P[f"{p}.layer_norm2.weight"], P[f"{p}.layer_norm2.bias"])
            y = ops.linear(y, P[f"{p}.linear1.weight"], P[f"{p}.linear1.bias"], epi=_ACT_EPI[c.hidden_act])
            if self.fp8 and i in self._w8:
                yq, ys = ops.quantize_rows_fp8(y.view(B * N, 4 * D))
                h = ops.linear_fp8(yq, ys, *self._w8[i], P[f"{p}.linear2.bias"], epi=EPI_GATE_RES, res=h.view(B * N, D)).view(B, N, D)
            else:
                h = ops.linear(y, P[f"{p}.linear2.weight"], P[f"{p}.linear2.bias"], epi=EPI_GATE_RES, res=h)
            hs.append(h)
        last = ops.layernorm_affine(h, P["final_layer_norm.weight"], P["final_layer_norm.bias"])
        rows = (torch.arange(B, dtype=torch.int32) * N + eos.cpu().to(torch.int32)).to(self.device).contiguous()
        pooled = ops.embedding(rows, last.view(B * N, D))
        if c.projection_dim is not None:        # text_projection (no bias), stable_diffusion/.../clip.py:107-108
            pooled = torch.cat([ops.small_linear(pooled[i:i + 16].contiguous(), P["text_projection.weight"], None)
                                for i in range(0, B, 16)], dim=0)
        return CLIPOutput(pooled_output=pooled, last_hidden_state=last, hidden_states=hs)

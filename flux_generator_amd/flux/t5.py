"""T5 encoder (T5-XXL for Flux) on MI355X — host-side mirror of the reference's flux/t5.py
(T5Config, T5Encoder.sanitize/__call__; encoder only, the decoder half is unused by the pipeline).

Per layer: RMSNorm kernel -> fused [q;k] projection GEMM + V^T GEMM (value projection written
transposed) -> head_dim-64 flash attention with the relative-position bias added to the logits
(scale 1.0, pads attended, flux/t5.py:153-155) -> out_proj GEMM with the residual in its epilogue ->
RMSNorm -> gated-GELU FFN as two GEMMs (wi_1, then wi_0 with the `res * gelu_erf(.)` epilogue) ->
wo GEMM + residual.  The bias table [H,T,T] depends only on T and is built once per length."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Union

import torch

from .. import _lib, ops
from ..ops import EPI_GATE_RES, EPI_GEGLU, FluxHipError, make_gemm_desc

BF16 = torch.bfloat16

_SHARED = [(".block.", ".layers."), (".k.", ".key_proj."), (".o.", ".out_proj."), (".q.", ".query_proj."),
           (".v.", ".value_proj."), ("shared.", "wte."), ("lm_head.", "lm_head.linear."),
           (".layer.0.layer_norm.", ".ln1."), (".layer.1.layer_norm.", ".ln2."), (".layer.2.layer_norm.", ".ln3."),
           (".final_layer_norm.", ".ln."),
           ("layers.0.layer.0.SelfAttention.relative_attention_bias.", "relative_attention_bias.embeddings.")]
_ENCODER = [(".layer.0.SelfAttention.", ".attention."), (".layer.1.DenseReluDense.", ".dense.")]


@dataclass
class T5Config:
    """flux/t5.py:33-67."""
    vocab_size: int
    num_layers: int
    num_heads: int
    relative_attention_num_buckets: int
    d_kv: int
    d_model: int
    feed_forward_proj: str
    tie_word_embeddings: bool
    d_ff: Optional[int] = None
    num_decoder_layers: Optional[int] = None
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6

    @classmethod
    def from_dict(cls, config):
        return cls(vocab_size=config["vocab_size"], num_layers=config["num_layers"], num_heads=config["num_heads"],
                   relative_attention_num_buckets=config["relative_attention_num_buckets"], d_kv=config["d_kv"],
                   d_model=config["d_model"], feed_forward_proj=config["feed_forward_proj"],
                   tie_word_embeddings=config["tie_word_embeddings"], d_ff=config.get("d_ff", 4 * config["d_model"]),
                   num_decoder_layers=config.get("num_decoder_layers", config["num_layers"]),
                   relative_attention_max_distance=config.get("relative_attention_max_distance", 128),
                   layer_norm_epsilon=config.get("layer_norm_epsilon", 1e-6))


# google/t5-v1_1-xxl encoder as shipped in FLUX.1 text_encoder_2/config.json
T5_XXL = dict(vocab_size=32128, num_layers=24, num_heads=64, relative_attention_num_buckets=32, d_kv=64, d_model=4096,
              feed_forward_proj="gated-gelu", tie_word_embeddings=False, d_ff=10240)


def relative_position_bucket(rpos: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """Bidirectional bucketing of flux/t5.py:78-97 (integer / log arithmetic on the host)."""
    nb = num_buckets // 2
    max_exact = nb // 2
    a = rpos.abs()
    scale = (nb - max_exact) / math.log(max_distance / max_exact)
    big = (torch.log(a.float() / max_exact) * scale).to(torch.int16).to(torch.int64)
    big = torch.minimum(max_exact + big, torch.tensor(nb - 1))
    return torch.where(a < max_exact, a, big) + (rpos > 0).long() * nb


class T5Encoder:
    def __init__(self, config: T5Config, device: Union[str, torch.device] = "cuda"):
        if config.d_kv != 64:
            raise ValueError("libfluxhip T5 attention is built for d_kv = 64")
        if not config.feed_forward_proj.startswith("gated") or config.feed_forward_proj.removeprefix("gated-") != "gelu":
            raise ValueError("only the gated-gelu feed-forward of T5 v1.1 is built")
        self.config = config
        if torch.device(device).type != "cuda":
            raise FluxHipError("T5Encoder needs a HIP device")
        self.device = _lib.bind_device(device)
        _lib.load()
        c, inner = config, config.d_kv * config.num_heads
        shp = {"wte.weight": (c.vocab_size, c.d_model),
               "encoder.relative_attention_bias.embeddings.weight": (c.relative_attention_num_buckets, c.num_heads),
               "encoder.ln.weight": (c.d_model,)}
        for i in range(c.num_layers):
            p = f"encoder.layers.{i}"
            shp[f"{p}.ln1.weight"] = (c.d_model,)
            shp[f"{p}.ln2.weight"] = (c.d_model,)
            for n in ("query_proj", "key_proj", "value_proj"):
                shp[f"{p}.attention.{n}.weight"] = (inner, c.d_model)
            shp[f"{p}.attention.out_proj.weight"] = (c.d_model, inner)
            shp[f"{p}.dense.wi_0.weight"] = (c.d_ff, c.d_model)
            shp[f"{p}.dense.wi_1.weight"] = (c.d_ff, c.d_model)
            shp[f"{p}.dense.wo.weight"] = (c.d_model, c.d_ff)
        self._params = {k: torch.empty(*v, dtype=BF16, device=self.device) for k, v in shp.items()}
        self._qk: Dict[int, torch.Tensor] = {}
        self._bias: Dict[int, torch.Tensor] = {}
        self.fp8 = False
        self._w8: Dict[str, tuple] = {}
        self.use_graph = True
        self._graphs: Dict[tuple, tuple] = {}

    def enable_fp8(self, enabled: bool = True) -> "T5Encoder":
        """The reference's `--quantize` covers the text towers too (txt2image.py:79-82: nn.quantize of every Linear whose
        in_dim % 512 == 0 - all of T5's).  Same redesign as the flow model's (include/fluxhip.h, fluxhip_gemm_fp8): OCP e4m3fn
        weights with one float32 scale per output channel, inputs quantised per token on the fly, products on the block-scaled
        fp8 MFMA; the [q;k] projection, out_proj and the three FFN Linears (98 % of the encoder's FLOPs).  The value
        projection stays bf16: it is written transposed with the activations as the per-batch "weight" operand, which the fp8
        entry point's per-batch scales do not describe.  RMSNorm, attention, residual stream: unchanged."""
        if enabled and not self._w8:
            if self.config.d_model % 128 or self.config.d_ff % 128:
                raise ValueError("fp8 needs d_model % 128 == 0 and d_ff % 128 == 0")
            P = self._params
            for i in range(self.config.num_layers):
                p = f"encoder.layers.{i}"
                self._w8[f"{p}.qk"] = ops.quantize_rows_fp8(self._qk[i])
                for n in ("attention.out_proj", "dense.wi_0", "dense.wi_1", "dense.wo"):
                    self._w8[f"{p}.{n}"] = ops.quantize_rows_fp8(P[f"{p}.{n}.weight"])
        self.fp8 = bool(enabled)
        return self

    def _lin(self, x: torch.Tensor, name: str, w: torch.Tensor, epi: int = ops.EPI_BIAS, res: Optional[torch.Tensor] = None):
        """x @ w^T (+ epilogue): bf16, or per-token e4m3 x per-channel e4m3 when enable_fp8() is on."""
        if not self.fp8:
            return ops.linear(x, w, None, epi=epi, res=res)
        K = x.shape[-1]
        xq, xs = ops.quantize_rows_fp8(x.reshape(-1, K))
        wq, ws = self._w8[name]
        r2 = None if res is None else res.reshape(-1, res.shape[-1])
        return ops.linear_fp8(xq, xs, wq, ws, None, epi=epi, res=r2).view(*x.shape[:-1], wq.shape[0])

    def parameters(self):
        return self._params

    def sanitize(self, weights):
        """flux/t5.py:232-241."""
        out = {}
        for k, w in weights.items():
            for old, new in _SHARED:
                k = k.replace(old, new)
            if k.startswith("encoder."):
                for old, new in _ENCODER:
                    k = k.replace(old, new)
            out[k] = w
        return out

    def init_random(self, seed: int = 0) -> "T5Encoder":
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, t in self._params.items():
            if t.dim() == 1:
                t.fill_(1.0)
            elif name.endswith("embeddings.weight") or name == "wte.weight":
                t.copy_((torch.randn(t.shape, generator=g, device=self.device) * (1.0 if name == "wte.weight" else 0.5)).to(BF16))
            else:
                k = 1.0 / math.sqrt(t.shape[1])
                flat = t.view(-1)
                for s in range(0, flat.numel(), 1 << 26):
                    n = min(1 << 26, flat.numel() - s)
                    flat[s:s + n] = ((torch.rand(n, generator=g, device=self.device) * 2 - 1) * k).to(BF16)
        return self.finalize()

    def load_weights(self, weights, strict: bool = True) -> "T5Encoder":
        items = weights.items() if isinstance(weights, dict) else weights
        seen = set()
        for k, w in items:
            if k not in self._params:
                if strict and not k.startswith(("decoder.", "lm_head.")):
                    raise ValueError(f"Unexpected parameter {k}")
                continue
            if tuple(self._params[k].shape) != tuple(w.shape):
                raise ValueError(f"Shape mismatch for {k}")
            self._params[k].copy_(w.to(device=self.device, dtype=BF16))
            seen.add(k)
        if strict and set(self._params) - seen:
            raise ValueError(f"Missing parameters: {sorted(set(self._params) - seen)[:5]} ...")
        return self.finalize()

    def finalize(self) -> "T5Encoder":
        P = self._params
        was8, self._w8 = self.fp8, {}
        self._qk = {i: torch.cat([P[f"encoder.layers.{i}.attention.query_proj.weight"],
                                  P[f"encoder.layers.{i}.attention.key_proj.weight"]], dim=0).contiguous()
                    for i in range(self.config.num_layers)}
        self._bias = {}
        self._graphs = {}                   # captured graphs hold the old fused [q;k] / fp8 tensors' addresses
        if was8:
            self.enable_fp8(True)           # new weights: quantise again
        return self

    def _position_bias(self, T: int) -> torch.Tensor:
        """RelativePositionBias.__call__ (flux/t5.py:99-116) -> bf16 [H, T, T], cached per length."""
        b = self._bias.get(T)
        if b is None:
            c = self.config
            rp = torch.arange(T)[None, :] - torch.arange(T)[:, None]
            bucket = relative_position_bucket(rp, c.relative_attention_num_buckets, c.relative_attention_max_distance)
            idx = bucket.to(torch.int32).reshape(-1).contiguous().to(self.device)
            emb = self._params["encoder.relative_attention_bias.embeddings.weight"]           # [buckets, H]
            # gather rows of the [buckets, H] table -> [T*T, H] -> [H, T, T]; H % 8 may not hold, so this tiny
            # index/permute (integer plumbing, once per sequence length) uses torch
            b = emb[idx.long()].view(T, T, c.num_heads).permute(2, 0, 1).contiguous()
            self._bias[T] = b
        return b

    MAX_GRAPHS = 4          # captured encoder graphs kept (LRU): one per (batch, length, fp8) seen

    def __call__(self, inputs: torch.Tensor) -> torch.Tensor:
        """T5Encoder.__call__ (flux/t5.py:243-244): tokens [B,T] -> [B,T,d_model] bf16.  The ~290 launches of the 24 layers are
        captured once per (batch, length) into a hipGraph over a static token buffer and replayed (use_graph, default on):
        eagerly the encoder is host-bound - 6.9 ms per prompt at S = 256 against the time its kernels take."""
        tokens = inputs.to(device=self.device, dtype=torch.int32).contiguous()
        if not getattr(self, "use_graph", True) or torch.cuda.is_current_stream_capturing():
            return self._forward(tokens)
        B, T = tokens.shape
        if T % 4:
            raise FluxHipError("sequence length must be a multiple of 4")
        key = (B, T, self.fp8)
        ent = self._graphs.get(key)
        if ent is None:
            self._position_bias(T)                      # host-side table build: outside the capture
            static = tokens.clone()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._forward(static)                   # warm-up: kernel attributes, allocator
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._forward(static)
            ent = (g, static, out)
            self._graphs[key] = ent
            while len(self._graphs) > self.MAX_GRAPHS:
                self._graphs.pop(next(iter(self._graphs)))
        else:
            self._graphs[key] = self._graphs.pop(key)   # LRU: most recent last
        g, static, out = ent
        static.copy_(tokens)
        g.replay()
        return out.clone()

    def _forward(self, tokens: torch.Tensor) -> torch.Tensor:
        c, P = self.config, self._params
        B, T = tokens.shape
        H, D, inner = c.num_heads, c.d_model, c.d_kv * c.num_heads
        if T % 4:
            raise FluxHipError("sequence length must be a multiple of 4")
        x = ops.embedding(tokens, P["wte.weight"])
        bias = self._position_bias(T)
        Tpad = (T + 63) // 64 * 64
        vt = torch.zeros(B, inner, Tpad, dtype=BF16, device=self.device)
        o = torch.empty(B, T, inner, dtype=BF16, device=self.device)
        for i in range(c.num_layers):
            p = f"encoder.layers.{i}"
            y = ops.rmsnorm(x, P[f"{p}.ln1.weight"], c.layer_norm_epsilon)
            qk = self._lin(y, f"{p}.qk", self._qk[i])                                    # [B,T,2*inner]
            ops.gemm(make_gemm_desc([dict(A=P[f"{p}.attention.value_proj.weight"].data_ptr(), W=y.data_ptr(),
                                          C=vt.data_ptr(), a_bstride=0, w_bstride=T * D, c_bstride=inner * Tpad, M=inner)],
                                    B, T, D, D, Tpad))
            st = (T * 2 * inner, 64, 2 * inner)
            ops.attention_masked(qk, qk[..., inner:], vt, o, B, H, T, T, Tpad, st, st, inner, 1.0, bias=bias)
            x = self._lin(o, f"{p}.attention.out_proj", P[f"{p}.attention.out_proj.weight"], EPI_GATE_RES, x)
            y = ops.rmsnorm(x, P[f"{p}.ln2.weight"], c.layer_norm_epsilon)
            lin = self._lin(y, f"{p}.dense.wi_1", P[f"{p}.dense.wi_1.weight"])
            h = self._lin(y, f"{p}.dense.wi_0", P[f"{p}.dense.wi_0.weight"], EPI_GEGLU, lin)      # gelu(wi_0 y) * wi_1 y
            x = self._lin(h, f"{p}.dense.wo", P[f"{p}.dense.wo.weight"], EPI_GATE_RES, x)
        return ops.rmsnorm(x, P["encoder.ln.weight"], c.layer_norm_epsilon)

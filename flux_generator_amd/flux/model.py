"""Flux MMDiT on MI355X: host-side mirror of the reference's ``flux/model.py``.

Same constructor argument (FluxParams), same ``__call__(img, img_ids, txt, txt_ids, timesteps, y,
guidance)`` contract and ValueErrors as the reference (flux/model.py:35-136); the body is a fixed
*launch plan* of libfluxhip kernels over a preallocated HBM workspace:

  * activations live in ONE packed token buffer x[B, T = S + L, D] with the txt rows first
    (the reference concatenates txt/img for attention in every double block and again before the
    single blocks, flux/layers.py:212-214, flux/model.py:129 — here nothing is ever concatenated);
  * the txt and img streams of a DoubleStreamBlock run as 2-group GEMM launches;
  * all 2*19 + 38 + 1 modulation Linears depend only on ``vec``: their weights are stored
    row-concatenated and evaluated by one HBM-bound launch per step;
  * q/k RMSNorm + RoPE + the V transpose are one launch, attention one launch; gated residuals,
    GELU and the single-block [attn | gelu(mlp)] concat are GEMM epilogues.
"""
from __future__ import annotations

import ctypes
import os
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Tuple, Union

import torch

from .. import _lib
from .. import ops
from ..ops import EPI_BIAS, EPI_GATE_RES, EPI_GELU_TANH, EPI_SPLIT_GELU, FluxHipError, make_gemm_desc

BF16 = torch.bfloat16


@dataclass
class FluxParams:
    """flux/model.py:20-32."""
    in_channels: int
    vec_in_dim: int
    context_in_dim: int
    hidden_size: int
    mlp_ratio: float
    num_heads: int
    depth: int
    depth_single_blocks: int
    axes_dim: List[int]
    theta: int
    qkv_bias: bool
    guidance_embed: bool


class Flux:
    def __init__(self, params: FluxParams, device: Union[str, torch.device] = "cuda"):
        # same argument checks as flux/model.py:42-50
        if params.hidden_size % params.num_heads != 0:
            raise ValueError(f"Hidden size {params.hidden_size} must be divisible by num_heads {params.num_heads}")
        pe_dim = params.hidden_size // params.num_heads
        if sum(params.axes_dim) != pe_dim:
            raise ValueError(f"Got {params.axes_dim} but expected positional dim {pe_dim}")
        if pe_dim != 128 or len(params.axes_dim) != 3:
            raise ValueError("libfluxhip attention is built for head_dim 128 and 3 position axes")
        self.params = params
        self.in_channels = params.in_channels
        self.out_channels = params.in_channels
        self.hidden_size = params.hidden_size
        self.num_heads = params.num_heads
        if torch.device(device).type != "cuda":
            raise FluxHipError("Flux needs a HIP device: there is no CPU fallback for the denoise path")
        self.device = _lib.bind_device(device)
        self._side = None            # side stream of the launch plan (modulation GEMV under the first blocks)
        self._t_cache = OrderedDict()   # modulation_tables: device copies of the timestep groups seen so far (LRU, 64)
        self.plan_epoch = 0          # bumped whenever the workspaces / launch plans are rebuilt (enable_fp8)
        _lib.load()
        self._alloc_parameters()
        self._ws: "OrderedDict[Tuple[int, int, int], dict]" = OrderedDict()   # per-(B, S, L) workspaces + launch plans, LRU
        self.fp8_mx = os.environ.get("FLUXHIP_FP8_MX", "1") != "0"   # fp8 mode: block-scaled hand-off GELU -> next Linear (0: per-token quantise passes)
        self.fp8_mx_min_rows = int(os.environ.get("FLUXHIP_FP8_MX_MIN_ROWS", "0"))
        self.fp8 = False             # enable_fp8(): e4m3 weights + per-token e4m3 activations on the fp8 matrix cores
        self._w8: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._lora: Dict[str, Tuple[torch.Tensor, torch.Tensor, float]] = {}   # attach_lora(): layer -> (A^T pad, B^T pad, scale)
        self._lora_zero: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}
        # test knob (tests/test_full_size_parity_gpu.py mutation check): (layer name, d) adds d to every E8M0 block-scale byte
        # right after that layer's block-scaled PRODUCER launch, i.e. the consumer of its output sees scales 2^d too large.
        # None in the product; set through `set_debug_mx_shift` (rebuilds the launch plans).
        self.debug_mx_shift: Optional[Tuple[str, int]] = None

    def set_debug_mx_shift(self, layer: Optional[str], shift: int = 1) -> "Flux":
        self.debug_mx_shift = None if layer is None else (layer, int(shift))
        self._ws.clear()
        self.plan_epoch += 1
        return self

    # ------------------------------------------------------------------ parameters
    def _alloc_parameters(self) -> None:
        P, D = self.params, self.params.hidden_size
        mlp = int(D * P.mlp_ratio)
        dev = self.device
        W: Dict[str, torch.Tensor] = {}

        # row-concatenated modulation table: [img_mod, txt_mod] x depth, modulation x singles, final adaLN
        rows, off = 0, {}
        for i in range(P.depth):
            for st in ("img", "txt"):
                off[f"double_blocks.{i}.{st}_mod.lin"] = rows
                rows += 6 * D
        for i in range(P.depth_single_blocks):
            off[f"single_blocks.{i}.modulation.lin"] = rows
            rows += 3 * D
        off["final_layer.adaLN_modulation.layers.1"] = rows
        rows += 2 * D
        self.mod_rows, self.mod_off = rows, off
        self.mod_w = torch.empty(rows, D, dtype=BF16, device=dev)
        self.mod_b = torch.empty(rows, dtype=BF16, device=dev)
        sizes = {k: (6 * D if "double" in k else 3 * D if "single" in k else 2 * D) for k in off}
        for k, o in off.items():
            W[f"{k}.weight"] = self.mod_w[o:o + sizes[k]]
            W[f"{k}.bias"] = self.mod_b[o:o + sizes[k]]

        def lin(name, out_d, in_d, bias=True):
            W[f"{name}.weight"] = torch.empty(out_d, in_d, dtype=BF16, device=dev)
            if bias:
                W[f"{name}.bias"] = torch.empty(out_d, dtype=BF16, device=dev)

        lin("img_in", D, P.in_channels)
        lin("txt_in", D, P.context_in_dim)
        emb = [("time_in", 256), ("vector_in", P.vec_in_dim)] + ([("guidance_in", 256)] if P.guidance_embed else [])
        for e, d in emb:
            lin(f"{e}.in_layer", D, d)
            lin(f"{e}.out_layer", D, D)
        hd = D // P.num_heads
        for i in range(P.depth):
            p = f"double_blocks.{i}"
            for st in ("img", "txt"):
                lin(f"{p}.{st}_attn.qkv", 3 * D, D, bias=P.qkv_bias)
                W[f"{p}.{st}_attn.norm.query_norm.weight"] = torch.empty(hd, dtype=BF16, device=dev)
                W[f"{p}.{st}_attn.norm.key_norm.weight"] = torch.empty(hd, dtype=BF16, device=dev)
                lin(f"{p}.{st}_attn.proj", D, D)
                lin(f"{p}.{st}_mlp.layers.0", mlp, D)
                lin(f"{p}.{st}_mlp.layers.2", D, mlp)
        for i in range(P.depth_single_blocks):
            p = f"single_blocks.{i}"
            lin(f"{p}.linear1", 3 * D + mlp, D)
            lin(f"{p}.linear2", D, D + mlp)
            W[f"{p}.norm.query_norm.weight"] = torch.empty(hd, dtype=BF16, device=dev)
            W[f"{p}.norm.key_norm.weight"] = torch.empty(hd, dtype=BF16, device=dev)
        lin("final_layer.linear", P.in_channels, D)
        self._params = W

    def parameters(self) -> Dict[str, torch.Tensor]:
        return self._params

    def init_random(self, seed: int = 0) -> "Flux":
        """Random init with the reference framework's defaults:
        U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for Linear weight and bias, 1 for RMSNorm scales."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        done = set()
        for name, t in self._params.items():
            if id(t) in done:
                continue
            if name.endswith("_norm.weight"):
                t.fill_(1.0)
                continue
            base = name.rsplit(".", 1)[0]
            fan_in = self._params[f"{base}.weight"].shape[-1]
            k = 1.0 / math.sqrt(fan_in)
            # generate in fp32 chunks to keep the uniform unbiased after the bf16 rounding
            flat = t.view(-1)
            step = 1 << 26
            for s in range(0, flat.numel(), step):
                n = min(step, flat.numel() - s)
                flat[s:s + n] = ((torch.rand(n, generator=g, device=self.device) * 2 - 1) * k).to(BF16)
        return self

    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Checkpoint key mapping of the reference (flux/model.py:85-97)."""
        new = {}
        for k, w in weights.items():
            if k.startswith("model.diffusion_model."):
                k = k[22:]
            if k.endswith(".scale"):
                k = k[:-6] + ".weight"
            for seq in ("img_mlp", "txt_mlp", "adaLN_modulation"):
                if f".{seq}." in k:
                    k = k.replace(f".{seq}.", f".{seq}.layers.")
                    break
            new[k] = w
        return new

    def load_weights(self, weights: Union[Dict[str, torch.Tensor], Iterable[Tuple[str, torch.Tensor]]],
                     strict: bool = True) -> "Flux":
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
            dst.copy_(w.to(device=self.device, dtype=BF16))
            seen.add(k)
            n = k[: -len(".weight")] if k.endswith(".weight") else None
            if n in self._w8:            # an fp8 copy exists already: requantise in place (launch plans hold its address)
                ops.quantize_rows_fp8(dst, out=self._w8[n][0], scale=self._w8[n][1])
        if strict:
            missing = set(self._params) - seen
            if missing:
                raise ValueError(f"Missing parameters: {sorted(missing)[:5]} ...")
        return self

    def broadcast_weights(self, src: int = 0) -> int:
        """Multi-GPU start-up without N disk reads: rank `src` holds the loaded weights, every other
        rank receives them over RCCL/xGMI (23.8 GB in ~0.2-0.4 s on the per-link-bound ring).  The modulation Linears
        are views into one table, which is sent once.  fp8 copies are re-quantised locally (bit-identical)."""
        from .. import parallel
        views = {f"{k}.{leaf}" for k in self.mod_off for leaf in ("weight", "bias")}      # slices of mod_w / mod_b
        tensors = [self.mod_w, self.mod_b] + [t for k, t in self._params.items() if k not in views]
        n = parallel.broadcast_tensors(tensors, src)
        for name, (q, sc) in self._w8.items():
            ops.quantize_rows_fp8(self._params[f"{name}.weight"], out=q, scale=sc)
        return n

    # ------------------------------------------------------------------ fp8 (BASELINE.json configs[4]; txt2image.py -q)
    _FP8_LAYERS = ("attn.qkv", "attn.proj", "mlp.layers.0", "mlp.layers.2", "linear1", "linear2")

    def enable_fp8(self, enabled: bool = True) -> "Flux":
        """The reference's `--quantize` (txt2image.py:26-28,79-82: nn.quantize of the Linears with in_dim % 512 == 0)
        redesigned for CDNA4: the transformer blocks' Linears (qkv / proj / MLP / linear1 / linear2 = 99.6 % of the
        forward's FLOPs) get OCP e4m3fn weights with one float32 scale per output channel, their inputs are quantised per
        token on the fly, and the products run on the block-scaled fp8 MFMA at twice the bf16 rate
        (include/fluxhip.h, fluxhip_gemm_fp8).  The residual stream, norms, modulation, attention and the small
        embedders stay bf16.  Weights are quantised once, here; launch plans are rebuilt."""
        if enabled and self._lora:
            # The reference quantises LoRALinear.linear and keeps the low-rank branch (nn.quantize walks into the wrapper);
            # the fp8 plan here has no matrix-addend epilogue, so the branches are folded into W first (W + scale B^T A^T in
            # bf16, LoRALinear.fuse) - the result the reference gets with --fuse-adapter --quantize.  Stated, not silent.
            import warnings
            warnings.warn(f"enable_fp8: folding {len(self._lora)} unfused LoRA branches into the weights before quantising "
                          "(the fp8 plan carries no separate low-rank branch)")
            self.fuse_attached_lora()
        if enabled and not self._w8:
            for name, w in self._params.items():
                if name.endswith(".weight") and name[: -len(".weight")].endswith(self._FP8_LAYERS) and w.dim() == 2:
                    if w.shape[1] % 128:
                        raise ValueError(f"{name}: fp8 needs in_dim % 128 == 0")
                    self._w8[name[: -len(".weight")]] = ops.quantize_rows_fp8(w)
        self.fp8 = bool(enabled)
        self._ws.clear()
        self.plan_epoch += 1         # captured graphs of the old plans (FluxPipeline._graphs) must not be replayed
        return self

    def fuse_lora(self, adapter: Dict[str, torch.Tensor], scale: float = 1.0) -> int:
        """LoRA adapters at inference (flux/lora.py:28-43, flux/flux.py:229-246): for every Linear `name` with
        `name.lora_a` [in, r] and `name.lora_b` [r, out] in `adapter`,  W <- W + (scale * lora_b^T @ lora_a^T).astype(bf16).
        One libfluxhip GEMM per layer — M = out, N = in, K = r zero-padded to the 64-wide K-step, alpha = scale,
        residual epilogue in place on the weight (the epilogue rounds the product to bf16 before the add, exactly the
        reference's two roundings).  Returns the number of fused layers."""
        names = sorted(k[: -len(".lora_a")] for k in adapter if k.endswith(".lora_a"))
        for n in names:
            if f"{n}.lora_b" not in adapter:
                raise ValueError(f"adapter has {n}.lora_a but no {n}.lora_b")
            if f"{n}.weight" not in self._params:
                raise ValueError(f"adapter targets unknown layer {n}")
        for n in names:
            W = self._params[f"{n}.weight"]                   # [out, in] (modulation layers: a view into the table)
            a, b = adapter[f"{n}.lora_a"], adapter[f"{n}.lora_b"]
            out_d, in_d = W.shape
            r = a.shape[1]
            if tuple(a.shape) != (in_d, r) or tuple(b.shape) != (r, out_d):
                raise ValueError(f"Shape mismatch for {n}: lora_a {tuple(a.shape)}, lora_b {tuple(b.shape)}, weight {tuple(W.shape)}")
            rp = (r + 63) // 64 * 64
            bt = torch.zeros(out_d, rp, dtype=BF16, device=self.device)     # lora_b^T, K-contiguous
            bt[:, :r] = b.to(device=self.device, dtype=BF16).t()
            ap = torch.zeros(in_d, rp, dtype=BF16, device=self.device)      # lora_a is already [in, r] = the "weight" operand
            ap[:, :r] = a.to(device=self.device, dtype=BF16)
            ops.gemm(make_gemm_desc([dict(A=bt.data_ptr(), W=ap.data_ptr(), C=W.data_ptr(), res=W.data_ptr(), M=out_d)],
                                    1, in_d, rp, rp, in_d, EPI_GATE_RES, alpha=float(scale)))
            if n in self._w8:            # fp8 copy of this layer already made: requantise IN PLACE (launch plans hold its address)
                ops.quantize_rows_fp8(W, out=self._w8[n][0], scale=self._w8[n][1])
        torch.cuda.synchronize(self.device)
        return len(names)

    def fuse_attached_lora(self) -> int:
        """Fold the branches `attach_lora` keeps separate into the weights (LoRALinear.fuse, flux/lora.py:28-43) and drop them:
        W <- W + (scale * B^T A^T).astype(bf16), the same in-place GEMM as `fuse_lora`.  Returns the number of layers folded."""
        n_l = len(self._lora)
        for n, (at, bt, sc) in self._lora.items():
            W = self._params[f"{n}.weight"]
            out_d, in_d = W.shape
            ap = at.t().contiguous()                          # [in, pad]: the "weight" operand of the update GEMM
            ops.gemm(make_gemm_desc([dict(A=bt.data_ptr(), W=ap.data_ptr(), C=W.data_ptr(), res=W.data_ptr(), M=out_d)],
                                    1, in_d, ap.shape[1], ap.shape[1], in_d, EPI_GATE_RES, alpha=float(sc)))
            if n in self._w8:
                ops.quantize_rows_fp8(W, out=self._w8[n][0], scale=self._w8[n][1])
        torch.cuda.synchronize(self.device)
        if n_l:
            self._lora = {}
            self._lora_zero.clear()
            self._ws.clear()
            self.plan_epoch += 1
        return n_l

    LORA_PAD = 64        # the rank is zero-padded to whole K-steps of the GEMM (exact: the padding multiplies zeros); attach_lora
                         # raises the instance's value to ceil(max rank / 64) * 64 (dreambooth.py --lora-rank is unbounded)

    def splits_for_adapter(self, adapter: Dict[str, torch.Tensor]):
        """(branch, fold): the adapter's entries that can run as unfused low-rank branches (the blocks' qkv / proj / MLP /
        linear1 / linear2 Linears) and those that cannot (the modulation Linears the reference's linear_to_lora_layers also
        wraps, flux/flux.py:229-239: here they are rows of ONE concatenated GEMV table, evaluated once per image)."""
        branch, fold = {}, {}
        for k, v in adapter.items():
            n = k.rsplit(".", 1)[0]
            (branch if (n not in self.mod_off and n.endswith(self._FP8_LAYERS)) else fold)[k] = v
        return branch, fold

    def attach_lora(self, adapter: Dict[str, torch.Tensor], scale: float = 1.0) -> int:
        """LoRA adapters kept as SEPARATE low-rank branches — what the reference runs when `--fuse-adapter` is absent
        (flux/lora.py:73-76, LoRALinear.__call__):  y = linear(x) + (scale * ((x @ lora_a) @ lora_b)).astype(x.dtype).
        Unlike `fuse_lora`, W is not touched, so an update smaller than half a bf16 ulp of W is not rounded away.  Per adapted
        Linear the launch plan gets two skinny GEMMs (u = x A with the rank padded to 64 columns; z = scale * u B) and the
        layer's own GEMM takes z as a matrix addend in its epilogue — before the fused activation / gate, where the reference
        adds it (include/fluxhip.h, fluxhip_gemm_group.add).  Layers: the transformer blocks' Linears (what
        FluxPipeline.linear_to_lora_layers adapts, flux/flux.py:229-239), incl. the modulation Linears?  No: those are
        evaluated by the GEMV over the concatenated table; an adapter that targets them is refused here (use fuse_lora).
        Returns the number of adapted layers; launch plans are rebuilt."""
        names = sorted(k[: -len(".lora_a")] for k in adapter if k.endswith(".lora_a"))
        for n in names:
            if f"{n}.lora_b" not in adapter:
                raise ValueError(f"adapter has {n}.lora_a but no {n}.lora_b")
            if f"{n}.weight" not in self._params:
                raise ValueError(f"adapter targets unknown layer {n}")
            if n in self.mod_off or not n.endswith(self._FP8_LAYERS):
                raise ValueError(f"{n}: only the blocks' qkv / proj / MLP / linear1 / linear2 Linears run an unfused branch; "
                                 "fold this one with fuse_lora (FluxPipeline.load_adapter does: Flux.splits_for_adapter)")
        if self.fp8 and names:
            raise ValueError("unfused LoRA branches run on the bf16 plan: enable_fp8(False) first, or fuse_lora "
                             "(FluxPipeline.load_adapter folds by itself when the flow model is already quantised)")
        lora = {}
        pad = max([64] + [(adapter[f"{n}.lora_a"].shape[1] + 63) // 64 * 64 for n in names])
        for n in names:
            W = self._params[f"{n}.weight"]
            a, b = adapter[f"{n}.lora_a"], adapter[f"{n}.lora_b"]
            out_d, in_d = W.shape
            r = a.shape[1]
            if tuple(a.shape) != (in_d, r) or tuple(b.shape) != (r, out_d):
                raise ValueError(f"Shape mismatch for {n}: lora_a {tuple(a.shape)}, lora_b {tuple(b.shape)}, weight {tuple(W.shape)}")
            at = torch.zeros(pad, in_d, dtype=BF16, device=self.device)     # rows = rank: the "weight" of u = x A
            at[:r] = a.to(device=self.device, dtype=BF16).t()
            bt = torch.zeros(out_d, pad, dtype=BF16, device=self.device)    # [out, rank]: the "weight" of z = u B
            bt[:, :r] = b.to(device=self.device, dtype=BF16).t()
            lora[n] = (at, bt, float(scale))
        self._lora = lora
        self.LORA_PAD = pad
        self._lora_zero.clear()
        self._ws.clear()
        self.plan_epoch += 1
        return len(names)

    # ------------------------------------------------------------------ workspace + launch plan
    MAX_WORKSPACES = 8      # unpinned (B, S, L) workspaces kept (LRU); ~1 GB each at 1024 x 1024

    def pin_workspace(self, B: int, S: int, L: int) -> None:
        """A captured hipGraph holds the addresses of this shape's workspace: it must outlive the graph."""
        ws = self._workspace(B, S, L)
        ws["pins"] = ws.get("pins", 0) + 1

    def release_workspace(self, B: int, S: int, L: int) -> None:
        """Undo one pin_workspace; an unpinned workspace becomes evictable again (and is dropped right away when the
        cache is over its bound)."""
        ws = self._ws.get((B, S, L))
        if ws is None:
            return
        ws["pins"] = max(0, ws.get("pins", 0) - 1)
        self._evict_workspaces()

    def _evict_workspaces(self) -> None:
        free = [k for k, w in self._ws.items() if not w.get("pins", 0)]
        for k in free[: max(0, len(free) - self.MAX_WORKSPACES)]:
            del self._ws[k]

    def _workspace(self, B: int, S: int, L: int) -> dict:
        key = (B, S, L)
        ws = self._ws.get(key)
        if ws is not None:
            self._ws.move_to_end(key)
            return ws
        P, D, dev = self.params, self.params.hidden_size, self.device
        mlp = int(D * P.mlp_ratio)
        T, H = S + L, P.num_heads
        Tpad = (T + 63) // 64 * 64

        def buf(*shape, dtype=BF16):
            return torch.empty(*shape, dtype=dtype, device=dev)

        ws = dict(
            B=B, S=S, L=L, T=T, Tpad=Tpad,
            in_img=buf(B, L, P.in_channels), in_txt=buf(B, S, P.context_in_dim), in_y=buf(B, P.vec_in_dim),
            in_t=buf(B), in_g=buf(B), in_ids=buf(B, T, 3, dtype=torch.int32),
            temb=buf(B, 256), h1=buf(B, D), vec=buf(B, D), mods=buf(B, self.mod_rows),
            x=buf(B, T, D), xm=buf(B, T, D), qkv=buf(B, T, 3 * D), attn=buf(B, T, D), hmlp=buf(B, T, mlp),
            cat=buf(B, T, D + mlp), Q=buf(B, H, T, 128), K=buf(B, H, T, 128), Vt=buf(B, H, 128, Tpad),
            rope=buf(B, T, 64, 2), xl=buf(B, L, D), pred=buf(B, L, P.in_channels),
        )
        if self._lora:  # unfused LoRA branches: u = x A [B, T, 64] and z = scale * u B [B, T, widest adapted output]
            ws["lora_u"] = buf(B, T, self.LORA_PAD)
            ws["lora_z"] = buf(B, T, max(self._params[f"{n}.weight"].shape[0] for n in self._lora))
        if self.fp8:   # one (e4m3 rows, per-token scale) scratch pair shared by every GEMM input of the step
            ws["a8"] = buf(B * T, D + mlp, dtype=torch.uint8)
            ws["asc"] = buf(B * T, dtype=torch.float32)
            # block-scaled ("MX") operands of mlp.layers.2 / linear2: written by the GELU epilogue of the GEMM before them
            # (include/fluxhip.h, fluxhip_fp8_mx) - whole 64-row groups per stream and image
            # (fp8_mx_min_rows / FLUXHIP_FP8_MX_MIN_ROWS: a row threshold for A/B runs.  With 256-row tiles only, one 512^2 image
            #  - B*T = 1280 - was 9 % slower block-scaled, 14.5 vs 13.2 ms per step; with the 128-row ping-pong tiles it is level,
            #  13.46 vs 13.51, and ahead from two images on: 22.6 vs 23.5 ms)
            ws["mx"] = self.fp8_mx and S % 64 == 0 and L % 64 == 0 and B * T >= self.fp8_mx_min_rows
            if ws["mx"]:
                ws["a8m"] = buf(B * T, D + mlp, dtype=torch.uint8)
                ws["amx"] = ops.mx_scale_buffer(B * T, D + mlp, dev)
        ws["plan"] = self._build_plan(ws)
        self._ws[key] = ws
        self._evict_workspaces()
        return ws

    def _build_plan(self, ws: dict) -> list:
        """The fixed sequence of libfluxhip launches of one Flux forward (flux/model.py:112-136)."""
        lib = _lib.load()
        P, D, Wt = self.params, self.params.hidden_size, self._params
        mlp = int(D * P.mlp_ratio)
        B, S, L, T, Tpad, H = ws["B"], ws["S"], ws["L"], ws["T"], ws["Tpad"], P.num_heads
        e = 2  # bytes per bf16
        ptr = {k: v.data_ptr() for k, v in ws.items() if isinstance(v, torch.Tensor)}
        w = lambda n: Wt[n].data_ptr()           # noqa: E731
        wo = lambda n: Wt[n].data_ptr() if n in Wt else None   # noqa: E731
        mp, NM = ptr["mods"], self.mod_rows
        plan: list = []
        keep: list = []   # descriptors must outlive the plan

        def call(fn, *args):
            plan.append((fn, args))

        # diagnostic: FLUXHIP_PLAN_TILES="3072x12288=558,..." forces tile | split << 8 for the block GEMMs of that N x K
        forced = {tuple(int(v) for v in e.split("=")[0].split("x")): int(e.split("=")[1])
                  for e in os.environ.get("FLUXHIP_PLAN_TILES", "").split(",") if e}

        def gemm(groups, nbatch, N, K, lda, ldc, epi=EPI_BIAS, names=None, rows=None, **kw):
            """names / rows: the layer name and first token row of every group - a launch whose layers carry an unfused LoRA
            branch (attach_lora) is preceded by u = x A and z = scale * u B and takes z as its matrix addend."""
            if names and any(n in self._lora for n in names):
                U, Z, ldz, R = ptr["lora_u"], ptr["lora_z"], ws["lora_z"].shape[-1], self.LORA_PAD
                ug, zg, scales = [], [], set()
                for g, n, r0 in zip(groups, names, rows):
                    if n in self._lora:
                        at, bt, sc = self._lora[n]
                        scales.add(sc)
                    else:               # a stream of this launch without an adapter: a zero branch (z = 0 exactly)
                        at = bt = None
                    if at is None:
                        key_ = ("zero", N, K)
                        if key_ not in self._lora_zero:
                            self._lora_zero[key_] = (torch.zeros(R, K, dtype=BF16, device=self.device),
                                                     torch.zeros(N, R, dtype=BF16, device=self.device))
                        at, bt = self._lora_zero[key_]
                    ug.append(dict(A=g["A"], W=at.data_ptr(), C=U + r0 * R * e, a_bstride=g.get("a_bstride", 0), c_bstride=T * R, M=g["M"]))
                    zg.append(dict(A=U + r0 * R * e, W=bt.data_ptr(), C=Z + r0 * ldz * e, a_bstride=T * R, c_bstride=T * ldz, M=g["M"]))
                    g.update(add=Z + r0 * ldz * e, add_bstride=T * ldz)
                if len(scales) > 1:
                    raise ValueError("the adapters of one launch's streams must share their scale")
                du = make_gemm_desc(ug, nbatch, R, K, lda, R)
                dz = make_gemm_desc(zg, nbatch, N, R, R, ldz, alpha=scales.pop())
                keep.extend([du, dz])
                call(lib.fluxhip_gemm_bf16, ctypes.byref(du))
                call(lib.fluxhip_gemm_bf16, ctypes.byref(dz))
                kw["ld_add"] = ldz
            if (N, K) in forced:
                kw["tile_cfg"] = forced[(N, K)]
            d = make_gemm_desc(groups, nbatch, N, K, lda, ldc, epi, **kw)
            keep.append(d)
            call(lib.fluxhip_gemm_bf16, ctypes.byref(d))

        def gemm8(src, K, groups, wnames, N, ldc, epi=EPI_BIAS, **kw):   # noqa: E306
            """fp8 variant of a block Linear over the packed token buffer: quantise the bf16 input rows `src`
            [B*T, K] per token into the shared scratch, then one fluxhip_gemm_fp8 launch.  `groups` as for gemm() with
            A given as the ROW offset of the group inside a batch (txt rows first); wnames = weight names per group."""
            if src is not None:      # None: the producer (fluxhip_ln_modulate_fp8) already wrote the e4m3 rows + scales
                call(lib.fluxhip_quantize_rows_fp8, src, ptr["a8"], ptr["asc"], B * T, K, K)
            gs, a_sc, w_sc = [], [], []
            for g, wn in zip(groups, wnames):
                wq, wscale = self._w8[wn]
                row0 = g.pop("row0")
                g.update(A=ptr["a8"] + row0 * K, W=wq.data_ptr(), a_bstride=T * K)
                gs.append(g)
                a_sc.append(ptr["asc"] + row0 * 4)
                w_sc.append(wscale.data_ptr())
            if (N, K) in forced:
                kw["tile_cfg"] = forced[(N, K)]
            d = make_gemm_desc(gs, B, N, K, K, ldc, epi, **kw)
            sc = ops.make_fp8_scales(a_sc, w_sc, T)
            keep.extend([d, sc])
            call(lib.fluxhip_gemm_fp8, ctypes.byref(d), ctypes.byref(sc))

        MX = bool(ws.get("mx"))

        def gemm8mx(K, groups, wnames, N, ldc, epi, produce, ld8, coloff=0, **kw):
            """Block-scaled forms of gemm8 (fluxhip_gemm_fp8_mx).  produce=True: per-token input from the shared scratch (K
            wide), the GELU'd output leaves as e4m3 + block scales in `a8m` viewed as [B*T, ld8] from column `coloff`;
            produce=False: the activation operand IS `a8m` ([B*T, ld8 = K]) with its block scales."""
            gs, a_sc, w_sc, c8, rows = [], [], [], [], []
            for g, wn in zip(groups, wnames):
                wq, wscale = self._w8[wn]
                row0 = g.pop("row0")
                rows.append(row0)
                if produce:
                    g.update(A=ptr["a8"] + row0 * K, W=wq.data_ptr(), a_bstride=T * K)
                    a_sc.append(ptr["asc"] + row0 * 4)
                    c8.append(ptr["a8m"] + row0 * ld8)
                else:
                    g.update(A=ptr["a8m"] + row0 * ld8, W=wq.data_ptr(), a_bstride=T * ld8)
                    a_sc.append(None)
                gs.append(g)
                w_sc.append(wscale.data_ptr())
            if (N, K) in forced:
                kw["tile_cfg"] = forced[(N, K)]
            d = make_gemm_desc(gs, B, N, K, K if produce else ld8, ldc, epi, **kw)
            sc = ops.make_fp8_scales(a_sc, w_sc, T)
            if produce:
                mxd = ops.make_fp8_mx(c8=c8, c8_bstride=T * ld8, ldc8=ld8, c8_coloff=coloff, c_mx=ptr["amx"], c_row0=rows,
                                      c_bstride=T, c_kstride=B * T)
            else:
                mxd = ops.make_fp8_mx(a_mx=ptr["amx"], a_row0=rows, a_bstride=T, a_kstride=B * T)
            keep.extend([d, sc, mxd])
            call(lib.fluxhip_gemm_fp8_mx, ctypes.byref(d), ctypes.byref(sc), ctypes.byref(mxd))
            if produce and self.debug_mx_shift is not None and self.debug_mx_shift[0] in wnames:
                amx, dshift = ws["amx"], self.debug_mx_shift[1]
                plan.append(("py", lambda: amx.add_(dshift)))      # (mutation knob: never set in the product)

        def small(x, wn, out, K, N, silu_in, accum):
            call(lib.fluxhip_small_linear_bf16, x, w(wn + ".weight"), wo(wn + ".bias"), out, B, N, K, silu_in, accum)

        # vec = time_in(temb(t)) [+ guidance_in(temb(g))] + vector_in(y)      flux/model.py:113-120
        call(lib.fluxhip_timestep_embedding_bf16, ptr["in_t"], ptr["temb"], B, 256, 1000.0, 10000.0)
        small(ptr["temb"], "time_in.in_layer", ptr["h1"], 256, D, 0, 0)
        small(ptr["h1"], "time_in.out_layer", ptr["vec"], D, D, 1, 0)
        if P.guidance_embed:
            call(lib.fluxhip_timestep_embedding_bf16, ptr["in_g"], ptr["temb"], B, 256, 1000.0, 10000.0)
            small(ptr["temb"], "guidance_in.in_layer", ptr["h1"], 256, D, 0, 0)
            small(ptr["h1"], "guidance_in.out_layer", ptr["vec"], D, D, 1, 1)
        small(ptr["in_y"], "vector_in.in_layer", ptr["h1"], P.vec_in_dim, D, 0, 0)
        small(ptr["h1"], "vector_in.out_layer", ptr["vec"], D, D, 1, 1)
        # every Modulation.lin(silu(vec)) of the step in one launch            flux/layers.py:136-137
        # (6.5 GB of weights at batch 1: pure HBM streaming, ~0.95 ms.)  The rows of the first JOIN_AT double blocks
        # are computed in line; the rest runs on a side stream underneath the first blocks' MFMA-bound GEMMs and is
        # joined before block JOIN_AT reads its rows.  Only for B == 1: a row range of the [B, NM] table is not a
        # dense [B, n] block for the GEMV's output addressing at larger batch, where it also matters less.
        JOIN_AT = int(os.environ.get("FLUXHIP_MOD_JOIN", "3"))
        split_rows = self.mod_off[f"double_blocks.{JOIN_AT}.img_mod.lin"] if (B == 1 and P.depth > JOIN_AT) else NM
        call(lib.fluxhip_small_linear_bf16, ptr["vec"], self.mod_w.data_ptr(), self.mod_b.data_ptr(), mp, B, split_rows, D, 1, 0)
        if split_rows < NM:
            plan.append(("side", (lib.fluxhip_small_linear_bf16,
                                  (ptr["vec"], self.mod_w.data_ptr() + split_rows * D * e, self.mod_b.data_ptr() + split_rows * e,
                                   mp + split_rows * e, B, NM - split_rows, D, 1, 0))))
        plan.append(("mod_end", ()))     # everything up to here depends on (t, y, guidance) only: see modulation_tables
        # pe = EmbedND(ids)                                                    flux/model.py:123-124
        call(lib.fluxhip_rope_table_bf16, ptr["in_ids"], ptr["rope"], B * T, 3, P.axes_dim[0], P.axes_dim[1],
             P.axes_dim[2], float(P.theta))
        # txt_in / img_in write straight into the packed token buffer          flux/model.py:112,121
        if S > 0:
            gemm([dict(A=ptr["in_txt"], W=w("txt_in.weight"), bias=wo("txt_in.bias"), C=ptr["x"],
                       a_bstride=S * P.context_in_dim, c_bstride=T * D, M=S)], B, D, P.context_in_dim,
                 P.context_in_dim, D)
        gemm([dict(A=ptr["in_img"], W=w("img_in.weight"), bias=wo("img_in.bias"), C=ptr["x"] + S * D * e,
                   a_bstride=L * P.in_channels, c_bstride=T * D, M=L)], B, D, P.in_channels, P.in_channels, D)

        def two_streams(A, lda, a_bs, C, ldc, c_bs, wname, M_txt_rows=S, res=None, gate_off=None, i_off=0, t_off=0,
                        prefix=""):
            """txt group (rows [0,S)) + img group (rows [S,T)) of one double-block Linear."""
            gs = []
            for st, row0, M, moff in (("txt", 0, S, t_off), ("img", S, L, i_off)):
                if M == 0:
                    continue
                g = dict(A=A + row0 * lda * e, W=w(f"{prefix}.{st}_{wname}.weight"), bias=wo(f"{prefix}.{st}_{wname}.bias"),
                         C=C + row0 * ldc * e, a_bstride=a_bs, c_bstride=c_bs, M=M)
                if res is not None:
                    g.update(res=res + row0 * ldc * e, gate=mp + (moff + gate_off) * e, gate_bstride=NM)
                gs.append(g)
            return gs

        def stream_names(wname, prefix):
            """(layer names, first rows) of the groups two_streams builds."""
            sel = [(f"{prefix}.{st}_{wname}", r0) for st, r0, M in (("txt", 0, S), ("img", S, L)) if M]
            return dict(names=[n for n, _ in sel], rows=[r for _, r in sel])

        def two_streams8(src, K, C, ldc, c_bs, wname, N, epi=EPI_BIAS, res=None, gate_off=None, i_off=0, t_off=0, prefix="",
                         mx=None):
            gs, wn = [], []
            for st, row0, M, moff in (("txt", 0, S, t_off), ("img", S, L, i_off)):
                if M == 0:
                    continue
                g = dict(row0=row0, bias=wo(f"{prefix}.{st}_{wname}.bias"), C=C + row0 * ldc * e, c_bstride=c_bs, M=M)
                if res is not None:
                    g.update(res=res + row0 * ldc * e, gate=mp + (moff + gate_off) * e, gate_bstride=NM)
                gs.append(g)
                wn.append(f"{prefix}.{st}_{wname}")
            if mx is None:
                gemm8(src, K, gs, wn, N, ldc, epi)
            else:
                gemm8mx(K, gs, wn, N, ldc, epi, mx == "produce", mlp if mx == "produce" else K)

        F8 = self.fp8

        def ln_mod(shift_txt, scale_txt, shift_img, scale_img, S_):
            """LayerNorm + modulate of the whole token buffer -> xm (bf16), or, in fp8 mode, straight to the e4m3 rows and
            per-token scales the next GEMM reads (the quantisation is fused into the producer; returns the GEMM's src)."""
            if F8:
                call(lib.fluxhip_ln_modulate_fp8, ptr["x"], ptr["a8"], ptr["asc"], B, T, D, S_, T * D, T * D,
                     shift_txt, scale_txt, shift_img, scale_img, NM, 1e-6)
                return None
            call(lib.fluxhip_ln_modulate_bf16, ptr["x"], ptr["xm"], B, T, D, S_, T * D, T * D,
                 shift_txt, scale_txt, shift_img, scale_img, NM, 1e-6)
            return ptr["xm"]

        for i in range(P.depth):                                              # flux/layers.py:181-231
            p = f"double_blocks.{i}"
            if i == JOIN_AT and split_rows < NM:
                plan.append(("join", ()))
            io, to = self.mod_off[f"{p}.img_mod.lin"], self.mod_off[f"{p}.txt_mod.lin"]
            src = ln_mod(mp + to * e, mp + (to + D) * e, mp + io * e, mp + (io + D) * e, S)
            if F8:
                two_streams8(src, D, ptr["qkv"], 3 * D, T * 3 * D, "attn.qkv", 3 * D, prefix=p)
            else:
                gemm(two_streams(ptr["xm"], D, T * D, ptr["qkv"], 3 * D, T * 3 * D, "attn.qkv", prefix=p), B, 3 * D, D, D, 3 * D,
                     **stream_names("attn.qkv", p))
            call(lib.fluxhip_qk_norm_rope_bf16, ptr["qkv"], 3 * D, B, T, S, H,
                 wo(f"{p}.txt_attn.norm.query_norm.weight"), wo(f"{p}.txt_attn.norm.key_norm.weight"),
                 w(f"{p}.img_attn.norm.query_norm.weight"), w(f"{p}.img_attn.norm.key_norm.weight"),
                 ptr["rope"], T * 128, ptr["Q"], ptr["K"], ptr["Vt"], Tpad, 1e-5)
            if F8 and MX:     # attention's finalize step quantises its own output: e4m3 + block scales, attn.proj's operand
                call(lib.fluxhip_attention_d128_mx, ptr["Q"], ptr["K"], ptr["Vt"], ptr["a8m"], D, ptr["amx"], B * T, B, H, T, Tpad,
                     128 ** -0.5)
                two_streams8(None, D, ptr["x"], D, T * D, "attn.proj", D, EPI_GATE_RES, res=ptr["x"], gate_off=2 * D,
                             i_off=io, t_off=to, prefix=p, mx="consume")
            else:
                call(lib.fluxhip_attention_d128_bf16, ptr["Q"], ptr["K"], ptr["Vt"], ptr["attn"], D, B, H, T, Tpad,
                     128 ** -0.5)
            if F8 and not MX:
                two_streams8(ptr["attn"], D, ptr["x"], D, T * D, "attn.proj", D, EPI_GATE_RES, res=ptr["x"], gate_off=2 * D,
                             i_off=io, t_off=to, prefix=p)
            elif not F8:
                gemm(two_streams(ptr["attn"], D, T * D, ptr["x"], D, T * D, "attn.proj", res=ptr["x"], gate_off=2 * D,
                                 i_off=io, t_off=to, prefix=p), B, D, D, D, D, EPI_GATE_RES, **stream_names("attn.proj", p))
            src = ln_mod(mp + (to + 3 * D) * e, mp + (to + 4 * D) * e, mp + (io + 3 * D) * e, mp + (io + 4 * D) * e, S)
            if F8 and MX:     # GELU'd hidden state leaves mlp.layers.0 as e4m3 + block scales: no bf16 round trip, no quantise pass
                two_streams8(None, D, ptr["hmlp"], mlp, T * mlp, "mlp.layers.0", mlp, EPI_GELU_TANH, prefix=p, mx="produce")
                two_streams8(None, mlp, ptr["x"], D, T * D, "mlp.layers.2", D, EPI_GATE_RES, res=ptr["x"],
                             gate_off=5 * D, i_off=io, t_off=to, prefix=p, mx="consume")
            elif F8:
                two_streams8(src, D, ptr["hmlp"], mlp, T * mlp, "mlp.layers.0", mlp, EPI_GELU_TANH, prefix=p)
                two_streams8(ptr["hmlp"], mlp, ptr["x"], D, T * D, "mlp.layers.2", D, EPI_GATE_RES, res=ptr["x"],
                             gate_off=5 * D, i_off=io, t_off=to, prefix=p)
            else:
                gemm(two_streams(ptr["xm"], D, T * D, ptr["hmlp"], mlp, T * mlp, "mlp.layers.0", prefix=p), B, mlp, D, D, mlp,
                     EPI_GELU_TANH, **stream_names("mlp.layers.0", p))
                gemm(two_streams(ptr["hmlp"], mlp, T * mlp, ptr["x"], D, T * D, "mlp.layers.2", res=ptr["x"], gate_off=5 * D,
                                 i_off=io, t_off=to, prefix=p), B, D, mlp, mlp, D, EPI_GATE_RES, **stream_names("mlp.layers.2", p))

        for i in range(P.depth_single_blocks):                                # flux/layers.py:262-284
            p = f"single_blocks.{i}"
            o = self.mod_off[f"{p}.modulation.lin"]
            src = ln_mod(None, None, mp + o * e, mp + (o + D) * e, 0)
            if F8 and MX:     # the GELU half of linear1 -> e4m3 + block scales at columns [D, D + mlp) of linear2's operand
                gemm8mx(D, [dict(row0=0, bias=wo(f"{p}.linear1.bias"), C=ptr["qkv"], c_bstride=T * 3 * D, M=T)],
                        [f"{p}.linear1"], 3 * D + mlp, 3 * D, EPI_SPLIT_GELU, True, D + mlp, coloff=D, n_split=3 * D,
                        C2=ptr["cat"], ldc2=D + mlp, c2_bstride=T * (D + mlp), c2_coloff=D)
            elif F8:
                gemm8(src, D, [dict(row0=0, bias=wo(f"{p}.linear1.bias"), C=ptr["qkv"], c_bstride=T * 3 * D, M=T)],
                      [f"{p}.linear1"], 3 * D + mlp, 3 * D, EPI_SPLIT_GELU, n_split=3 * D, C2=ptr["cat"], ldc2=D + mlp,
                      c2_bstride=T * (D + mlp), c2_coloff=D)
            else:
                gemm([dict(A=ptr["xm"], W=w(f"{p}.linear1.weight"), bias=wo(f"{p}.linear1.bias"), C=ptr["qkv"],
                           a_bstride=T * D, c_bstride=T * 3 * D, M=T)], B, 3 * D + mlp, D, D, 3 * D, EPI_SPLIT_GELU,
                     names=[f"{p}.linear1"], rows=[0],
                     n_split=3 * D, C2=ptr["cat"], ldc2=D + mlp, c2_bstride=T * (D + mlp), c2_coloff=D)
            call(lib.fluxhip_qk_norm_rope_bf16, ptr["qkv"], 3 * D, B, T, 0, H, None, None,
                 w(f"{p}.norm.query_norm.weight"), w(f"{p}.norm.key_norm.weight"),
                 ptr["rope"], T * 128, ptr["Q"], ptr["K"], ptr["Vt"], Tpad, 1e-5)
            if F8 and MX:     # attention output -> block-scaled columns [0, D) of linear2's operand (the GELU half is there already)
                call(lib.fluxhip_attention_d128_mx, ptr["Q"], ptr["K"], ptr["Vt"], ptr["a8m"], D + mlp, ptr["amx"], B * T, B, H, T,
                     Tpad, 128 ** -0.5)
                gemm8mx(D + mlp, [dict(row0=0, bias=wo(f"{p}.linear2.bias"), C=ptr["x"], res=ptr["x"],
                                       gate=mp + (o + 2 * D) * e, gate_bstride=NM, c_bstride=T * D, M=T)],
                        [f"{p}.linear2"], D, D, EPI_GATE_RES, False, D + mlp)
                continue
            call(lib.fluxhip_attention_d128_bf16, ptr["Q"], ptr["K"], ptr["Vt"], ptr["cat"], D + mlp, B, H, T, Tpad,
                 128 ** -0.5)
            if F8:
                gemm8(ptr["cat"], D + mlp, [dict(row0=0, bias=wo(f"{p}.linear2.bias"), C=ptr["x"], res=ptr["x"],
                                                 gate=mp + (o + 2 * D) * e, gate_bstride=NM, c_bstride=T * D, M=T)],
                      [f"{p}.linear2"], D, D, EPI_GATE_RES)
            else:
                gemm([dict(A=ptr["cat"], W=w(f"{p}.linear2.weight"), bias=wo(f"{p}.linear2.bias"), C=ptr["x"], res=ptr["x"],
                           gate=mp + (o + 2 * D) * e, gate_bstride=NM, a_bstride=T * (D + mlp), c_bstride=T * D, M=T)],
                     B, D, D + mlp, D + mlp, D, EPI_GATE_RES, names=[f"{p}.linear2"], rows=[0])

        # LastLayer on the img rows                                           flux/layers.py:298-302
        o = self.mod_off["final_layer.adaLN_modulation.layers.1"]
        call(lib.fluxhip_ln_modulate_bf16, ptr["x"] + S * D * e, ptr["xl"], B, L, D, 0, T * D, L * D,
             None, None, mp + o * e, mp + (o + D) * e, NM, 1e-6)
        gemm([dict(A=ptr["xl"], W=w("final_layer.linear.weight"), bias=wo("final_layer.linear.bias"), C=ptr["pred"],
                   M=B * L)], 1, P.in_channels, D, D, P.in_channels)
        plan.append(("keepalive", keep))
        return plan

    # ------------------------------------------------------------------ forward
    def __call__(self, img: torch.Tensor, img_ids: torch.Tensor, txt: torch.Tensor, txt_ids: torch.Tensor,
                 timesteps: torch.Tensor, y: torch.Tensor, guidance: Optional[torch.Tensor] = None) -> torch.Tensor:
        if img.ndim != 3 or txt.ndim != 3:
            raise ValueError("Input img and txt tensors must have 3 dimensions.")
        if self.params.guidance_embed and guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")
        B, L, _ = img.shape
        S = txt.shape[1]
        ws = self._workspace(B, S, L)
        ws["in_img"].copy_(img)
        ws["in_txt"].copy_(txt)
        ws["in_y"].copy_(y)
        ws["in_t"].copy_(timesteps)
        if guidance is not None:
            ws["in_g"].copy_(guidance)
        if S > 0:
            ws["in_ids"][:, :S].copy_(txt_ids)
        ws["in_ids"][:, S:].copy_(img_ids)
        self.run_plan(ws)
        return ws["pred"].clone()

    def profile_plan(self, ws: dict, with_shape: bool = False, with_bytes: bool = False) -> list:
        """Run the plan eagerly with a HIP event pair around every launch (events are recorded on the
        stream the kernels are launched on).  Returns [(kernel label, ms, flops)] in launch order;
        GEMM labels carry the tile configuration the library selected.  with_bytes: a fourth field, the launch's
        algorithmic HBM bytes (bf16 GEMMs only, 0 elsewhere)."""
        lib = _lib.load()
        stream = torch.cuda.current_stream()
        recs = []
        for fn, args in ws["plan"]:
            if fn in ("keepalive", "join", "mod_end"):
                continue
            if fn == "py":
                args()
                continue
            if fn == "side":        # timed in line here (the graph runs it on the side stream)
                fn, args = args
            label, flops, nbytes = fn.__name__, 0.0, 0.0
            if fn.__name__ == "fluxhip_gemm_bf16":
                d = args[0]._obj
                m_total = sum(d.g[i].M for i in range(d.ngroups)) * d.nbatch
                flops = 2.0 * m_total * d.N * d.K
                # algorithmic HBM bytes of the launch: every operand once - activations, one weight panel per
                # group, the output, and the residual / gate / bias the epilogue reads
                nbytes = 2.0 * (m_total * d.K + d.ngroups * d.N * d.K + m_total * d.N + d.ngroups * d.N)
                for i in range(d.ngroups):
                    if d.g[i].res:
                        nbytes += 2.0 * d.g[i].M * d.nbatch * d.N
                    if d.g[i].gate:
                        nbytes += 2.0 * d.nbatch * d.N
                code = lib.fluxhip_gemm_tile_cfg(args[0])      # tile cfg | split-K factor << 8
                label = f"fluxhip_gemm_bf16/cfg{code & 255}" + (f"s{code >> 8}" if (code >> 8) > 1 else "")
                if with_shape:
                    label += f" N{d.N} K{d.K}"
            elif fn.__name__ == "fluxhip_gemm_fp8":
                d = args[0]._obj
                m_total = sum(d.g[i].M for i in range(d.ngroups)) * d.nbatch
                flops = 2.0 * m_total * d.N * d.K
                code = lib.fluxhip_gemm_fp8_tile_cfg(args[0])
                label = f"fluxhip_gemm_fp8/cfg{code & 255}" + (f"s{code >> 8}" if (code >> 8) > 1 else "")
            elif fn.__name__ == "fluxhip_gemm_fp8_mx":
                d = args[0]._obj
                flops = 2.0 * sum(d.g[i].M for i in range(d.ngroups)) * d.nbatch * d.N * d.K
                label = "fluxhip_gemm_fp8_mx/" + ("A-block-scaled" if args[2]._obj.a_mx else "gelu->e4m3+scales")
                if with_shape:
                    label += f" N{d.N} K{d.K}"
            elif fn.__name__ == "fluxhip_attention_d128_bf16":
                Bq, Hq, Tq = args[5], args[6], args[7]
                flops = 4.0 * Bq * Hq * Tq * Tq * 128
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = fn(*args, stream.cuda_stream)
            e1.record(stream)
            if rc != 0:
                raise FluxHipError(f"{fn.__name__} failed with code {rc}")
            recs.append((label, e0, e1, flops, nbytes))
        torch.cuda.synchronize()
        if with_bytes:
            return [(l, e0.elapsed_time(e1), f, nb) for l, e0, e1, f, nb in recs]
        return [(l, e0.elapsed_time(e1), f) for l, e0, e1, f, nb in recs]

    def run_plan(self, ws: dict, skip_mod: bool = False) -> None:
        """Enqueue the launches of one forward on the current stream.  skip_mod: ws["mods"] already holds this step's
        modulation table (modulation_tables), so the launches before the plan's "mod_end" marker are left out."""
        cur = torch.cuda.current_stream()
        stream = cur.cuda_stream
        skipping = skip_mod
        for fn, args in ws["plan"]:
            if fn == "mod_end":
                skipping = False
                continue
            if skipping or fn == "keepalive":
                continue
            if fn == "py":          # debug_mx_shift only: a torch op on the current stream
                args()
                continue
            if fn == "side":        # fork: launch on the side stream, ordered after everything enqueued so far
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                self._side.wait_stream(cur)
                fn, args = args
                rc = fn(*args, self._side.cuda_stream)
            elif fn == "join":
                if not skip_mod:
                    cur.wait_stream(self._side)
                continue
            else:
                rc = fn(*args, stream)
            if rc != 0:
                raise FluxHipError(f"{fn.__name__} failed with code {rc}")

    def modulation_tables(self, timesteps, y: torch.Tensor, guidance: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The modulation tables (every Modulation.lin(silu(vec)) and the LastLayer adaLN, flux/layers.py:134-137,
        298-300) of SEVERAL denoise steps at once: [n_steps, B, mod_rows].

        vec = time_in(temb(t)) [+ guidance_in(temb(g))] + vector_in(y) (flux/model.py:113-120) does not depend on the
        latents, so all steps' tables are known before the loop starts.  The 6.5 GB of modulation weights are then
        streamed from HBM once per group of 4 / B steps instead of once per step (the GEMV kernel stays HBM-bound with
        up to 4 activation rows in registers per weight row; its 16-row variant is VALU-bound), and the per-step forward
        starts at the first block.  Every row is computed by the same kernel code with the same accumulation order as
        the in-line launches of the plan, so the tables are bit-identical to what `__call__` computes step by step."""
        lib = _lib.load()
        P, D, NM = self.params, self.params.hidden_size, self.mod_rows
        if P.guidance_embed and guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")
        B = y.shape[0]
        ts = [float(t) for t in timesteps]
        out = torch.empty(len(ts), B, NM, dtype=BF16, device=self.device)
        if B > 16:
            raise ValueError("modulation_tables: batch > 16 (use the in-line path)")
        stream = torch.cuda.current_stream().cuda_stream
        Wt = self._params
        w = lambda n: Wt[n].data_ptr()                                  # noqa: E731
        wo = lambda n: Wt[n].data_ptr() if n in Wt else None           # noqa: E731

        def chk(rc, name):
            if rc != 0:
                raise FluxHipError(f"{name} failed with code {rc}")

        per = max(1, 4 // B)                                            # steps per pass over the modulation weights
        for k0 in range(0, len(ts), per):
            n = min(per, len(ts) - k0)
            R = n * B                                                   # row r = step (k0 + r // B), image r % B
            tkey = (tuple(ts[k0:k0 + n]), B)
            t_all = self._t_cache.get(tkey)          # schedules repeat from image to image: no host->device copy then
            if t_all is None:
                t_all = torch.tensor(tkey[0], dtype=torch.float32, device=self.device).to(BF16).repeat_interleave(B).contiguous()
                self._t_cache[tkey] = t_all
                while len(self._t_cache) > 64:
                    self._t_cache.popitem(last=False)
            else:
                self._t_cache.move_to_end(tkey)
            y_all = y.to(BF16).repeat(n, 1).contiguous()
            temb = torch.empty(R, 256, dtype=BF16, device=self.device)
            h1 = torch.empty(R, D, dtype=BF16, device=self.device)
            vec = torch.empty(R, D, dtype=BF16, device=self.device)

            def small(x, wn, o, K, N, silu_in, accum):
                chk(lib.fluxhip_small_linear_bf16(x.data_ptr(), w(wn + ".weight"), wo(wn + ".bias"), o.data_ptr(), R, N, K,
                                                  silu_in, accum, stream), "fluxhip_small_linear_bf16")

            chk(lib.fluxhip_timestep_embedding_bf16(t_all.data_ptr(), temb.data_ptr(), R, 256, 1000.0, 10000.0, stream),
                "fluxhip_timestep_embedding_bf16")
            small(temb, "time_in.in_layer", h1, 256, D, 0, 0)
            small(h1, "time_in.out_layer", vec, D, D, 1, 0)
            if P.guidance_embed:
                g_all = guidance.to(BF16).repeat(n).contiguous()
                chk(lib.fluxhip_timestep_embedding_bf16(g_all.data_ptr(), temb.data_ptr(), R, 256, 1000.0, 10000.0, stream),
                    "fluxhip_timestep_embedding_bf16")
                small(temb, "guidance_in.in_layer", h1, 256, D, 0, 0)
                small(h1, "guidance_in.out_layer", vec, D, D, 1, 1)
            small(y_all, "vector_in.in_layer", h1, P.vec_in_dim, D, 0, 0)
            small(h1, "vector_in.out_layer", vec, D, D, 1, 1)
            # silu(vec) once, outside the weight-streaming loop (in-line, one step: silu_in = 1 on the fly; same bits)
            chk(lib.fluxhip_silu_bf16(vec.data_ptr(), h1.data_ptr(), R * D, stream), "fluxhip_silu_bf16")
            chk(lib.fluxhip_small_linear_bf16(h1.data_ptr(), self.mod_w.data_ptr(), self.mod_b.data_ptr(),
                                              out[k0:k0 + n].data_ptr(), R, NM, D, 0, 0, stream), "fluxhip_small_linear_bf16")
        return out

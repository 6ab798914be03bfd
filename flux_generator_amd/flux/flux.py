"""FluxPipeline — the reference's generator surface (flux/flux.py:22-193) over the HIP hot path.

Kept verbatim: ``FluxPipeline(name, t5_padding=True)``, ``generate_latents`` (a generator whose first
yield is the conditioning tuple and whose next ``num_steps`` yields are x_t), ``decode``,
``generate_images``, ``tokenize``, ``ensure_models_are_loaded``, ``reload_text_encoders``.
Added for the north-star wording: ``FluxPipeline(model="schnell")`` alias and ``.generate()``.
Training / LoRA methods are out of the hot-path scope.

Execution: each denoise step = one launch plan of libfluxhip kernels (flux/model.py) + one Euler
kernel.  With ``use_graph=True`` (default) the plan of a given (B, S, L) shape is captured once into
a hipGraph and replayed per step, which removes ~350 host launches per step from the critical path.
Eager torch has no lazy evaluation: every ``next()`` returns an enqueued step; the caller
synchronises when it reads (the analogue of the reference's ``mx.eval``).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Optional, Tuple

import torch

from .. import _lib, ops, parallel
from .sampler import FluxSampler
from .utils import load_ae, load_clip, load_clip_tokenizer, load_flow_model, load_t5, load_t5_tokenizer

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, **kw):
        return x


class FluxPipeline:
    MAX_GRAPHS = 8      # captured hipGraphs kept per pipeline (LRU); each owns a private memory pool

    def __init__(self, name: Optional[str] = None, t5_padding: bool = True, model: Optional[str] = None,
                 device: str = "cuda", use_graph: bool = True, broadcast_weights: Optional[bool] = None):
        name = name if name is not None else model
        if name is None:
            raise ValueError("FluxPipeline needs a model name ('flux-schnell' or 'flux-dev')")
        if not name.startswith("flux-"):
            name = "flux-" + name
        self.dtype = torch.bfloat16
        self.name = name
        self.t5_padding = t5_padding
        self.device = _lib.bind_device(device)
        self.use_graph = use_graph

        # multi-GPU: broadcast_weights=True (or FLUX_BROADCAST_WEIGHTS=1) -> only rank 0 reads the checkpoints, the other
        # ranks receive flow + AE weights over RCCL/xGMI; default: every rank reads its own copy
        if broadcast_weights is None:
            broadcast_weights = os.environ.get("FLUX_BROADCAST_WEIGHTS", "0") == "1"
        self.ae = load_ae(name, device=device, from_rank0=broadcast_weights)
        self.flow = load_flow_model(name, device=device, from_rank0=broadcast_weights)
        # T5-XXL / CLIP (~10 GB) are built on first use: under torchrun only rank 0 ever evaluates them
        # (parallel.shard_generation_inputs), so the other ranks never allocate or load them
        self._t5 = self._clip = None
        self.text_fp8 = False        # quantize_text(): the towers are built lazily, the flag is applied when they are
        self.clip_tokenizer = load_clip_tokenizer(name)
        self.t5_tokenizer = load_t5_tokenizer(name)
        self.sampler = FluxSampler(name)
        self._graphs = OrderedDict()
        self._graph_epoch = 0

    # ------------------------------------------------------------------ text encoders (lazy; rank 0 only under torchrun)
    @property
    def t5(self):
        if self._t5 is None:
            self._t5 = load_t5(self.name, device=self.device).enable_fp8(self.text_fp8)
        return self._t5

    @t5.setter
    def t5(self, m):
        self._t5 = m

    @property
    def clip(self):
        if self._clip is None:
            self._clip = load_clip(self.name, device=self.device).enable_fp8(self.text_fp8)
        return self._clip

    @clip.setter
    def clip(self, m):
        self._clip = m

    def _encodes_text(self) -> bool:
        return not parallel.active() or parallel.world()[0] == 0

    def ensure_models_are_loaded(self):
        if self._encodes_text():
            self.t5, self.clip      # noqa: B018  (build them now)
        torch.cuda.synchronize(self.device)

    def quantize_text(self, enabled: bool = True) -> None:
        """`--quantize` for the text towers (txt2image.py:79-82 quantises flow, t5 and clip): fp8 Linears in T5 (all but the
        value projection) and in CLIP (the layers' second MLP Linear: the reference's in_dim % 512 predicate).  Applies to
        towers already built and to the ones built later (they are lazy: rank 0 only under torchrun)."""
        self.text_fp8 = bool(enabled)
        for m in (self._t5, self._clip):
            if m is not None:
                m.enable_fp8(self.text_fp8)

    def reload_text_encoders(self):
        if self._encodes_text():
            self._t5 = load_t5(self.name, device=self.device).enable_fp8(self.text_fp8)
            self._clip = load_clip(self.name, device=self.device).enable_fp8(self.text_fp8)

    # ------------------------------------------------------------------ captured graphs: bounded LRU
    def _graph_get(self, key):
        ent = self._graphs.get(key)
        if ent is not None:
            self._graphs.move_to_end(key)
        return ent

    def _graph_put(self, key, ent):
        """Insert a captured graph; the least recently used ones beyond MAX_GRAPHS are dropped TOGETHER with the flow
        model's workspace of their shape (a hipGraph pins its private pool and the launch plan's buffers: a long-lived
        server that sees many image sizes must give both back)."""
        self._graphs[key] = ent
        while len(self._graphs) > self.MAX_GRAPHS:
            old, _ = self._graphs.popitem(last=False)
            if old[0] != "decode":
                self.flow.release_workspace(*old[:3])

    def tokenize(self, text):
        t5_tokens = self.t5_tokenizer.encode(text, pad=self.t5_padding)
        if t5_tokens.shape[1] % 4:      # kernels want T % 4 == 0: only reachable with --no-t5-padding
            fill = max(getattr(self.t5_tokenizer, "pad_token", 0), 0)
            extra = 4 - t5_tokens.shape[1] % 4
            t5_tokens = torch.nn.functional.pad(t5_tokens, (0, extra), value=fill)
        clip_tokens = self.clip_tokenizer.encode(text)
        return t5_tokens, clip_tokens

    def _prepare_latent_images(self, x: torch.Tensor):
        """flux/flux.py:53-71: 2x2 pack (HIP kernel) + (0,row,col) position ids."""
        b, h, w, c = x.shape
        packed = ops.pack_latents(x.contiguous())
        j, k = torch.meshgrid(torch.arange(h // 2, dtype=torch.int32, device=x.device),
                              torch.arange(w // 2, dtype=torch.int32, device=x.device), indexing="ij")
        x_ids = torch.stack([torch.zeros_like(j), j, k], dim=-1).reshape(1, h * w // 4, 3).repeat(b, 1, 1)
        return packed, x_ids.contiguous()

    def _prepare_conditioning(self, n_images, t5_tokens, clip_tokens):
        """flux/flux.py:73-85."""
        txt = self.t5(t5_tokens)
        if len(txt) == 1 and n_images > 1:
            txt = txt.expand(n_images, *txt.shape[1:]).contiguous()
        txt_ids = torch.zeros((n_images, txt.shape[1], 3), dtype=torch.int32, device=txt.device)
        vec = self.clip(clip_tokens).pooled_output
        if len(vec) == 1 and n_images > 1:
            vec = vec.expand(n_images, *vec.shape[1:]).contiguous()
        return txt, txt_ids, vec

    # ------------------------------------------------------------------ denoise step execution
    def _flow_step(self, x_t, x_ids, txt, txt_ids, vec, t_vec, guidance, mods: Optional[torch.Tensor] = None) -> torch.Tensor:
        """pred = flow(...) — eagerly, or by replaying the captured hipGraph of this shape.  mods: this step's
        modulation table [B, mod_rows] from `Flux.modulation_tables` (graph path only); the replayed graph then holds the
        forward without the vec / modulation launches."""
        if not self.use_graph:
            return self.flow(img=x_t, img_ids=x_ids, txt=txt, txt_ids=txt_ids, y=vec, timesteps=t_vec, guidance=guidance)
        B, L, _ = x_t.shape
        S = txt.shape[1]
        ws = self.flow._workspace(B, S, L)
        ws["in_img"].copy_(x_t)
        ws["in_txt"].copy_(txt)
        ws["in_y"].copy_(vec)
        ws["in_t"].copy_(t_vec)
        ws["in_g"].copy_(guidance)
        if S > 0:
            ws["in_ids"][:, :S].copy_(txt_ids)
        ws["in_ids"][:, S:].copy_(x_ids)
        skip_mod = mods is not None
        if skip_mod:
            ws["mods"].copy_(mods)
        if self._graph_epoch != self.flow.plan_epoch:       # the flow model rebuilt its plans (enable_fp8): old graphs are stale
            self._graphs = OrderedDict((k, v) for k, v in self._graphs.items() if k[0] == "decode")
            self._graph_epoch = self.flow.plan_epoch      # (enable_fp8 cleared the flow model's workspaces, pins included)
        key = (B, S, L, "premod") if skip_mod else (B, S, L, "full")
        g = self._graph_get(key)
        if g is None:
            # warm up once on a side stream (first-touch attribute calls), then capture
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.flow.run_plan(ws, skip_mod=skip_mod)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.flow.run_plan(ws, skip_mod=skip_mod)
            self.flow.pin_workspace(B, S, L)
            self._graph_put(key, g)
        g.replay()
        return ws["pred"]

    def _denoising_loop(self, x_t, x_ids, txt, txt_ids, vec, num_steps: int = 35, guidance: float = 4.0,
                        start: float = 1, stop: float = 0):
        """flux/flux.py:87-126.  On the graph path at batch <= 2 the modulation tables of all steps are computed up
        front, 4 / B steps per pass over the modulation weights (they depend on (t, vec, guidance) only, never on the
        latents): bit-identical to computing them inside every step, 6.5 GB less HBM traffic per further step, and no
        GEMV sharing the CUs with the first blocks' single-round GEMMs.  At larger batch the GEMV cannot cover several
        steps per pass and stays in line in the step graph."""
        B = len(x_t)

        def scalar(x):
            return torch.full((B,), x, dtype=self.dtype, device=x_t.device)

        guidance = scalar(guidance)
        timesteps = self.sampler.timesteps(num_steps, x_t.shape[1], start=start, stop=stop)
        mods = None
        if self.use_graph and 0 < B <= 2 and num_steps > 1:
            # (round 6: these launches on a side stream under the PREVIOUS image's VAE decode - HBM-bound GEMV beside MFMA-bound
            #  convs - measured 0.2 ms per image SLOWER, profiles/r06_negative_experiments.json; they stay in line)
            mods = self.flow.modulation_tables(timesteps[:num_steps], vec, guidance)
        for i in range(num_steps):
            t, t_prev = timesteps[i], timesteps[i + 1]
            pred = self._flow_step(x_t, x_ids, txt, txt_ids, vec, scalar(t), guidance, None if mods is None else mods[i])
            x_t = self.sampler.step(pred, x_t, t, t_prev)
            yield x_t

    def generate_latents(self, text: str, n_images: int = 1, num_steps: int = 35, guidance: float = 4.0,
                         latent_size: Tuple[int, int] = (64, 64), seed=None):
        """flux/flux.py:128-155.  Under torch.distributed (one process per GPU, launched by torchrun) the batch is
        sharded by image: rank 0 alone runs T5 / CLIP and broadcasts txt / vec over RCCL, every rank
        draws the full x_T from the job's seed and keeps its rows, and all yields are the LOCAL images
        (`self.shard` = their [lo, hi) range in the batch; gather with `gather_images`)."""
        if parallel.active():
            def cond():
                t5_tokens, clip_tokens = self.tokenize(text)
                txt, _, vec = self._prepare_conditioning(1 if isinstance(text, str) else n_images, t5_tokens, clip_tokens)
                return txt, vec
            x_T, txt, vec, self.shard = parallel.shard_generation_inputs(n_images, (*latent_size, 16), seed, self.device,
                                                                         cond, dtype=self.dtype)
            n_local = x_T.shape[0]
            txt_ids = torch.zeros((n_local, txt.shape[1], 3), dtype=torch.int32, device=self.device)
            if n_local == 0:             # more ranks than images: this rank only takes part in the collectives
                empty = torch.empty(0, latent_size[0] * latent_size[1] // 4, 64, dtype=self.dtype, device=self.device)
                yield (empty, torch.empty(0, empty.shape[1], 3, dtype=torch.int32, device=self.device), txt, txt_ids, vec)
                for _ in range(num_steps):
                    yield empty
                return
            x_T, x_ids = self._prepare_latent_images(x_T)
            yield (x_T, x_ids, txt, txt_ids, vec)
            yield from self._denoising_loop(x_T, x_ids, txt, txt_ids, vec, num_steps=num_steps, guidance=guidance)
            return
        self.shard = (0, n_images)
        gen = None
        if seed is not None:
            gen = torch.Generator(device=self.device).manual_seed(seed)
        x_T = self.sampler.sample_prior((n_images, *latent_size, 16), dtype=self.dtype, key=gen, device=self.device)
        x_T, x_ids = self._prepare_latent_images(x_T)
        t5_tokens, clip_tokens = self.tokenize(text)
        txt, txt_ids, vec = self._prepare_conditioning(n_images, t5_tokens, clip_tokens)
        yield (x_T, x_ids, txt, txt_ids, vec)
        yield from self._denoising_loop(x_T, x_ids, txt, txt_ids, vec, num_steps=num_steps, guidance=guidance)

    def gather_images(self, images: torch.Tensor, n_images: int):
        """Decoded float images [n_local,H,W,3] of this rank -> uint8 (truncating, txt2image.py:133) -> gathered to
        rank 0 in batch order over RCCL.  Returns uint8 [n_images,H,W,3] on rank 0, None elsewhere; with one process
        it is just the uint8 conversion."""
        return parallel.gather_images(parallel.to_uint8(images).contiguous(), n_images)

    def decode(self, x: torch.Tensor, latent_size: Tuple[int, int] = (64, 64), precision: Optional[str] = None) -> torch.Tensor:
        """flux/flux.py:157-162: [b,L,64] -> [b,8h,8w,3] float in [0,1] (unpack, VAE decode, clip fused).
        The decode runs in the reference's float32 arithmetic (AutoEncoder precision "fp32", the default) unless
        precision="bf16" is asked for.  With use_graph the ~200 launches of a decode are captured once per shape."""
        precision = precision or self.ae.precision
        if not self.use_graph:
            return self.ae.decode_packed(x, latent_size, precision)
        key = ("decode", x.shape[0], tuple(latent_size), precision, self.ae._store.epoch)
        ent = self._graph_get(key)
        if ent is None:
            for k in [k for k in self._graphs if k[0] == "decode" and k[4] != key[4]]:
                del self._graphs[k]           # decode graphs of an older parameter epoch can never be replayed again
            static_in = x.to(self.dtype).contiguous().clone()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.ae.decode_packed(static_in, latent_size, precision)      # warm-up (attribute calls, workspaces)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = self.ae.decode_packed(static_in, latent_size, precision)
            ent = (g, static_in, static_out)
            self._graph_put(key, ent)
        g, static_in, static_out = ent
        static_in.copy_(x)
        g.replay()
        return static_out.clone()

    def generate_images(self, text: str, n_images: int = 1, num_steps: int = 35, guidance: float = 4.0,
                        latent_size: Tuple[int, int] = (64, 64), seed=None, reload_text_encoders: bool = True,
                        progress: bool = True):
        """flux/flux.py:164-193."""
        latents = self.generate_latents(text, n_images, num_steps, guidance, latent_size, seed)
        next(latents)
        if reload_text_encoders:
            self.reload_text_encoders()
        x_t = None
        for x_t in tqdm(latents, total=num_steps, disable=not progress, leave=True):
            pass
        images = []
        for i in tqdm(range(len(x_t)), disable=not progress, desc="generate images"):
            images.append(self.decode(x_t[i:i + 1], latent_size))
        images = (torch.cat(images, dim=0) if images else
                  torch.empty(0, latent_size[0] * 8, latent_size[1] * 8, 3, device=self.device))
        torch.cuda.synchronize(self.device)
        return images        # under torchrun: this rank's images (self.shard); gather_images() collects them on rank 0

    def load_adapter(self, adapter_file: str, fuse: bool = False) -> int:
        """txt2image.py:30-37 (`load_adapter`): read a LoRA adapter saved by the reference's dreambooth.py (safetensors of
        `<layer>.lora_a` / `.lora_b` with metadata lora_rank / lora_blocks) and apply it to the flow model.  fuse=True
        (`--fuse-adapter`): W <- W + (scale B^T A^T).astype(bf16), one GEMM per layer at load time, none per step
        (LoRALinear.fuse, flux/lora.py:28-43).  fuse=False: the reference keeps LoRALinear layers, and so does this - the
        low-rank branch stays separate and is added before the layer's activation / gate (`Flux.attach_lora`), so an update
        below half a bf16 ulp of W is not rounded away."""
        from safetensors import safe_open
        weights = {}
        with safe_open(adapter_file, framework="pt") as f:
            meta = f.metadata() or {}
            for k in f.keys():
                weights[k] = f.get_tensor(k)
        rank = int(meta.get("lora_rank", 0))
        for k, v in weights.items():
            if k.endswith(".lora_a") and rank and v.shape[1] != rank:
                raise ValueError(f"{k}: rank {v.shape[1]} does not match the adapter's lora_rank {rank}")
        if fuse or self.flow.fp8:
            # in place on the weight tensors: captured hipGraphs keep pointing at the (now updated) weights.  An adapter loaded
            # AFTER --quantize is folded as well (the fp8 plan carries no separate branch; Flux.enable_fp8 says the same).
            if not fuse:
                import warnings
                warnings.warn("load_adapter: the flow model runs the fp8 plan - the adapter is folded into the weights")
            n = self.flow.fuse_lora(weights, scale=1.0)         # LoRALinear.from_base default scale (flux/lora.py:15)
            self.adapter_layers = dict(branches=0, folded=n)
            return n
        # The reference wraps EVERY nn.Linear of a block (linear_to_lora_layers, flux/flux.py:229-239), the modulation Linears
        # included, and dreambooth.py saves them all.  Block Linears keep their separate low-rank branch; the modulation
        # Linears are rows of the one concatenated GEMV table evaluated once per image, so their update is folded into the
        # table (bf16, like LoRALinear.fuse) - the only deviation from the unfused reference, of the size of one bf16 rounding
        # of those weights.
        if self.flow._lora:
            # a second unfused adapter: `attach_lora` REPLACES the block branches while the first adapter's modulation deltas
            # are already folded into the table - the two would be mixed silently.  The first adapter's branches are folded
            # into the weights first (what its modulation half already is), so both adapters apply in full.
            import warnings
            warnings.warn(f"load_adapter: folding the {len(self.flow._lora)} branches of the adapter already attached before "
                          "attaching the next one")
            self.flow.fuse_attached_lora()
        branch, fold = self.flow.splits_for_adapter(weights)
        n = self.flow.fuse_lora(fold, scale=1.0) if fold else 0
        self.adapter_layers = dict(branches=len(branch) // 2, folded=n)
        return n + self.flow.attach_lora(branch, scale=1.0)     # (plan_epoch bump: stale step graphs are dropped)

    def generate(self, *args, **kwargs):
        """Alias of generate_images (BASELINE.json north_star wording)."""
        return self.generate_images(*args, **kwargs)

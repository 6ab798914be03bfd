"""StableDiffusion / StableDiffusionXL pipelines over the HIP hot path (mirror of the reference's
stable_diffusion/stable_diffusion/__init__.py:19-306): same class names, constructor arguments and
generator methods (``generate_latents`` yields x_t per step — no conditioning yield —, ``decode``).
image2image / VAE encoder are out of the hot-path scope.

Multi-GPU: under torch.distributed (one process per GPU) ``generate_latents`` shards the batch by image
exactly like FluxPipeline — job seed from rank 0, text towers evaluated on rank 0 ONLY and their outputs broadcast over
RCCL, the prior AND the ancestral sampler's per-step noise drawn for the full batch on every rank and sliced, yields are the
LOCAL images (``self.shard``), ``gather_images`` collects uint8 images on rank 0.  No collective inside a step."""
from __future__ import annotations

import time
from collections import OrderedDict
from typing import Optional, Tuple

import torch

from .. import _lib, ops, parallel
from .model_io import (_DEFAULT_MODEL, load_autoencoder, load_diffusion_config, load_text_encoder, load_tokenizer,
                       load_unet)
from .sampler import SimpleEulerAncestralSampler, SimpleEulerSampler


class _LazyTower:
    """A text tower built on first use: under torchrun only rank 0 ever evaluates it, the other ranks never allocate it."""

    def __init__(self, load):
        self._load, self._m = load, None

    def get(self):
        if self._m is None:
            self._m = self._load()
        return self._m


class StableDiffusion:
    MAX_GRAPHS = 6      # captured hipGraphs kept (LRU): each owns a private pool of ~1-6 GB at batch 16

    def __init__(self, model: str = _DEFAULT_MODEL, float16: bool = False, device: str = "cuda", use_graph: bool = True,
                 storage: Optional[str] = None):
        # float16=True: the reference's float16 arithmetic (__init__.py:20-27; flux_app.py:77-79 passes it) - UNet and text
        # towers store IEEE half and multiply on v_mfma_f32_16x16x32_f16 with fp32 accumulate / norms / softmax.
        # float16=False (the reference's DEFAULT): its FLOAT32 UNet / CLIP (__init__.py:18-23).  Round 6: that arithmetic is built -
        # float32 master parameters, every Linear / conv on the float32-faithful split-bf16 kernels the VAE decoders use (three
        # MFMA passes over hi / lo planes, float32 accumulation), float32 norms / activations / softmax (unet_f32.py,
        # flux/clip.py `_call_f32`), float32 latents and sampler.  It is the arithmetic path, not a tuned one (3x the MFMA work).
        # storage="bfloat16" (or FLUXHIP_SD_STORAGE=bfloat16) remains as the explicit, NAMED opt-in to bf16 storage: same kernels
        # as float16, float32 range, 8-bit significand (rel-L2 7e-3 against float32 on the full-size UNet where float16
        # measures ~1e-3).  Nothing narrower than what the caller asked for runs silently.  The VAE decode is float32-faithful
        # in every mode.
        import os
        if float16:
            # an explicit float16=True wins over the FLUXHIP_SD_STORAGE environment default (flux_app.py always passes it: the
            # server must keep building its pipelines with that variable exported); only an explicit storage= can contradict it
            if storage not in (None, "float16"):
                raise ValueError(f"float16=True stores IEEE half; storage='{storage}' contradicts it")
            self.dtype = torch.float16
        else:
            storage = storage or os.environ.get("FLUXHIP_SD_STORAGE") or "float32"
            if storage not in ("float32", "bfloat16"):
                raise ValueError(f"float16=False computes in float32 (storage='float32', the default) or, by explicit request, "
                                 f"with bfloat16 storage; got storage='{storage}'")
            self.dtype = torch.float32 if storage == "float32" else torch.bfloat16
        self.float16 = bool(float16)
        self.device = _lib.bind_device(device)
        self.use_graph = use_graph
        self._graphs = OrderedDict()
        self.shard = None
        self.diffusion_config = load_diffusion_config(model)
        self.unet = load_unet(model, float16, device=device, dtype=self.dtype)
        self._towers = {"text_encoder": _LazyTower(lambda: load_text_encoder(model, float16, device=device, dtype=self.dtype))}
        self.autoencoder = load_autoencoder(model, False, device=device)
        self.sampler = SimpleEulerSampler(self.diffusion_config)
        self._set_sampler_dtype()
        self.tokenizer = load_tokenizer(model)

    def _set_sampler_dtype(self):
        """The reference derives the step coefficients in the eps dtype (sampler.py:77-78,90-96) and rounds the timesteps to
        the pipeline dtype (__init__.py:94-96): float16 under float16=True, float32 otherwise."""
        self.sampler.coef_dtype = torch.float16 if self.float16 else torch.float32

    @property
    def text_encoder(self):
        if "text_encoder" not in self._towers:        # StableDiffusionXL renames it (reference: `del self.text_encoder`)
            raise AttributeError("text_encoder")
        return self._towers["text_encoder"].get()

    def _encodes_text(self) -> bool:
        return not parallel.active() or parallel.world()[0] == 0

    def ensure_models_are_loaded(self):
        if self._encodes_text():
            for t in self._towers.values():
                t.get()
        torch.cuda.synchronize(self.device)

    def _graph_get(self, key):
        ent = self._graphs.get(key)
        if ent is not None:
            self._graphs.move_to_end(key)
        return ent

    def _graph_put(self, key, ent):
        self._graphs[key] = ent
        while len(self._graphs) > self.MAX_GRAPHS:
            self._graphs.popitem(last=False)       # graph + its static buffers + its private pool go together

    def _tokenize(self, tokenizer, text: str, negative_text: Optional[str] = None):
        """__init__.py:34-46."""
        tokens = [tokenizer.tokenize(text)]
        if negative_text is not None:
            tokens += [tokenizer.tokenize(negative_text)]
        N = max(len(t) for t in tokens)
        return torch.tensor([t + [0] * (N - len(t)) for t in tokens], dtype=torch.int32)

    def _get_text_conditioning(self, text: str, n_images: int = 1, cfg_weight: float = 7.5, negative_text: str = ""):
        """__init__.py:48-65."""
        tokens = self._tokenize(self.tokenizer, text, (negative_text if cfg_weight > 1 else None))
        conditioning = self.text_encoder(tokens).last_hidden_state
        if n_images > 1:
            conditioning = conditioning.repeat_interleave(n_images, dim=0)
        return conditioning

    def _denoising_step(self, x_t, t, t_prev, conditioning, cfg_weight: float = 7.5, text_time=None, noise=None,
                        key: Optional[torch.Generator] = None, coef_dev: Optional[torch.Tensor] = None, shard=None,
                        new_conditioning: bool = True):
        """__init__.py:67-82.  One UNet step is ~1700 kernel launches; with use_graph they are captured ONCE per
        (shape, cfg) into a hipGraph over static input buffers: the timestep is a device tensor and the sampler's
        (ca, cb, cc) live in a float32[3] device buffer (fluxhip_axpbypcz_dev_bf16), so every step of every run
        replays the same graph.  The ancestral sampler's per-step noise is drawn from the run's seeded generator
        `key` outside the graph and copied into a static buffer.  coef_dev: this step's row of
        `sampler.coeff_table` (device float32[3]; `_denoising_loop` uploads the whole table once per run) — without it the
        coefficients are computed and uploaded here.  shard: (lo, hi, n_total) of a multi-GPU run (noise rows).
        The cross-attention K / V^T of all transformer layers depend on the conditioning only: they live in static buffers
        next to it (`UNetModel.text_kv`, two GEMMs per layer width) and are projected OUTSIDE the graph, when the
        conditioning changes (new_conditioning: `_denoising_loop` passes False after the first step of a run)."""
        if noise is None and self.sampler.needs_noise:
            noise = self.sampler.draw_noise(x_t, key, shard)
        if not self.use_graph:
            return self._denoising_step_eager(x_t, t, t_prev, conditioning, cfg_weight, text_time, noise)
        key_ = ("step", tuple(x_t.shape), tuple(conditioning.shape), float(cfg_weight), text_time is not None,
                noise is not None)
        ent = self._graph_get(key_)
        nb = len(x_t) * (2 if cfg_weight > 1 else 1)
        if ent is None:
            sx, sc = x_t.clone(), conditioning.clone()
            stt = None if text_time is None else (text_time[0].clone(), text_time[1].clone())
            sn = None if noise is None else noise.clone()
            st = torch.full((nb,), float(t), dtype=torch.float32, device=self.device)
            scoef = torch.tensor(self.sampler.coeffs(t, t_prev), dtype=torch.float32, device=self.device)
            skv = self.unet.text_kv(self.unet.pad_encoder_states(sc))
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._denoising_step_dev(sx, st, scoef, sc, cfg_weight, stt, sn, skv)      # warm-up
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._denoising_step_dev(sx, st, scoef, sc, cfg_weight, stt, sn, skv)
            ent = (g, sx, sc, stt, sn, st, scoef, out, skv, {"cond": None})
            self._graph_put(key_, ent)
            new_conditioning = True
        g, sx, sc, stt, sn, st, scoef, out, skv, seen = ent
        sx.copy_(x_t)
        # the graph entry is shared by every run of this shape on the pipeline: its static conditioning (and the K / V^T
        # projected from it) is refreshed whenever the caller's tensor is not the one last written - identity + version, so
        # two interleaved generate_latents generators (an abandoned or resumed one, concurrent requests) cannot denoise
        # with each other's prompt - not only when the caller says this is the first step of a run
        token = (conditioning.data_ptr(), conditioning._version, tuple(conditioning.shape))
        if new_conditioning or seen["cond"] != token:
            sc.copy_(conditioning)
            self.unet.text_kv(self.unet.pad_encoder_states(sc), out=skv)
            seen["cond"] = token
        if stt is not None:
            stt[0].copy_(text_time[0])
            stt[1].copy_(text_time[1])
        if sn is not None:
            sn.copy_(noise)
        st.fill_(float(t))
        if coef_dev is None:
            coef_dev = torch.tensor(self.sampler.coeffs(t, t_prev), dtype=torch.float32).to(self.device)
        scoef.copy_(coef_dev)                  # device -> device: nothing host-side between two replays
        g.replay()
        return out.clone()

    def _eps(self, x_t, t_unet, conditioning, cfg_weight, text_time, text_kv=None):
        """UNet evaluation with classifier-free guidance: CFG doubles the batch (text first, negative second)."""
        x_unet = torch.cat([x_t] * 2, dim=0) if cfg_weight > 1 else x_t
        eps = self.unet(x_unet, t_unet, encoder_x=conditioning, text_time=text_time, text_kv=text_kv)
        if cfg_weight > 1:
            eps_text, eps_neg = eps.chunk(2)
            # eps_neg + w (eps_text - eps_neg) = (1 - w) eps_neg + w eps_text
            eps = ops.axpbypcz(eps_neg.contiguous(), eps_text.contiguous(), None, 1.0 - cfg_weight, cfg_weight)
        return eps

    def _denoising_step_dev(self, x_t, t_unet, coef, conditioning, cfg_weight, text_time, noise, text_kv=None):
        return self.sampler.step_dev(self._eps(x_t, t_unet, conditioning, cfg_weight, text_time, text_kv), x_t, coef, noise)

    def _denoising_step_eager(self, x_t, t, t_prev, conditioning, cfg_weight: float = 7.5, text_time=None, noise=None):
        nb = len(x_t) * (2 if cfg_weight > 1 else 1)
        t_unet = torch.full((nb,), float(t), dtype=torch.float32, device=x_t.device)
        return self.sampler.step(self._eps(x_t, t_unet, conditioning, cfg_weight, text_time), x_t, t, t_prev, noise)

    def _denoising_loop(self, x_T, T, conditioning, num_steps: int = 50, cfg_weight: float = 7.5, text_time=None,
                        key: Optional[torch.Generator] = None, shard=None):
        """__init__.py:84-100."""
        x_t = x_T
        steps = self.sampler.timesteps(num_steps, start_time=T, dtype=torch.float16 if self.float16 else torch.float32)
        coefs = self.sampler.coeff_table(steps, self.device) if self.use_graph else None
        first = True                           # the conditioning (and its K / V^T projections) is new to the step graph once per run
        for i, (t, t_prev) in enumerate(steps):
            if len(x_t) == 0:          # more ranks than images: keep the job generator in step, nothing to compute
                if self.sampler.needs_noise:
                    self.sampler.draw_noise(x_t, key, shard)
                yield x_t
                continue
            x_t = self._denoising_step(x_t, t, t_prev, conditioning, cfg_weight, text_time, key=key,
                                       coef_dev=None if coefs is None else coefs[i], shard=shard, new_conditioning=first)
            first = False
            yield x_t

    # ------------------------------------------------------------------ multi-GPU front half
    def _job_inputs(self, n_images: int, latent_size, seed, make_conditioning):
        """seed, generator, the base conditioning tensors (as computed for ONE image, on rank 0 only, broadcast), the
        local slice of the prior and `shard` = (lo, hi, n_total) — or shard None without a process group."""
        ch = self.autoencoder.latent_channels
        if not parallel.active():
            seed = int(time.time()) if seed is None else seed
            g = torch.Generator(device=self.device).manual_seed(seed)
            self.shard = (0, n_images)
            return g, make_conditioning(), None, None
        seed = parallel.broadcast_seed(int(time.time()) if seed is None else seed, self.device)
        g = torch.Generator(device=self.device).manual_seed(seed)
        cond = parallel.broadcast_from(make_conditioning, self.device)
        rank, W = parallel.world()
        lo, hi = parallel.shard_range(n_images, rank, W)
        self.shard = (lo, hi)
        x_T = self.sampler.sample_prior((n_images, *latent_size, ch), dtype=self.dtype, key=g, device=self.device,
                                        rows=(lo, hi))
        return g, cond, x_T, (lo, hi, n_images)

    def gather_images(self, images: torch.Tensor, n_images: int):
        """Decoded float images [n_local,H,W,3] of this rank -> uint8 (truncating, like the reference's CLI) -> gathered to
        rank 0 in batch order over RCCL (None on the other ranks); with one process it is just the conversion."""
        return parallel.gather_images(parallel.to_uint8(images).contiguous(), n_images)

    def generate_latents(self, text: str, n_images: int = 1, num_steps: int = 50, cfg_weight: float = 7.5,
                         negative_text: str = "", latent_size: Tuple[int, int] = (64, 64), seed=None):
        """__init__.py:102-129."""
        g, cond, x_T, shard = self._job_inputs(n_images, latent_size, seed,
                                               lambda: [self._get_text_conditioning(text, 1, cfg_weight, negative_text)])
        n_local = n_images if shard is None else shard[1] - shard[0]
        conditioning = cond[0].repeat_interleave(n_local, dim=0) if n_local != 1 else cond[0]
        if x_T is None:
            x_T = self.sampler.sample_prior((n_images, *latent_size, self.autoencoder.latent_channels), dtype=self.dtype,
                                            key=g, device=self.device)
        yield from self._denoising_loop(x_T, self.sampler.max_time, conditioning, num_steps, cfg_weight, key=g, shard=shard)

    def decode(self, x_t, precision: Optional[str] = None):
        """__init__.py:166-169: clip(vae.decode(x_t) / 2 + 0.5, 0, 1), fused into the last conv.  float32 arithmetic
        like the reference's VAE (load_autoencoder(model, False), __init__.py:25) unless precision="bf16"."""
        precision = precision or self.autoencoder.precision
        if not self.use_graph:
            return self.autoencoder.decode_image(x_t, precision)
        key = ("decode", tuple(x_t.shape), precision, self.autoencoder._store.epoch)
        ent = self._graph_get(key)
        if ent is None:
            for k in [k for k in self._graphs if k[0] == "decode" and k[3] != key[3]]:
                del self._graphs[k]           # decode graphs of an older parameter epoch can never be replayed again
            sx = x_t.to(self.dtype).contiguous().clone()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.autoencoder.decode_image(sx, precision)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self.autoencoder.decode_image(sx, precision)
            ent = (g, sx, out)
            self._graph_put(key, ent)
        g, sx, out = ent
        sx.copy_(x_t)
        g.replay()
        return out.clone()


class StableDiffusionXL(StableDiffusion):
    def __init__(self, model: str = _DEFAULT_MODEL, float16: bool = False, device: str = "cuda", use_graph: bool = True,
                 storage: Optional[str] = None):
        super().__init__(model, float16, device, use_graph, storage)
        self.sampler = SimpleEulerAncestralSampler(self.diffusion_config)
        self._set_sampler_dtype()
        self._towers = {"text_encoder_1": self._towers["text_encoder"],
                        "text_encoder_2": _LazyTower(lambda: load_text_encoder(model, float16, model_key="text_encoder_2",
                                                                               device=device, dtype=self.dtype))}
        self.tokenizer_1 = self.tokenizer
        del self.tokenizer
        self.tokenizer_2 = load_tokenizer(model, merges_key="tokenizer_2_merges", vocab_key="tokenizer_2_vocab")

    @property
    def text_encoder_1(self):
        return self._towers["text_encoder_1"].get()

    @property
    def text_encoder_2(self):
        return self._towers["text_encoder_2"].get()

    def _get_text_conditioning(self, text: str, n_images: int = 1, cfg_weight: float = 7.5, negative_text: str = ""):
        """__init__.py:206-229."""
        neg = negative_text if cfg_weight > 1 else None
        c1 = self.text_encoder_1(self._tokenize(self.tokenizer_1, text, neg))
        c2 = self.text_encoder_2(self._tokenize(self.tokenizer_2, text, neg))
        conditioning = torch.cat([c1.hidden_states[-2], c2.hidden_states[-2]], dim=-1)
        pooled = c2.pooled_output
        if n_images > 1:
            conditioning = conditioning.repeat_interleave(n_images, dim=0)
            pooled = pooled.repeat_interleave(n_images, dim=0)
        return conditioning, pooled

    def generate_latents(self, text: str, n_images: int = 1, num_steps: int = 2, cfg_weight: float = 0.0,
                         negative_text: str = "", latent_size: Tuple[int, int] = (64, 64), seed=None):
        """__init__.py:231-267."""
        g, cond, x_T, shard = self._job_inputs(n_images, latent_size, seed,
                                               lambda: list(self._get_text_conditioning(text, 1, cfg_weight, negative_text)))
        n_local = n_images if shard is None else shard[1] - shard[0]
        conditioning, pooled = cond
        if n_local != 1:
            conditioning = conditioning.repeat_interleave(n_local, dim=0)
            pooled = pooled.repeat_interleave(n_local, dim=0)
        time_ids = torch.tensor([[512, 512, 0, 0, 512, 512.0]] * len(pooled), device=self.device)
        text_time = (pooled, time_ids)
        if x_T is None:
            x_T = self.sampler.sample_prior((n_images, *latent_size, self.autoencoder.latent_channels), dtype=self.dtype,
                                            key=g, device=self.device)
        yield from self._denoising_loop(x_T, self.sampler.max_time, conditioning, num_steps, cfg_weight,
                                        text_time=text_time, key=g, shard=shard)

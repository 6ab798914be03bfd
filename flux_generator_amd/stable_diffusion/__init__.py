"""StableDiffusion / StableDiffusionXL pipelines over the HIP hot path (mirror of the reference's
stable_diffusion/stable_diffusion/__init__.py:19-306): same class names, constructor arguments and
generator methods (``generate_latents`` yields x_t per step — no conditioning yield —, ``decode``).
image2image / VAE encoder are out of the hot-path scope."""
from __future__ import annotations

import time
from typing import Optional, Tuple

import torch

from .. import _lib, ops
from .model_io import (_DEFAULT_MODEL, load_autoencoder, load_diffusion_config, load_text_encoder, load_tokenizer,
                       load_unet)
from .sampler import SimpleEulerAncestralSampler, SimpleEulerSampler


class StableDiffusion:
    def __init__(self, model: str = _DEFAULT_MODEL, float16: bool = False, device: str = "cuda", use_graph: bool = True):
        # the HIP path computes in bf16 storage / fp32 accumulate whatever `float16` says
        self.dtype = torch.bfloat16
        self.device = _lib.bind_device(device)
        self.use_graph = use_graph
        self._graphs = {}
        self.diffusion_config = load_diffusion_config(model)
        self.unet = load_unet(model, float16, device=device)
        self.text_encoder = load_text_encoder(model, float16, device=device)
        self.autoencoder = load_autoencoder(model, False, device=device)
        self.sampler = SimpleEulerSampler(self.diffusion_config)
        self.tokenizer = load_tokenizer(model)

    def ensure_models_are_loaded(self):
        torch.cuda.synchronize(self.device)

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
                        key: Optional[torch.Generator] = None):
        """__init__.py:67-82.  One UNet step is ~1700 kernel launches; with use_graph they are captured ONCE per
        (shape, cfg) into a hipGraph over static input buffers: the timestep is a device tensor and the sampler's
        (ca, cb, cc) live in a float32[3] device buffer (fluxhip_axpbypcz_dev_bf16), so every step of every run
        replays the same graph.  The ancestral sampler's per-step noise is drawn from the run's seeded generator
        `key` outside the graph and copied into a static buffer."""
        if noise is None and self.sampler.needs_noise:
            noise = self.sampler.draw_noise(x_t, key)
        if not self.use_graph:
            return self._denoising_step_eager(x_t, t, t_prev, conditioning, cfg_weight, text_time, noise)
        key_ = ("step", tuple(x_t.shape), tuple(conditioning.shape), float(cfg_weight), text_time is not None,
                noise is not None)
        ent = self._graphs.get(key_)
        nb = len(x_t) * (2 if cfg_weight > 1 else 1)
        if ent is None:
            sx, sc = x_t.clone(), conditioning.clone()
            stt = None if text_time is None else (text_time[0].clone(), text_time[1].clone())
            sn = None if noise is None else noise.clone()
            st = torch.full((nb,), float(t), dtype=torch.float32, device=self.device)
            scoef = torch.tensor(self.sampler.coeffs(t, t_prev), dtype=torch.float32, device=self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._denoising_step_dev(sx, st, scoef, sc, cfg_weight, stt, sn)      # warm-up
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._denoising_step_dev(sx, st, scoef, sc, cfg_weight, stt, sn)
            ent = (g, sx, sc, stt, sn, st, scoef, out)
            self._graphs[key_] = ent
        g, sx, sc, stt, sn, st, scoef, out = ent
        sx.copy_(x_t)
        sc.copy_(conditioning)
        if stt is not None:
            stt[0].copy_(text_time[0])
            stt[1].copy_(text_time[1])
        if sn is not None:
            sn.copy_(noise)
        st.fill_(float(t))
        scoef.copy_(torch.tensor(self.sampler.coeffs(t, t_prev), dtype=torch.float32), non_blocking=False)
        g.replay()
        return out.clone()

    def _eps(self, x_t, t_unet, conditioning, cfg_weight, text_time):
        """UNet evaluation with classifier-free guidance: CFG doubles the batch (text first, negative second)."""
        x_unet = torch.cat([x_t] * 2, dim=0) if cfg_weight > 1 else x_t
        eps = self.unet(x_unet, t_unet, encoder_x=conditioning, text_time=text_time)
        if cfg_weight > 1:
            eps_text, eps_neg = eps.chunk(2)
            # eps_neg + w (eps_text - eps_neg) = (1 - w) eps_neg + w eps_text
            eps = ops.axpbypcz(eps_neg.contiguous(), eps_text.contiguous(), None, 1.0 - cfg_weight, cfg_weight)
        return eps

    def _denoising_step_dev(self, x_t, t_unet, coef, conditioning, cfg_weight, text_time, noise):
        return self.sampler.step_dev(self._eps(x_t, t_unet, conditioning, cfg_weight, text_time), x_t, coef, noise)

    def _denoising_step_eager(self, x_t, t, t_prev, conditioning, cfg_weight: float = 7.5, text_time=None, noise=None):
        nb = len(x_t) * (2 if cfg_weight > 1 else 1)
        t_unet = torch.full((nb,), float(t), dtype=torch.float32, device=x_t.device)
        return self.sampler.step(self._eps(x_t, t_unet, conditioning, cfg_weight, text_time), x_t, t, t_prev, noise)

    def _denoising_loop(self, x_T, T, conditioning, num_steps: int = 50, cfg_weight: float = 7.5, text_time=None,
                        key: Optional[torch.Generator] = None):
        """__init__.py:84-100."""
        x_t = x_T
        for t, t_prev in self.sampler.timesteps(num_steps, start_time=T):
            x_t = self._denoising_step(x_t, t, t_prev, conditioning, cfg_weight, text_time, key=key)
            yield x_t

    def generate_latents(self, text: str, n_images: int = 1, num_steps: int = 50, cfg_weight: float = 7.5,
                         negative_text: str = "", latent_size: Tuple[int, int] = (64, 64), seed=None):
        """__init__.py:102-129."""
        seed = int(time.time()) if seed is None else seed
        g = torch.Generator(device=self.device).manual_seed(seed)
        conditioning = self._get_text_conditioning(text, n_images, cfg_weight, negative_text)
        x_T = self.sampler.sample_prior((n_images, *latent_size, self.autoencoder.latent_channels), dtype=self.dtype,
                                        key=g, device=self.device)
        yield from self._denoising_loop(x_T, self.sampler.max_time, conditioning, num_steps, cfg_weight, key=g)

    def decode(self, x_t, precision: Optional[str] = None):
        """__init__.py:166-169: clip(vae.decode(x_t) / 2 + 0.5, 0, 1), fused into the last conv.  float32 arithmetic
        like the reference's VAE (load_autoencoder(model, False), __init__.py:25) unless precision="bf16"."""
        precision = precision or self.autoencoder.precision
        if not self.use_graph:
            return self.autoencoder.decode_image(x_t, precision)
        key = ("decode", tuple(x_t.shape), precision, self.autoencoder._store.epoch)
        ent = self._graphs.get(key)
        if ent is None:
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
            self._graphs[key] = ent
        g, sx, out = ent
        sx.copy_(x_t)
        g.replay()
        return out.clone()


class StableDiffusionXL(StableDiffusion):
    def __init__(self, model: str = _DEFAULT_MODEL, float16: bool = False, device: str = "cuda", use_graph: bool = True):
        super().__init__(model, float16, device, use_graph)
        self.sampler = SimpleEulerAncestralSampler(self.diffusion_config)
        self.text_encoder_1 = self.text_encoder
        self.tokenizer_1 = self.tokenizer
        del self.tokenizer, self.text_encoder
        self.text_encoder_2 = load_text_encoder(model, float16, model_key="text_encoder_2", device=device)
        self.tokenizer_2 = load_tokenizer(model, merges_key="tokenizer_2_merges", vocab_key="tokenizer_2_vocab")

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
        seed = int(time.time()) if seed is None else seed
        g = torch.Generator(device=self.device).manual_seed(seed)
        conditioning, pooled = self._get_text_conditioning(text, n_images, cfg_weight, negative_text)
        time_ids = torch.tensor([[512, 512, 0, 0, 512, 512.0]] * len(pooled), device=self.device)
        text_time = (pooled, time_ids)
        x_T = self.sampler.sample_prior((n_images, *latent_size, self.autoencoder.latent_channels), dtype=self.dtype,
                                        key=g, device=self.device)
        yield from self._denoising_loop(x_T, self.sampler.max_time, conditioning, num_steps, cfg_weight,
                                        text_time=text_time, key=g)

"""SimpleEulerSampler / SimpleEulerAncestralSampler (mirror of the reference's
stable_diffusion/stable_diffusion/sampler.py).  The sigma table and the per-step coefficients are host
scalars (float32, like the reference); the update itself is one HBM-bound libfluxhip kernel."""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops
from .config import DiffusionConfig


def _linspace(a, b, num):
    """sampler.py:8-10."""
    x = torch.arange(0, num, dtype=torch.float32) / (num - 1)
    return (b - a) * x + a


def _interp(y: torch.Tensor, x_new: torch.Tensor) -> torch.Tensor:
    """sampler.py:13-23."""
    x_low = x_new.to(torch.int32).long()
    x_high = torch.clamp(x_low + 1, max=len(y) - 1)
    delta = x_new - x_low
    return y[x_low] * (1 - delta) + delta * y[x_high]


class SimpleEulerSampler:
    def __init__(self, config: DiffusionConfig):
        if config.beta_schedule == "linear":
            betas = _linspace(config.beta_start, config.beta_end, config.num_train_steps)
        elif config.beta_schedule == "scaled_linear":
            betas = _linspace(config.beta_start ** 0.5, config.beta_end ** 0.5, config.num_train_steps).square()
        else:
            raise NotImplementedError(f"{config.beta_schedule} is not implemented.")
        alphas_cumprod = torch.cumprod(1 - betas, dim=0)
        self._sigmas = torch.cat([torch.zeros(1), ((1 - alphas_cumprod) / alphas_cumprod).sqrt()])
        # dtype the step coefficients are derived in.  The reference casts sigma / sigma_prev to the eps dtype FIRST and
        # evaluates everything derived from them in it (sampler.py:77-78,90-96): float16 under float16=True (the pipeline
        # sets this), float32 otherwise.
        self.coef_dtype = torch.float32

    @property
    def max_time(self):
        return len(self._sigmas) - 1

    def sample_prior(self, shape, dtype=torch.bfloat16, key: Optional[torch.Generator] = None, device="cuda", rows=None):
        """sampler.py:56-60: N(0,1) * sigma_max / sqrt(sigma_max^2 + 1).  rows=(lo, hi): multi-GPU — the FULL batch is
        drawn (every rank, same seed: "one seed -> one batch" for any world size) and rows [lo, hi) are kept."""
        noise = torch.randn(shape, generator=key, device=device, dtype=torch.float32)
        if rows is not None:
            noise = noise[rows[0]:rows[1]]
        s = self._sigmas[-1]
        return (noise * float(s * torch.rsqrt(s.square() + 1))).to(dtype).contiguous()

    def sigmas(self, t) -> torch.Tensor:
        return _interp(self._sigmas, torch.as_tensor(t, dtype=torch.float32))

    def timesteps(self, num_steps: int, start_time=None, dtype=torch.float32):
        """sampler.py:70-74 (the timestep values are rounded to `dtype` like the reference's astype)."""
        start_time = start_time or (len(self._sigmas) - 1)
        assert 0 < start_time <= (len(self._sigmas) - 1)
        steps = _linspace(start_time, 0, num_steps + 1).to(dtype).to(torch.float32).tolist()
        return list(zip(steps, steps[1:]))

    def _coeffs(self, t, t_prev):
        sigma, sigma_prev = self.sigmas(t).to(self.coef_dtype), self.sigmas(t_prev).to(self.coef_dtype)
        c1, dt = (sigma.square() + 1).sqrt(), sigma_prev - sigma              # evaluated in coef_dtype like the reference
        inv = torch.rsqrt(sigma_prev.square() + 1)
        return float(c1.float() * inv.float()), float(dt.float() * inv.float()), 0.0

    needs_noise = False

    def coeffs(self, t, t_prev):
        """(ca, cb, cc) of x' = ca x + cb eps + cc noise for the step t -> t_prev (host floats)."""
        return self._coeffs(t, t_prev)

    def coeff_table(self, steps, device) -> torch.Tensor:
        """float32 [len(steps), 3] on `device`: the (ca, cb, cc) of every (t, t_prev) of a run, uploaded ONCE — the
        captured step graph reads its coefficients from a device buffer that is refreshed by a device-side copy of one
        row, so no per-step host->device transfer sits between graph replays."""
        return torch.tensor([self._coeffs(t, tp) for t, tp in steps], dtype=torch.float32).to(device)

    def step(self, eps_pred: torch.Tensor, x_t: torch.Tensor, t, t_prev, noise: Optional[torch.Tensor] = None):
        """sampler.py:76-85: ((sigma^2+1)^.5 x + eps (sigma_prev - sigma)) (sigma_prev^2+1)^-.5"""
        ca, cb, _ = self._coeffs(t, t_prev)
        return ops.axpbypcz(x_t, eps_pred, None, ca, cb)

    def step_dev(self, eps_pred: torch.Tensor, x_t: torch.Tensor, coef: torch.Tensor,
                 noise: Optional[torch.Tensor] = None):
        """`step` with the coefficients in a device buffer (one captured graph for all steps)."""
        return ops.axpbypcz_dev(x_t, eps_pred, noise if self.needs_noise else None, coef)


class SimpleEulerAncestralSampler(SimpleEulerSampler):
    needs_noise = True

    def _coeffs(self, t, t_prev):
        sigma, sigma_prev = self.sigmas(t).to(self.coef_dtype), self.sigmas(t_prev).to(self.coef_dtype)
        sigma2, sigma_prev2 = sigma.square(), sigma_prev.square()
        sigma_up = (sigma_prev2 * (sigma2 - sigma_prev2) / sigma2).sqrt()
        sigma_down = (sigma_prev2 - sigma_up ** 2).sqrt()
        inv = torch.rsqrt(sigma_prev2 + 1).float()
        return float((sigma2 + 1).sqrt().float() * inv), float((sigma_down - sigma).float() * inv), float(sigma_up.float() * inv)

    def draw_noise(self, x_t: torch.Tensor, key: Optional[torch.Generator] = None, shard=None) -> torch.Tensor:
        """The fresh N(0,1) of sampler.py:100, drawn from the run's seeded generator (the reference's
        mx.random.seed(seed) fixes this noise too, __init__.py:242-243).  shard=(lo, hi, n_total): multi-GPU — every
        rank draws the step's FULL-batch noise from the job's generator and keeps its rows."""
        if shard is not None:
            lo, hi, n = shard
            full = torch.randn((n, *x_t.shape[1:]), generator=key, device=x_t.device, dtype=torch.float32)
            return full[lo:hi].to(x_t.dtype).contiguous()
        return torch.randn(x_t.shape, generator=key, device=x_t.device, dtype=torch.float32).to(x_t.dtype)

    def step(self, eps_pred, x_t, t, t_prev, noise: Optional[torch.Tensor] = None,
             key: Optional[torch.Generator] = None):
        """sampler.py:89-105; draws the fresh N(0,1) from `key` unless `noise` is given (parity tests)."""
        ca, cb, cc = self._coeffs(t, t_prev)
        if noise is None:
            noise = self.draw_noise(x_t, key)
        return ops.axpbypcz(x_t, eps_pred, noise.contiguous(), ca, cb, cc)

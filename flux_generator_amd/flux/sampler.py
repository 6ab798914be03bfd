"""FluxSampler — flow-matching Euler sampler (mirror of the reference's flux/sampler.py).

Schedules are host scalars (negligible work, flux/sampler.py:15-31); the Euler update runs as a
libfluxhip kernel (flux/sampler.py:56-57); the prior is torch's Philox generator on the device —
MLX's RNG stream is not reproducible outside MLX, so parity is defined on identical ``x_T``
(SURVEY.md Appendix A).
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import Optional

import torch

from .. import ops


class FluxSampler:
    def __init__(self, name: str, base_shift: float = 0.5, max_shift: float = 1.15):
        self._base_shift = base_shift
        self._max_shift = max_shift
        self._schnell = "schnell" in name

    def _time_shift(self, x: float, t: float) -> float:
        """flux/sampler.py:15-20 (t = 0 maps to 0: 1/t = inf in IEEE arithmetic)."""
        x1, x2 = 256, 4096
        t1, t2 = self._base_shift, self._max_shift
        exp_mu = math.exp((x - x1) * (t2 - t1) / (x2 - x1) + t1)
        if t == 0:
            return 0.0
        return exp_mu / (exp_mu + (1 / t - 1))

    @lru_cache
    def timesteps(self, num_steps: int, image_sequence_length: int, start: float = 1, stop: float = 0):
        """flux/sampler.py:22-31: float32 linspace, time-shifted unless schnell; python list."""
        t = torch.linspace(start, stop, num_steps + 1, dtype=torch.float32).tolist()
        if not self._schnell:
            t = [float(torch.tensor(self._time_shift(image_sequence_length, v), dtype=torch.float32)) for v in t]
        return t

    def sample_prior(self, shape, dtype=torch.bfloat16, key: Optional[torch.Generator] = None, device="cuda"):
        """flux/sampler.py:44-45: N(0,1) in ``dtype``."""
        return torch.randn(shape, generator=key, device=device, dtype=torch.float32).to(dtype)

    def add_noise(self, x, t, noise=None, key=None):
        """flux/sampler.py:47-54 (training / img2img helper; plain tensor arithmetic)."""
        noise = noise if noise is not None else torch.randn(x.shape, generator=key, device=x.device).to(x.dtype)
        t = t.reshape([-1] + [1] * (x.ndim - 1))
        return x * (1 - t) + t * noise

    def step(self, pred: torch.Tensor, x_t: torch.Tensor, t: float, t_prev: float) -> torch.Tensor:
        """flux/sampler.py:56-57: x_t + (t_prev - t) * pred, one HBM-bound kernel. The python scalar
        takes the array dtype first (MLX weak scalar typing): dt is rounded to bf16."""
        dt = float(torch.tensor(t_prev - t, dtype=x_t.dtype))
        return ops.euler_step(x_t, pred, dt)

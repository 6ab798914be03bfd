"""Multi-GPU batch sharding: one process per GPU, torch.distributed over
RCCL/xGMI ("nccl" backend on ROCm; "gloo" in the CPU tests).

Images of a batch never interact inside the denoise/decode path (every op is per-sample, batch is
dim 0 — flux/flux.py:76-83,142), so the path shards by image with NO data-path collective:

  * conditioning (txt [P,S,4096], vec [P,768]) is computed on rank 0 and BROADCAST (2-4 MB/prompt);
  * x_T: every rank draws the FULL batch from the same Philox seed and keeps its rows, which preserves
    the reference's "one seed -> one batch" semantics (flux/flux.py:138-142) without communication;
  * decoded uint8 images are GATHERed to rank 0 (0.79 MB each at 512x512).

The reference's only collective (gradient all-reduce in dreambooth.py:198,227) belongs to training
and is out of scope.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n images: the first n % W ranks take one extra."""
    q, r = divmod(n, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_conditioning(txt: Optional[torch.Tensor], vec: Optional[torch.Tensor], shapes=None, src: int = 0,
                           device=None, dtype=torch.bfloat16):
    """Rank `src` passes tensors; the others pass None + `shapes` = (txt_shape, vec_shape)."""
    rank, W = world()
    if W == 1:
        return txt, vec
    if rank != src:
        txt = torch.empty(shapes[0], dtype=dtype, device=device)
        vec = torch.empty(shapes[1], dtype=dtype, device=device)
    # bf16 / int16 are not supported by every gloo build: ship the raw bytes
    for t in (txt, vec):
        dist.broadcast(t.view(torch.uint8), src=src)
    return txt, vec


def sample_prior_sharded(shape, seed: int, device, dtype=torch.bfloat16) -> torch.Tensor:
    """Full-batch draw on every rank (same seed), local rows kept."""
    rank, W = world()
    g = torch.Generator(device=device).manual_seed(seed)
    full = torch.randn(shape, generator=g, device=device, dtype=torch.float32).to(dtype)
    lo, hi = shard_range(shape[0], rank, W)
    return full[lo:hi].contiguous()


def gather_images(local: torch.Tensor, n_total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """local uint8 [n_local,H,W,3] -> rank dst gets [n_total,H,W,3] in batch order, others None."""
    rank, W = world()
    if W == 1:
        return local
    sizes = [shard_range(n_total, r, W) for r in range(W)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((nmax, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs: Optional[List[torch.Tensor]] = [torch.empty_like(pad) for _ in range(W)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def to_uint8(images: torch.Tensor) -> torch.Tensor:
    """(x*255).astype(uint8): float->uint8 TRUNCATION like the reference (txt2image.py:133,144)."""
    return (images * 255).to(torch.uint8)

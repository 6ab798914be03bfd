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


class CollectiveStepFailed(RuntimeError):
    """Raised on EVERY rank when a rank-local step in front of a collective failed on one rank (the failing rank raises its
    own exception): all ranks leave the collective sequence at the same point, so the process group stays usable."""


def all_ok(ok: bool) -> bool:
    """True iff every rank reports ok (one 4-byte all-reduce) - the agreement point in front of a collective whose
    participants could otherwise diverge.  Without a process group: `ok`."""
    if not active():
        return bool(ok)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(t)
    return int(t.item()) == 0


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def active() -> bool:
    """True when a process group exists (torchrun launch): the sharded code path and its collectives are used even at
    world_size 1, so a single-GPU box exercises exactly the code an 8-GPU node runs."""
    return dist.is_available() and dist.is_initialized()


def _host_staged() -> bool:
    """gloo carries host memory only (it is the backend of the CPU tests and of the two-ranks-on-one-GPU test): device
    tensors are staged through the host there.  RCCL ("nccl") moves device memory directly over xGMI."""
    return dist.get_backend() == "gloo"


def _broadcast(t: torch.Tensor, src: int) -> None:
    """In-place broadcast of a contiguous tensor."""
    if t.is_cuda and _host_staged():
        h = t.cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n images: the first n % W ranks take one extra."""
    q, r = divmod(n, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_conditioning(txt: Optional[torch.Tensor], vec: Optional[torch.Tensor], shapes=None, src: int = 0,
                           device=None, dtype=torch.bfloat16):
    """Rank `src` passes tensors; the others pass None + `shapes` = (txt_shape, vec_shape)."""
    rank, W = world()
    if not active():
        return txt, vec
    if rank != src:
        txt = torch.empty(shapes[0], dtype=dtype, device=device)
        vec = torch.empty(shapes[1], dtype=dtype, device=device)
    # bf16 / int16 are not supported by every gloo build: ship the raw bytes
    for t in (txt, vec):
        _broadcast(t.view(torch.uint8), src)
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
    if not active():
        return local
    sizes = [shard_range(n_total, r, W) for r in range(W)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((nmax, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    dev = pad.device
    if pad.is_cuda and _host_staged():
        pad = pad.cpu()
    bufs: Optional[List[torch.Tensor]] = [torch.empty_like(pad) for _ in range(W)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0).to(dev)


def broadcast_seed(seed: Optional[int], device, src: int = 0) -> int:
    """One seed for the whole job: rank `src`'s value (drawn from the clock-seeded default generator when the caller
    passed None, like the reference's unseeded mx.random) is broadcast so every rank draws the same full batch."""
    rank, W = world()
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    if not active():
        return int(seed)
    t = torch.tensor([int(seed)], dtype=torch.int64, device=device)
    _broadcast(t, src)
    return int(t.item())


def shard_generation_inputs(n_images: int, latent_shape, seed: Optional[int], device, make_conditioning,
                            dtype=torch.bfloat16, src: int = 0):
    """The multi-GPU front half of FluxPipeline.generate_latents (flux/flux.py:138-147) for W ranks:

      * one seed for the job (broadcast_seed), full-batch prior on every rank, local rows kept;
      * `make_conditioning()` -> (txt [P,S,Dt], vec [P,Dv]) runs on rank `src` ONLY (the T5 / CLIP encoders are not
        even evaluated on the other ranks); its shapes, then its bytes, are broadcast over RCCL/xGMI;
      * P == 1 (one prompt for the whole batch): every rank expands it to its local image count;
        P == n_images (one prompt per image): every rank keeps rows [lo, hi).

    Returns (x_T_local [n_local,*latent_shape], txt_local, vec_local, (lo, hi)).  n_local may be 0."""
    rank, W = world()
    seed = broadcast_seed(seed, device, src)
    x_T = sample_prior_sharded((n_images, *latent_shape), seed, device, dtype)
    lo, hi = shard_range(n_images, rank, W)
    txt = vec = None
    meta = torch.zeros(5, dtype=torch.int64, device=device)
    err = None
    if rank == src:
        # a failure of the rank-`src`-only text conditioning (missing checkpoint, OOM in T5 ...) must reach EVERY rank: the others
        # are about to enter the broadcast below and would wait in it for ever.  The header carries the outcome (-1 = failed).
        try:
            txt, vec = make_conditioning()
            txt, vec = txt.to(device=device, dtype=dtype).contiguous(), vec.to(device=device, dtype=dtype).contiguous()
            meta = torch.tensor([*txt.shape, *vec.shape], dtype=torch.int64, device=device)
        except Exception as e:                   # noqa: BLE001 - re-raised below, on every rank
            if not active():
                raise
            err = e
            meta = torch.full((5,), -1, dtype=torch.int64, device=device)
    if active():
        _broadcast(meta, src)
    m = [int(v) for v in meta.tolist()]
    if m[0] < 0:
        raise CollectiveStepFailed(f"text conditioning failed on rank {src}" + (f": {type(err).__name__}: {err}" if err else "")) from err
    txt, vec = broadcast_conditioning(txt, vec, shapes=(tuple(m[:3]), tuple(m[3:])), src=src, device=device, dtype=dtype)
    if txt.shape[0] == 1:
        txt, vec = txt.expand(hi - lo, -1, -1).contiguous(), vec.expand(hi - lo, -1).contiguous()
    elif txt.shape[0] == n_images:
        txt, vec = txt[lo:hi].contiguous(), vec[lo:hi].contiguous()
    else:
        raise ValueError(f"conditioning batch {txt.shape[0]} is neither 1 nor n_images={n_images}")
    return x_T, txt, vec, (lo, hi)


def broadcast_tensors(tensors, src: int = 0, bucket_bytes: int = 1 << 28) -> int:
    """In-place broadcast of a list of (contiguous) parameter tensors from rank `src` — SURVEY.md §8(e).2: the optional
    one-time weight broadcast when only rank `src` reads the checkpoint from disk.  xGMI rings are per-link bound
    (~153 GB/s), so large tensors go as they are (one collective each, no staging copy) and the many small ones
    (biases, norm scales) are coalesced into `bucket_bytes` staging buffers instead of one launch-bound collective each.
    Returns the number of bytes broadcast.  No-op without a process group."""
    if not active():
        return 0
    rank, _ = world()
    total, small = 0, []

    def flush():
        if not small:
            return
        flat = torch.cat([t.reshape(-1).view(torch.uint8) for t in small])
        _broadcast(flat, src)
        if rank != src:
            off = 0
            for t in small:
                n = t.numel() * t.element_size()
                t.reshape(-1).view(torch.uint8).copy_(flat[off:off + n])
                off += n
        small.clear()

    pending = 0
    for t in tensors:
        if not t.is_contiguous():
            raise ValueError("broadcast_tensors needs contiguous tensors (views of one buffer: pass the buffer)")
        n = t.numel() * t.element_size()
        total += n
        if n >= (1 << 22):
            _broadcast(t.view(-1).view(torch.uint8), src)     # raw bytes: bf16 is not supported by every gloo build
        else:
            small.append(t)
            pending += n
            if pending >= bucket_bytes:
                flush()
                pending = 0
    flush()
    return total


_DTYPES = [torch.bfloat16, torch.float32, torch.float16, torch.int32, torch.int64, torch.uint8]


def broadcast_from(make, device, src: int = 0):
    """`make()` -> list of tensors runs on rank `src` ONLY (text towers: the other ranks never build or evaluate them);
    count / dtypes / shapes travel in one int64 header, then the raw bytes, over RCCL (gloo in the CPU tests).
    Returns the list on every rank.  Without a process group it is just `make()`."""
    rank, W = world()
    if not active():
        return [t.to(device) for t in make()]
    MAXT, MAXD = 8, 6
    hdr = torch.zeros(1 + MAXT * (2 + MAXD), dtype=torch.int64, device=device)
    ts = None
    err = None
    if rank == src:
        try:                                      # (a failure here is broadcast in the header: see shard_generation_inputs)
            ts = [t.to(device).contiguous() for t in make()]
            if len(ts) > MAXT or any(t.dim() > MAXD for t in ts):
                raise ValueError("broadcast_from: at most 8 tensors of at most 6 dims")
            h = [len(ts)]
            for t in ts:
                h += [_DTYPES.index(t.dtype), t.dim()] + list(t.shape) + [0] * (MAXD - t.dim())
            hdr[: len(h)] = torch.tensor(h, dtype=torch.int64)
        except Exception as e:                    # noqa: BLE001 - re-raised below, on every rank
            err = e
            hdr[0] = -1
    _broadcast(hdr, src)
    h = [int(v) for v in hdr.tolist()]
    if h[0] < 0:
        raise CollectiveStepFailed(f"broadcast_from: the producer failed on rank {src}" + (f": {type(err).__name__}: {err}" if err else "")) from err
    if rank != src:
        ts = []
        for i in range(h[0]):
            o = 1 + i * (2 + MAXD)
            ts.append(torch.empty(h[o + 2: o + 2 + h[o + 1]], dtype=_DTYPES[h[o]], device=device))
    for t in ts:
        if t.numel():
            _broadcast(t.view(-1).view(torch.uint8), src)
    return ts


def to_uint8(images: torch.Tensor) -> torch.Tensor:
    """(x*255).astype(uint8): float->uint8 TRUNCATION like the reference (txt2image.py:133,144)."""
    return (images * 255).to(torch.uint8)


# ---------------------------------------------------------------------------------------------- host-side placement
def partition_cpus(cpus, local_rank: int, local_world: int):
    """Contiguous, balanced slice of a sorted CPU list for one of `local_world` ranks sharing it (every rank gets at
    least one CPU; the first len % world ranks take one extra)."""
    cpus = sorted(cpus)
    if local_world <= 1 or len(cpus) < local_world:
        return cpus
    lo, hi = shard_range(len(cpus), local_rank, local_world)
    return cpus[lo:hi]


def _parse_cpulist(text: str):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(index: int) -> int:
    """NUMA node of GPU `index` (its PCI function's sysfs `numa_node`), or -1 when the platform does not say."""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            return int(f.read().strip())
    except Exception:
        return -1


def bind_rank_to_cpus(local_rank: int, local_world: int, gpu_index: Optional[int] = None) -> dict:
    """One process per GPU also means one launch thread per GPU: eight ranks whose threads wander over all sockets pay
    cross-socket latency on every graph launch and fight each other's OpenMP pools.  Pin this rank to its share of the
    CPUs of ITS GPU's NUMA node (sysfs; all allowed CPUs when the node is unknown), split evenly among the ranks on that
    node.  Best effort, never fatal; returns what was done for the bench record."""
    import os
    info = {"numa_node": -1, "cpus": None, "bound": False}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        node = gpu_numa_node(gpu_index if gpu_index is not None else local_rank) if torch.cuda.is_available() else -1
        pool, peers, me = allowed, local_world, local_rank
        if node >= 0:
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                node_cpus = [c for c in _parse_cpulist(f.read()) if c in set(allowed)]
            same = [r for r in range(local_world) if gpu_numa_node(r) == node] if torch.cuda.device_count() >= local_world else []
            if node_cpus and local_rank in same:
                pool, peers, me = node_cpus, len(same), same.index(local_rank)
        mine = partition_cpus(pool, me, peers)
        if mine and len(mine) < len(allowed):
            os.sched_setaffinity(0, mine)
            info["bound"] = True
        info.update(numa_node=node, cpus=f"{mine[0]}-{mine[-1]} ({len(mine)})" if mine else None)
    except Exception as e:      # containers without sysfs, restricted cpusets, ...
        info["error"] = f"{type(e).__name__}: {e}"
    return info

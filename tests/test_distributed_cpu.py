"""world_size-2 gloo tests of the multi-GPU sharding logic on CPU."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    from flux_generator_amd.parallel import shard_range
    for n in (1, 2, 7, 8, 32, 33):
        for W in (1, 2, 4, 8):
            parts = [shard_range(n, r, W) for r in range(W)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flux_generator_amd import parallel as P
    n = 5
    # conditioning broadcast from rank 0
    if rank == 0:
        g = torch.Generator().manual_seed(3)
        txt, vec = torch.randn(n, 6, 16, generator=g).bfloat16(), torch.randn(n, 8, generator=g).bfloat16()
        txt, vec = P.broadcast_conditioning(txt, vec)
    else:
        txt, vec = P.broadcast_conditioning(None, None, shapes=((n, 6, 16), (n, 8)), device="cpu")
    # same seed -> same full batch on every rank, each keeps its rows
    x = P.sample_prior_sharded((n, 4, 4, 16), seed=1234, device="cpu")
    lo, hi = P.shard_range(n, rank, world)
    full = torch.randn((n, 4, 4, 16), generator=torch.Generator().manual_seed(1234)).bfloat16()
    ok = torch.equal(x, full[lo:hi])
    # "decode" = something that depends on the local rows, then gather in batch order
    imgs = P.to_uint8((x.float()[..., :3].clamp(-1, 1) + 1) * 0.5)
    allimg = P.gather_images(imgs, n)
    if rank == 0:
        want = P.to_uint8((full.float()[..., :3].clamp(-1, 1) + 1) * 0.5)
        ok = ok and torch.equal(allimg, want)
    else:
        ok = ok and allimg is None
    q.put((rank, ok, float(txt.float().sum()), float(vec.float().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_shard_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]      # both ranks hold the same conditioning


def _worker_inputs(rank, world, port, q, n, per_image_prompts):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flux_generator_amd import parallel as P
    calls = []

    def cond():                       # the T5 / CLIP stand-in: must run on rank 0 only
        calls.append(rank)
        g = torch.Generator().manual_seed(11)
        P_ = n if per_image_prompts else 1
        return torch.randn(P_, 6, 16, generator=g), torch.randn(P_, 8, generator=g)

    x, txt, vec, (lo, hi) = P.shard_generation_inputs(n, (4, 4, 16), None if rank else 77, "cpu", cond)
    g = torch.Generator().manual_seed(11)
    P_ = n if per_image_prompts else 1
    ftxt, fvec = torch.randn(P_, 6, 16, generator=g).bfloat16(), torch.randn(P_, 8, generator=g).bfloat16()
    full = torch.randn((n, 4, 4, 16), generator=torch.Generator().manual_seed(77)).bfloat16()   # rank 0's seed wins
    ok = (lo, hi) == P.shard_range(n, rank, world) and torch.equal(x, full[lo:hi]) and calls == ([0] if rank == 0 else [])
    if per_image_prompts:
        ok = ok and torch.equal(txt, ftxt[lo:hi]) and torch.equal(vec, fvec[lo:hi])
    else:
        ok = ok and txt.shape == (hi - lo, 6, 16) and all(torch.equal(txt[i], ftxt[0]) for i in range(hi - lo))
        ok = ok and vec.shape == (hi - lo, 8) and all(torch.equal(vec[i], fvec[0]) for i in range(hi - lo))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,per_image", [(5, False), (4, True), (1, False)])      # n = 1: rank 1 gets no image
def test_two_rank_generation_inputs(n, per_image):
    """The front half of FluxPipeline.generate_latents under torchrun: seed + conditioning from rank 0, prior slice."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + n * 7 + per_image) % 2000
    procs = [ctx.Process(target=_worker_inputs, args=(r, 2, port, q, n, per_image)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def test_uint8_truncates_like_reference():
    from flux_generator_amd.parallel import to_uint8
    assert to_uint8(torch.tensor([0.0, 0.999, 1.0, 0.5])).tolist() == [0, 254, 255, 127]

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


def _run_bench(*argv, env=None):
    import json
    import subprocess
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)
    last = [l for l in r.stdout.splitlines() if l.strip()]
    return r, (json.loads(last[-1]) if last and last[-1].startswith("{") else None)


def test_bench_gpus2_launches_two_ranks():
    """A plain `python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) re-executes itself under torch.distributed.run and
    two ranks join (--dry-run: gloo, no kernels); the JSON line is the last line of stdout and reports n_gpus = 2."""
    r, out = _run_bench("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out is not None and out["n_gpus"] == 2 and out["ranks_joined"] == 2 and out["dry_run"] is True
    assert out["conditioning_evaluated_on_ranks"] == [0] and len(out["per_rank_ms"]) == 2
    assert out["config"]["global_batch"] == 2 and out["scaling"] == "weak"


def test_bench_gpus8_dry_run_and_cpu_placement():
    """The launch an 8-GPU node gets, on CPU: `python bench.py --gpus 8 --dry-run` starts 8 ranks (gloo), every rank joins the
    broadcast / barrier / max-reduce / gather skeleton, and every rank pins itself to its own share of the CPUs
    (parallel.bind_rank_to_cpus: disjoint slices when there are at least 8 CPUs)."""
    r, out = _run_bench("--gpus", "8", "--dry-run", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out is not None and out["n_gpus"] == 8 and out["ranks_joined"] == 8 and len(out["per_rank_ms"]) == 8
    assert out["conditioning_evaluated_on_ranks"] == [0] and out["config"]["global_batch"] == 8
    pl = out["placement"]
    assert len(pl) == 8 and all(p is not None and "error" not in p for p in pl)
    if len(os.sched_getaffinity(0)) >= 8:
        assert all(p["bound"] for p in pl) and len({p["cpus"] for p in pl}) == 8


def test_partition_cpus():
    from flux_generator_amd.parallel import _parse_cpulist, partition_cpus
    cpus = _parse_cpulist("0-5,8,10-12")
    assert cpus == [0, 1, 2, 3, 4, 5, 8, 10, 11, 12]
    parts = [partition_cpus(cpus, r, 4) for r in range(4)]
    assert sum(parts, []) == cpus and [len(p) for p in parts] == [3, 3, 2, 2]
    assert partition_cpus([3, 1], 0, 4) == [1, 3]          # fewer CPUs than ranks: everybody keeps them all
    assert partition_cpus(cpus, 0, 1) == cpus


def test_bench_refuses_world_size_mismatch():
    """--gpus must agree with the launcher's WORLD_SIZE: a mismatch is an error, not a silent 1-rank run."""
    r, out = _run_bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and out is None and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def _worker_sd(rank, world, port, q):
    """The stable_diffusion/ pipelines' sharding pieces (flux_generator_amd/stable_diffusion/__init__.py `_job_inputs`,
    sampler rows / shard arguments) on CPU: broadcast_from evaluates the text-tower stand-in on rank 0 only; the prior and
    the ancestral sampler's per-step noise are full-batch draws from the job seed, sliced."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flux_generator_amd import parallel as P
    from flux_generator_amd.stable_diffusion.config import DiffusionConfig
    from flux_generator_amd.stable_diffusion.sampler import SimpleEulerAncestralSampler
    calls = []

    def towers():
        calls.append(rank)
        g = torch.Generator().manual_seed(5)
        return [torch.randn(2, 7, 16, generator=g).bfloat16(), torch.randn(2, 12, generator=g), torch.arange(6, dtype=torch.int32)]

    seed = P.broadcast_seed(41 if rank == 0 else 999, "cpu")
    cond = P.broadcast_from(towers, "cpu")
    g5 = torch.Generator().manual_seed(5)
    ok = calls == ([0] if rank == 0 else []) and seed == 41
    ok = ok and torch.equal(cond[0], torch.randn(2, 7, 16, generator=g5).bfloat16()) and cond[0].dtype == torch.bfloat16
    ok = ok and torch.equal(cond[1], torch.randn(2, 12, generator=g5)) and torch.equal(cond[2], torch.arange(6, dtype=torch.int32))
    n = 5
    lo, hi = P.shard_range(n, rank, world)
    sam = SimpleEulerAncestralSampler(DiffusionConfig())
    g = torch.Generator().manual_seed(seed)
    x = sam.sample_prior((n, 4, 4, 4), dtype=torch.float32, key=g, device="cpu", rows=(lo, hi))
    n1 = sam.draw_noise(x, g, (lo, hi, n))
    n2 = sam.draw_noise(x, g, (lo, hi, n))
    gf = torch.Generator().manual_seed(seed)                     # what ONE process draws for the whole batch
    fx = sam.sample_prior((n, 4, 4, 4), dtype=torch.float32, key=gf, device="cpu")
    f1 = sam.draw_noise(fx, gf)
    f2 = sam.draw_noise(fx, gf)
    ok = ok and torch.equal(x, fx[lo:hi]) and torch.equal(n1, f1[lo:hi]) and torch.equal(n2, f2[lo:hi])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sd_sharding_pieces():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_sd, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def _worker_bcast_params(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flux_generator_amd import parallel as P
    g = torch.Generator().manual_seed(9)
    shapes = [(3000, 1024), (7,), (33, 5), (1 << 21,), (2, 2)]       # one tensor above the 4 MiB direct-send threshold
    want = [torch.randn(*s, generator=g).bfloat16() for s in shapes]
    have = [w.clone() if rank == 0 else torch.zeros_like(w) for w in want]
    nbytes = P.broadcast_tensors(have, 0, bucket_bytes=64)         # tiny bucket: several flushes
    ok = nbytes == sum(w.numel() * 2 for w in want) and all(torch.equal(a, b) for a, b in zip(have, want))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_weight_broadcast():
    """parallel.broadcast_tensors (the optional one-time weight broadcast when only rank 0 reads the checkpoint)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_bcast_params, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


# ------------------------------------------------------------------------------------------ sharded HTTP surface
class _StubFlux:
    """Stand-in for FluxPipeline with the SAME sharding calls the real one makes (flux_generator_amd/parallel.py): text
    conditioning on rank 0 only + broadcast, same-seed prior slice, per-rank decode, uint8 gather to rank 0."""

    def __init__(self, rank):
        self.rank, self.cond_calls, self.shard = rank, 0, None

    def generate_latents(self, prompt, n_images=1, num_steps=2, latent_size=(8, 8), guidance=4.0, seed=None):
        from flux_generator_amd import parallel as P

        def cond():
            self.cond_calls += 1
            g = torch.Generator().manual_seed(len(prompt))
            return torch.randn(1, 6, 16, generator=g), torch.randn(1, 8, generator=g)

        x, txt, vec, self.shard = P.shard_generation_inputs(n_images, (*latent_size, 16), seed, "cpu", cond)
        yield (x, None, txt, None, vec)
        for _ in range(num_steps):
            x = (x.float() * 0.5 + txt.float().mean() * 0.0).to(x.dtype)
            yield x

    def decode(self, x, latent_size):
        img = torch.sigmoid(x.float()[..., :3])                                   # [n, h, w, 3] in (0, 1)
        return img.repeat_interleave(8, dim=1).repeat_interleave(8, dim=2)        # [n, 8h, 8w, 3]

    def gather_images(self, images, n_images):
        from flux_generator_amd import parallel as P
        return P.gather_images(P.to_uint8(images).contiguous(), n_images)


def _worker_http(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import flux_app
    stub = _StubFlux(rank)
    flux_app.api.init_pipeline = lambda model: stub                # noqa: E731 - every rank builds "the same pipeline"
    if rank != 0:
        served = flux_app.worker_loop(flux_app.api)
        q.put((rank, served, stub.cond_calls, None))
    else:
        import base64
        import io
        import numpy as np
        from fastapi.testclient import TestClient
        from PIL import Image
        client = TestClient(flux_app.get_app())
        got = []
        for n, seed in ((3, 5), (1, 9)):                           # n = 1: rank 1 has no image of the request
            r = client.post("/sdapi/v1/txt2img", json=dict(prompt="a cat", width=64, height=64, steps=2, batch_size=n, seed=seed,
                                                           model="schnell"))
            assert r.status_code == 200, r.text
            imgs = [np.asarray(Image.open(io.BytesIO(base64.b64decode(b)))) for b in r.json()["images"]]
            # what ONE process computes for the same seed: the full-batch prior, two halvings, the stub decode
            full = torch.randn((n, 8, 8, 16), generator=torch.Generator().manual_seed(seed)).bfloat16()
            x = full
            for _ in range(2):
                x = (x.float() * 0.5).to(x.dtype)
            want = (_StubFlux(0).decode(x, (8, 8)) * 255).to(torch.uint8).numpy()
            got.append(len(imgs) == n and all(np.array_equal(a, b) for a, b in zip(imgs, want)))
        opts = client.get("/sdapi/v1/options").json()
        got.append(opts["sd_device"].startswith("2 x MI355X") and opts["sd_backend"] == "Flux MLX")
        flux_app.shutdown_workers()
        q.put((rank, 2, stub.cond_calls, got))
    dist.barrier()
    dist.destroy_process_group()


def test_http_surface_shards_a_request_over_two_ranks():
    """`torchrun --nproc-per-node 2 flux_app.py` in miniature (gloo, stub pipelines that make the real parallel.* calls):
    rank 0 serves POST /sdapi/v1/txt2img through FastAPI's TestClient, rank 1 sits in flux_app.worker_loop; the request's
    batch_size * n_iter images (flux_app.py:123-204 of the reference: one batch) are split by parallel.shard_range, the text
    conditioning runs on rank 0 only, and the response carries all images in batch order, equal to the single-process result
    of the same seed.  Unmeasured on hardware (no multi-GPU box); the RCCL world-1 twin is tests/test_configs_gpu.py."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_http, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, served0, cond0, got), (r1, served1, cond1, _) = res
    assert got == [True, True, True], got
    assert served1 == 2, "the worker rank must have served both requests and left its loop on shutdown"
    assert cond0 == 2 and cond1 == 0, "text conditioning is computed on rank 0 only"


# ------------------------------------------------------------------------------------------ sharded HTTP surface: fail-stop
class _FaultyFlux(_StubFlux):
    """_StubFlux with injected rank-local failures: `fail_cond` (rank 0's text conditioning raises), `fail_decode` (this rank's
    decode raises) - armed per request through the prompt text so both ranks see the same schedule."""

    def generate_latents(self, prompt, n_images=1, num_steps=2, latent_size=(8, 8), guidance=4.0, seed=None):
        from flux_generator_amd import parallel as P

        def cond():
            self.cond_calls += 1
            if "bad-prompt" in prompt:
                raise ValueError("tokenizer exploded")
            g = torch.Generator().manual_seed(len(prompt))
            return torch.randn(1, 6, 16, generator=g), torch.randn(1, 8, generator=g)

        x, txt, vec, self.shard = P.shard_generation_inputs(n_images, (*latent_size, 16), seed, "cpu", cond)
        self._prompt = prompt
        yield (x, None, txt, None, vec)
        for _ in range(num_steps):
            x = (x.float() * 0.5).to(x.dtype)
            yield x

    def decode(self, x, latent_size):
        if f"decode-fails-on-{self.rank}" in self._prompt:
            raise MemoryError("decode ran out of memory")
        return super().decode(x, latent_size)


def _worker_http_faults(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FLUX_APP_NO_EXIT="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import flux_app
    stub = _FaultyFlux(rank)
    flux_app.api.init_pipeline = lambda model: stub                # noqa: E731
    if rank != 0:
        served = flux_app.worker_loop(flux_app.api)
        q.put((rank, served, None))
    else:
        from fastapi.testclient import TestClient
        client = TestClient(flux_app.get_app())
        codes = []
        for prompt in ("bad-prompt", "decode-fails-on-1", "decode-fails-on-0", "a cat"):
            r = client.post("/sdapi/v1/txt2img", json=dict(prompt=prompt, width=64, height=64, steps=2, batch_size=2, seed=3,
                                                           model="schnell"))
            codes.append((r.status_code, r.json().get("detail", "") if r.status_code != 200 else len(r.json()["images"])))
        flux_app.shutdown_workers()
        q.put((rank, 4, codes))
    dist.barrier()
    dist.destroy_process_group()


def test_http_surface_rank_local_failures_do_not_strand_the_peers():
    """Advisor (round 5), flux_app.py: after the agreed `init_pipeline`, a failure on ONE rank - rank 0's text conditioning, any
    rank's decode - left the other ranks inside the next collective.  Now the conditioning outcome travels in the broadcast
    header (parallel.shard_generation_inputs) and the denoise / decode stretch is agreed on in front of the gather: every such
    request is an HTTP 500 on rank 0, the worker is back in its loop, and the NEXT request is served by both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_http_faults, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, _, codes), (_, served1, _) = res
    assert [c for c, _ in codes] == [500, 500, 500, 200], codes
    assert "tokenizer exploded" in codes[0][1] and "another rank" in codes[1][1] and "out of memory" in codes[2][1], codes
    assert codes[3][1] == 2
    assert served1 == 4, "the worker rank must have gone through all four requests and left its loop on shutdown"

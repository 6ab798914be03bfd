#!/usr/bin/env python3
"""Headline benchmark: images/sec (+ denoise-step ms) for Flux-schnell 512x512 2-step on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``.  For N > 1 the job is one rank per GPU over RCCL:
either the caller launches it through ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (then
WORLD_SIZE must equal N), or a plain ``python bench.py --gpus N`` re-executes itself under torch.distributed.run
(127.0.0.1 rendezvous, a free port).  ``--dry-run`` runs the launch / sharding / collective skeleton on CPU (gloo, no
kernels) so the N-rank path is testable without GPUs.  A "step" is ONE pass of the hot path over one batch of
synthetic input on every rank: B images/GPU x (2 denoise steps [Flux forward + Euler] + VAE decode),
inputs (x_T, txt, vec) already resident in HBM.  Weak scaling: every rank generates its own images,
no data-path collective; the only collective is the timing barrier/max.

The JSON line also carries
  roofline     — the dominant kernel (the bf16 MFMA GEMM tile config with the largest share of a forward):
                 algorithmic FLOPs of its launches / their HIP-event time, measured live here, vs the dense
                 bf16 MFMA peak; `traffic` = its HBM bytes per launch from the committed PMC passes;
  cpu_baseline — the CPU oracle (a port; the reference's MLX cannot run here) timed on the host
                 cores on a bounded sample of the same workload, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

os.environ.setdefault("FLUX_ALLOW_RANDOM_INIT", "1")   # random-init weights of the named architecture (no checkpoints here): stated in `data`

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0    # dense, block-scaled K=128 fp8 MFMA (same guide)


def flux_forward_flops(L: int, S: int, D: int = 3072, depth: int = 19, singles: int = 38) -> float:
    """Algorithmic FLOPs (2*MAC) of one Flux forward per image."""
    T = L + S
    per_tok = 2 * D * (3 * D + D + 2 * 4 * D)          # qkv + proj + mlp (= 226.5 MFLOP at D=3072)
    attn = 4 * T * T * D
    f = depth * (per_tok * T + attn + 2 * 2 * D * 6 * D) + singles * (per_tok * T + attn + 2 * D * 3 * D)
    return f + 2 * (64 * L + 4096 * S) * D + 2 * 64 * L * D


def cpu_baseline_sample(T: int, threads: int, S: int = 256) -> dict:
    """Oracle (port) on the host cores: 3 double + 8 single blocks (after a warm-up block) at full Flux width, T tokens,
    bf16-representable weights, fp32 math; extrapolated to 2 x (19 double + 38 single) per image."""
    from oracle import flux_oracle as O
    torch.set_num_threads(threads)
    P = O.FluxParams(depth=1, depth_single_blocks=1)
    shapes = {k: v for k, v in O.flux_weight_shapes(P).items() if k.startswith(("double_blocks.0", "single_blocks.0"))}
    W = O.init_weights(shapes, seed=0)
    g = torch.Generator().manual_seed(0)
    img, txt = torch.randn(1, T - S, 3072, generator=g), torch.randn(1, S, 3072, generator=g)
    vec = torch.randn(1, 3072, generator=g)
    ids = torch.zeros(1, T, 3, dtype=torch.int32)
    pe = O.embed_nd(ids, P.axes_dim, P.theta)
    ND, NS = 3, 8      # repetitions: ~10-20 s of CPU work on the GPU box's host cores
    with torch.no_grad():
        i2, t2 = O.double_stream_block(W, "double_blocks.0", 24, img, txt, vec, pe)     # warm-up (thread pool, caches)
        t0 = time.perf_counter()
        for _ in range(ND):
            i2, t2 = O.double_stream_block(W, "double_blocks.0", 24, img, txt, vec, pe)
        t_double = (time.perf_counter() - t0) / ND
        x = torch.cat([t2, i2], dim=1)
        t0 = time.perf_counter()
        for _ in range(NS):
            x = O.single_stream_block(W, "single_blocks.0", 24, x, vec, pe)
        t_single = (time.perf_counter() - t0) / NS
    # the VAE decode of the same image (64 x 64 x 16 latents -> 512 x 512 x 3) in the reference's float32, once
    A = O.AutoEncoderParams()
    WA = O.init_weights(O.decoder_weight_shapes(A), seed=1)
    lat = int(round(((T - S) * 4) ** 0.5))
    z = torch.randn(1, (lat // 2) ** 2, 64, generator=g)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.pipeline_decode(A, WA, z, (lat, lat))
        t_vae = time.perf_counter() - t0
    per_image = 2 * (19 * t_double + 38 * t_single) + t_vae
    return {"value": 1.0 / per_image, "unit": "images/sec", "cores": threads, "kind": "port",
            "blocks_timed": ND + NS, "blocks_total": 2 * (19 + 38), "vae_decode_s": t_vae,
            "sample": f"oracle fp32: {ND} x DoubleStreamBlock ({t_double:.2f} s each) + {NS} x SingleStreamBlock ({t_single:.2f} s each) "
                      f"at full width, T={T}, after one warm-up block, extrapolated to 2 steps x (19+38) blocks; + ONE float32 VAE decode "
                      f"of the {lat}x{lat} latents ({t_vae:.1f} s, timed whole, cold)"}


def pmc_traffic(label: str, workload: str) -> dict:
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (tools/profile_round*.sh -> tools/pmc_summary.py; counters cannot be read from inside this process).  The newest
    profiles/rNN_hbm_traffic_pmc.csv is used, and only when it was collected on THIS workload (its `# workload:` header; files
    without one were taken on the headline, "flux-schnell B1 T1280"): a launch at another M moves other bytes, so any other
    workload reports null, as does a summary without a row for this kernel."""
    import csv, glob, re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic_pmc.csv")))
    m = re.search(r"cfg(\d+)", label)
    if not m or "fp8" in label or not files:
        return {"traffic": None}
    path = files[-1]
    with open(path) as f:
        head = [l for l in f if l.startswith("#")]
    taken_on = next((l.split(":", 1)[1].strip() for l in head if l.startswith("# workload:")), "flux-schnell B1 T1280")
    if taken_on != workload:
        return {"traffic": None, "traffic_note": f"{os.path.basename(path)} was collected on '{taken_on}', this run is '{workload}'"}
    import ctypes
    from flux_generator_amd import _lib
    bm, bn, th = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    if _lib.load().fluxhip_gemm_tile_shape(int(m.group(1)), bm, bn, th) != 0:
        return {"traffic": None}
    best = None
    with open(path) as f:
        for row in csv.DictReader(l for l in f if not l.startswith("#")):
            t = re.match(r"gemm_nt_kernel<(\d+), (\d+), \d+, \d+, (\d+)", row["kernel"])
            if t and (int(t.group(1)), int(t.group(2)), int(t.group(3))) == (bm.value, bn.value, 0):
                if best is None or int(row["launches"]) > int(best["launches"]):
                    best = row
    if best is None:
        return {"traffic": None}
    return {"traffic": float(best["avg_total_MB"]) * 1e6, "traffic_unit": "HBM+MALL bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
            "traffic_source": f"profiles/{os.path.basename(path)}: {best['kernel']}"}


def private_l2_floor(flow, ws, label: str):
    """Bytes per launch (mean over the launches labelled `label`) that MUST cross the L2 -> fabric boundary FETCH_SIZE /
    WRITE_SIZE count, given 8 XCDs with private L2s and the kernel's own block -> tile map (gemm_core.h, reduce-scatter
    brick map: blocks in (split, N-tile, M-tile) order, M fastest, cut into 8 contiguous ranges): per XCD the distinct
    (split, M-tile) activation panels and (split, N-tile) weight panels of its range, once each; the output, the
    residual, and for S > 1 the fp32 partials a block writes through for its S - 1 peers and reads back from them.
    None for labels that are not split-K bf16 GEMMs (the unsplit maps differ)."""
    import ctypes, re
    from flux_generator_amd import _lib
    lib = _lib.load()
    m = re.search(r"cfg(\d+)s(\d+)$", label)
    if not m:
        return None
    cfg, S = int(m.group(1)), int(m.group(2))
    bm, bn, th = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    if lib.fluxhip_gemm_tile_shape(cfg, bm, bn, th) != 0:
        return None
    bm, bn = bm.value, bn.value
    tot, cnt = 0.0, 0
    for fn, a in ws["plan"]:
        if getattr(fn, "__name__", "") != "fluxhip_gemm_bf16":
            continue
        code = lib.fluxhip_gemm_tile_cfg(a[0])
        if (code & 255) != cfg or (code >> 8) != S:
            continue
        d = a[0]._obj
        tm = sum((d.g[i].M + bm - 1) // bm for i in range(d.ngroups)) * d.nbatch
        tn = (d.N + bn - 1) // bn
        nblk = tm * tn * S
        kb = d.K // S * 2                                   # bytes of one panel row over one K range
        byts, q8, r8, lo = 0.0, nblk // 8, nblk % 8, 0
        for x in range(8):
            hi = lo + q8 + (1 if x < r8 else 0)
            pa, pw = set(), set()
            for lin in range(lo, hi):
                s_, t = divmod(lin, tm * tn)
                pa.add((s_, t % tm)); pw.add((s_, t // tm))
            byts += len(pa) * bm * kb + len(pw) * bn * kb
            lo = hi
        m_total = sum(d.g[i].M for i in range(d.ngroups)) * d.nbatch
        byts += 2.0 * m_total * d.N * (2 if d.g[0].res else 1)                       # output (+ residual)
        byts += 2.0 * nblk * bm * bn * 4.0 * (S - 1) / S                              # partials: written through + read back
        tot += byts; cnt += 1
    return tot / cnt if cnt else None


def time_other_configs(pipe, dev, reps: int = 3) -> list:
    """One driver-timed line per remaining BASELINE.json config, measured inside the headline run (N = 1, rank 0), after
    the headline's own measurements: C5's per-GPU shape (Flux-schnell fp8 blocks, 1024x1024, batch 4), C3 (Flux-dev
    1024x1024, batch 1, S = 512, guidance 7) and C4 (sdxl-turbo 512x512, 1 step, batch 16).  Each entry: one denoise step
    (graph replay, HIP events, `reps` replays) and one VAE decode of that batch, the algorithmic TFLOP/s of the step and its
    fraction of the dense MFMA peak of the dtype.  `value` of the JSON line stays the headline (C2)."""
    import warnings
    from flux_generator_amd.flux.flux import FluxPipeline
    out = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn):
        fn()                                         # capture / warm outside the timed region
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, r

    def flux_line(p, tag, B, lat, S, guidance, nsteps, dtype, peak):
        L = (lat // 2) ** 2
        g = torch.Generator(device=dev).manual_seed(99)
        x_T = torch.randn(B, lat, lat, 16, generator=g, device=dev).to(torch.bfloat16)
        txt = (torch.randn(B, S, 4096, generator=g, device=dev) * 0.1).to(torch.bfloat16)
        vec = torch.randn(B, 768, generator=g, device=dev).to(torch.bfloat16)
        txt_ids = torch.zeros(B, S, 3, dtype=torch.int32, device=dev)
        x, x_ids = p._prepare_latent_images(x_T)
        tv = torch.full((B,), 1.0, dtype=torch.bfloat16, device=dev)
        gv = torch.full((B,), guidance, dtype=torch.bfloat16, device=dev)

        def step():
            pred = p._flow_step(x, x_ids, txt, txt_ids, vec, tv, gv)
            return p.sampler.step(pred, x, 1.0, 0.5)

        step_ms, x1 = timed(step)
        dec_ms, img = timed(lambda: p.decode(x1, (lat, lat)))
        tfl = B * flux_forward_flops(L, S) / 1e12
        ok = bool(torch.isfinite(img).all())
        out.append({"workload": tag, "dtype": dtype, "batch": B, "denoise_step_ms": step_ms, "vae_decode_ms": dec_ms,
                    "ms": nsteps * step_ms + dec_ms, "images_per_sec": B / ((nsteps * step_ms + dec_ms) * 1e-3),
                    "denoise_steps_per_image": nsteps, "tflop_per_step": tfl, "tflops": tfl / (step_ms * 1e-3),
                    "frac": tfl / (step_ms * 1e-3) / peak, "peak": peak, "finite": ok, "replays": reps,
                    "note": "step = graph replay of one forward incl. its modulation GEMV + Euler; ms = steps x step + decode "
                            "(fp32-faithful VAE)"})

    def c5():
        pipe.flow.enable_fp8(True)
        try:
            flux_line(pipe, "C5 per-GPU shape: Flux-schnell fp8 blocks, 1024x1024 4-step, batch 4", 4, 128, 256, 4.0, 4,
                      "fp8 e4m3 block Linears (bf16 elsewhere)", MFMA_FP8_PEAK_TFLOPS)
        finally:
            pipe.flow.enable_fp8(False)
            pipe._graphs.clear()
            pipe.flow._ws.clear()
            torch.cuda.empty_cache()

    def c3():
        dev_pipe = FluxPipeline("flux-dev", device=str(dev))
        flux_line(dev_pipe, "C3: Flux-dev 1024x1024 28-step, guidance 7, S = 512, batch 1 (ONE step + decode timed)", 1, 128, 512,
                  7.0, 28, "bf16", MFMA_BF16_PEAK_TFLOPS)
        del dev_pipe
        torch.cuda.empty_cache()

    def c4():
        from flux_generator_amd.stable_diffusion import StableDiffusionXL
        sd = StableDiffusionXL("stabilityai/sdxl-turbo", float16=True)
        B = 16
        g = torch.Generator(device=dev).manual_seed(0)
        x_T = sd.sampler.sample_prior((B, 64, 64, 4), dtype=sd.dtype, key=g, device=dev)
        cond = torch.randn(B, 77, 2048, generator=g, device=dev).to(sd.dtype)
        pooled = torch.randn(B, 1280, generator=g, device=dev).to(sd.dtype)
        tt = (pooled, torch.tensor([[512, 512, 0, 0, 512, 512.0]] * B, device=dev))
        (t, tp), = sd.sampler.timesteps(1)
        step_ms, x1 = timed(lambda: sd._denoising_step(x_T, t, tp, cond, 0.0, tt))
        dec_ms, img = timed(lambda: sd.decode(x1))
        tfl = 1.59 * B                                # TFLOP per UNet evaluation at 64x64 latents (tools/bench_sdxl.py)
        out.append({"workload": "C4: sdxl-turbo 512x512 1-step, batch 16 (UNet step + VAE decode)", "dtype": str(sd.dtype).replace("torch.", ""),
                    "batch": B, "denoise_step_ms": step_ms, "vae_decode_ms": dec_ms, "ms": step_ms + dec_ms,
                    "images_per_sec": B / ((step_ms + dec_ms) * 1e-3), "denoise_steps_per_image": 1, "tflop_per_step": tfl,
                    "tflops": tfl / (step_ms * 1e-3), "frac": tfl / (step_ms * 1e-3) / MFMA_BF16_PEAK_TFLOPS,
                    "peak": MFMA_BF16_PEAK_TFLOPS, "finite": bool(torch.isfinite(img).all()), "replays": reps,
                    "note": "UNet and latents in the arithmetic `dtype` names (float16=True: the reference's flux_app.py setting, "
                            "v_mfma_f32_16x16x32_f16 at the bf16 rate), fp32-faithful VAE decode"})
        del sd
        torch.cuda.empty_cache()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, fn in (("C5", c5), ("C3", c3), ("C4", c4)):
            try:
                fn()
            except Exception as ex:      # one side measurement failing must not lose the others (nor the headline line)
                out.append({"workload": name, "error": f"{type(ex).__name__}: {ex}"})
    return out


def relaunch_under_torchrun(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` (one rank per GPU)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def dry_run(args) -> None:
    """The N-rank job's skeleton on CPU (gloo): process group, job seed + full-batch prior slice + rank-0 conditioning
    broadcast (parallel.shard_generation_inputs), the timing barrier / max-reduce, the uint8 gather — everything but
    the kernels.  Prints the JSON line with "dry_run": true; `value` is null."""
    import torch.distributed as dist
    from flux_generator_amd import parallel
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("gloo")
    dev = torch.device("cpu")
    placement = parallel.bind_rank_to_cpus(int(os.environ.get("LOCAL_RANK", rank)), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    placements = [None] * world
    dist.all_gather_object(placements, placement)
    B, lat = args.batch, 8
    S = 512 if args.model == "flux-dev" else 256
    calls = []

    def cond():
        calls.append(rank)
        g = torch.Generator().manual_seed(1234)
        return (torch.randn(1, S, 64, generator=g) * 0.1).to(torch.bfloat16), torch.randn(1, 32, generator=g).to(torch.bfloat16)

    x_T, txt, vec, shard = parallel.shard_generation_inputs(B * world, (lat, lat, 16), 1234, dev, cond)
    assert x_T.shape[0] == B and shard == (rank * B, rank * B + B) and (calls == [0] if rank == 0 else calls == [])
    dist.barrier()
    t0 = time.perf_counter()
    img = (x_T.float().mean(dim=-1, keepdim=True).expand(-1, -1, -1, 3) * 0 + (rank + 1.5) / 255.0).contiguous()
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    allel = [torch.zeros_like(el) for _ in range(world)]
    dist.all_gather(allel, el)
    gathered = parallel.gather_images(parallel.to_uint8(img).contiguous(), B * world)
    if rank == 0:
        assert gathered.shape == (B * world, lat, lat, 3)
        assert [int(gathered[r * B, 0, 0, 0]) for r in range(world)] == list(range(1, world + 1))
        print(json.dumps({"metric": "dry run (no kernels)", "value": None, "unit": "images/sec", "n_gpus": world, "dry_run": True,
                          "steps": args.steps, "warmup": args.warmup, "scaling": "weak", "backend": "gloo",
                          "ranks_joined": world, "conditioning_evaluated_on_ranks": [0], "placement": placements,
                          "per_rank_ms": [float(t.item()) * 1e3 for t in allel],
                          "config": {"workload": "launch / sharding / collective skeleton", "global_batch": B * world}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def parity_full_size_record():
    """The committed full-size parity numbers (not re-measured by the bench: the oracle forwards take minutes of host time):
    what `pytest tests/test_full_size_parity_gpu.py` measured for this tree on a GPU box, copied to profiles/ by the round's
    profile script.  Every number is against an oracle whose parity with the MLX reference is UNPINNED (oracle/UNVERIFIED.md)."""
    for name in ("r06_parity_full_size.json", "r05_parity_full_size.json", "r04_parity_full_size.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                rec = json.load(f)
            rec["source"] = f"profiles/{name} (committed; measured by tests/test_full_size_parity_gpu.py, rel-L2 unless named otherwise)"
            rec["oracle"] = "oracle/flux_oracle.py, oracle/sd_oracle.py: CPU restatement of the reference, parity with MLX unpinned"
            return rec
    return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 1; 4 with --fp8)")
    ap.add_argument("--image-size", type=int, default=None, help="default 512; 1024 with --fp8")
    ap.add_argument("--denoise-steps", type=int, default=None, help="default 2; 4 with --fp8")
    ap.add_argument("--model", default="flux-schnell", help="flux-schnell (headline) or flux-dev (BASELINE.json configs[2])")
    ap.add_argument("--fp8", action="store_true", help="BASELINE.json configs[4]: e4m3 weights + per-token e4m3 activations on the "
                    "fp8 matrix cores; defaults become 1024x1024, 4 denoise steps, batch 4 per GPU")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C3 / C4 / C5 lines (`other_configs`) of the default N = 1 run")
    ap.add_argument("--profile-only", action="store_true", help="run a few steps for rocprofv3, print nothing else")
    ap.add_argument("--dump-plan-bytes", default=None, metavar="PATH", help="with --profile-only: write the algorithmic HBM bytes per launch "
                    "of every GEMM tile configuration of the plan (JSON) - tools/pmc_summary.py puts them next to the counter bytes")
    ap.add_argument("--guidance", type=float, default=None, help="default 4.0 (7.0 for flux-dev, BASELINE.json configs[2])")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo skeleton of the N-rank job: launch, sharding, broadcast, "
                    "barrier, max-reduce and gather, no kernels (tests/test_distributed_cpu.py)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args.gpus))
    args.batch = args.batch or (4 if args.fp8 else 1)
    args.image_size = args.image_size or (1024 if args.fp8 else 512)
    args.denoise_steps = args.denoise_steps or (4 if args.fp8 else 2)
    args.guidance = args.guidance if args.guidance is not None else (7.0 if args.model == "flux-dev" else 4.0)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"`python bench.py --gpus N` or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if args.dry_run:
        return dry_run(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_FORCE_DIST=1: create the RCCL process group even at world size 1, so a single-GPU box runs exactly the
    # broadcast / barrier / max-reduce / gather code an 8-GPU launch runs
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    share_gpu = os.environ.get("BENCH_SHARE_GPU") == "1"
    # (two processes on ONE GPU need no special mode: the reduce-scatter split-K hand-off completes without co-residency,
    # include/fluxhip.h fluxhip_gemm_set_splitk_mode)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # BENCH_SHARE_GPU=1 (diagnostic, single-GPU boxes): every rank uses cuda:0 and the collectives go over gloo (RCCL refuses
        # two ranks on one device).  Exercises the N-rank launch, sharding, barrier / max-reduce and gather with real kernels;
        # the throughput of such a run is NOT a scaling number (the ranks time-share one GPU) and the JSON says so.
        if share_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    # N > 1: pin this rank's threads to its share of the CPUs of its GPU's NUMA node (8 ranks x graph launches + OpenMP pools)
    from flux_generator_amd import parallel as _par
    placement = (_par.bind_rank_to_cpus(int(os.environ.get("LOCAL_RANK", local_rank)), int(os.environ.get("LOCAL_WORLD_SIZE", world)), dev.index)
                 if world > 1 else None)
    if placement and placement.get("cpus") and "OMP_NUM_THREADS" in os.environ:
        torch.set_num_threads(max(1, min(torch.get_num_threads(), int(placement["cpus"].split("(")[1].rstrip(")")))))

    from flux_generator_amd.flux.flux import FluxPipeline
    import warnings
    warnings.simplefilter("ignore")
    pipe = FluxPipeline(args.model, device=str(dev), use_graph=not args.no_graph)
    if args.fp8:
        pipe.flow.enable_fp8()
    peak = MFMA_FP8_PEAK_TFLOPS if args.fp8 else MFMA_BF16_PEAK_TFLOPS

    B = args.batch
    lat = args.image_size // 8
    # T5 sequence length: 256 for schnell, 512 for dev (flux/utils.py:210 of the reference; txt2image.py pads to it)
    L, S = (lat // 2) ** 2, (512 if args.model == "flux-dev" else 256)
    # synthetic conditioning, resident in HBM before the timed region.  Multi-GPU: the job is ONE
    # batch of B x world images of one prompt, sharded by image exactly like FluxPipeline.generate_latents under
    # torchrun (flux_generator_amd/parallel.py): rank 0 holds the (synthetic) T5 / CLIP embeddings and broadcasts them
    # over RCCL, every rank draws the full-batch prior from the job seed and keeps its rows.
    from flux_generator_amd import parallel

    def synthetic_conditioning():
        g = torch.Generator(device=dev).manual_seed(1234)
        return ((torch.randn(1, S, 4096, generator=g, device=dev) * 0.1).to(torch.bfloat16),
                torch.randn(1, 768, generator=g, device=dev).to(torch.bfloat16))

    x_T, txt, vec, shard = parallel.shard_generation_inputs(B * world, (lat, lat, 16), 1234, dev, synthetic_conditioning)
    assert x_T.shape[0] == B and shard == (rank * B, rank * B + B)
    txt_ids = torch.zeros(B, S, 3, dtype=torch.int32, device=dev)

    def one_pass():
        x, x_ids = pipe._prepare_latent_images(x_T)
        for x in pipe._denoising_loop(x, x_ids, txt, txt_ids, vec, num_steps=args.denoise_steps, guidance=args.guidance):
            pass
        return pipe.decode(x, (lat, lat))

    def sync_all():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        img = one_pass()
    if args.profile_only:
        for _ in range(args.steps):
            one_pass()
        torch.cuda.synchronize()
        if args.dump_plan_bytes:
            import ctypes
            from flux_generator_amd import _lib
            ws = pipe.flow._workspace(B, S, L)
            agg = {}
            for label, _, _, nb in pipe.flow.profile_plan(ws, with_bytes=True):
                a = agg.setdefault(label, [0, 0.0])
                a[0] += 1; a[1] += nb
            out = {}
            for label, (cnt, nb) in agg.items():
                ent = {"launches_per_forward": cnt, "algorithmic_MB_per_launch": nb / cnt / 1e6 if nb else None}
                mm = __import__("re").search(r"cfg(\d+)", label)
                if mm:
                    bm, bn, th = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                    if _lib.load().fluxhip_gemm_tile_shape(int(mm.group(1)), bm, bn, th) == 0:
                        ent.update(bm=bm.value, bn=bn.value)
                out[label] = ent
            with open(args.dump_plan_bytes, "w") as f:
                json.dump({"workload": f"{args.model} B{B} T{L + S}", "labels": out}, f, indent=1)
        return
    sync_all()
    t0 = time.perf_counter()
    host_us = None
    ev_start, ev_first = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_start.record()
    for i in range(args.steps):
        h0 = time.perf_counter()
        img = one_pass()
        if i == 0:
            host_us = (time.perf_counter() - h0) * 1e6     # host time to ENQUEUE one step (graph launches + glue); no sync here
            ev_first.record()                              # GPU-side end of this rank's first step after the common barrier
    sync_all()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [elapsed / args.steps * 1e3]
    rank_diag = [{"rank": rank, "first_step_ms": ev_start.elapsed_time(ev_first), "enqueue_host_us_per_step": host_us,
                  "placement": placement}]
    if use_dist:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device="cpu" if share_gpu else dev, dtype=torch.float64)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        gathered_diag = [None] * world
        dist.all_gather_object(gathered_diag, rank_diag[0])
        rank_diag = gathered_diag
    assert img.shape == (B, args.image_size, args.image_size, 3) and bool(torch.isfinite(img).all())
    # after the timed region: uint8 images of every rank gathered on rank 0 (RCCL), as the CLI does before saving
    gathered = pipe.gather_images(img, B * world)
    if rank == 0:
        assert gathered.shape == (B * world, args.image_size, args.image_size, 3) and gathered.dtype == torch.uint8

    # ---- denoise-step latency (one Flux forward + Euler) with HIP events on the launch stream
    x, x_ids = pipe._prepare_latent_images(x_T)
    tvec = torch.full((B,), 1.0, dtype=torch.bfloat16, device=dev)
    gvec = torch.full((B,), args.guidance, dtype=torch.bfloat16, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    pipe._flow_step(x, x_ids, txt, txt_ids, vec, tvec, gvec)      # capture outside the timed region (the image loop may use the other graph)
    e0.record()
    for _ in range(reps):
        pred = pipe._flow_step(x, x_ids, txt, txt_ids, vec, tvec, gvec)
        pipe.sampler.step(pred, x, 1.0, 0.5)
    e1.record()
    torch.cuda.synchronize()
    step_ms = e0.elapsed_time(e1) / reps
    # the same step as the image loop runs it: all steps' modulation tables computed up front (one pass over the
    # modulation weights per image), the per-step graph then starts at the first block
    mod_ms = step_ms_loop = None
    nst_ok = args.denoise_steps > 1
    if B <= 2 and nst_ok and not args.no_graph:
        nst = args.denoise_steps
        ts_all = pipe.sampler.timesteps(nst, x.shape[1])[:nst]
        pipe.flow.modulation_tables(ts_all, vec, gvec)
        e0.record()
        for _ in range(reps):
            mods = pipe.flow.modulation_tables(ts_all, vec, gvec)
        e1.record()
        torch.cuda.synchronize()
        mod_ms = e0.elapsed_time(e1) / reps
        pipe._flow_step(x, x_ids, txt, txt_ids, vec, tvec, gvec, mods[0])
        e0.record()
        for _ in range(reps):
            pred = pipe._flow_step(x, x_ids, txt, txt_ids, vec, tvec, gvec, mods[0])
            pipe.sampler.step(pred, x, 1.0, 0.5)
        e1.record()
        torch.cuda.synchronize()
        step_ms_loop = e0.elapsed_time(e1) / reps
    decode_ms = {}
    for prec in ("fp32", "bf16"):        # "fp32" = the reference's VAE arithmetic (fp32-faithful bf16x3 kernels): the one `value` uses
        pipe.decode(x, (lat, lat), precision=prec)
        e0.record()
        for _ in range(reps):
            pipe.decode(x, (lat, lat), precision=prec)
        e1.record()
        torch.cuda.synchronize()
        decode_ms[prec] = e0.elapsed_time(e1) / reps
    assert pipe.ae.precision == "fp32"

    # ---- roofline of the dominant kernel: per-launch HIP events over one eager pass of the plan
    ws = pipe.flow._workspace(B, S, L)
    recs = pipe.flow.profile_plan(ws, with_bytes=True)
    recs = pipe.flow.profile_plan(ws, with_bytes=True)
    by = {}
    for label, ms, fl, nb in recs:
        a = by.setdefault(label, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += fl; a[3] += nb
    dom = max(by, key=lambda k: by[k][1])
    n, ms, fl, nb = by[dom]
    ach = fl / (ms * 1e-3) / 1e12
    roofline = {"bound": "mfma", "kernel": dom, "launches_per_forward": n, "avg_launch_ms": ms / n,
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "algorithmic_flop_per_launch": fl / n,
                "algorithmic_bytes_per_launch": (nb / n) if nb else None}
    roofline.update(pmc_traffic(dom, f"{args.model} B{B} T{L + S}"))
    if roofline.get("traffic") and nb:
        # counter bytes / algorithmic bytes of the same launches: a regression in re-reads shows here (tools/pmc_summary.py writes
        # the same ratio per kernel into profiles/rNN_hbm_traffic_pmc.csv)
        roofline["traffic_ratio"] = roofline["traffic"] / (nb / n)
        floor = private_l2_floor(pipe.flow, ws, dom)
        if floor:
            roofline["traffic_floor_8_private_l2"] = floor
            roofline["traffic_over_floor"] = roofline["traffic"] / floor
            roofline["traffic_note"] = ("FETCH_SIZE / WRITE_SIZE count requests on the fabric side of the 8 per-XCD L2s, Infinity-Cache hits "
                                        "included (MI355X_MICROARCH.md, HBM): `traffic_floor_8_private_l2` is what the launch's own brick map "
                                        "must move at that boundary - every XCD its K range of the activation panel and its own weight "
                                        "panels, plus the split-K partials written through and read back - DESIGN.md 3.1")
    breakdown = {k: {"launches": v[0], "ms": round(v[1], 3), "tflops": (round(v[2] / (v[1] * 1e-3) / 1e12, 1) if v[2] else None)}
                 for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])}

    if rank == 0:
        total_images = world * args.steps * B
        fwd_tflop = flux_forward_flops(L, S) / 1e12
        out = {
            "metric": f"images/sec, {args.model.replace('flux-', 'Flux-')} {args.image_size}x{args.image_size} {args.denoise_steps}-step (denoise-step ms in config)",
            "value": total_images / elapsed, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "per_rank_ms_per_step": {"min": min(per_rank_ms), "max": max(per_rank_ms), "ranks": per_rank_ms},
            "first_step_skew_ms": max(d["first_step_ms"] for d in rank_diag) - min(d["first_step_ms"] for d in rank_diag),
            "per_rank": rank_diag,
            "collectives": (("gloo; ALL RANKS SHARE ONE GPU (BENCH_SHARE_GPU diagnostic): not a scaling measurement" if share_gpu else
                             f"RCCL {'.'.join(str(v) for v in torch.cuda.nccl.version())} (torch.distributed nccl backend)") if use_dist else None),
            "dtype": "fp8 e4m3 (block Linears: weights per-channel; activations per-token out of LayerNorm, block-scaled (E8M0 per 32) out of attention / GELU; residual stream / attention / VAE as in bf16 mode)" if args.fp8 else "bf16",
            "data": "synthetic",
            "config": {"workload": f"{args.model.replace('flux-', 'Flux-')} {args.image_size}x{args.image_size} {args.denoise_steps}-step, "
                                   f"batch {B}/GPU, random-init weights, synthetic x_T/txt/vec resident in HBM; "
                                   f"per step: {args.denoise_steps} x (Flux forward + Euler) + VAE decode (fp32-faithful)",
                       "global_batch": B * world, "parallelism": f"dp{world} (batch sharded by image; txt/vec broadcast from rank 0 before and uint8 gather after the timed region, no collective inside)",
                       "denoise_step_ms": step_ms, "denoise_step_ms_in_loop": step_ms_loop, "modulation_tables_ms_per_image": mod_ms, "vae_decode_ms": decode_ms["fp32"],
                       "vae_precision": "fp32-faithful, like the reference's fp32 AE (bf16 hi/lo planes, 3 MFMA passes, fp32 accumulate / norms / softmax)",
                       "vae_decode_ms_bf16_storage_optin": decode_ms["bf16"],
                       "flux_forward_tflop_per_image": fwd_tflop, "denoise_mfma_frac": B * fwd_tflop / (step_ms * 1e-3) / peak,
                       "denoise_mfma_frac_in_loop": (B * fwd_tflop / ((step_ms_loop + mod_ms / args.denoise_steps) * 1e-3) / peak) if step_ms_loop else None,
                       "hip_graph": not args.no_graph, "kernel_breakdown_one_forward": breakdown,
                       "parity_full_size": parity_full_size_record()},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_sample(L + S, torch.get_num_threads(), S)
        headline = args.model == "flux-schnell" and not args.fp8 and args.image_size == 512 and B == 1
        if world == 1 and headline and not args.no_other_configs and not args.no_graph:
            try:
                out["other_configs"] = time_other_configs(pipe, dev)
            except Exception as ex:      # the headline line must survive a failure of the side measurements
                out["other_configs"] = [{"error": f"{type(ex).__name__}: {ex}"}]
    # RCCL prints its version banner through C stdio (flushed at exit when stdout is a pipe).  Every rank flushes it
    # now, then a barrier, then rank 0 prints: the JSON line is the LAST line of the job's output.
    def flush_c_stdio():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    flush_c_stdio()
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()
    if rank == 0:
        flush_c_stdio()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

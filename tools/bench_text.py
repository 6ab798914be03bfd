#!/usr/bin/env python3
"""Text-encode step of the Flux pipelines (flux/flux.py:73-85; the tensors rank 0 broadcasts over RCCL): T5-XXL encoder
(24 layers, d_model 4096, 64 x 64 heads, d_ff 10240) at S = 256 (schnell) / 512 (dev) and CLIP-L (12 layers, 768) at 77 tokens,
random-init weights, tokens resident on the host like the reference's.  Prints one JSON line: ms per prompt (HIP events,
eager launches: the towers run once per job, there is no graph), algorithmic TFLOP and the fraction of the dense bf16 MFMA peak."""
import json, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("FLUX_ALLOW_RANDOM_INIT", "1")
warnings.simplefilter("ignore")
from flux_generator_amd.flux.utils import load_clip, load_t5

PEAK = 2500.0
dev = torch.device("cuda:0")
t5, clip = load_t5("flux-dev", device=dev), load_clip("flux-dev", device=dev)
out = {"workload": "Flux text towers: T5-XXL encoder + CLIP-L, batch 1, random-init, bf16", "peak_tflops": PEAK}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t5_flop(S, L=24, D=4096, F=10240, H=64):
    return L * (2 * S * D * (4 * D + 3 * F) + 4 * S * S * D)


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for S in (256, 512):
    tok = torch.randint(2, 32000, (1, S), generator=torch.Generator().manual_seed(S))
    ms = timed(lambda: t5(tok))
    tf = t5_flop(S) / 1e12
    out[f"t5_xxl_S{S}"] = {"ms": ms, "tflop": tf, "tflops": tf / (ms * 1e-3), "frac": tf / (ms * 1e-3) / PEAK}
ctok = torch.randint(1, 49000, (1, 77), generator=torch.Generator().manual_seed(1))
ctok[:, 0], ctok[:, 20:] = 49406, 49407
ms = timed(lambda: clip(ctok))
tf = 12 * (2 * 77 * 768 * 12 * 768 + 4 * 77 * 77 * 768) / 1e12
out["clip_l_77"] = {"ms": ms, "tflop": tf, "tflops": tf / (ms * 1e-3), "frac": tf / (ms * 1e-3) / PEAK}
out["encode_ms_per_prompt_schnell"] = out["t5_xxl_S256"]["ms"] + out["clip_l_77"]["ms"]
out["encode_ms_per_prompt_dev"] = out["t5_xxl_S512"]["ms"] + out["clip_l_77"]["ms"]
out["note"] = ("weights of T5-XXL are 9.4 GB bf16: at batch 1 the encoder is HBM-bound on them (>= 1.5 ms at 6.3 TB/s); "
               "eager launches, host-bound below ~12 us per kernel")
print(json.dumps(out))

#!/usr/bin/env python3
"""BASELINE.json config C4: stabilityai/sdxl-turbo 512x512 1-step, batch 16, 1 x MI355X (random-init weights,
synthetic conditioning resident in HBM).  Prints one JSON line (images/sec, UNet step ms, decode ms)."""
import json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("FLUX_ALLOW_RANDOM_INIT", "1")
warnings.simplefilter("ignore")
from flux_generator_amd.stable_diffusion import StableDiffusionXL

B = int(os.environ.get("SDXL_BATCH", "16"))
steps = int(os.environ.get("SDXL_ITERS", "5"))
dev = torch.device("cuda:0")
# SDXL_F32=1: the reference's DEFAULT float16=False = float32 arithmetic on the float32-faithful kernels
pipe = StableDiffusionXL("stabilityai/sdxl-turbo", float16=os.environ.get("SDXL_F32") != "1")
g = torch.Generator(device=dev).manual_seed(0)
x_T = pipe.sampler.sample_prior((B, 64, 64, 4), dtype=pipe.dtype, key=g, device=dev)
cond = torch.randn(B, 77, 2048, generator=g, device=dev).to(pipe.dtype)
pooled = torch.randn(B, 1280, generator=g, device=dev).to(pipe.dtype)
tt = (pooled, torch.tensor([[512, 512, 0, 0, 512, 512.0]] * B, device=dev))
(t, tp), = pipe.sampler.timesteps(1)

def one():
    x = pipe._denoising_step(x_T, t, tp, cond, 0.0, tt)
    return pipe.decode(x)

for _ in range(2):
    img = one()
torch.cuda.synchronize()
if os.environ.get("PROFILE_ONLY"):
    one(); torch.cuda.synchronize(); sys.exit(0)
t0 = time.perf_counter()
for _ in range(steps):
    img = one()
torch.cuda.synchronize()
el = time.perf_counter() - t0
e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
pipe.decode(x_T, precision="bf16")
e0.record(); x = pipe._denoising_step(x_T, t, tp, cond, 0.0, tt); e1.record(); pipe.decode(x); e2.record()
pipe.decode(x, precision="bf16"); e3.record()
torch.cuda.synchronize()
unet_flop = 1.59e12 * B
print(json.dumps({"workload": f"sdxl-turbo 512x512 1-step batch {B}", "unet_dtype": str(pipe.dtype), "images_per_sec": B * steps / el,
                  "unet_step_ms": e0.elapsed_time(e1), "vae_decode_ms": e1.elapsed_time(e2),
                  "vae_precision": "fp32-faithful (the reference's float32 VAE)", "vae_decode_ms_bf16_storage_optin": e2.elapsed_time(e3),
                  "unet_tflops": unet_flop / (e0.elapsed_time(e1) * 1e-3) / 1e12, "finite": bool(torch.isfinite(img).all())}))

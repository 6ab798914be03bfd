#!/usr/bin/env python3
"""SD 2.1-base 512x512 with classifier-free guidance (UNet batch 2 per image), 1 x MI355X, random-init weights, synthetic
conditioning: UNet step ms (hipGraph replay), images/sec for a 50-step run incl. the float32-faithful decode."""
import json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("FLUX_ALLOW_RANDOM_INIT", "1")
warnings.simplefilter("ignore")
from flux_generator_amd.stable_diffusion import StableDiffusion

B = int(os.environ.get("SD_BATCH", "1")); NSTEP = int(os.environ.get("SD_STEPS", "50"))
dev = torch.device("cuda:0")
pipe = StableDiffusion("stabilityai/stable-diffusion-2-1-base", float16=True)
g = torch.Generator(device=dev).manual_seed(0)
x_T = pipe.sampler.sample_prior((B, 64, 64, 4), dtype=pipe.dtype, key=g, device=dev)
cond = torch.randn(2 * B, 77, 1024, generator=g, device=dev).to(pipe.dtype)      # [text, negative]
def run():
    x = x_T
    for x in pipe._denoising_loop(x_T, pipe.sampler.max_time, cond, NSTEP, 7.5):
        pass
    return pipe.decode(x)
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); img = run(); torch.cuda.synchronize(); el = time.perf_counter() - t0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
steps = pipe.sampler.timesteps(NSTEP)
e0.record()
x = x_T
for i, (t, tp) in enumerate(steps[:10]):
    x = pipe._denoising_step(x, t, tp, cond, 7.5, new_conditioning=(i == 0))
e1.record(); torch.cuda.synchronize()
print(json.dumps({"workload": f"sd-2.1-base 512x512 {NSTEP}-step CFG 7.5 batch {B}", "images_per_sec": B / el, "unet_step_ms": e0.elapsed_time(e1) / 10,
                  "finite": bool(torch.isfinite(img).all())}))

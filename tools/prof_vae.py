#!/usr/bin/env python3
"""Flux VAE decode only (for rocprofv3): 5 eager decodes of one 64x64x16 latent.  VAE_PREC=fp32|bf16 (default fp32 =
the reference's arithmetic on the fp32-faithful split-bf16 kernels), VAE_B = batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
os.environ.setdefault("FLUX_ALLOW_RANDOM_INIT", "1")
from flux_generator_amd.flux.utils import load_ae
ae = load_ae("flux-schnell", device="cuda")
prec = os.environ.get("VAE_PREC", "fp32")
x = torch.randn(int(os.environ.get("VAE_B", "1")), 1024, 64, device="cuda").to(torch.bfloat16)
for _ in range(5):
    y = ae.decode_packed(x, (64, 64), precision=prec)
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ with the CPU oracle (fp32 math, bf16-representable
weights and inputs).  The reference itself cannot run here (no MLX), so these are goldens OF THE
ORACLE: they freeze its behaviour and give the HIP path fixed targets that do not depend on the
oracle code at test time.  Weights are regenerated from a seed (not stored) to keep fixtures small.

    python tests/golden/make_golden.py        # rewrites the .pt files
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import flux_oracle as O  # noqa: E402

BF = torch.bfloat16


def flux_cfg(guidance):
    return dict(in_channels=64, vec_in_dim=64, context_in_dim=128, hidden_size=256, mlp_ratio=4.0, num_heads=2,
                depth=1, depth_single_blocks=2, axes_dim=[16, 56, 56], theta=10_000, qkv_bias=True,
                guidance_embed=guidance)


VAE_CFG = dict(resolution=32, in_channels=3, ch=128, out_ch=3, ch_mult=[1, 2], num_res_blocks=1, z_channels=16,
               scale_factor=0.3611, shift_factor=0.1159)


def flux_weights(cfg, seed):
    P = O.FluxParams(**cfg)
    W = O.init_weights(O.flux_weight_shapes(P), seed=seed, norm_jitter=0.2)
    return P, {k: v.to(BF).float() for k, v in W.items()}


def vae_weights(seed):
    A = O.AutoEncoderParams(**VAE_CFG)
    W = O.init_weights(O.decoder_weight_shapes(A), seed=seed, norm_jitter=0.2)
    return A, {k: v.to(BF).float() for k, v in W.items()}


def _flux_inputs(cfg, B, S, h, w, steps, guidance, name, seed):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(B, h, w, 16, generator=g).to(BF)
    txt = (torch.randn(B, S, cfg["context_in_dim"], generator=g) * 0.5).to(BF)
    vec = torch.randn(B, cfg["vec_in_dim"], generator=g).to(BF)
    return dict(cfg=cfg, weight_seed=seed + 100, z=z, txt=txt, vec=vec, steps=steps, guidance=guidance, name=name,
                latent=(h, w))


def run_flux(inp):
    P, W = flux_weights(inp["cfg"], inp["weight_seed"])
    x, ids = O.prepare_latent_images(inp["z"])
    B, S = inp["txt"].shape[:2]
    tids = torch.zeros(B, S, 3, dtype=torch.int32)
    lat = O.denoising_loop(P, W, inp["name"], x.float(), ids, inp["txt"].float(), tids, inp["vec"].float(),
                           inp["steps"], inp["guidance"])
    out = {f"x_step{i}": v for i, v in enumerate(lat)}
    out["timesteps"] = torch.tensor(O.timesteps(inp["name"], inp["steps"], x.shape[1]))
    return out


def run_vae(inp):
    A, W = vae_weights(inp["weight_seed"])
    return {"image": O.pipeline_decode(A, W, inp["x"], inp["latent"])}


CASES = {"flux_tiny_schnell": run_flux, "flux_tiny_dev": run_flux, "vae_tiny": run_vae}


def make_inputs():
    g = torch.Generator().manual_seed(77)
    return {
        "flux_tiny_schnell": _flux_inputs(flux_cfg(False), 1, 16, 8, 8, 2, 4.0, "flux-schnell", 1),
        "flux_tiny_dev": _flux_inputs(flux_cfg(True), 2, 24, 8, 12, 3, 3.5, "flux-dev", 2),
        "vae_tiny": dict(weight_seed=9, latent=(8, 8), x=torch.randn(1, 16, 64, generator=g).to(BF)),
    }


if __name__ == "__main__":
    for name, inp in make_inputs().items():
        exp = CASES[name](inp)
        torch.save({"inputs": inp, "expected": exp}, os.path.join(HERE, f"{name}.pt"))
        print(name, {k: tuple(v.shape) for k, v in exp.items()})

#!/usr/bin/env python3
"""In-situ A/B of tile | split-K choices for the block GEMMs of the Flux launch plan (GPU box).

For every candidate `N x K = tile | splits << 8` (FLUXHIP_PLAN_TILES, read when a plan is built) the full-size model's plan
is rebuilt and run eagerly with a HIP event pair around every launch (Flux.profile_plan); the summed time of the launches
of that N x K is reported next to the library's own pick.  In situ = with the real operand traffic around the launch —
split-K hand-offs in particular do not time the same in a weight-rotating micro-benchmark.

usage: python tools/plan_sweep.py [--model flux-schnell] [--size 512] [--batch 1] "3072x3072=47,819,564" "3072x15360=819,1073"
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="flux-schnell")
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("sweeps", nargs="+")
a = ap.parse_args()

from flux_generator_amd.flux.model import Flux
from flux_generator_amd.flux.utils import configs
P = configs[a.model].params
model = Flux(P, device="cuda").init_random(0)
B, S, L = a.batch, (512 if a.model == "flux-dev" else 256), (a.size // 16) ** 2


def measure(shape_key):
    model._ws.clear()
    ws = model._workspace(B, S, L)
    for k in ("in_img", "in_txt", "in_y", "x"):
        ws[k].normal_()
    ws["in_t"].fill_(0.5)
    ws["in_g"].fill_(4.0)
    ws["in_ids"].zero_()
    model.profile_plan(ws)
    tot, lab, n, whole = 0.0, None, 0, 0.0
    for _ in range(a.reps):
        for label, ms, fl in model.profile_plan(ws, with_shape=True):
            whole += ms
            if label.endswith(shape_key):
                tot += ms; n += 1; lab = label.split()[0]
    return tot / a.reps, n // a.reps, lab, whole / a.reps


for sw in a.sweeps:
    shape, cands = sw.split("=")
    N, K = (int(v) for v in shape.split("x"))
    key = f" N{N} K{K}"
    os.environ.pop("FLUXHIP_PLAN_TILES", None)
    ms, n, lab, whole = measure(key)
    fl = 2.0 * B * (S + L) * N * K * n
    print(f"{shape}: auto -> {lab}: {n} launches {ms:.3f} ms ({fl / ms / 1e9:.0f} TFLOP/s), forward {whole:.2f} ms", flush=True)
    for c in cands.split(","):
        os.environ["FLUXHIP_PLAN_TILES"] = f"{N}x{K}={c}"
        try:
            ms, n, lab, whole = measure(key)
            print(f"    forced {int(c) & 255}s{int(c) >> 8}: {ms:.3f} ms ({fl / ms / 1e9:.0f} TFLOP/s), forward {whole:.2f} ms", flush=True)
        except Exception as e:
            print(f"    forced {c}: {type(e).__name__}: {e}", flush=True)
    os.environ.pop("FLUXHIP_PLAN_TILES", None)

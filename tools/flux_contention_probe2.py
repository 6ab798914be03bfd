#!/usr/bin/env python3
"""Diagnostic: do 10-forward dependent chains of the full-size Flux model keep their bits while a co-tenant process uses the
GPU?  Co-tenant kinds: "torch" (plain elementwise torch kernels, no libfluxhip), "flux" (the same forward, synchronised every
forward so its queue stays short), "fluxrun" (forwards enqueued without synchronisation).  usage: <kind> | hammer <kind>"""
import os, subprocess, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
BF = torch.bfloat16
hammer = sys.argv[1] == "hammer"
kind = sys.argv[2] if hammer else sys.argv[1]
secs = float(os.environ.get("HAMMER_SECONDS", "40"))
if hammer and kind == "torch":
    a = torch.randn(64 << 20, device=dev)
    open(os.environ["HAMMER_READY"], "w").close()
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        for _ in range(20):
            a.mul_(1.0001).add_(0.001)
        torch.cuda.synchronize()
        n += 40
    print(f"hammer(torch): {n} kernels in {time.time() - t0:.1f} s", flush=True)
    sys.exit(0)

from flux_generator_amd.flux.model import Flux
from flux_generator_amd.flux.utils import configs
P = configs["flux-schnell"].params
model = Flux(P, device=dev).init_random(0)
g = torch.Generator().manual_seed(3)
B, S, L = 1, 256, 1024
img = torch.randn(B, L, 64, generator=g).to(BF).to(dev)
txt = (torch.randn(B, S, P.context_in_dim, generator=g) * 0.5).to(BF).to(dev)
vec = torch.randn(B, P.vec_in_dim, generator=g).to(BF).to(dev)
ii, jj = torch.meshgrid(torch.arange(32, dtype=torch.int32), torch.arange(32, dtype=torch.int32), indexing="ij")
img_ids = torch.stack([torch.zeros_like(ii), ii, jj], dim=-1).reshape(1, L, 3).to(dev)
txt_ids = torch.zeros(B, S, 3, dtype=torch.int32, device=dev)
t = torch.full((B,), 0.5, dtype=BF, device=dev)
model(img, img_ids, txt, txt_ids, t, vec)
torch.cuda.synchronize()
if hammer:
    open(os.environ["HAMMER_READY"], "w").close()
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        model(img, img_ids, txt, txt_ids, t, vec)
        n += 1
        if kind == "flux" or n % 20 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"hammer({kind}): {n} forwards in {time.time() - t0:.1f} s", flush=True)
    sys.exit(0)


def chain(n=10):
    x, outs = img, []
    for i in range(n):
        pred = model(x, img_ids, txt, txt_ids, t, vec)
        x = (img + 0.25 * pred).to(BF)
        outs.append(pred.clone())
    torch.cuda.synchronize()
    return outs


ref = chain()
print("alone repeatable:", all(torch.equal(p, q) for p, q in zip(ref, chain())), flush=True)
ready = f"/tmp/hammer_ready_{kind}"
if os.path.exists(ready):
    os.remove(ready)
child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "hammer", kind], env=dict(os.environ, HAMMER_READY=ready))
while not os.path.exists(ready):
    time.sleep(0.1)
    assert child.poll() is None, "hammer died"
time.sleep(1.0)
t0 = time.time()
for r in range(8):
    got = chain()
    bad = [i for i in range(10) if not torch.equal(got[i], ref[i])]
    mag = max((float((got[i].float() - ref[i].float()).abs().max()) for i in bad), default=0.0)
    print(f"[{kind}] chain {r} at {time.time() - t0:5.1f} s (hammer alive: {child.poll() is None}): first differing forward "
          f"{bad[0] if bad else None}, {len(bad)} differ, max |d| {mag:.3g}", flush=True)
print("hammer exit code:", child.wait(), flush=True)
print("alone again:", all(torch.equal(p, q) for p, q in zip(ref, chain())), flush=True)

#!/usr/bin/env python3
"""Diagnostic: is bit-repeatability under GPU sharing a property of libfluxhip's kernels or of the platform?
  A: main = dependent chain of Flux forwards,          co-tenant = torch.matmul loop (hipBLASLt kernels)
  B: main = dependent chain of torch.matmul + torch ops, co-tenant = Flux forward loop
usage: A | B | hammer <matmul|flux>"""
import os, subprocess, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
BF = torch.bfloat16
mode = sys.argv[1]
secs = float(os.environ.get("HAMMER_SECONDS", "35"))


def flux_setup():
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
    fwd = lambda x: model(x, img_ids, txt, txt_ids, t, vec)      # noqa: E731
    fwd(img)
    torch.cuda.synchronize()
    return fwd, img


def matmul_setup():
    g = torch.Generator().manual_seed(1)
    Ws = [(torch.randn(4096, 4096, generator=g) / 64).to(BF).to(dev) for _ in range(8)]
    x0 = torch.randn(2048, 4096, generator=g).to(BF).to(dev)

    def fwd(x):
        for W in Ws * 6:
            x = torch.nn.functional.gelu(x @ W)
            x = x + 0.5
        return x
    fwd(x0)
    torch.cuda.synchronize()
    return fwd, x0


if mode == "hammer":
    fwd, x0 = flux_setup() if sys.argv[2] == "flux" else matmul_setup()
    open(os.environ["HAMMER_READY"], "w").close()
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        fwd(x0)
        torch.cuda.synchronize()
        n += 1
    print(f"hammer({sys.argv[2]}): {n} forwards in {time.time() - t0:.1f} s", flush=True)
    sys.exit(0)

fwd, x0 = flux_setup() if mode == "A" else matmul_setup()


def chain(n=10):
    x, outs = x0, []
    for i in range(n):
        y = fwd(x)
        x = (x0 + 0.25 * y[..., : x0.shape[-1]]).to(BF) if mode == "A" else y
        outs.append(y.clone())
    torch.cuda.synchronize()
    return outs


ref = chain()
print(f"[{mode}] alone repeatable:", all(torch.equal(p, q) for p, q in zip(ref, chain())), flush=True)
ready = f"/tmp/hammer_ready_p{mode}"
if os.path.exists(ready):
    os.remove(ready)
child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "hammer", "matmul" if mode == "A" else "flux"], env=dict(os.environ, HAMMER_READY=ready))
while not os.path.exists(ready):
    time.sleep(0.1)
    assert child.poll() is None, "hammer died"
time.sleep(1.0)
t0 = time.time()
for r in range(8):
    got = chain()
    bad = [i for i in range(10) if not torch.equal(got[i], ref[i])]
    print(f"[{mode}] chain {r} at {time.time() - t0:5.1f} s (hammer alive: {child.poll() is None}): first differing {bad[0] if bad else None}, {len(bad)} differ", flush=True)
print("hammer exit code:", child.wait(), flush=True)

#!/usr/bin/env python3
"""Diagnostic: the reduce-scatter split-K GEMM of the Flux plan (1280 x 3072 x 12288 / 15360, 256x192 tile, S = 3, gate-residual
epilogue) while ANOTHER process keeps the GPU busy with the same launches (neither grid is resident as a whole: the hand-off's
bounded poll + orphan completion carries both).  Every result is compared with the bits of the uncontended run; the first
mismatch is located (rows / columns / magnitude).   usage: rs_contention_probe.py [iters] [hammer]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flux_generator_amd import _lib, ops

dev = torch.device("cuda:0")
BF = torch.bfloat16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
hammer = len(sys.argv) > 2 and sys.argv[2] == "hammer"


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(dev)


M, N = 1280, 3072
probs = []
for K in (12288, 15360):
    probs.append((rnd(M, K, seed=K), rnd(N, K, seed=K + 1, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4), rnd(N, seed=5)))
run = lambda p: ops.linear(p[0], p[1], p[2], epi=ops.EPI_GATE_RES, res=p[3], gate=p[4])      # noqa: E731
if hammer:
    t0 = time.time()
    n = 0
    open(os.environ["HAMMER_READY"], "w").close()
    while time.time() - t0 < float(os.environ.get("HAMMER_SECONDS", "60")):
        for _ in range(50):
            for p in probs:
                run(p)
        n += 100
        torch.cuda.synchronize()
    print(f"hammer: {n} launches in {time.time() - t0:.1f} s", flush=True)
    sys.exit(0)

lib = _lib.load()
want = [run(p).clone() for p in probs]
torch.cuda.synchronize()
for it in range(50):                       # uncontended repeatability first
    for p, w in zip(probs, want):
        assert torch.equal(run(p), w), "not repeatable even alone"
print("alone: 100 launches repeatable")
ready = "/tmp/hammer_ready"
if os.path.exists(ready):
    os.remove(ready)
child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "0", "hammer"], env=dict(os.environ, HAMMER_SECONDS="30", HAMMER_READY=ready))
while not os.path.exists(ready):
    time.sleep(0.05)
    assert child.poll() is None, "hammer died"
time.sleep(0.5)
bad = 0
t0 = time.time()
for it in range(iters):
    for pi, (p, w) in enumerate(zip(probs, want)):
        y = run(p)
        if not torch.equal(y, w):
            bad += 1
            if bad <= 3:
                d = (y.float() - w.float()).abs()
                rows = torch.nonzero(d.amax(dim=1)).flatten()
                cols = torch.nonzero(d.amax(dim=0)).flatten()
                print(f"iter {it} problem {pi}: {int((d > 0).sum())} elements differ, max |d| {float(d.max()):.4g}; rows {int(rows.min())}..{int(rows.max())} "
                      f"({rows.numel()}), cols {int(cols.min())}..{int(cols.max())} ({cols.numel()})")
                # per 256x192 tile and 16-col fragment: where?
                tiles = {}
                nz = torch.nonzero(d)
                for r, c in nz[:: max(1, nz.shape[0] // 2000)].tolist():
                    tiles.setdefault((r // 256, c // 192), set()).add(((r % 256) // 16, (c % 192) // 16))
                for k, v in sorted(tiles.items())[:6]:
                    print("   tile", k, "fragments (row16, col16):", sorted(v)[:24])
torch.cuda.synchronize()
print(f"contended: {bad} of {2 * iters} launches differ from the uncontended bits; {time.time() - t0:.1f} s")
child.wait()

#!/usr/bin/env python3
"""Stress of the reduce-scatter split-K hand-off against STALE reads (GPU box): the same split-K launch over and over on the same
tiles - the same workgroup-to-CU placement, the same slab addresses, a consumer whose caches are warm with the previous launch's
partials - with the activation operand changing every launch, compared bit for bit with the result of the same launch on a quiet chip with cold caches.  A partial read from a cache instead of from its producer shows up as a mismatch.  A second stream keeps the chip's
memory system busy (uneven load).
What it can and cannot see: a build made on purpose with PLAIN (L1-cached) loads of the partials and no acquire passes it as well - a
131 KB stream per block evicts itself from the 32 KiB L1 before it is re-read (the MI355X guide says as much: ">= 64 KB streaming reads
apparently self-evict").  So this is a guard for the L2 / fabric side of the hand-off and for its determinism, NOT evidence about the L1
side; that rests on the form itself (write-through stores + sc1 loads: cdna guide, Guideline 16 and correctness table).
usage: rs_stale_stress.py [launches]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flux_generator_amd import ops, _lib

lib = _lib.load()
dev = torch.device("cuda:0")
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
g = torch.Generator(device=dev).manual_seed(0)
M, N, K = 1280, 3072, 15360
xs = [torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16) for _ in range(5)]
w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
res = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
gate = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
# reference bits: the reduce-scatter launch itself on a QUIET chip with cold caches (512 MiB written between launches), three times per
# operand (must agree bit for bit), and within bf16 rounding of the chain hand-off's result (another fp32 summation order)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
lib.fluxhip_gemm_set_splitk_mode(0)
refs = []
for x in xs:
    outs = []
    for rep in range(3):
        flush.fill_(rep)
        torch.cuda.synchronize()
        o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.linear(x, w, b, out=o, epi=ops.EPI_GATE_RES, res=res, gate=gate)
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "the quiet-chip reference is not reproducible"
    refs.append(outs[0])
lib.fluxhip_gemm_set_splitk_mode(1)
for x, r in zip(xs, refs):
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.linear(x, w, b, out=o, epi=ops.EPI_GATE_RES, res=res, gate=gate)
    rel = float((o.float() - r.float()).norm() / r.float().norm())
    assert rel < 2e-3, f"reduce-scatter reference vs chain: {rel}"
lib.fluxhip_gemm_set_splitk_mode(0)
n_rs0 = lib.fluxhip_gemm_rs_launches()
side = torch.cuda.Stream()
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
bad_t = torch.zeros((), dtype=torch.int64, device=dev)       # launches whose result differs, counted on the device (no host sync in the loop)
first_bad = torch.full((), -1, dtype=torch.int64, device=dev)
for it in range(n_launch):
    k = (it * 2 + it // 7) % 5
    if it % 3 == 0:
        with torch.cuda.stream(side):
            big[: (64 + 37 * (it % 5)) << 20].add_(1)          # uneven background traffic
    ops.linear(xs[k], w, b, out=out, epi=ops.EPI_GATE_RES, res=res, gate=gate)
    ne = (out != refs[k]).any()
    bad_t += ne
    first_bad = torch.where((first_bad < 0) & ne, torch.full_like(first_bad, it), first_bad)
torch.cuda.synchronize()
bad = int(bad_t)
print(f"{n_launch} reduce-scatter launches ({lib.fluxhip_gemm_rs_launches() - n_rs0} took the hand-off), {bad} differ from the quiet-chip result"
      + (f" (first: launch {int(first_bad)})" if bad else ""))
sys.exit(1 if bad else 0)

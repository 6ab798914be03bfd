#!/usr/bin/env python3
"""Fit the tile picker's time model  time = rounds x (K/64 x t_step + t_fixed) [+ split-K hop]  to gemm_tune.py sweeps.
usage: fit_tiles.py sweep_m1280.txt sweep_m4352.txt   (lines as printed by tools/gemm_tune.py)"""
import re, sys
import numpy as np
SHAPES = {"qkv(2grp)": (9216, 3072), "proj(2grp)": (3072, 3072), "mlp0(2grp)": (12288, 3072), "mlp2(2grp)": (3072, 12288),
          "linear1": (21504, 3072), "linear2": (3072, 15360)}
src = open("/root/repo/flux_generator_amd/csrc/gemm.hip").read()
cf = re.findall(r"make_cfg<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, \d+)?>\(\)", src)
cfgs = {i + 1: tuple(int(v) for v in c) for i, c in enumerate(cf)}
data = {}
for path in sys.argv[1:]:
    for line in open(path):
        m = re.match(r"(\S+)\s+M=(\d+) N=(\d+) K=(\d+):(.*)BEST", line)
        if not m:
            continue
        M, N, K = int(m.group(2)), int(m.group(3)), int(m.group(4))
        for c, v in re.findall(r"c(\d+)=([\d.]+)", m.group(5)):
            data.setdefault(int(c), []).append((M, N, K, float(v)))
for c, rows in sorted(data.items()):
    bm, bn, wm, wn, ns, pipe = cfgs[c]
    lds = (ns * bm + (ns + 1 if pipe >= 3 else ns) * bn) * 128
    bpc = 2 if lds <= 80 * 1024 else 1
    A, y = [], []
    for M, N, K, tf in rows:
        tm = sum((g + bm - 1) // bm for g in (256, M - 256))
        t = tm * ((N + bn - 1) // bn)
        rounds = -(-t // (256 * bpc))
        A.append([rounds * K / 64, rounds]); y.append(2.0 * M * N * K / (tf * 1e12) * 1e6)
    A, y = np.array(A), np.array(y)
    sol, *_ = np.linalg.lstsq(A / y[:, None], np.ones(len(y)), rcond=None)
    err = np.abs(A @ sol / y - 1)
    print(f"cfg {c:2d} {bm}x{bn} pipe {pipe} bpc {bpc}: t_step {sol[0]:.3f} us  t_fixed {sol[1]:5.2f} us   err mean {err.mean()*100:.0f}% max {err.max()*100:.0f}%")

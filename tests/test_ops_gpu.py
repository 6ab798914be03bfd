"""Per-kernel parity: libfluxhip (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (floating point; the path computes in bf16 storage / fp32 accumulate):
  * vs the oracle run in fp32 on the SAME bf16-rounded inputs: rel-L2 <= 4e-3 for one op
    (bf16 output rounding alone is ~2e-3 rel-L2), exact for pure data movement;
  * index/permutation kernels (pack / unpack) are bit-exact.
"""

import pytest
import torch

from conftest import rel_l2
from oracle import flux_oracle as O

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
TOL = 4e-3


def rnd(*shape, scale=1.0, seed=0, dev="cuda"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(dev)


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 256), (200, 136, 128), (1280, 3072, 512),
                                   (37, 64, 64), (1024, 64, 3072)])
@pytest.mark.parametrize("cfg", [c for c in range(0, 58) if c not in (34, 35, 42, 48, 56)])   # 34/35/42: duplicates; 48 / 56: phase-timed diagnostics
def test_gemm_bias(dev, M, N, K, cfg):
    from flux_generator_amd import ops
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = ops.linear(x, w, b, tile_cfg=cfg)
    ref = O.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())
    assert rel_l2(y, ref) < TOL


def test_gemm_is_not_transposed(dev):
    """A = I check with an asymmetric W (catches row/col swaps of the MFMA C layout)."""
    from flux_generator_amd import ops
    K = 128
    x = torch.eye(K, dtype=BF, device=dev)
    w = (torch.arange(192 * K, dtype=torch.float32).reshape(192, K) % 251 - 125).to(BF).to(dev)
    y = ops.linear(x, w)
    assert torch.equal(y.float().cpu(), w.float().cpu().t())


def test_gemm_epilogues(dev):
    from flux_generator_amd import ops
    M, N, K = 320, 256, 192
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    xf, wf, bf = x.float().cpu(), w.float().cpu(), b.float().cpu()
    lin = O.linear(xf, wf, bf)
    assert rel_l2(ops.linear(x, w, b, epi=ops.EPI_GELU_TANH), O.gelu_tanh(lin)) < TOL
    assert rel_l2(ops.linear(x, w, b, epi=ops.EPI_SILU), O.silu(lin)) < TOL
    res, gate = rnd(M, N, seed=4), rnd(N, seed=5)
    out = ops.linear(x, w, b, epi=ops.EPI_GATE_RES, res=res, gate=gate)
    assert rel_l2(out, res.float().cpu() + gate.float().cpu() * lin) < TOL
    # in-place residual (C aliases res), as the denoise path uses it
    buf = res.clone()
    ops.linear(x, w, b, epi=ops.EPI_GATE_RES, out=buf, res=buf, gate=gate)
    assert torch.equal(buf, out)


@pytest.mark.parametrize("N", [260, 512])
@pytest.mark.parametrize("cfg", [0, 4, 36, 40, 44, 50, 57])
def test_gemm_epilogue_paths(dev, N, cfg):
    """N = 260 is only 8-byte addressable per row -> the direct-store epilogue; N = 512 goes through
    the LDS-transposed 16-byte epilogue.  Both must give the same fused results."""
    from flux_generator_amd import ops
    M, K = 300, 128
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    lin = O.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())
    res, gate = rnd(M, N, seed=4), rnd(N, seed=5)
    assert rel_l2(ops.linear(x, w, b, tile_cfg=cfg), lin) < TOL
    assert rel_l2(ops.linear(x, w, b, epi=ops.EPI_GELU_TANH, tile_cfg=cfg), O.gelu_tanh(lin)) < TOL
    out = ops.linear(x, w, b, epi=ops.EPI_GATE_RES, res=res, gate=gate, tile_cfg=cfg)
    assert rel_l2(out, res.float().cpu() + gate.float().cpu() * lin) < TOL
    out = ops.linear(x, w, b, epi=ops.EPI_GATE_RES, res=res, tile_cfg=cfg)
    assert rel_l2(out, res.float().cpu() + lin) < TOL


@pytest.mark.parametrize("cfg,S", [(41, 2), (46, 2), (46, 4), (43, 2), (49, 2), (51, 3), (36, 3), (40, 2), (45, 3), (10, 2), (15, 4), (7, 2)])
def test_gemm_split_k(dev, cfg, S):
    """Split-K (tile_cfg = cfg | S << 8): S blocks per output tile pass an fp32 partial through the workspace in a
    fixed order.  Must match the unsplit kernel, be repeatable bit for bit, and leave the counters clean."""
    from flux_generator_amd import ops
    M, N, K = 600, 520, 1024
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    res, gate = rnd(M, N, seed=4), rnd(N, seed=5)
    lin = O.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())
    ref = res.float().cpu() + gate.float().cpu() * lin
    outs = [ops.linear(x, w, b, epi=ops.EPI_GATE_RES, res=res, gate=gate, tile_cfg=cfg | (S << 8)) for _ in range(3)]
    assert rel_l2(outs[0], ref) < TOL
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    plain = ops.linear(x, w, b, epi=ops.EPI_GATE_RES, res=res, gate=gate, tile_cfg=cfg)
    assert rel_l2(outs[0], plain.float().cpu()) < 1e-2


@pytest.mark.parametrize("cfg,S", [(51, 3), (49, 4), (46, 2), (55, 2)])
def test_gemm_split_k_no_stale_partials(dev, cfg, S):
    """The partial-tile slots are reused by every launch: alternating two different problems through the same
    workspace must give each its own result every time (a reducer that read a stale partial of the previous launch from
    a cache would reproduce the OTHER problem's contribution), also with other traffic in between."""
    from flux_generator_amd import ops
    M, N, K = 1280, 3072, 4096
    xs = [rnd(M, K, seed=s) for s in (1, 2)]
    w = rnd(N, K, seed=3, scale=K ** -0.5)
    first = [ops.linear(x, w, None, tile_cfg=cfg | (S << 8)).clone() for x in xs]
    assert not torch.equal(first[0], first[1])
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for it in range(6):
        i = it & 1
        if it % 3 == 2:
            junk.random_(0, 255)                # churn L2 / MALL between launches
        y = ops.linear(xs[i], w, None, tile_cfg=cfg | (S << 8))
        assert torch.equal(y, first[i]), f"iteration {it}"
    assert rel_l2(first[0], O.linear(xs[0].float().cpu(), w.float().cpu(), None)) < TOL


def _rs_launches():
    from flux_generator_amd import _lib
    return int(_lib.load().fluxhip_gemm_rs_launches())


@pytest.mark.parametrize("cfg,S", [(49, 2), (49, 4), (51, 2), (51, 3), (51, 4), (56, 3)])
def test_gemm_split_k_reduce_scatter(dev, cfg, S):
    """The reduce-scatter split-K hand-off (the tiles that carry a FLAG_RS kernel: 256 x 256, 256 x 192 and its phase-stamped
    twin; whole grid resident): every supported (tile, S) — ownership by
    fragment rows (MI % S == 0) or by fragment columns (256 x 192 with S = 3) — with ragged M / N edges and the fused
    epilogues that the Flux plan runs through it; the launch must really take the reduce-scatter path, agree with the chain
    hand-off up to fp32 summation order, be repeatable bit for bit and leave the counters clean."""
    from flux_generator_amd import _lib, ops
    lib = _lib.load()
    M, N, K = 600, 520, 1024
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    res, gate = rnd(M, N, seed=4), rnd(N, seed=5)
    lin = O.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())
    code = cfg | (S << 8)
    n0 = _rs_launches()
    outs = [ops.linear(x, w, b, epi=ops.EPI_GATE_RES, res=res, gate=gate, tile_cfg=code) for _ in range(3)]
    assert _rs_launches() == n0 + 3, "the launch did not use the reduce-scatter hand-off"
    assert rel_l2(outs[0], res.float().cpu() + gate.float().cpu() * lin) < TOL
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    assert rel_l2(ops.linear(x, w, b, tile_cfg=code), lin) < TOL
    assert rel_l2(ops.linear(x, w, b, epi=ops.EPI_GELU_TANH, tile_cfg=code), O.gelu_tanh(lin)) < TOL
    assert lib.fluxhip_gemm_set_splitk_mode(1) == 0
    try:
        n1 = _rs_launches()
        chain = ops.linear(x, w, b, epi=ops.EPI_GATE_RES, res=res, gate=gate, tile_cfg=code)
        assert _rs_launches() == n1
    finally:
        assert lib.fluxhip_gemm_set_splitk_mode(0) == 0
    assert rel_l2(outs[0], chain.float().cpu()) < 4e-3
    again = ops.linear(x, w, b, epi=ops.EPI_GATE_RES, res=res, gate=gate, tile_cfg=code)      # counters were left clean
    assert torch.equal(again, outs[0])


@pytest.mark.parametrize("cfg,epi,N,K", [(51, "bias", 9216, 3072), (49, "gelu", 12288, 3072), (47, "gate_res", 3072, 3072),
                                            (50, "split_gelu", 21504, 3072), (51 | (3 << 8), "gate_res", 3072, 12288)])
def test_gemm_lean_kernels_match_the_generic_ones(dev, cfg, epi, N, K):
    """The single-epilogue (FLAG_LEAN) twins of the Flux plan's tiles: each of the five launch kinds of a batch-1 forward
    (qkv, mlp0, attn.proj, linear1 with the split-GELU epilogue, the split-K 3 reduce-scatter projections) must take its lean
    kernel and give the SAME BITS as the generic kernel; anything a lean kernel has no code for stays on the generic one."""
    from flux_generator_amd import _lib, ops
    from flux_generator_amd.ops import make_gemm_desc
    lib = _lib.load()
    M = 1280
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    res, gate = rnd(M, 3072, seed=4), rnd(3072, seed=5)

    def run():
        if epi == "split_gelu":      # linear1: q|k|v columns to one buffer, GELU(mlp) columns to another
            c1 = torch.empty(M, 9216, dtype=BF, device=dev)
            c2 = torch.zeros(M, 3072 + 12288, dtype=BF, device=dev)
            d = make_gemm_desc([dict(A=x.data_ptr(), W=w.data_ptr(), bias=b.data_ptr(), C=c1.data_ptr(), M=M)], 1, N, K, K, 9216,
                               ops.EPI_SPLIT_GELU, n_split=9216, C2=c2.data_ptr(), ldc2=3072 + 12288, c2_coloff=3072, tile_cfg=cfg)
            ops.gemm(d)
            return torch.cat([c1, c2], dim=1)
        kw = dict(bias={}, gelu=dict(epi=ops.EPI_GELU_TANH), gate_res=dict(epi=ops.EPI_GATE_RES, res=res, gate=gate))[epi]
        return ops.linear(x, w, b, tile_cfg=cfg, **kw)

    n0 = int(lib.fluxhip_gemm_lean_launches()) + _rs_launches()
    lean = run()
    assert int(lib.fluxhip_gemm_lean_launches()) + _rs_launches() == n0 + 1, "the launch did not take its lean kernel"
    assert lib.fluxhip_gemm_set_lean(0) == 0 and lib.fluxhip_gemm_set_splitk_mode(1 if cfg >> 8 else 0) == 0
    try:
        n1 = int(lib.fluxhip_gemm_lean_launches()) + _rs_launches()
        generic = run()
        assert int(lib.fluxhip_gemm_lean_launches()) + _rs_launches() == n1
    finally:
        assert lib.fluxhip_gemm_set_lean(1) == 0 and lib.fluxhip_gemm_set_splitk_mode(0) == 0
    if cfg >> 8:      # split-K: reduce-scatter vs chain differ in fp32 summation order
        assert rel_l2(lean, generic.float().cpu()) < 4e-3
    else:
        assert torch.equal(lean, generic)
    # a launch the lean kernel has no code for (another activation) stays generic and correct
    n2 = int(lib.fluxhip_gemm_lean_launches())
    y = ops.linear(x, w, b, epi=ops.EPI_SILU, tile_cfg=cfg & 255)
    assert int(lib.fluxhip_gemm_lean_launches()) == n2
    lin = O.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())
    assert rel_l2(y, lin * torch.sigmoid(lin)) < TOL


def test_gemm_split_k_tiles_without_rs_kernel_use_the_chain(dev):
    """Tiles without a FLAG_RS instantiation (e.g. 256 x 224, 128 x 256 ping-pong) split through the chain hand-off."""
    from flux_generator_amd import ops
    M, N, K = 600, 520, 1024
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = O.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())
    for cfg, S in ((50, 2), (55, 2), (52, 4)):
        n0 = _rs_launches()
        y = ops.linear(x, w, b, tile_cfg=cfg | (S << 8))
        assert _rs_launches() == n0 and rel_l2(y, ref) < TOL


def test_gemm_split_k_narrow_rows_stay_on_the_chain(dev):
    """Outputs that are only 8-byte addressable per row (N = 260) take the direct-store epilogue, which writes every fragment a
    block holds: such launches must NOT use the reduce-scatter hand-off (a block would store the partial sums of fragments it
    does not own).  They fall back to the chain and stay correct."""
    from flux_generator_amd import ops
    M, N, K = 600, 260, 1024
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    n0 = _rs_launches()
    y = ops.linear(x, w, b, tile_cfg=51 | (3 << 8))
    assert _rs_launches() == n0
    assert rel_l2(y, O.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())) < TOL


def test_gemm_split_k_reduce_scatter_under_uneven_load(dev):
    """The S blocks of a tile wait for EACH OTHER: run the Flux shapes (linear2: 1280 x 3072 x 15360, S = 3, 240 blocks) while a
    second stream keeps taking CUs and memory bandwidth away (blocks of one tile then start far apart), alternating two
    problems through the same workspace so that a consumer's L1 / L2 hold the PREVIOUS launch's slab lines.  Every result
    must equal its first, bit for bit, 30 launches long."""
    from flux_generator_amd import ops
    M, N, K = 1280, 3072, 15360
    xs = [rnd(M, K, seed=s) for s in (1, 2)]
    w = rnd(N, K, seed=3, scale=K ** -0.5)
    res = rnd(M, N, seed=4)
    n0 = _rs_launches()
    first = [ops.linear(x, w, None, epi=ops.EPI_GATE_RES, res=res).clone() for x in xs]
    assert _rs_launches() == n0 + 2, "the picker did not choose a reduce-scatter split for linear2 at batch 1"
    assert rel_l2(first[0], res.float().cpu() + O.linear(xs[0].float().cpu(), w.float().cpu(), None)) < TOL
    side = torch.cuda.Stream(device=dev)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    a, bmat = rnd(4096, 4096, seed=7), rnd(4096, 4096, seed=8)
    for it in range(30):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                     # competing work: a copy kernel and a plain GEMM, varying per iteration
            if it % 3 != 1:
                big[: (64 + 32 * (it % 5)) << 20].add_(1)
            if it % 2:
                ops.linear(a, bmat, None)
        y = ops.linear(xs[it & 1], w, None, epi=ops.EPI_GATE_RES, res=res)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(y, first[it & 1]), f"iteration {it}"


@pytest.mark.parametrize("cfg,S", [(49, 2), (49, 4), (51, 3), (51, 4), (56, 3)])
def test_gemm_split_k_reduce_scatter_orphan_path_same_bits(dev, cfg, S):
    """The wait-free completion of the reduce-scatter hand-off (a block whose peers have not arrived in time publishes its own
    slice, sets its orphan bit and exits; the tile's last arriver finishes the orphaned slices from the workspace).  With a
    poll time of 0 every block that is not already complete on arrival takes that path; the results must be the SAME BITS as
    with the default poll time (same summation order, same epilogue arithmetic), launch after launch, and the counters must be
    left clean for the next launch (ragged M / N edges, gate-residual epilogue, both ownership layouts)."""
    from flux_generator_amd import _lib, ops
    lib = _lib.load()
    M, N, K = 600, 520, 2048
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    res, gate = rnd(M, N, seed=4), rnd(N, seed=5)
    code = cfg | (S << 8)
    run = lambda: ops.linear(x, w, b, epi=ops.EPI_GATE_RES, res=res, gate=gate, tile_cfg=code)      # noqa: E731
    want = run().clone()
    lin = O.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())
    assert rel_l2(want, res.float().cpu() + gate.float().cpu() * lin) < TOL
    assert lib.fluxhip_gemm_set_rs_timeout_us(0) == 0
    try:
        n0 = _rs_launches()
        for it in range(6):
            assert torch.equal(run(), want), f"orphan path, launch {it}"
        assert _rs_launches() == n0 + 6
    finally:
        assert lib.fluxhip_gemm_set_rs_timeout_us(100) == 0
    assert torch.equal(run(), want)                       # ... and back on the fast path with clean counters


@pytest.mark.parametrize("timeout_us", [100, 0, 5])
def test_gemm_split_k_reduce_scatter_grid_larger_than_the_chip(dev, timeout_us):
    """The situation that used to be fatal (a process sharing its GPU, masked CUs): the blocks of a tile are NOT all resident.
    Forced here on an exclusive GPU with a 480-block reduce-scatter grid on 256 CUs (mode 2 lifts the launcher's grid <= CUs
    condition): in XCD-brick order the first 256 blocks are all first / second splits and wait for third splits that cannot
    start before somebody leaves.  The bounded poll + orphan hand-off must complete, give the bits of the resident case for
    every tile (same problem solved as two resident halves), and repeat."""
    from flux_generator_amd import _lib, ops
    lib = _lib.load()
    M, N, K = 2560, 3072, 3072
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    res, gate = rnd(M, N, seed=4), rnd(N, seed=5)
    code = 51 | (3 << 8)                                   # 256 x 192 tiles: 10 x 16 tiles x 3 splits = 480 blocks
    halves = [ops.linear(x[i * 1280:(i + 1) * 1280], w, b, epi=ops.EPI_GATE_RES, res=res[i * 1280:(i + 1) * 1280], gate=gate,
                         tile_cfg=code) for i in range(2)]       # 240 blocks each: resident, fast path
    want = torch.cat(halves, dim=0)
    assert lib.fluxhip_gemm_set_splitk_mode(2) == 0 and lib.fluxhip_gemm_set_rs_timeout_us(timeout_us) == 0
    try:
        n0 = _rs_launches()
        for it in range(4):
            y = ops.linear(x, w, b, epi=ops.EPI_GATE_RES, res=res, gate=gate, tile_cfg=code)
            torch.cuda.synchronize()
            assert torch.equal(y, want), f"launch {it}"
        assert _rs_launches() == n0 + 4, "the launch did not take the reduce-scatter kernel"
    finally:
        assert lib.fluxhip_gemm_set_splitk_mode(0) == 0 and lib.fluxhip_gemm_set_rs_timeout_us(100) == 0


def test_gemm_split_k_workspace_is_serialised_across_streams(dev):
    """The split-K workspace serves one launch at a time: when the launching stream changes, the library orders the new
    stream after the work already enqueued on the previous one (two reduce-scatter launches issued back to back on two
    streams without any caller-side dependency must both be right)."""
    from flux_generator_amd import ops
    M, N, K = 1280, 3072, 12288
    xs = [rnd(M, K, seed=s) for s in (1, 2)]
    w = rnd(N, K, seed=3, scale=K ** -0.5)
    res = rnd(M, N, seed=4)
    want = [ops.linear(x, w, None, epi=ops.EPI_GATE_RES, res=res).clone() for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for it in range(8):
        outs = []
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs.append(ops.linear(xs[i], w, None, epi=ops.EPI_GATE_RES, res=res))
        torch.cuda.synchronize()
        assert torch.equal(outs[0], want[0]) and torch.equal(outs[1], want[1]), f"iteration {it}"


def test_gemm_split_k_reduce_scatter_grouped_batched(dev):
    """Reduce-scatter split-K under the 2-group (txt / img), batched launch shape of the double blocks' mlp2 with per-batch
    gates, forced onto 256 x 256 tiles with S = 4 (ownership by fragment rows)."""
    from flux_generator_amd import ops
    B, S_, L, D, N = 2, 256, 256, 2048, 512
    T = S_ + L
    x = rnd(B, T, D, seed=1)
    wt, wi = rnd(N, D, seed=2, scale=D ** -0.5), rnd(N, D, seed=3, scale=D ** -0.5)
    bt, bi = rnd(N, seed=4), rnd(N, seed=5)
    res = rnd(B, T, N, seed=6)
    gates = rnd(B, 2 * N, seed=7)
    out = torch.empty(B, T, N, dtype=BF, device=dev)
    e = 2
    gs = [dict(A=x.data_ptr(), W=wt.data_ptr(), bias=bt.data_ptr(), C=out.data_ptr(), res=res.data_ptr(), gate=gates.data_ptr(),
               gate_bstride=2 * N, a_bstride=T * D, c_bstride=T * N, M=S_),
          dict(A=x.data_ptr() + S_ * D * e, W=wi.data_ptr(), bias=bi.data_ptr(), C=out.data_ptr() + S_ * N * e,
               res=res.data_ptr() + S_ * N * e, gate=gates.data_ptr() + N * e, gate_bstride=2 * N, a_bstride=T * D,
               c_bstride=T * N, M=L)]
    n0 = _rs_launches()
    ops.gemm(ops.make_gemm_desc(gs, B, N, D, D, N, ops.EPI_GATE_RES, tile_cfg=49 | (4 << 8)))
    assert _rs_launches() == n0 + 1
    xf, rf, gf = x.float().cpu(), res.float().cpu(), gates.float().cpu()
    ref = torch.empty(B, T, N)
    ref[:, :S_] = rf[:, :S_] + gf[:, None, :N] * O.linear(xf[:, :S_], wt.float().cpu(), bt.float().cpu())
    ref[:, S_:] = rf[:, S_:] + gf[:, None, N:] * O.linear(xf[:, S_:], wi.float().cpu(), bi.float().cpu())
    assert rel_l2(out, ref) < TOL


def test_gemm_split_k_oversubscribed(dev):
    """More split-K blocks than the chip can hold at once: consumers only ever wait for lower block ids."""
    from flux_generator_amd import ops
    M, N, K = 4096, 2048, 512
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    y = ops.linear(x, w, None, tile_cfg=4 | (2 << 8))          # 64x64 tiles: 2048 tiles x 2 splits
    assert rel_l2(y, O.linear(x.float().cpu(), w.float().cpu(), None)) < TOL


def test_gemm_auto_split_k_shapes(dev):
    """The shapes the tile picker splits at batch 1 (N = 3072, long K) against the oracle."""
    from flux_generator_amd import ops
    M, N, K = 1280, 3072, 12288
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = ops.linear(x, w, b)
    assert rel_l2(y, O.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())) < TOL


def test_gemm_grouped_split_batched(dev):
    """Two weight groups over one packed [txt;img] buffer, per-batch gates, split-GELU epilogue."""
    from flux_generator_amd import ops
    B, S, L, D, N = 2, 40, 150, 128, 256
    T = S + L
    x = rnd(B, T, D, seed=1)
    wt, wi = rnd(N, D, seed=2, scale=D ** -0.5), rnd(N, D, seed=3, scale=D ** -0.5)
    bt, bi = rnd(N, seed=4), rnd(N, seed=5)
    out = torch.zeros(B, T, N, dtype=BF, device=dev)
    e = 2
    groups = [dict(A=x.data_ptr(), W=wt.data_ptr(), bias=bt.data_ptr(), C=out.data_ptr(), a_bstride=T * D,
                   c_bstride=T * N, M=S),
              dict(A=x.data_ptr() + S * D * e, W=wi.data_ptr(), bias=bi.data_ptr(), C=out.data_ptr() + S * N * e,
                   a_bstride=T * D, c_bstride=T * N, M=L)]
    ops.gemm(ops.make_gemm_desc(groups, B, N, D, D, N))
    xf = x.float().cpu()
    ref = torch.cat([O.linear(xf[:, :S], wt.float().cpu(), bt.float().cpu()),
                     O.linear(xf[:, S:], wi.float().cpu(), bi.float().cpu())], dim=1)
    assert rel_l2(out, ref) < TOL

    # gated residual with per-batch gate vectors living in a wider table
    NM = 3 * N
    table = rnd(B, NM, seed=6)
    res = rnd(B, T, N, seed=7)
    buf = res.clone()
    for g, off in zip(groups, (0, N)):
        g.update(res=g["C"] - out.data_ptr() + buf.data_ptr(), C=g["C"] - out.data_ptr() + buf.data_ptr(),
                 gate=table.data_ptr() + off * e, gate_bstride=NM)
    ops.gemm(ops.make_gemm_desc(groups, B, N, D, D, N, ops.EPI_GATE_RES))
    tb = table.float().cpu()
    gate = torch.cat([tb[:, None, 0:N].expand(B, S, N), tb[:, None, N:2 * N].expand(B, L, N)], dim=1)
    assert rel_l2(buf, res.float().cpu() + gate * ref) < TOL

    # split epilogue: first n_split columns raw, the rest GELU'd into a second buffer at a column offset
    n_split, off2, ld2 = 128, 64, 64 + (N - 128)
    c1 = torch.zeros(B, T, n_split, dtype=BF, device=dev)
    c2 = torch.zeros(B, T, ld2, dtype=BF, device=dev)
    g = dict(A=x.data_ptr(), W=wi.data_ptr(), bias=bi.data_ptr(), C=c1.data_ptr(), a_bstride=T * D,
             c_bstride=T * n_split, M=T)
    ops.gemm(ops.make_gemm_desc([g], B, N, D, D, n_split, ops.EPI_SPLIT_GELU, n_split=n_split, C2=c2.data_ptr(),
                                ldc2=ld2, c2_bstride=T * ld2, c2_coloff=off2))
    full = O.linear(xf, wi.float().cpu(), bi.float().cpu())
    assert rel_l2(c1, full[..., :n_split]) < TOL
    assert rel_l2(c2[..., off2:], O.gelu_tanh(full[..., n_split:])) < TOL
    assert torch.count_nonzero(c2[..., :off2]) == 0


def test_gemm_bad_args(dev):
    from flux_generator_amd import ops
    x, w = rnd(64, 72), rnd(64, 72)   # K % 64 != 0 -> -1, raised as FluxHipError
    with pytest.raises(ops.FluxHipError):
        ops.linear(x, w)


# ------------------------------------------------------------------ small linear
@pytest.mark.parametrize("B,N,K", [(1, 3072, 256), (1, 1030, 768), (3, 513, 3072), (8, 256, 128)])
@pytest.mark.parametrize("silu_in", [False, True])
def test_small_linear(dev, B, N, K, silu_in):
    from flux_generator_amd import ops
    x, w, b = rnd(B, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = ops.small_linear(x, w, b, silu_in=silu_in)
    xi = x.float().cpu()
    if silu_in:
        xi = O.silu(x.cpu()).float()      # silu output is a bf16 tensor in the reference
    ref = O.linear(xi, w.float().cpu(), b.float().cpu())
    assert rel_l2(y, ref) < TOL
    prev = rnd(B, N, seed=9)
    acc = ops.small_linear(x, w, b, out=prev.clone(), silu_in=silu_in, accum=True)
    assert rel_l2(acc, prev.float().cpu() + ref) < TOL


# ------------------------------------------------------------------ adaLN
@pytest.mark.parametrize("D", [256, 384, 3072])
def test_ln_modulate(dev, D):
    from flux_generator_amd import ops
    B, S, L = 2, 5, 11
    T = S + L
    x = rnd(B, T, D, seed=1, scale=3.0) + 0.5
    mods = rnd(B, 4 * D, seed=2, scale=0.3)
    out = torch.empty_like(x)
    mp, e = mods.data_ptr(), 2
    ops.ln_modulate(x, out, B, T, D, S, T * D, T * D, mp, mp + D * e, mp + 2 * D * e, mp + 3 * D * e, 4 * D)
    xf, m = x.float().cpu(), mods.float().cpu()
    ln = O.layer_norm(xf)
    ref = torch.cat([(1 + m[:, None, D:2 * D]) * ln[:, :S] + m[:, None, 0:D],
                     (1 + m[:, None, 3 * D:]) * ln[:, S:] + m[:, None, 2 * D:3 * D]], dim=1)
    assert rel_l2(out, ref) < 6e-3   # four bf16 op-boundary roundings are reproduced inside the kernel
    # bf16-faithful oracle (same op boundaries as MLX): tighter
    xb, mb = x.cpu(), mods.cpu()
    lnb = O.layer_norm(xb)
    refb = torch.cat([(1 + mb[:, None, D:2 * D]) * lnb[:, :S] + mb[:, None, 0:D],
                      (1 + mb[:, None, 3 * D:]) * lnb[:, S:] + mb[:, None, 2 * D:3 * D]], dim=1)
    assert rel_l2(out, refb) < 2e-3


# ------------------------------------------------------------------ qk norm + rope + V^T
@pytest.mark.parametrize("T,S", [(96, 32), (77, 13)])
def test_qk_norm_rope_vt(dev, T, S):
    from flux_generator_amd import ops
    B, H = 2, 3
    D = H * 128
    qkv = rnd(B, T, 3 * D, seed=1)
    ws = [(1 + 0.2 * rnd(128, seed=10 + i).float()).to(BF) for i in range(4)]   # q_txt, k_txt, q_img, k_img
    ids = torch.zeros(B, T, 3, dtype=torch.int32)
    ids[:, S:, 1] = torch.arange(T - S, dtype=torch.int32) // 7
    ids[:, S:, 2] = torch.arange(T - S, dtype=torch.int32) % 7
    rope = ops.rope_table(ids.to(dev), [16, 56, 56], 10000.0)
    Tpad = (T + 63) // 64 * 64
    Q = torch.empty(B, H, T, 128, dtype=BF, device=dev)
    K = torch.empty_like(Q)
    Vt = torch.full((B, H, 128, Tpad), 7.0, dtype=BF, device=dev)
    ops.qk_norm_rope(qkv, 3 * D, B, T, S, H, ws[0], ws[1], ws[2], ws[3], rope, T * 128, Q, K, Vt, Tpad)

    f = qkv.float().cpu()
    q, k, v = [O._split_heads(t, H) for t in torch.chunk(f, 3, dim=-1)]
    pe = O.embed_nd(ids, [16, 56, 56], 10000).to(BF).float()

    def norm(t, wt, wi):
        return torch.cat([O.rms_norm(t[:, :, :S], wt.float().cpu()), O.rms_norm(t[:, :, S:], wi.float().cpu())], dim=2)

    qr = O.apply_rope(norm(q, ws[0], ws[2]), pe)
    kr = O.apply_rope(norm(k, ws[1], ws[3]), pe)
    assert rel_l2(Q, qr) < TOL and rel_l2(K, kr) < TOL
    # V^T: pure data movement (exact), zero padded to Tpad keys, keys permuted inside every group of 16 as
    # [0-3, 8-11, 4-7, 12-15] (the attention kernel's PV fragment order, include/fluxhip.h)
    want = torch.zeros(B, H, 128, Tpad)
    want[..., :T] = v.transpose(-1, -2)
    perm = ops.vt_key_permutation(Tpad)
    assert perm[:16].tolist() == [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15] and int(perm[16]) == 16
    assert torch.equal(Vt.float().cpu(), want[..., perm])
    # rope table itself (cos, sin) vs the oracle's rotation matrices, both rounded to bf16
    assert rel_l2(rope[..., 0], pe[:, 0, :, :, 0, 0]) < 1e-3 and rel_l2(rope[..., 1], pe[:, 0, :, :, 1, 0]) < 1e-3


# ------------------------------------------------------------------ attention

@pytest.mark.parametrize("variant", [4, 5, 0x104, 0x105])
@pytest.mark.parametrize("B,H,T", [(1, 2, 128), (2, 3, 200), (1, 2, 65), (1, 2, 192), (1, 24, 1280), (1, 3, 1000), (2, 2, 577)])
def test_attention_d128_w64_variants(dev, variant, B, H, T):
    """The 64-queries-per-wave head_dim-128 kernel (csrc/attention.hip, attn128_w64_kernel): 256 (variant 4) and 128
    (variant 5, two KV wave sets merged through LDS) queries per workgroup, ragged query / key tails, odd tile counts — and
    its online-rescale fallback path forced on (+ 0x100) — against the float32 oracle attention on the same inputs."""
    from flux_generator_amd import _lib, ops
    lib = _lib.load()
    q, k, v = rnd(B, H, T, 128, seed=1), rnd(B, H, T, 128, seed=2), rnd(B, H, T, 128, seed=3)
    Tp = (T + 63) // 64 * 64
    vt = torch.zeros(B, H, 128, Tp, dtype=BF, device=dev)
    vt[..., :T] = v.transpose(-1, -2)
    vt = vt[..., ops.vt_key_permutation(Tp, dev)].contiguous()
    o = torch.full((B, T, H * 128), 7.0, dtype=BF, device=dev)
    assert lib.fluxhip_attention_set_variant(variant) == 0
    try:
        ops.attention_d128(q, k, vt, o, H * 128, B, H, T, Tp, 128 ** -0.5)
        torch.cuda.synchronize()
    finally:
        lib.fluxhip_attention_set_variant(0)
    ref = O.sdpa(q.float().cpu(), k.float().cpu(), v.float().cpu(), 128 ** -0.5).transpose(1, 2).reshape(B, T, H * 128)
    assert rel_l2(o, ref) < 6e-3


def test_attention_d128_w64_overflow_falls_back(dev):
    """The fast path of attn128_w64_kernel keeps each query's softmax reference at the row maximum of its FIRST KV tile; a
    later logit more than ~100 (log2 domain) above it would overflow exp2, so the kernel must notice and recompute the
    workgroup with the online-rescale loop.  Spiked keys late in the sequence (raw q.k ~ +1500 over everything before) force
    exactly that: the result must match the oracle, where a missed overflow gives inf / NaN."""
    from flux_generator_amd import _lib, ops
    lib = _lib.load()
    B, H, T = 1, 2, 640
    q, k, v = rnd(B, H, T, 128, seed=4), rnd(B, H, T, 128, seed=5), rnd(B, H, T, 128, seed=6)
    k[0, 0, 500] = q[0, 0, 17] * 1.5              # query 17 of head 0 meets a key aligned with it in tile 7
    k[0, 1, 300] = q[0, 1, 200] * 1.2
    Tp = (T + 63) // 64 * 64
    vt = torch.zeros(B, H, 128, Tp, dtype=BF, device=dev)
    vt[..., :T] = v.transpose(-1, -2)
    vt = vt[..., ops.vt_key_permutation(Tp, dev)].contiguous()
    ref = O.sdpa(q.float().cpu(), k.float().cpu(), v.float().cpu(), 1.0).transpose(1, 2).reshape(B, T, H * 128)
    for variant in (4, 5):
        o = torch.zeros(B, T, H * 128, dtype=BF, device=dev)
        assert lib.fluxhip_attention_set_variant(variant) == 0
        try:
            ops.attention_d128(q, k, vt, o, H * 128, B, H, T, Tp, 1.0)     # scale 1: logits of +-30 ordinarily, ~190 / ~150 at the spikes
            torch.cuda.synchronize()
        finally:
            lib.fluxhip_attention_set_variant(0)
        assert bool(torch.isfinite(o.float()).all()) and rel_l2(o, ref) < 6e-3


@pytest.mark.parametrize("B,H,T", [(1, 2, 128), (2, 3, 200), (1, 1, 1280), (1, 2, 65), (1, 2, 192), (1, 24, 1280), (2, 48, 512)])   # last: 384 workgroups -> single wave set; 192/200/1280: KV tiles split over two wave sets
def test_attention(dev, B, H, T):
    from flux_generator_amd import ops
    Tpad = (T + 63) // 64 * 64
    q, k, v = rnd(B, H, T, 128, seed=1), rnd(B, H, T, 128, seed=2), rnd(B, H, T, 128, seed=3)
    vt = torch.zeros(B, H, 128, Tpad, dtype=BF, device=dev)
    vt[..., :T] = v.transpose(-1, -2)
    vt = vt[..., ops.vt_key_permutation(Tpad, dev)].contiguous()           # the layout fluxhip_qk_norm_rope_bf16 writes
    o = torch.empty(B, T, H * 128, dtype=BF, device=dev)
    ops.attention_d128(q, k, vt, o, H * 128, B, H, T, Tpad, 128 ** -0.5)
    ref = O.sdpa(q.float().cpu(), k.float().cpu(), v.float().cpu(), 128 ** -0.5).transpose(1, 2).reshape(B, T, -1)
    assert rel_l2(o, ref) < 6e-3      # P is rounded to bf16 before the PV product
    # independent check of the oracle's sdpa itself
    ind = torch.nn.functional.scaled_dot_product_attention(q.float().cpu(), k.float().cpu(), v.float().cpu())
    assert rel_l2(ref, ind.transpose(1, 2).reshape(B, T, -1)) < 1e-5


def test_attention_forced_rescale(dev):
    """A key that spikes late forces the online-softmax rescale branch (rare on random data)."""
    from flux_generator_amd import ops
    B, H, T = 1, 1, 256
    q, k, v = rnd(B, H, T, 128, seed=1), rnd(B, H, T, 128, seed=2, scale=0.1), rnd(B, H, T, 128, seed=3)
    k[0, 0, 200] = q[0, 0, 17] * 2.0          # query 17 meets its spike in the 4th key tile
    k[0, 0, 70] = q[0, 0, 140] * 1.5
    vt = v.transpose(-1, -2)[..., ops.vt_key_permutation(T, dev)].contiguous()
    o = torch.empty(B, T, 128, dtype=BF, device=dev)
    ops.attention_d128(q, k, vt, o, 128, B, H, T, T, 128 ** -0.5)
    ref = O.sdpa(q.float().cpu(), k.float().cpu(), v.float().cpu(), 128 ** -0.5).transpose(1, 2).reshape(B, T, -1)
    assert rel_l2(o, ref) < 6e-3
    assert rel_l2(o[0, 17], ref[0, 17]) < 1e-2 and rel_l2(o[0, 140], ref[0, 140]) < 1e-2


# ------------------------------------------------------------------ embeddings, sampler, pack
def test_timestep_embedding(dev):
    from flux_generator_amd import ops
    t = torch.tensor([1.0, 0.75, 0.5, 0.25, 0.0, 0.3141], dtype=BF)
    got = ops.timestep_embedding(t.to(dev), 256).float().cpu()
    ref = O.timestep_embedding(t, 256).float()
    # cos/sin of arguments up to 1000 rad, then bf16: allow one bf16 ulp on a handful of entries
    assert (got - ref).abs().max() <= 2 ** -7
    assert ((got - ref).abs() > 0).float().mean() < 0.05


def test_euler_and_pack(dev):
    from flux_generator_amd import ops
    x, p = rnd(2, 96, 64, seed=1), rnd(2, 96, 64, seed=2)
    for dt in (-0.5, -0.0117):
        dtb = float(torch.tensor(dt, dtype=BF))
        got = ops.euler_step(x, p, dtb)
        ref = O.euler_step(p.cpu(), x.cpu(), 0.0, dt)          # bf16 oracle: identical op boundaries
        assert torch.equal(got.cpu(), ref)
    z = rnd(2, 12, 20, 16, seed=3)
    packed = ops.pack_latents(z)
    ref_p, ids = O.prepare_latent_images(z.cpu())
    assert torch.equal(packed.cpu(), ref_p)                      # bit exact
    assert torch.equal(ops.unpack_latents(packed, 12, 20).cpu(), z.cpu())
    aff = ops.unpack_latents(packed, 12, 20, 0.3611, 0.1159)
    assert rel_l2(aff, z.float().cpu() / 0.3611 + 0.1159) < TOL


def test_gemm_pingpong_falls_back_for_4gib_operands(dev):
    """The ping-pong tiles address dense operands as scalar base + 32-bit byte offset; an activation whose rows span
    4 GiB or more (here 2112 rows spaced 2 MiB apart) is routed to the plain-ring tile of the same shape instead
    (same result as the product path's own pick on a compact copy)."""
    from flux_generator_amd import ops
    from flux_generator_amd.ops import make_gemm_desc
    M, N, K, lda = 2112, 384, 128, 1 << 20                       # M * lda * 2 bytes = 4.1 GiB
    big = torch.empty(M * lda, dtype=BF, device=dev)
    A = big.view(M, lda)[:, :K]
    a = rnd(M, K, seed=1).to(dev)
    A.copy_(a)
    w, b = rnd(N, K, seed=2, scale=K ** -0.5).to(dev), rnd(N, seed=3).to(dev)
    ref = ops.linear(a.contiguous(), w, b)
    for cfg in (51, 49, 55):
        out = torch.zeros(M, N, dtype=BF, device=dev)
        ops.gemm(make_gemm_desc([dict(A=A.data_ptr(), W=w.data_ptr(), bias=b.data_ptr(), C=out.data_ptr(), M=M)], 1, N, K, lda, N,
                                tile_cfg=cfg))
        assert rel_l2(out, ref) < 2e-3, cfg
    del big


def test_gemm_picker_keeps_the_flux_plan(dev):
    """The tile picker's time model was refitted on 111 shapes in round 3 (csrc/gemm.hip, kCands); the choices for the six
    block GEMMs of the batch-1 Flux plan were established IN SITU (tools/plan_sweep.py) and are the constraint of that fit:
    256x192 for qkv, 128x128 for attn.proj, 256x256 for mlp0, 256x224 for linear1, 256x192 split-K 3 (reduce-scatter) for the
    two K >= 12288 projections.  Also: the mid-sized SDXL projection the old model mis-ranked gets the 128x160 tile of round 6."""
    import ctypes
    from flux_generator_amd import _lib
    from flux_generator_amd.ops import make_gemm_desc
    lib = _lib.load()

    def pick(groups, N, K):
        d = make_gemm_desc([dict(A=0, W=0, C=0, M=m) for m in groups], 1, N, K, K, N)
        c = lib.fluxhip_gemm_tile_cfg(ctypes.byref(d))
        return c & 255, c >> 8

    assert pick([256, 1024], 9216, 3072) == (51, 1)
    assert pick([256, 1024], 3072, 3072) == (47, 1)
    assert pick([256, 1024], 12288, 3072) == (49, 1)
    assert pick([256, 1024], 3072, 12288) == (51, 3)
    assert pick([1280], 21504, 3072) == (50, 1)
    assert pick([1280], 3072, 15360) == (51, 3)
    assert pick([4096], 1280, 1280) == (57, 1)       # round 6: 128x160 - 1280 = 8 x 160, 256 tiles (128x256 before: 160 tiles)


@pytest.mark.parametrize("M,C", [(4096, 1280), (600, 320), (16384, 640), (130, 64)])
def test_gemm_geglu_pair_epilogue(dev, M, C):
    """EPI_GEGLU_PAIR: the two Linears of the UNet's GEGLU (stable_diffusion/.../unet.py:74-78) as one launch over interleaved
    value / gate rows; must give the SAME BITS as linear1 followed by linear2 with the EPI_GEGLU epilogue (ragged M edges, both
    tiles that carry the pair kernel), and agree with the float32 oracle; requests it cannot serve fail loudly."""
    from flux_generator_amd import ops
    K, N = C, 4 * C
    x = rnd(M, K, seed=1)
    w1, w2 = rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, K, seed=3, scale=K ** -0.5)
    b1, b2 = rnd(N, seed=4), rnd(N, seed=5)
    a = ops.linear(x, w1, b1)
    two = ops.linear(x, w2, b2, epi=ops.EPI_GEGLU, res=a)
    wp, bp = ops.interleave_geglu(w1, w2), ops.interleave_geglu(b1, b2)
    for cfg in (0, 49, 55):
        one = ops.linear(x, wp, bp, epi=ops.EPI_GEGLU_PAIR, tile_cfg=cfg)
        assert one.shape == (M, N) and torch.equal(one, two), cfg
    xf = x.float().cpu()
    va = O.linear(xf, w1.float().cpu(), b1.float().cpu())
    vg = O.linear(xf, w2.float().cpu(), b2.float().cpu())
    assert rel_l2(two, va * torch.nn.functional.gelu(vg)) < TOL
    with pytest.raises(ops.FluxHipError):
        ops.linear(x, wp, bp, epi=ops.EPI_GEGLU_PAIR, tile_cfg=51)          # a tile without the pair kernel

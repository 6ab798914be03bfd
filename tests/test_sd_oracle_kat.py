"""Pins for the stable_diffusion/ CPU oracle (no GPU): known answers from SURVEY.md Appendix B and
independent re-implementations."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import flux_oracle as O
from oracle import sd_oracle as S


def test_sigma_table_kat():
    s = S.EulerSampler(S.DiffusionConfig())
    assert float(s._sigmas[0]) == 0.0
    assert float(s._sigmas[1]) == pytest.approx(0.0291672, abs=1e-6)
    assert float(s._sigmas[500]) == pytest.approx(1.612886, abs=2e-6)
    assert float(s._sigmas[1000]) == pytest.approx(14.614641, abs=2e-5)
    assert s.max_time == 1000 and s.prior_scale() == pytest.approx(0.997667, abs=1e-6)
    assert s.timesteps(2) == [(1000.0, 500.0), (500.0, 0.0)]
    assert float(s.sigmas(499.5)) == pytest.approx(float((s._sigmas[499] + s._sigmas[500]) / 2), rel=1e-6)
    with pytest.raises(NotImplementedError):
        S.EulerSampler(S.DiffusionConfig(beta_schedule="cosine"))


def test_euler_steps_formulas():
    s, a = S.EulerSampler(S.DiffusionConfig()), S.EulerAncestralSampler(S.DiffusionConfig())
    x, e, n = torch.randn(4), torch.randn(4), torch.randn(4)
    t, tp = 800.0, 400.0
    sg, sp = float(s.sigmas(t)), float(s.sigmas(tp))
    want = (math.sqrt(sg * sg + 1) * x + e * (sp - sg)) / math.sqrt(sp * sp + 1)
    assert torch.allclose(s.step(e, x, t, tp), want, atol=1e-5)
    up = math.sqrt(sp * sp * (sg * sg - sp * sp) / (sg * sg))
    down = math.sqrt(sp * sp - up * up)
    want = (math.sqrt(sg * sg + 1) * x + e * (down - sg) + n * up) / math.sqrt(sp * sp + 1)
    assert torch.allclose(a.step(e, x, t, tp, n), want, atol=1e-5)


def test_ancestral_step_float16_cast_order():
    """The reference runs the UNet (and therefore the sampler arithmetic) in float16: sigma and sigma_prev are cast to the
    eps dtype first and sigma^2, sigma_up, sigma_down are float16 expressions of those (sampler.py:88-105).  An independent
    numpy float16 evaluation of exactly that sequence, op by op, must match the oracle's float16 run bit for bit, and it
    must DIFFER from evaluating the coefficients in float32 and casting afterwards (what the oracle did before)."""
    import numpy as np
    a = S.EulerAncestralSampler(S.DiffusionConfig())
    g = torch.Generator().manual_seed(3)
    x, e, n = (torch.randn(64, generator=g).half() for _ in range(3))
    differs = 0
    for t, tp in [(1000.0, 750.0), (750.0, 500.0), (500.0, 250.0), (999.0, 333.0)]:
        h = np.float16
        sg, sp = h(float(a.sigmas(t))), h(float(a.sigmas(tp)))
        sg2, sp2 = h(sg * sg), h(sp * sp)
        up = h(np.sqrt(h(h(sp2 * h(sg2 - sp2)) / sg2)))
        down = h(np.sqrt(h(sp2 - h(up * up))))
        dt = h(down - sg)
        xn, en, nn_ = x.numpy(), e.numpy(), n.numpy()
        y = (h(np.sqrt(h(sg2 + h(1)))) * xn).astype(h) + (en * dt).astype(h)
        y = (y + (nn_ * up).astype(h)).astype(h)
        want = (y * h(h(1) / np.sqrt(h(sp2 + h(1))))).astype(h)
        got = a.step(e, x, t, tp, n)
        assert got.dtype == torch.float16
        # torch.rsqrt in half is evaluated in float and rounded once; numpy's 1/sqrt above rounds twice: allow 1 ulp there
        assert np.max(np.abs(got.numpy().astype(np.float32) - want.astype(np.float32))) <= 2 * float(np.spacing(h(np.max(np.abs(want)))))
        _, _, up32, down32 = a.coefficients(t, tp, torch.float32)
        differs += int(float(up32.half()) != float(up) or float(down32.half()) != float(down))
    assert differs > 0, "float16-first and float32-then-cast coefficients never differ: the test does not exercise the cast order"


def test_sinusoidal_matches_diffusers_timesteps():
    """== diffusers Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0): 10000^(-i/half)."""
    t = torch.tensor([999.0, 1.0])
    for dims in (320, 256):
        half = dims // 2
        freq = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
        want = torch.cat([torch.cos(t[:, None] * freq), torch.sin(t[:, None] * freq)], -1)
        assert torch.allclose(S.sinusoidal_encoding(t, dims), want, atol=2e-3)   # fp32 args up to 999 rad


def test_param_counts_and_structure():
    xl = S.UNetConfig(block_out_channels=(320, 640, 1280), layers_per_block=(2, 2, 2), transformer_layers_per_block=(1, 2, 10),
                      num_attention_heads=(5, 10, 20), cross_attention_dim=(2048,) * 3,
                      down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                      up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"), addition_embed_type="text_time",
                      addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816)
    n = sum(math.prod(v) for v in S.unet_weight_shapes(xl).values())
    assert round(n / 1e9, 3) == 2.567                      # stable_diffusion/README.md:104 "2.6B parameters"
    sd21 = sum(math.prod(v) for v in S.unet_weight_shapes(S.UNetConfig(
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"))).values())
    assert round(sd21 / 1e6) == 866                        # SD 2.1 UNet: 866 M parameters
    down, up = S._block_plan(xl)
    assert [b["resnets"] for b in up][0] == [(2560, 1280), (2560, 1280), (1920, 1280)]
    assert [b["resnets"] for b in up][2] == [(960, 320), (640, 320), (640, 320)]


def test_transformer_block_vs_torch_modules():
    g = torch.Generator().manual_seed(0)
    C, H, N, enc, Sx = 128, 2, 10, 96, 7
    shapes = {k[len("mid_blocks.1.transformer_blocks.0."):]: v for k, v in S.unet_weight_shapes(
        S.UNetConfig(block_out_channels=(C,), layers_per_block=(1,), transformer_layers_per_block=(1,), num_attention_heads=(H,),
                     cross_attention_dim=(enc,), down_block_types=("DownBlock2D",), up_block_types=("UpBlock2D",))).items()
              if k.startswith("mid_blocks.1.transformer_blocks.0.")}
    W = {"b." + k: v for k, v in O.init_weights(shapes, seed=1, norm_jitter=0.3).items()}
    x, mem = torch.randn(2, N, C, generator=g), torch.randn(2, Sx, enc, generator=g)
    got = S.transformer_block(W, "b", H, x, mem)

    def attn(p, q_in, kv):
        m = torch.nn.MultiheadAttention(C, H, bias=False, batch_first=True, kdim=kv.shape[-1], vdim=kv.shape[-1])
        if kv.shape[-1] == C:
            m.in_proj_weight.data = torch.cat([W[f"b.{p}.query_proj.weight"], W[f"b.{p}.key_proj.weight"], W[f"b.{p}.value_proj.weight"]])
        else:
            m.q_proj_weight.data, m.k_proj_weight.data, m.v_proj_weight.data = (W[f"b.{p}.query_proj.weight"], W[f"b.{p}.key_proj.weight"],
                                                                                W[f"b.{p}.value_proj.weight"])
        m.out_proj.weight.data = W[f"b.{p}.out_proj.weight"]
        o, _ = m(q_in, kv, kv, need_weights=False)
        return o + W[f"b.{p}.out_proj.bias"]

    ln = lambda t, k: F.layer_norm(t, (C,), W[f"b.norm{k}.weight"], W[f"b.norm{k}.bias"], 1e-5)   # noqa: E731
    y = x + attn("attn1", ln(x, 1), ln(x, 1))
    y = y + attn("attn2", ln(y, 2), mem)
    n3 = ln(y, 3)
    ff = F.linear(F.linear(n3, W["b.linear1.weight"], W["b.linear1.bias"]) * F.gelu(F.linear(n3, W["b.linear2.weight"], W["b.linear2.bias"])),
                  W["b.linear3.weight"], W["b.linear3.bias"])
    assert rel_l2(got, y + ff) < 1e-5

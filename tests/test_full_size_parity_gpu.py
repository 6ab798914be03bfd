"""Full-DEPTH, full-SIZE Flux parity against the CPU oracle — the headline configuration itself (BASELINE.json
configs[1]: Flux-schnell, 19 double + 38 single blocks, width 3072, 512 x 512 -> L = 1024, S = 256, T = 1280, batch 1).

Everything else in tests/ compares depth <= 2 + 3 models (or single full-width blocks) with the oracle and checks the
57-block model through properties; here the whole model runs on both sides, so the compounding of the bf16-storage
rounding through 57 gated residual blocks at width 3072 is MEASURED (round 4, profiles/r04_parity_full_size.json):

  a1  one forward (t = 1.0) against the oracle in bf16 mode = the REFERENCE'S OWN ARITHMETIC (the MLX pipeline is bf16,
      flux/flux.py:24: every op returns a bf16 array)                     rel-L2(pred)   <= 1e-2    measured 5.2e-3
      ... and against the fp32 oracle: no further from it than the reference's arithmetic is
          rel-L2(HIP, fp32) <= 1.1 x rel-L2(bf16 oracle, fp32) + 5e-4, and <= 2e-2          measured 1.415e-2 vs 1.416e-2
      What this says: at depth 57 BOTH bf16 pipelines sit 1.4e-2 from the fp32 answer and 5e-3 from each other — most
      of the distance is the bf16 rounding of the residual stream, which two implementations that agree to < 1 ulp
      round the SAME way (test_c2_which_rounding_compounds isolates it).  An fp32 residual stream would move the HIP path
      towards the fp32 oracle and AWAY from what the reference computes, so it is not what the path does.
  a2  the 2-step schnell loop through FluxPipeline._denoising_loop (hipGraph replay + hoisted modulation tables, i.e. the
      product's own loop) against the fp32 oracle loop                     rel-L2(latents) <= 2e-2  measured 5.7e-3
  a3  decode of the final latents (fp32-faithful VAE, 64 x 64 x 16 -> 512 x 512 x 3) against oracle.pipeline_decode on
      the SAME latents                                                     max-abs <= 1/255         measured 2.5e-5
      (end to end against the all-fp32-oracle image: max-abs 1.1e-2, mean-abs 1.2e-3 — reported, not asserted)
  b   `enable_fp8()` forward vs the fp32 oracle on the DE-QUANTISED weights   rel-L2(pred) <= 2.5e-2  measured 1.55e-2
  c   the same forward with every modulation bias drawn from U(-0.5, 0.5) (gates / shifts / scales of O(0.3), where the
      default init leaves them at O(0.03) and every block close to the identity): the residual stream then really is
      rewritten 57 times: HIP vs the float32 oracle <= 1.1 x (bf16 oracle vs float32) + 5e-4 (measured 1.534e-2 vs
      1.535e-2); HIP vs the bf16 oracle <= 2e-2 (measured 1.25e-2: the two bf16 pipelines decorrelate when the blocks do more)
  d   STORED oracle outputs (tests/golden/full_size/, tools/make_full_size_golden.py): Flux-dev 1024 x 1024 (T = 4608, guidance
      embedding), the fp8 plan at B = 4 / T = 4352 on de-quantised weights, the SDXL UNet at batch 16 - see the end of the file;
      the `which rounding compounds` diagnostic of round 4 is a mode of that tool

Both sides use the SAME weights: drawn on the GPU (`init_random`, bf16-representable), fetched to the host one tensor
at a time while the oracle walks the blocks (`DeviceWeights`), so only one block's fp32 weights are resident
on the host.  References: flux/model.py:99-136, flux/flux.py:87-126,157-162, flux/sampler.py:22-31,56-57.
The measured numbers are written to gpurun_out/parity_full_size.json (committed copy: profiles/r04_parity_full_size.json).
"""
import json
import os
import time
import warnings
from collections.abc import Mapping

import pytest
import torch

from conftest import ROOT, rel_l2
from oracle import flux_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
RESULTS = {}
# bounds of the stored-oracle tests: 1.5 x the error measured when the fixtures were generated (see each fixture's `measured`)
C3_BOUND_VS_BF16, C3_BOUND_VS_FP32, C5_BOUND, C4_BOUND = 8.6e-3, 2e-2, 2.3e-2, 1.3e-3      # measured 5.74e-3, 1.426e-2, 1.54e-2 (worst image 1.55e-2), 8.6e-4
# round 6, the "live" fixtures (modulation biases U(-0.5, 0.5): every block rewrites the stream, tests/full_size_cases.py):
# measured 1.220e-2 / 1.460e-2 (C3) and 2.998e-2, worst image 3.03e-2 (C5); the committed mutation (one layer's E8M0 scales one
# exponent up) measured 8.52e-2 at generation
C3_LIVE_BOUND_VS_BF16, C3_LIVE_BOUND_VS_FP32, C5_LIVE_BOUND = 1.83e-2, 2.2e-2, 4.5e-2


class DeviceWeights(Mapping):
    """The oracle's weight dict, backed by the HIP model's own device tensors: a key is converted on the GPU and copied
    into a pinned host staging buffer when the oracle asks for it (two buffers per shape, used alternately: every oracle
    op consumes its weight before the next one of that shape is requested), so one block's weights are resident at a time
    and nothing is page-faulted per fetch.  `dequant`: name -> (e4m3 rows, scale) for the layers whose fp8 copy is what
    the HIP path multiplies with."""
    _stage = {}

    def __init__(self, params, dequant=None, dtype=torch.float32):
        self.p, self.dq, self.dtype = params, dequant or {}, dtype

    def __getitem__(self, k):
        base = k[: -len(".weight")] if k.endswith(".weight") else None
        if base in self.dq:
            q, sc = self.dq[base]
            src = (q.view(torch.float8_e4m3fn).float() * sc.float()[:, None]).to(self.dtype)
        else:
            src = self.p[k].to(self.dtype)
        key = (tuple(src.shape), self.dtype)
        slot = DeviceWeights._stage.setdefault(key, [0, [torch.empty(src.shape, dtype=self.dtype, pin_memory=True) for _ in range(2)]])
        slot[0] ^= 1
        buf = slot[1][slot[0]]
        buf.copy_(src)
        return buf

    def __iter__(self):
        return iter(self.p)

    def __len__(self):
        return len(self.p)


def _save():
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_full_size.json"), "w") as f:
        json.dump(RESULTS, f, indent=1, sort_keys=True)


def _inputs(P, S, lat, seed=11):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(1, lat, lat, 16, generator=g).to(BF)
    img, img_ids = O.prepare_latent_images(z)
    txt = (torch.randn(1, S, P.context_in_dim, generator=g) * 0.5).to(BF)
    txt_ids = torch.zeros(1, S, 3, dtype=torch.int32)
    vec = torch.randn(1, P.vec_in_dim, generator=g).to(BF)
    return img, img_ids, txt, txt_ids, vec


@pytest.fixture(scope="module")
def schnell(dev):
    """The bench's own model: FluxPipeline('flux-schnell') with the seed-0 random init (no checkpoint in this image)."""
    from flux_generator_amd.flux import FluxPipeline
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pipe = FluxPipeline("flux-schnell")
    P = pipe.flow.params
    assert (P.depth, P.depth_single_blocks, P.hidden_size, P.guidance_embed) == (19, 38, 3072, False)
    OP = O.FluxParams(**{k: getattr(P, k) for k in O.FluxParams.__dataclass_fields__})
    st = dict(pipe=pipe, OP=OP, inputs=_inputs(P, 256, 64))
    yield st
    _save()


def _oracle_forward(OP, W, inputs, t, guidance=None, dtype=torch.float32):
    img, img_ids, txt, txt_ids, vec = inputs
    tt = torch.full((1,), t, dtype=BF).to(dtype)
    gd = None if guidance is None else torch.full((1,), guidance, dtype=BF).to(dtype)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = O.flux_forward(OP, W, img.to(dtype), img_ids, txt.to(dtype), txt_ids, tt, vec.to(dtype), gd)
    return ref, time.perf_counter() - t0


def test_c2_full_depth_forward_loop_decode(dev, schnell):
    """a1 - a4 of the module docstring."""
    pipe, OP, inputs = schnell["pipe"], schnell["OP"], schnell["inputs"]
    img, img_ids, txt, txt_ids, vec = inputs
    W = DeviceWeights(pipe.flow.parameters())
    d = [a.to(dev) for a in inputs]

    # ---- a1: one forward, eager plan (Flux.__call__) and the oracle at t = 1.0 (step 0 of the schnell schedule)
    ts = pipe.sampler.timesteps(2, img.shape[1])
    assert ts == O.timesteps("flux-schnell", 2, img.shape[1]) == [1.0, 0.5, 0.0]
    got = pipe.flow(d[0], d[1], d[2], d[3], torch.full((1,), ts[0], dtype=BF, device=dev), d[4])
    ref0, secs = _oracle_forward(OP, W, inputs, ts[0])
    e_fwd = rel_l2(got, ref0)
    print(f"[a1] full-depth forward T=1280: rel-L2 vs fp32 oracle {e_fwd:.3e}  (oracle {secs:.0f} s on {torch.get_num_threads()} threads)")
    RESULTS["c2_forward_rel_l2_vs_fp32"] = e_fwd
    RESULTS["oracle_forward_seconds"] = secs
    RESULTS["host_threads"] = torch.get_num_threads()

    # ---- a4: the reference's own arithmetic (oracle in bf16: every op output rounded to bf16) against the same fp32 run
    Wb = DeviceWeights(pipe.flow.parameters(), dtype=BF)
    ref16, secs16 = _oracle_forward(OP, Wb, inputs, ts[0], dtype=BF)
    e_ref16 = rel_l2(ref16, ref0)
    e_hip16 = rel_l2(got, ref16)
    print(f"[a4] bf16 oracle vs fp32 oracle {e_ref16:.3e};  HIP vs bf16 oracle {e_hip16:.3e}  ({secs16:.0f} s)")
    RESULTS["c2_bf16_oracle_vs_fp32"] = e_ref16
    RESULTS["c2_forward_rel_l2_vs_bf16_oracle"] = e_hip16

    # ---- a2: the product's 2-step loop (graph replay, hoisted modulation tables) vs the fp32 oracle loop
    xs = list(pipe._denoising_loop(d[0], d[1], d[2], d[3], d[4], num_steps=2, guidance=4.0))
    x1_ref = O.euler_step(ref0, img.float(), ts[0], ts[1])
    e_x1 = rel_l2(xs[0], x1_ref)
    ref1, _ = _oracle_forward(OP, W, (x1_ref, img_ids, txt, txt_ids, vec), ts[1])
    x2_ref = O.euler_step(ref1, x1_ref, ts[1], ts[2])
    e_x2 = rel_l2(xs[1], x2_ref)
    print(f"[a2] latents after step 1: {e_x1:.3e}, after the 2-step loop: {e_x2:.3e}")
    RESULTS["c2_latents_step1_rel_l2"] = e_x1
    RESULTS["c2_latents_loop_rel_l2"] = e_x2

    # ---- a3: decode of the SAME (HIP) latents on both sides
    image = pipe.decode(xs[1], (64, 64))
    AP = O.AutoEncoderParams(**{k: getattr(pipe.ae.params, k) for k in O.AutoEncoderParams.__dataclass_fields__})
    WA = {k: v.float().cpu() for k, v in pipe.ae.parameters().items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        ref_img = O.pipeline_decode(AP, WA, xs[1].float().cpu(), (64, 64))
    d_img = float((image.float().cpu() - ref_img).abs().max())
    with torch.no_grad():                                       # and end to end: oracle latents through the oracle decoder
        ref_e2e = O.pipeline_decode(AP, WA, x2_ref.to(BF).float(), (64, 64))
    d_e2e = float((image.float().cpu() - ref_e2e).abs().max())
    m_e2e = float((image.float().cpu() - ref_e2e).abs().mean())
    print(f"[a3] 512x512 image: max-abs vs oracle decode of the same latents {d_img:.3e} "
          f"({time.perf_counter() - t0:.0f} s); end to end vs the all-oracle image max-abs {d_e2e:.3e} mean-abs {m_e2e:.3e}")
    RESULTS["c2_image_max_abs_same_latents"] = d_img
    RESULTS["c2_image_max_abs_end_to_end"] = d_e2e
    RESULTS["c2_image_mean_abs_end_to_end"] = m_e2e
    _save()

    schnell["ref0"] = ref0
    assert image.shape == (1, 512, 512, 3)
    assert e_hip16 <= 1e-2, "full-depth forward vs the reference's (bf16) arithmetic"
    assert e_fwd <= 1.1 * e_ref16 + 5e-4 and e_fwd <= 2e-2, "HIP path is further from fp32 than the reference's bf16 arithmetic"
    assert e_x1 <= 2e-2 and e_x2 <= 2e-2, "2-step loop"
    assert d_img <= 1.0 / 255, "decode of identical latents"


def test_c2_full_depth_live_modulation(dev, schnell):
    """c: gates / shifts / scales of O(0.3) — every block really rewrites the residual stream (the default init leaves
    the modulation at O(0.03)).  Runs after the default-weights test (it edits the modulation biases in place)."""
    pipe, OP, inputs = schnell["pipe"], schnell["OP"], schnell["inputs"]
    flow = pipe.flow
    keep = flow.mod_b.clone()
    try:
        g = torch.Generator(device=dev).manual_seed(5)
        flow.mod_b.copy_(((torch.rand(flow.mod_b.shape, generator=g, device=dev) - 0.5)).to(BF))
        d = [a.to(dev) for a in inputs]
        got = flow(d[0], d[1], d[2], d[3], torch.full((1,), 0.5, dtype=BF, device=dev), d[4])
        ref, secs = _oracle_forward(OP, DeviceWeights(flow.parameters()), inputs, 0.5)
        ref16, _ = _oracle_forward(OP, DeviceWeights(flow.parameters(), dtype=BF), inputs, 0.5, dtype=BF)
        e, e16, eh = rel_l2(got, ref), rel_l2(ref16, ref), rel_l2(got, ref16)
        print(f"[c] live modulation (bias U(-0.5,0.5)): HIP vs bf16 oracle {eh:.3e}; HIP vs fp32 oracle {e:.3e}; "
              f"bf16 oracle vs fp32 {e16:.3e}  ({secs:.0f} s)")
        RESULTS["c2_live_modulation_rel_l2_vs_bf16_oracle"] = eh
        RESULTS["c2_live_modulation_rel_l2_vs_fp32"] = e
        RESULTS["c2_live_modulation_bf16_oracle_vs_fp32"] = e16
        _save()
        # with every block really rewriting the stream the two bf16 pipelines decorrelate: each sits ~1.5e-2 from float32
        # (measured 1.534e-2 / 1.535e-2) and they are 1.25e-2 apart (independent errors of that size would be 2.2e-2 apart)
        assert eh <= 2e-2
        assert e <= 1.1 * e16 + 5e-4 and e <= 2e-2
    finally:
        flow.mod_b.copy_(keep)


def test_c2_full_depth_fp8_forward(dev, schnell):
    """b: the `--quantize` path (e4m3 weights per output channel, per-token e4m3 activations, fp8 MFMA) at full depth vs
    the fp32 oracle on the de-quantised weights (so what is measured is the activation quantisation + bf16 storage)."""
    pipe, OP, inputs = schnell["pipe"], schnell["OP"], schnell["inputs"]
    flow = pipe.flow
    flow.enable_fp8(True)
    try:
        d = [a.to(dev) for a in inputs]
        got = flow(d[0], d[1], d[2], d[3], torch.full((1,), 1.0, dtype=BF, device=dev), d[4])
        W = DeviceWeights(flow.parameters(), dequant=flow._w8)
        ref, secs = _oracle_forward(OP, W, inputs, 1.0)
        e = rel_l2(got, ref)
        print(f"[b] fp8 full-depth forward vs fp32 oracle on de-quantised weights: {e:.3e}  ({secs:.0f} s)")
        RESULTS["c2_fp8_forward_rel_l2_vs_dequant_fp32"] = e
        _save()
        assert e <= 2.5e-2          # measured 1.55e-2 (round 4), 1.5 x
    finally:
        flow.enable_fp8(False)


# ---------------------------------------------------------------------------------------------------------------------
# Stored oracle outputs (round 5): the full-size configurations whose oracle forward takes minutes of host time are compared
# with vectors the oracle produced ONCE on a GPU box (tools/make_full_size_golden.py -> tests/golden/full_size/*.pt; inputs and
# weights are regenerated from their seeds, and the weights' fingerprint must match the fixture's).  No test in this file is
# skipped any more: the two env-gated live-oracle runs of round 4 live in that tool (`c3`, `rounding`).
# Bounds = 1.5 x the error measured when the fixture was generated (recorded inside the fixture as `measured`).
import full_size_cases as FC      # noqa: E402


def _check_hash(case, gold, what):
    assert case["hash"] == gold["weight_hash"], (
        f"{what}: the regenerated weights do not fingerprint like the ones the stored oracle output was computed with "
        "(another torch / device Philox stream?): regenerate with tools/make_full_size_golden.py")


@pytest.mark.parametrize("live", [False, True], ids=["default_init", "live_modulation"])
def test_c3_dev_1024_forward_vs_stored_oracle(dev, live):
    """BASELINE.json configs[2]'s shape - Flux-dev, S = 512, L = 4096, T = 4608, guidance 7 - one full-depth forward against
    the stored fp32 oracle output and the stored output of the oracle in the reference's own (bf16) arithmetic
    (flux/model.py:99-136).  Two weight sets: `init_random` as drawn (modulation O(0.03): blocks close to the identity) and
    the same weights with O(0.3) modulation (live: every one of the 57 blocks rewrites the residual stream, so a defect inside
    a block reaches the output at full size)."""
    gold = FC.load_golden("c3_dev_t4608_live.pt" if live else "c3_dev_t4608.pt")
    case = FC.c3_case(dev, live=live)
    _check_hash(case, gold, "c3")
    got = FC.c3_forward(case, dev)
    e, eh = rel_l2(got, gold["ref_fp32"].float()), rel_l2(got, gold["ref_bf16"].float())
    m = gold["measured"]
    tag = "c3_live" if live else "c3"
    print(f"[{tag}] Flux-dev T=4608: HIP vs stored fp32 oracle {e:.3e} (at generation {m['hip_vs_fp32']:.3e}); vs stored bf16 oracle "
          f"{eh:.3e} ({m['hip_vs_bf16_oracle']:.3e}); bf16 oracle vs fp32 {m['bf16_oracle_vs_fp32']:.3e}")
    RESULTS[f"{tag}_dev1024_forward_rel_l2_vs_fp32"] = e
    RESULTS[f"{tag}_dev1024_forward_rel_l2_vs_bf16_oracle"] = eh
    RESULTS[f"{tag}_dev1024_bf16_oracle_vs_fp32"] = m["bf16_oracle_vs_fp32"]
    _save()
    assert eh <= (C3_LIVE_BOUND_VS_BF16 if live else C3_BOUND_VS_BF16)
    assert e <= 1.1 * m["bf16_oracle_vs_fp32"] + 5e-4 and e <= (C3_LIVE_BOUND_VS_FP32 if live else C3_BOUND_VS_FP32)


@pytest.mark.parametrize("live", [False, True], ids=["default_init", "live_modulation"])
def test_c5_fp8_b4_forward_vs_stored_oracle(dev, live):
    """BASELINE.json configs[4] at its per-GPU shape - Flux-schnell, fp8 plan (block-scaled hand-off), B = 4 DISTINCT
    images, T = 4352 - against the stored fp32 oracle output on the de-quantised weights (txt2image.py:79-82): an oracle
    bound for the batch-4 launch plan itself.  default_init: gates of O(0.03) scale every block's contribution down - the fp8
    plan measured 1.543e-2 where the bf16 plan on the same weights measured 1.537e-2, i.e. that fixture cannot see the e4m3
    arithmetic.  live_modulation (round 6): O(0.3) gates - the two plans are then 4.2e-2 apart (each 3.0-3.1e-2 from float32:
    independent errors), and the mutation check below fails this very assertion."""
    gold = FC.load_golden("c5_fp8_b4_t4352_live.pt" if live else "c5_fp8_b4_t4352.pt")
    case = FC.c5_case(dev, live=live)
    _check_hash(case, gold, "c5")
    got = FC.c5_forward(case, dev)
    ref = gold["ref_fp32"].float()
    e = rel_l2(got, ref)
    per = [rel_l2(got[i], ref[i]) for i in range(4)]
    m = gold["measured"]
    tag = "c5_live" if live else "c5"
    print(f"[{tag}] fp8 B=4 T=4352 vs stored fp32 oracle on de-quantised weights: {e:.3e} (at generation {m['fp8_vs_dequant_fp32']:.3e}); "
          f"per image {[f'{v:.3e}' for v in per]}; the bf16 plan on the same weights was {m['bf16_plan_vs_dequant_fp32']:.3e}, "
          f"the two plans {m['fp8_vs_bf16_plan']:.3e} apart")
    RESULTS[f"{tag}_fp8_b4_forward_rel_l2_vs_dequant_fp32"] = e
    bound = C5_LIVE_BOUND if live else C5_BOUND
    ok = bool(torch.isfinite(got).all()) and e <= bound and max(per) <= bound
    if live:
        # Mutation check (review of round 5): ONE layer's E8M0 block scales shifted by one exponent (Flux.set_debug_mx_shift:
        # the consumer of single_blocks.19.linear1's GELU half reads scales 2x too large) must make THIS test's assertion fail.
        case["flow"].set_debug_mx_shift(FC.C5_MUTATED_LAYER, 1)
        try:
            bad = FC.c5_forward(case, dev)
        finally:
            case["flow"].set_debug_mx_shift(None)
        eb = rel_l2(bad, ref)
        perb = [rel_l2(bad[i], ref[i]) for i in range(4)]
        print(f"[{tag}] mutated ({FC.C5_MUTATED_LAYER}: E8M0 + 1): {eb:.3e} (at generation {m['mutated_fp8_vs_dequant_fp32']:.3e}); bound {bound:.1e}")
        RESULTS[f"{tag}_mutated_rel_l2"] = eb
        again = FC.c5_forward(case, dev)                         # the knob is off again: the unmutated bits are back
        assert torch.equal(again, got)
        assert eb > bound and min(perb) > bound, "the stored vector does not see a one-exponent error in one layer's block scales"
    _save()
    assert ok, (e, per, bound)


def test_c4_sdxl_b16_vs_stored_oracle(dev):
    """BASELINE.json configs[3] at its batch: the full-size SDXL UNet in float16 on 16 DISTINCT latents / text states; images
    0 and 11 of the batch against the stored fp32 oracle outputs (stable_diffusion/stable_diffusion/unet.py:403-460)."""
    gold = FC.load_golden("c4_sdxl_b16.pt")
    case = FC.c4_case(dev)
    _check_hash(case, gold, "c4")
    got = FC.c4_forward(case, dev)
    assert got.shape == (16, 64, 64, 4) and bool(torch.isfinite(got).all())
    for i, ref in gold["ref_fp32"].items():
        e = rel_l2(got[i:i + 1], ref)
        print(f"[c4] SDXL UNet float16 B=16, image {i} vs stored fp32 oracle: {e:.3e} (at generation {gold['measured']['per_image'][str(i)]:.3e})")
        RESULTS[f"c4_sdxl_b16_image{i}_rel_l2_vs_fp32"] = e
        assert e <= C4_BOUND
    _save()


def test_c4_sdxl_float32_arithmetic_vs_stored_oracle(dev):
    """The full-size SDXL UNet in the reference's FLOAT32 arithmetic (float16=False, its default: stable_diffusion/__init__.py:19-25)
    on the float32-faithful kernels (flux_generator_amd/stable_diffusion/unet_f32.py) against the SAME stored float32 oracle
    outputs the float16 test above uses: the float16 case's weights and inputs, upcast exactly, are what the oracle saw.  Where
    float16 arithmetic measures 8.6e-4, float32 arithmetic must be at the oracle's own rounding level (bound 1e-4)."""
    from flux_generator_amd.stable_diffusion.config import UNetConfig
    from flux_generator_amd.stable_diffusion.unet import UNetModel
    gold = FC.load_golden("c4_sdxl_b16.pt")
    case = FC.c4_case(dev)
    _check_hash(case, gold, "c4")
    model = UNetModel(UNetConfig(**case["kw"]), device=dev, dtype=torch.float32)
    model.load_weights({k: v.float() for k, v in case["model"].parameters().items()})
    del case["model"]
    torch.cuda.empty_cache()
    for i, ref in gold["ref_fp32"].items():
        t0 = time.perf_counter()
        got = model(case["x"][i:i + 1].float().to(dev), case["t"][i:i + 1].to(dev), case["enc"][i:i + 1].float().to(dev),
                    text_time=(case["pooled"][i:i + 1].float().to(dev), case["tid"][i:i + 1].to(dev)))
        torch.cuda.synchronize()
        e = rel_l2(got, ref)
        print(f"[c4 float32] SDXL UNet float32 arithmetic, image {i} vs stored fp32 oracle: {e:.3e}  ({time.perf_counter() - t0:.2f} s incl. first-use set-up)")
        RESULTS[f"c4_sdxl_float32_image{i}_rel_l2_vs_fp32"] = e
        assert got.dtype == torch.float32 and got.shape == (1, 64, 64, 4) and e <= 1e-4
    _save()


def test_c3_dev_1024_loop_steps_vs_stored_oracle(dev):
    """BASELINE.json configs[2] through the PRODUCT's loop (review of round 5, weak #3: the 28-step shifted-schedule loop at full size
    was covered by properties and a tiny golden only): the first three latents of FluxPipeline._denoising_loop(num_steps=28) -
    time-shifted schedule (flux/sampler.py:22-31), modulation tables of all 28 steps hoisted, graph replay, Euler kernel - at
    T = 4608 with live modulation, against the stored float32 oracle loop (flux/flux.py:87-126).  Bounds = 1.5 x the error
    measured at generation, per step (1.65e-3 / 2.92e-3 / 4.10e-3; the reference is stored as float16, which adds ~3e-4)."""
    gold = FC.load_golden("c3_dev_t4608_loop3.pt")
    case = FC.c3_loop_case(dev)
    _check_hash(case, gold, "c3 loop")
    assert case["pipe"].sampler.timesteps(28, 4096)[: FC.C3_LOOP_STEPS + 1] == gold["measured"]["timesteps"]
    got = FC.c3_loop_forward(case, dev)
    for i, (g, r) in enumerate(zip(got, gold["ref_fp32"])):
        e, at_gen = rel_l2(g, r.float()), gold["measured"]["latents_rel_l2"][i]
        print(f"[c3 loop] latents after step {i + 1} of 28: {e:.3e} (at generation {at_gen:.3e})")
        RESULTS[f"c3_loop_step{i + 1}_latents_rel_l2_vs_fp32"] = e
        assert e <= 1.5 * at_gen + 1e-4
    _save()

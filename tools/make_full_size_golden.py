#!/usr/bin/env python3
"""Generate tests/golden/full_size/*.pt on a GPU box: the CPU oracle's outputs for the full-size configurations whose
oracle forward is too long for the driver's `-m gpu` run (minutes of host time each), stored once and compared against by
tests/test_full_size_parity_gpu.py on every run.

For every case (tests/full_size_cases.py) this script
  1. draws the model's seeded weights ON THE DEVICE (`init_random(seed)`; C5: + the fp8 quantiser) and fingerprints them,
  2. runs the HIP forward (so the fixture also records the error measured at generation time),
  3. runs the oracle (oracle/flux_oracle.py, oracle/sd_oracle.py) on the host cores with the SAME weights, fetched tensor
     by tensor through pinned staging (tests/test_full_size_parity_gpu.py::DeviceWeights; C5: the de-quantised e4m3 weights),
  4. stores ONLY outputs + fingerprint + the measured numbers (float16 / bfloat16 tensors, <= 2 MB per case).
Inputs are regenerated from their seeds by the tests; nothing under /root/reference is read (it does not exist on the box).

Also here (moved out of tests/ in round 5 so that the default GPU suite has no skipped tests): the diagnostic
`which rounding compounds` run of round 4 (fp32 oracle with only the residual stream rounded to bf16 between blocks).

usage (GPU box):  python tools/make_full_size_golden.py [c4] [c3] [c5] [c3live] [c5live] [c3loop] [rounding]      (default: c4 c3 c5)
(`c3live` / `c5live`, round 6: the same cases with O(0.3) modulation - tests/full_size_cases.py - and, for c5, the measured
effect of the committed mutation: one layer's E8M0 block scales shifted by one exponent)
Writes tests/golden/full_size/<case>.pt and gpurun_out/full_size_golden/<case>.pt + summary.json (gpurun merges the latter
back; copy the .pt files into tests/golden/full_size/ and commit them)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("FLUX_ALLOW_RANDOM_INIT", "1")

import torch  # noqa: E402

import full_size_cases as FC  # noqa: E402
from oracle import flux_oracle as O  # noqa: E402

BF = torch.bfloat16
OUT_DIRS = [FC.GOLDEN_DIR, os.path.join(ROOT, "gpurun_out", "full_size_golden")]
SUMMARY = {}


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def save(name, obj):
    for d in OUT_DIRS:
        os.makedirs(d, exist_ok=True)
        torch.save(obj, os.path.join(d, name))
    with open(os.path.join(OUT_DIRS[1], "summary.json"), "w") as f:
        json.dump(SUMMARY, f, indent=1, sort_keys=True)


def device_weights(params, dequant=None, dtype=torch.float32):
    from test_full_size_parity_gpu import DeviceWeights
    return DeviceWeights(params, dequant=dequant, dtype=dtype)


def oracle_flux(OP, W, inputs, t, guidance=None, dtype=torch.float32):
    """One oracle forward per image (the oracle materialises the [H, T, T] scores: 1.8 GB per image at T = 4352)."""
    img, img_ids, txt, txt_ids, vec = inputs
    outs = []
    t0 = time.perf_counter()
    for i in range(img.shape[0]):
        tt = torch.full((1,), t, dtype=BF).to(dtype)
        gd = None if guidance is None else torch.full((1,), guidance, dtype=BF).to(dtype)
        with torch.no_grad():
            outs.append(O.flux_forward(OP, W, img[i:i + 1].to(dtype), img_ids[i:i + 1], txt[i:i + 1].to(dtype), txt_ids[i:i + 1],
                                       tt, vec[i:i + 1].to(dtype), gd))
        print(f"   oracle image {i}: {time.perf_counter() - t0:.0f} s", flush=True)
    return torch.cat(outs, 0), time.perf_counter() - t0


def make_c3(dev, live=False):
    case = FC.c3_case(dev, live=live)
    P = case["P"]
    OP = O.FluxParams(**{k: getattr(P, k) for k in O.FluxParams.__dataclass_fields__})
    got = FC.c3_forward(case, dev)
    ref, secs = oracle_flux(OP, device_weights(case["flow"].parameters()), case["inputs"], case["t"], case["guidance"])
    ref16, secs16 = oracle_flux(OP, device_weights(case["flow"].parameters(), dtype=BF), case["inputs"], case["t"], case["guidance"], dtype=BF)
    m = dict(hip_vs_fp32=rel_l2(got, ref), hip_vs_bf16_oracle=rel_l2(got, ref16), bf16_oracle_vs_fp32=rel_l2(ref16, ref),
             oracle_seconds=secs, oracle_bf16_seconds=secs16, host_threads=torch.get_num_threads())
    tag = "c3_live" if live else "c3"
    print(tag, m, flush=True)
    SUMMARY[tag] = m
    save("c3_dev_t4608_live.pt" if live else "c3_dev_t4608.pt",
         dict(ref_fp32=ref.to(torch.float16), ref_bf16=ref16.to(BF), weight_hash=case["hash"], measured=m,
              meta="Flux-dev init_random(4)" + (" + modulation biases U(-0.5, 0.5) (live_modulation seed 14)" if live else "") +
                   ", inputs seed 2, S=512 L=4096, t=timesteps(28)[1], guidance 7; "
                   "ref_fp32 = fp32 oracle (stored float16), ref_bf16 = oracle in the reference's bf16 arithmetic"))


def make_c3loop(dev):
    """Three steps of Flux-dev's 28-step loop at 1024 x 1024: the product's loop against the float32 oracle loop
    (flux/flux.py:87-126, flux/sampler.py:22-57)."""
    case = FC.c3_loop_case(dev)
    P = case["P"]
    OP = O.FluxParams(**{k: getattr(P, k) for k in O.FluxParams.__dataclass_fields__})
    got = FC.c3_loop_forward(case, dev)
    img, img_ids, txt, txt_ids, vec = case["inputs"]
    ts = O.timesteps("flux-dev", 28, img.shape[1])
    assert ts == case["pipe"].sampler.timesteps(28, img.shape[1])
    W = device_weights(case["pipe"].flow.parameters())
    x, refs, t0 = img.float(), [], time.perf_counter()
    for i in range(FC.C3_LOOP_STEPS):
        pred, _ = oracle_flux(OP, W, (x, img_ids, txt, txt_ids, vec), ts[i], case["guidance"])
        x = O.euler_step(pred, x, ts[i], ts[i + 1])
        refs.append(x.clone())
    m = dict(latents_rel_l2=[rel_l2(g, r) for g, r in zip(got, refs)], timesteps=ts[: FC.C3_LOOP_STEPS + 1],
             oracle_seconds=time.perf_counter() - t0, host_threads=torch.get_num_threads())
    print("c3_loop", m, flush=True)
    SUMMARY["c3_loop"] = m
    save("c3_dev_t4608_loop3.pt", dict(ref_fp32=[r.to(torch.float16) for r in refs], weight_hash=case["hash"], measured=m,
                                       meta="FluxPipeline('flux-dev') seed-0 weights + live_modulation(15), inputs seed 8, S=512 L=4096, "
                                            "guidance 7: latents after steps 1..3 of the 28-step schedule, float32 oracle loop (stored float16)"))


def make_c5(dev, live=False):
    case = FC.c5_case(dev, live=live)
    P = case["P"]
    flow = case["flow"]
    OP = O.FluxParams(**{k: getattr(P, k) for k in O.FluxParams.__dataclass_fields__})
    got = FC.c5_forward(case, dev)
    flow.enable_fp8(False)
    got16 = FC.c5_forward(case, dev)
    flow.enable_fp8(True)
    ref, secs = oracle_flux(OP, device_weights(flow.parameters(), dequant=flow._w8), case["inputs"], case["t"])
    m = dict(fp8_vs_dequant_fp32=rel_l2(got, ref), bf16_plan_vs_dequant_fp32=rel_l2(got16, ref), fp8_vs_bf16_plan=rel_l2(got, got16),
             per_image=[rel_l2(got[i], ref[i]) for i in range(4)], oracle_seconds=secs, host_threads=torch.get_num_threads())
    # the mutation the stored vector must be able to see: ONE layer's E8M0 block scales one exponent too large
    flow.set_debug_mx_shift(FC.C5_MUTATED_LAYER, 1)
    bad = FC.c5_forward(case, dev)
    flow.set_debug_mx_shift(None)
    m["mutated_layer"] = FC.C5_MUTATED_LAYER
    m["mutated_fp8_vs_dequant_fp32"] = rel_l2(bad, ref)
    m["mutated_per_image"] = [rel_l2(bad[i], ref[i]) for i in range(4)]
    tag = "c5_live" if live else "c5"
    print(tag, m, flush=True)
    SUMMARY[tag] = m
    save("c5_fp8_b4_t4352_live.pt" if live else "c5_fp8_b4_t4352.pt",
         dict(ref_fp32=ref.to(torch.float16), weight_hash=case["hash"], measured=m,
              meta="Flux-schnell init_random(3)" + (" + modulation biases U(-0.5, 0.5) (live_modulation seed 13)" if live else "") +
                   " + enable_fp8, inputs seed 6, B=4 S=256 L=4096, t=0.75; "
                   "fp32 oracle on the DE-QUANTISED e4m3 weights (stored float16)"))


def make_c4(dev):
    from oracle import sd_oracle as S
    case = FC.c4_case(dev)
    got = FC.c4_forward(case, dev)
    ocfg = S.UNetConfig(**case["kw"])
    Wc = {k: v.float().cpu() for k, v in case["model"].parameters().items()}
    refs, t0 = {}, time.perf_counter()
    for i in FC.C4_ORACLE_IMAGES:
        with torch.no_grad():
            refs[i] = S.unet_forward(ocfg, Wc, case["x"][i:i + 1].float(), case["t"][i:i + 1], case["enc"][i:i + 1].float(),
                                     (case["pooled"][i:i + 1].float(), case["tid"][i:i + 1]))
    m = dict(per_image={str(i): rel_l2(got[i:i + 1], refs[i]) for i in refs}, oracle_seconds=time.perf_counter() - t0,
             host_threads=torch.get_num_threads())
    print("c4", m, flush=True)
    SUMMARY["c4"] = m
    save("c4_sdxl_b16.pt", dict(ref_fp32={i: r.to(torch.float32) for i, r in refs.items()}, weight_hash=case["hash"], measured=m,
                                meta="SDXL UNet init_random(5) float16, 16 distinct inputs seed 21, t=999; fp32 oracle of images 0 and 11"))


def make_rounding(dev):
    """Round 4's `which rounding compounds` diagnostic at C2's shape (was tests/...::test_c2_which_rounding_compounds)."""
    import warnings
    from flux_generator_amd.flux import FluxPipeline
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pipe = FluxPipeline("flux-schnell")
    P = pipe.flow.params
    OP = O.FluxParams(**{k: getattr(P, k) for k in O.FluxParams.__dataclass_fields__})
    inputs = FC.flux_inputs(P, 1, 256, 64, seed=11)
    W = device_weights(pipe.flow.parameters())
    ref0, _ = oracle_flux(OP, W, inputs, 1.0)
    img, img_ids, txt, txt_ids, y = inputs
    rb = lambda x: x.to(BF).float()      # noqa: E731
    with torch.no_grad():
        tt = torch.full((1,), 1.0, dtype=BF)
        x_img = O.linear(img.float(), W["img_in.weight"], W["img_in.bias"])
        vec = O.mlp_embedder(W, "time_in", O.timestep_embedding(tt, 256).float()) + O.mlp_embedder(W, "vector_in", y.float())
        x_txt = O.linear(txt.float(), W["txt_in.weight"], W["txt_in.bias"])
        pe = O.embed_nd(torch.cat([txt_ids, img_ids], dim=1), OP.axes_dim, OP.theta).to(BF).float()
        x_img, x_txt = rb(x_img), rb(x_txt)
        for i in range(OP.depth):
            x_img, x_txt = O.double_stream_block(W, f"double_blocks.{i}", OP.num_heads, x_img, x_txt, vec, pe)
            x_img, x_txt = rb(x_img), rb(x_txt)
        x = torch.cat([x_txt, x_img], dim=1)
        for i in range(OP.depth_single_blocks):
            x = rb(O.single_stream_block(W, f"single_blocks.{i}", OP.num_heads, x, vec, pe))
        out = O.last_layer(W, x[:, x_txt.shape[1]:], vec)
    SUMMARY["c2_fp32_oracle_stream_rounded_vs_fp32"] = rel_l2(out, ref0)
    print("rounding", SUMMARY["c2_fp32_oracle_stream_rounded_vs_fp32"], flush=True)
    save("_rounding_diagnostic.pt", dict(measured=SUMMARY["c2_fp32_oracle_stream_rounded_vs_fp32"]))
    for d in OUT_DIRS[:1]:
        os.remove(os.path.join(d, "_rounding_diagnostic.pt"))


if __name__ == "__main__":
    assert torch.cuda.is_available(), "needs a GPU box (the weights are drawn from the device's Philox stream)"
    dev = torch.device("cuda:0")
    which = sys.argv[1:] or ["c4", "c3", "c5"]
    for w in which:
        t0 = time.perf_counter()
        dict(c3=make_c3, c5=make_c5, c4=make_c4, rounding=make_rounding, c3live=lambda d: make_c3(d, True),
             c5live=lambda d: make_c5(d, True), c3loop=make_c3loop)[w](dev)
        torch.cuda.empty_cache()
        print(f"{w}: {time.perf_counter() - t0:.0f} s", flush=True)

"""Checkpoint round trips: tiny models are written to disk in the key / layout conventions of the files the
reference downloads (tests/disk_layouts.py), loaded through the product's loaders (FLUX_SCHNELL / AE /
FLUX_TEXT_DIR / SD_WEIGHTS_DIR + sanitize / map_*_weights), and must reproduce the oracle fed the SAME tensors.

CPU tests: the pure key-mapping functions invert the on-disk conventions (names and shapes).
GPU tests (`-m gpu`): file -> loader -> HIP model -> forward parity with the oracle.
Reference: flux/model.py:85-97, flux/autoencoder.py:336-345, flux/utils.py:98-191, flux/t5.py:232-241,
flux/clip.py:96-125, stable_diffusion/stable_diffusion/model_io.py:49-164.
"""
import json

import pytest
import torch
from safetensors.torch import save_file

from conftest import rel_l2
from disk_layouts import ae_to_disk, flux_to_disk, unet_to_disk, vae_to_disk
from oracle import flux_oracle as O
from oracle import sd_oracle as S
from oracle import text_oracle as T

BF = torch.bfloat16

FLUX_KW = dict(in_channels=64, vec_in_dim=64, context_in_dim=128, hidden_size=256, mlp_ratio=4.0, num_heads=2, depth=2,
               depth_single_blocks=2, axes_dim=[16, 56, 56], theta=10_000, qkv_bias=True, guidance_embed=True)
AE_KW = dict(resolution=64, in_channels=3, ch=128, out_ch=3, ch_mult=[1, 2], num_res_blocks=1, z_channels=16,
             scale_factor=0.3611, shift_factor=0.1159)
UNET_KW = dict(block_out_channels=(64, 128), layers_per_block=(1, 1), transformer_layers_per_block=(1, 2),
               num_attention_heads=(1, 2), cross_attention_dim=(128, 128),
               down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"), up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"),
               addition_embed_type="text_time", addition_time_embed_dim=32, projection_class_embeddings_input_dim=48 + 6 * 32)
VAE_KW = dict(block_out_channels=(128, 256), layers_per_block=1, scaling_factor=0.13025)


def _flux_weights():
    return O.init_weights(O.flux_weight_shapes(O.FluxParams(**FLUX_KW)), seed=0, norm_jitter=0.2)


# ------------------------------------------------------------------------------------------ CPU: key mappings
def test_flux_sanitize_inverts_disk_layout():
    from flux_generator_amd.flux.model import Flux
    W = _flux_weights()
    disk = flux_to_disk(W, prefix="model.diffusion_model.")
    assert "model.diffusion_model.double_blocks.0.img_mlp.0.weight" in disk
    assert "model.diffusion_model.double_blocks.1.txt_attn.norm.key_norm.scale" in disk
    assert "model.diffusion_model.final_layer.adaLN_modulation.1.weight" in disk
    back = Flux.sanitize(None, disk)
    assert set(back) == set(W) and all(back[k].shape == W[k].shape for k in W)
    assert set(Flux.sanitize(None, flux_to_disk(W))) == set(W)         # also without the prefix


def test_ae_sanitize_inverts_disk_layout():
    from flux_generator_amd.flux.autoencoder import AutoEncoder, AutoEncoderParams, decoder_weight_shapes
    W = O.init_weights(O.decoder_weight_shapes(O.AutoEncoderParams(**AE_KW)), seed=1)
    disk = ae_to_disk(W)
    assert disk["decoder.mid.attn_1.q.weight"].dim() == 4 and disk["decoder.conv_in.weight"].shape[1] == 16
    back = AutoEncoder.sanitize(None, disk)
    want = decoder_weight_shapes(AutoEncoderParams(**AE_KW))
    for k, shp in want.items():
        assert tuple(back[k].shape) == shp and torch.equal(back[k], W[k])


def test_sd_mappings_invert_disk_layout():
    from flux_generator_amd.stable_diffusion import model_io
    from flux_generator_amd.stable_diffusion.config import AutoencoderConfig, UNetConfig
    from flux_generator_amd.stable_diffusion.unet import unet_weight_shapes
    from flux_generator_amd.stable_diffusion.vae import vae_decoder_weight_shapes
    W = O.init_weights(S.unet_weight_shapes(S.UNetConfig(**UNET_KW)), seed=2)
    for conv_proj in (False, True):
        disk = unet_to_disk(W, conv_proj)
        assert any("ff.net.0.proj.weight" in k for k in disk) and any("to_out.0.bias" in k for k in disk)
        back = {}
        for k, v in disk.items():
            back.update(model_io.map_unet_weights(k, v))
        want = unet_weight_shapes(UNetConfig(**UNET_KW))
        assert set(back) == set(want)
        for k, shp in want.items():
            assert tuple(back[k].shape) == shp and torch.equal(back[k], W[k]), k
    Wv = O.init_weights(S.vae_decoder_weight_shapes(S.AutoencoderConfig(**VAE_KW)), seed=3)
    back = {}
    for k, v in vae_to_disk(Wv).items():
        back.update(model_io.map_vae_weights(k, v))
    want = vae_decoder_weight_shapes(AutoencoderConfig(**VAE_KW))
    for k, shp in want.items():
        assert tuple(back[k].shape) == shp and torch.equal(back[k], Wv[k]), k
    assert "quant_proj.weight" in back and "encoder.conv_in.weight" in back      # present on disk, skipped by the loader


# ------------------------------------------------------------------------------------------ GPU: file -> model -> parity
@pytest.mark.gpu
def test_flux_and_ae_checkpoints_load_and_match_oracle(dev, tmp_path, monkeypatch):
    from flux_generator_amd.flux import utils
    from flux_generator_amd.flux.autoencoder import AutoEncoderParams
    from flux_generator_amd.flux.model import FluxParams
    W = {k: v.to(BF) for k, v in _flux_weights().items()}                   # flux1-*.safetensors are bf16
    Wa = O.init_weights(O.decoder_weight_shapes(O.AutoEncoderParams(**AE_KW)), seed=1, norm_jitter=0.2)    # ae: fp32
    fpath, apath = str(tmp_path / "flux1-dev.safetensors"), str(tmp_path / "ae.safetensors")
    save_file(flux_to_disk(W, prefix="model.diffusion_model."), fpath)
    save_file(ae_to_disk(Wa), apath)
    spec = utils.ModelSpec(params=FluxParams(**FLUX_KW), ae_params=AutoEncoderParams(**AE_KW), ckpt_path=fpath, ae_path=apath,
                           repo_id=None, repo_flow=None, repo_ae=None)
    monkeypatch.setitem(utils.configs, "flux-dev", spec)
    model = utils.load_flow_model("flux-dev", device=dev)
    ae = utils.load_ae("flux-dev", device=dev)
    OP = O.FluxParams(**FLUX_KW)
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1, 16, 16, 16, generator=g).to(BF)
    img, ids = O.prepare_latent_images(z)
    txt = (torch.randn(1, 32, 128, generator=g) * 0.5).to(BF)
    tids = torch.zeros(1, 32, 3, dtype=torch.int32)
    vec = torch.randn(1, 64, generator=g).to(BF)
    t, gd = torch.full((1,), 0.75, dtype=BF), torch.full((1,), 3.5, dtype=BF)
    Wf = {k: v.float() for k, v in W.items()}
    ref = O.flux_forward(OP, Wf, img.float(), ids, txt.float(), tids, t, vec.float(), gd)
    got = model(img.to(dev), ids.to(dev), txt.to(dev), tids.to(dev), t.to(dev), vec.to(dev), gd.to(dev))
    assert rel_l2(got, ref) < 1e-2
    im = ae.decode_packed(img.to(dev), (16, 16))
    refim = O.pipeline_decode(O.AutoEncoderParams(**AE_KW), Wa, img.float(), (16, 16))
    assert float((im.cpu() - refim).abs().max()) <= 1.0 / 255
    # strict loading: a missing tensor is an error, not silently uninitialised memory
    bad = flux_to_disk(W)
    bad.pop("img_in.bias")
    save_file(bad, fpath)
    with pytest.raises(ValueError):
        utils.load_flow_model("flux-dev", device=dev)


@pytest.mark.gpu
def test_sd_checkpoints_load_and_match_oracle(dev, tmp_path, monkeypatch):
    from flux_generator_amd.stable_diffusion import model_io
    from flux_generator_amd.stable_diffusion.config import AutoencoderConfig, UNetConfig
    key = "stabilityai/sdxl-turbo"
    ocfg, ovae = S.UNetConfig(**UNET_KW), S.AutoencoderConfig(**VAE_KW)
    W = {k: v.to(BF).float() for k, v in O.init_weights(S.unet_weight_shapes(ocfg), seed=2, norm_jitter=0.2).items()}
    Wv = O.init_weights(S.vae_decoder_weight_shapes(ovae), seed=3, norm_jitter=0.2)
    root = tmp_path / key
    (root / "unet").mkdir(parents=True)
    (root / "vae").mkdir()
    save_file({k: v.to(torch.float16) if False else v for k, v in unet_to_disk(W).items()},
              str(root / "unet" / "diffusion_pytorch_model.safetensors"))
    save_file(vae_to_disk(Wv), str(root / "vae" / "diffusion_pytorch_model.safetensors"))
    monkeypatch.setenv("SD_WEIGHTS_DIR", str(tmp_path))
    monkeypatch.setitem(model_io._MODELS, key, {**model_io._MODELS[key], "unet_config": UNetConfig(**UNET_KW),
                                                "vae_config": AutoencoderConfig(**VAE_KW)})
    unet = model_io.load_unet(key, device=dev)
    vae = model_io.load_autoencoder(key, device=dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 16, 4, generator=g).to(BF)
    enc = torch.randn(2, 7, 128, generator=g).to(BF)
    t = torch.tensor([999.0, 999.0])
    tt = (torch.randn(2, 48, generator=g).to(BF), torch.tensor([[512, 512, 0, 0, 512, 512.0]] * 2))
    ref = S.unet_forward(ocfg, W, x.float(), t, enc.float(), (tt[0].float(), tt[1]))
    got = unet(x.to(dev), t.to(dev), enc.to(dev), text_time=(tt[0].to(dev), tt[1].to(dev)))
    assert rel_l2(got, ref) < 1.5e-2
    z = torch.randn(1, 8, 8, 4, generator=g).to(BF)
    im = vae.decode_image(z.to(dev))
    assert float((im.cpu() - S.sd_decode(ovae, Wv, z.float())).abs().max()) <= 1.0 / 255
    # the same files into the float32 model (float16=False, the reference's default: model_io.py:171-174 keeps the checkpoint's
    # float32): float32 master parameters, float32 arithmetic, float32 latents into the VAE
    unet32 = model_io.load_unet(key, device=dev, dtype=torch.float32)
    assert all(v.dtype == torch.float32 for v in unet32.parameters().values())
    got32 = unet32(x.float().to(dev), t.to(dev), enc.float().to(dev), text_time=(tt[0].float().to(dev), tt[1].to(dev)))
    assert got32.dtype == torch.float32 and rel_l2(got32, ref) < 1e-4
    im32 = vae.decode_image(z.float().to(dev))
    assert float((im32.cpu() - S.sd_decode(ovae, Wv, z.float())).abs().max()) <= 1e-4


@pytest.mark.gpu
def test_text_checkpoints_and_tokenizers_load_from_hub_layout(dev, tmp_path, monkeypatch):
    """FLUX_TEXT_DIR in the hub layout: text_encoder/{config.json,model.safetensors} (CLIP), text_encoder_2/
    {config.json, model-0000x-of-00002.safetensors, model.safetensors.index.json} (sharded T5), tokenizer/{vocab.json,
    merges.txt}, tokenizer_2/spiece.model — written from transformers' own state dicts."""
    tr = pytest.importorskip("transformers")
    spm = pytest.importorskip("sentencepiece")
    from flux_generator_amd.flux import utils
    d = tmp_path / "hub"
    for sub in ("text_encoder", "text_encoder_2", "tokenizer", "tokenizer_2"):
        (d / sub).mkdir(parents=True)
    # ---- CLIP
    ccfg = tr.CLIPTextConfig(vocab_size=300, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                             max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=299, bos_token_id=298,
                             pad_token_id=299)
    torch.manual_seed(1)
    hf_clip = tr.CLIPTextModel(ccfg).eval()
    save_file({k: v.contiguous() for k, v in hf_clip.state_dict().items()}, str(d / "text_encoder" / "model.safetensors"))
    json.dump(dict(ccfg.to_dict(), architectures=["CLIPTextModel"]), open(d / "text_encoder" / "config.json", "w"))
    # ---- T5, two shards + index
    tcfg = tr.T5Config(vocab_size=200, d_model=256, d_kv=64, d_ff=448, num_layers=2, num_heads=4,
                       relative_attention_num_buckets=32, feed_forward_proj="gated-gelu", tie_word_embeddings=False)
    torch.manual_seed(2)
    hf_t5 = tr.T5EncoderModel(tcfg).eval()
    sd = {k: v.contiguous() for k, v in hf_t5.state_dict().items() if k != "encoder.embed_tokens.weight"}
    names = sorted(sd)
    shards = {"model-00001-of-00002.safetensors": names[: len(names) // 2], "model-00002-of-00002.safetensors": names[len(names) // 2:]}
    for fn, ks in shards.items():
        save_file({k: sd[k] for k in ks}, str(d / "text_encoder_2" / fn))
    json.dump({"metadata": {}, "weight_map": {k: fn for fn, ks in shards.items() for k in ks}},
              open(d / "text_encoder_2" / "model.safetensors.index.json", "w"))
    json.dump(tcfg.to_dict(), open(d / "text_encoder_2" / "config.json", "w"))
    # ---- tokenizers
    pieces = ["<|startoftext|>", "<|endoftext|>", "a</w>", "cat</w>", "c", "a", "t", "t</w>", "ca", "at</w>"]
    json.dump({p: i for i, p in enumerate(pieces)}, open(d / "tokenizer" / "vocab.json", "w"))
    open(d / "tokenizer" / "merges.txt", "w").write("#version: 0.2\nc a\nca t</w>\n")
    corpus = tmp_path / "c.txt"
    corpus.write_text("\n".join(["a photo of a cat", "a painting of a dog on the moon", "the cat sat on the mat"] * 20))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(d / "tokenizer_2" / "spiece"), vocab_size=24,
                                   model_type="unigram", hard_vocab_limit=False, pad_id=0, eos_id=1, unk_id=2, bos_id=-1,
                                   minloglevel=2)
    monkeypatch.setenv("FLUX_TEXT_DIR", str(d))

    clip = utils.load_clip("flux-schnell", device=dev)
    tokens = torch.randint(1, 298, (2, 13), generator=torch.Generator().manual_seed(6))
    tokens[:, 0] = 298
    tokens[0, 6:] = 299
    tokens[1, 12] = 299
    Wc = {k: v.float() for k, v in clip.sanitize({k: v for k, v in hf_clip.state_dict().items() if "position_ids" not in k}).items()}
    ref = T.clip_text_model(T.CLIPTextModelConfig(num_layers=2, model_dims=128, num_heads=2, max_length=77, vocab_size=300), Wc, tokens)
    got = clip(tokens)
    assert rel_l2(got.pooled_output, ref.pooled_output) < 2e-2 and rel_l2(got.last_hidden_state, ref.last_hidden_state) < 2e-2

    t5 = utils.load_t5("flux-schnell", device=dev)
    tok = torch.randint(0, 200, (2, 24), generator=torch.Generator().manual_seed(3))
    Wt = {k: v.float() for k, v in t5.sanitize(dict(hf_t5.state_dict())).items()}
    ocfg = T.T5Config(vocab_size=200, num_layers=2, num_heads=4, relative_attention_num_buckets=32, d_kv=64, d_model=256, d_ff=448)
    assert rel_l2(t5(tok), T.t5_encoder(ocfg, Wt, tok)) < 2e-2

    ctok = utils.load_clip_tokenizer("flux-schnell")
    assert ctok.tokenize("a cat") == [0, 2, 3, 1]
    ttok = utils.load_t5_tokenizer("flux-schnell")
    ids = ttok.tokenize("a photo of a cat")
    assert len(ids) == 256 and 1 in ids and ids[-1] == 0


def test_random_init_is_an_explicit_opt_in(monkeypatch):
    """A missing checkpoint raises (the reference's hf_hub_download would fail without network, flux/utils.py:102-110) unless
    FLUX_ALLOW_RANDOM_INIT=1 asks for random weights (tests, bench.py): a mis-configured server must not answer with noise."""
    import warnings
    from flux_generator_amd.flux.utils import random_init_or_raise
    monkeypatch.delenv("FLUX_ALLOW_RANDOM_INIT", raising=False)
    with pytest.raises(FileNotFoundError, match="FLUX_ALLOW_RANDOM_INIT=1"):
        random_init_or_raise("flux-schnell flow model", "set FLUX_SCHNELL")
    monkeypatch.setenv("FLUX_ALLOW_RANDOM_INIT", "1")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        random_init_or_raise("flux-schnell flow model", "set FLUX_SCHNELL")
    assert len(w) == 1 and "random-init" in str(w[0].message)

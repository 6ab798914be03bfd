"""Pins for oracle/text_oracle.py (no GPU): the restated T5 encoder / CLIP text model against the
INDEPENDENT implementations in `transformers` (random tiny configs, weights copied through the
reference's own sanitize key mapping), plus tokenizer behaviour on synthetic vocabularies."""
import math

import pytest
import torch

from conftest import rel_l2
from oracle import text_oracle as T


def _sanitize_t5(sd):
    from flux_generator_amd.flux.t5 import _ENCODER, _SHARED
    out = {}
    for k, w in sd.items():
        for a, b in _SHARED:
            k = k.replace(a, b)
        if k.startswith("encoder."):
            for a, b in _ENCODER:
                k = k.replace(a, b)
        out[k] = w
    return out


def test_t5_vs_transformers():
    tr = pytest.importorskip("transformers")
    hf_cfg = tr.T5Config(vocab_size=100, d_model=64, d_kv=16, d_ff=96, num_layers=2, num_heads=4,
                         relative_attention_num_buckets=8, relative_attention_max_distance=16,
                         feed_forward_proj="gated-gelu", layer_norm_epsilon=1e-6, tie_word_embeddings=False)
    torch.manual_seed(0)
    hf = tr.T5EncoderModel(hf_cfg).eval()
    for blk in hf.encoder.block:                       # the reference uses the EXACT erf GELU (flux/t5.py:172-176)
        blk.layer[1].DenseReluDense.act = torch.nn.GELU()
    W = {k: v for k, v in _sanitize_t5(hf.state_dict()).items()}
    cfg = T.T5Config(vocab_size=100, num_layers=2, num_heads=4, relative_attention_num_buckets=8, d_kv=16, d_model=64,
                     d_ff=96, relative_attention_max_distance=16)
    assert set(T.t5_weight_shapes(cfg)) <= set(W)
    tokens = torch.randint(0, 100, (2, 24))
    with torch.no_grad():
        want = hf(input_ids=tokens).last_hidden_state
        got = T.t5_encoder(cfg, W, tokens)
    assert rel_l2(got, want) < 1e-5


def test_clip_vs_transformers():
    tr = pytest.importorskip("transformers")
    hf_cfg = tr.CLIPTextConfig(vocab_size=120, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                               num_attention_heads=2, max_position_embeddings=20, hidden_act="quick_gelu",
                               eos_token_id=119, bos_token_id=118, pad_token_id=119)
    torch.manual_seed(1)
    hf = tr.CLIPTextModel(hf_cfg).eval()
    from flux_generator_amd.flux.clip import CLIPTextModel as P   # only its sanitize (pure key mapping) is used
    W = P.sanitize(None, {k: v for k, v in hf.state_dict().items() if "position_ids" not in k})
    cfg = T.CLIPTextModelConfig(num_layers=2, model_dims=128, num_heads=2, max_length=20, vocab_size=120)
    assert set(T.clip_weight_shapes(cfg)) == set(W)
    tokens = torch.randint(0, 118, (2, 12))
    tokens[:, 0] = 118
    tokens[0, 7:] = 119
    tokens[1, 11] = 119
    with torch.no_grad():
        hfo = hf(input_ids=tokens, output_hidden_states=True)
        got = T.clip_text_model(cfg, W, tokens)
    assert rel_l2(got.last_hidden_state, hfo.last_hidden_state) < 1e-5
    assert rel_l2(got.pooled_output, hfo.pooler_output) < 1e-5          # EOS = argmax(token id)
    assert rel_l2(got.hidden_states[-2], hfo.hidden_states[-2]) < 1e-5


def test_sd_clip_with_projection_vs_transformers():
    """The stable_diffusion/ variant (stable_diffusion/stable_diffusion/clip.py): exact-erf "gelu" tower with the
    bias-free text_projection on the pooled EOS row, against transformers' CLIPTextModelWithProjection, through the
    product's map_clip_text_encoder_weights (pure key mapping)."""
    tr = pytest.importorskip("transformers")
    hf_cfg = tr.CLIPTextConfig(vocab_size=120, hidden_size=128, intermediate_size=512, num_hidden_layers=3,
                               num_attention_heads=2, max_position_embeddings=20, hidden_act="gelu", projection_dim=96,
                               eos_token_id=119, bos_token_id=118, pad_token_id=119)
    torch.manual_seed(2)
    hf = tr.CLIPTextModelWithProjection(hf_cfg).eval()
    from flux_generator_amd.stable_diffusion.clip import map_clip_text_encoder_weights
    W = {}
    for k, v in hf.state_dict().items():
        if "position_ids" not in k:
            W.update(map_clip_text_encoder_weights(k, v))
    cfg = T.CLIPTextModelConfig(num_layers=3, model_dims=128, num_heads=2, max_length=20, vocab_size=120, hidden_act="gelu",
                                projection_dim=96)
    assert set(T.clip_weight_shapes(cfg)) == set(W)
    tokens = torch.randint(0, 118, (2, 12))
    tokens[:, 0] = 118
    tokens[0, 7:] = 119
    tokens[1, 11] = 119
    with torch.no_grad():
        hfo = hf(input_ids=tokens, output_hidden_states=True)
        got = T.clip_text_model(cfg, W, tokens)
    assert rel_l2(got.last_hidden_state, hfo.last_hidden_state) < 1e-5
    assert got.pooled_output.shape == (2, 96) and rel_l2(got.pooled_output, hfo.text_embeds) < 1e-5
    assert rel_l2(got.hidden_states[-2], hfo.hidden_states[-2]) < 1e-5
    # the product's config parser: projection only for "...WithProjection" checkpoints (model_io.py:246-257)
    from flux_generator_amd.flux.clip import CLIPTextModelConfig as PC
    d = dict(num_hidden_layers=3, hidden_size=128, num_attention_heads=2, max_position_embeddings=20, vocab_size=120,
             hidden_act="gelu", projection_dim=96)
    assert PC.from_dict({**d, "architectures": ["CLIPTextModelWithProjection"]}).projection_dim == 96
    assert PC.from_dict({**d, "architectures": ["CLIPTextModel"]}).projection_dim is None


def test_relative_position_buckets_kat():
    b = T.relative_position_bucket(torch.arange(-40, 41), True, 32, 128)
    assert int(b[40]) == 0 and int(b[41]) == 17 and int(b[39]) == 1          # 0, +1 (offset 16), -1
    assert int(b[40 + 7]) == 16 + 7 and int(b[40 - 7]) == 7                   # exact range |d| < 8
    assert int(b.max()) <= 31 and int(b[0]) == 8 + int(math.log(40 / 8) / math.log(128 / 8) * 8)
    from flux_generator_amd.flux.t5 import relative_position_bucket as prod
    assert torch.equal(prod(torch.arange(-300, 300), 32, 128), T.relative_position_bucket(torch.arange(-300, 300), True, 32, 128))


def test_clip_tokenizer_bpe():
    from flux_generator_amd.flux.tokenizers import CLIPTokenizer
    merges = [("l", "o"), ("lo", "w</w>"), ("e", "r</w>"), ("n", "e"), ("ne", "w"), ("new", "er</w>"), ("t", "h"), ("th", "e</w>")]
    ranks = {m: i for i, m in enumerate(merges)}
    pieces = ["<|startoftext|>", "<|endoftext|>", "low</w>", "newer</w>", "the</w>", "l", "o", "w", "e", "r", "n", "t", "h",
              "w</w>", "r</w>", "e</w>", "er</w>", "lo", "ne", "new", "th", "s</w>", "'s</w>", "1</w>", "!</w>", "!!</w>"]
    vocab = {p: i for i, p in enumerate(pieces)}
    tok = CLIPTokenizer(ranks, vocab, max_length=8)
    assert tok.bpe("low") == ["low</w>"] and tok.bpe("newer") == ["newer</w>"] and tok.bpe("lower") == ["lo", "w", "er</w>"]
    ids = tok.tokenize("The  LOW newer")
    assert ids == [0, vocab["the</w>"], vocab["low</w>"], vocab["newer</w>"], 1]
    long_ids = tok.tokenize("low " * 20)
    assert len(long_ids) == 8 and long_ids[-1] == 1                          # truncated, EOS kept
    enc = tok.encode(["low", "the low newer"])
    assert enc.shape == (2, 5) and enc.dtype == torch.int32 and enc[0].tolist() == [0, vocab["low</w>"], 1, 1, 1]


def test_t5_tokenizer_sentencepiece(tmp_path):
    spm = pytest.importorskip("sentencepiece")
    corpus = tmp_path / "c.txt"
    corpus.write_text("\n".join(["a photo of a cat", "a painting of a dog on the moon", "the cat sat on the mat"] * 20))
    prefix = str(tmp_path / "m")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=prefix, vocab_size=24, model_type="unigram", hard_vocab_limit=False,
                                   pad_id=0, eos_id=1, unk_id=2, bos_id=-1, minloglevel=2)
    from flux_generator_amd.flux.tokenizers import T5Tokenizer
    tok = T5Tokenizer(prefix + ".model", max_length=16)
    ids = tok.tokenize("a photo of a cat")
    assert len(ids) == 16 and ids[-1] == 0 and 1 in ids and tok.pad_token == 0 and tok.eos_token == 1 and tok.bos_token == -1
    assert tok.tokenize("a cat", pad=False)[-1] == 1
    e = tok.encode(["a cat", "a painting of a dog"], pad=False)
    assert e.shape[0] == 2 and e.dtype == torch.int32

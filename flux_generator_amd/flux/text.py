"""Stand-ins for the text side of the pipeline (tokenizers, T5-XXL, CLIP-L).

The text encoders sit immediately BEFORE the denoise hot path and are the first "next" row of the
scope table; their checkpoints and vocabularies are not available offline.  These
classes only produce conditioning tensors with the reference's shapes and dtypes
(txt [B,S,4096] bf16, pooled vec [B,768] bf16; flux/flux.py:73-85) deterministically from the
prompt, so the pipeline surface can be driven end to end.  They are NOT the reference's encoders.
"""
from __future__ import annotations

import hashlib
from typing import List

import torch


class HashTokenizer:
    """Deterministic whitespace/hash tokenizer with the reference's padding contract
    (flux/tokenizers.py:122-185: T5 pads to max_length; CLIP pads to 77)."""

    def __init__(self, max_length: int, vocab: int, pad_with_eos: bool = False):
        self.max_length = max_length
        self.vocab = vocab
        self.pad_with_eos = pad_with_eos      # CLIP pads with EOS = the highest id (argmax pooling)

    def tokenize(self, text: str) -> List[int]:
        toks = [int.from_bytes(hashlib.sha1(w.encode()).digest()[:4], "little") % (self.vocab - 3) + 2
                for w in text.lower().split()]
        eos = self.vocab - 1 if self.pad_with_eos else 1
        return toks[: self.max_length - 1] + [eos]

    def encode(self, text, pad: bool = True) -> torch.Tensor:
        texts = [text] if isinstance(text, str) else list(text)
        rows = [self.tokenize(t) for t in texts]
        if self.pad_with_eos:                        # CLIPTokenizer.encode pads to the longest row with EOS
            n, fill = max(len(r) for r in rows), self.vocab - 1
        else:
            n, fill = (self.max_length if pad else max(len(r) for r in rows)), 0
        n = (n + 3) // 4 * 4 if not self.pad_with_eos else n
        return torch.tensor([r + [fill] * (n - len(r)) for r in rows], dtype=torch.int32)


class _SyntheticEncoder:
    def __init__(self, dim: int, device, table: int = 4096, seed: int = 7, scale: float = 0.1):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.table = (torch.randn(table, dim, generator=g) * scale).to(torch.bfloat16).to(device)
        self.device = torch.device(device)

    def _embed(self, tokens: torch.Tensor) -> torch.Tensor:
        return self.table[(tokens.to(self.device).long() % self.table.shape[0])]


class SyntheticT5(_SyntheticEncoder):
    def __call__(self, tokens: torch.Tensor) -> torch.Tensor:   # [B,S] -> [B,S,dim]
        return self._embed(tokens)


class _Pooled:
    def __init__(self, pooled):
        self.pooled_output = pooled


class SyntheticCLIP(_SyntheticEncoder):
    def __init__(self, dim: int, device):
        super().__init__(dim, device, seed=11, scale=1.0)

    def __call__(self, tokens: torch.Tensor) -> _Pooled:          # [B,77] -> .pooled_output [B,dim]
        return _Pooled(self._embed(tokens).float().mean(dim=1).to(torch.bfloat16))

"""Offline stand-in for the tokenizers' vocabulary files.

The real tokenizers (flux/tokenizers.py: CLIP BPE, T5 SentencePiece) need vocab.json / merges.txt / spiece.model,
which are not available without the hub.  When those files are not configured the loaders substitute this
deterministic hash tokenizer AND SAY SO (a warning): it keeps the reference's padding contract so the pipelines can be
driven end to end with random-init weights, but its token ids are not the models' ids."""
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

"""Tokenizers of the Flux pipeline (same classes / methods as the reference's flux/tokenizers.py):
CLIPTokenizer = lower-casing regex pre-tokeniser + byte-pair merges over a (vocab, merge-rank) pair,
T5Tokenizer = SentencePiece wrapper with pad-to-max_length.  ``encode`` returns int32 CPU tensors."""
from __future__ import annotations

from typing import Dict, List, Tuple

import regex
import torch


class CLIPTokenizer:
    _PATTERN = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""

    def __init__(self, bpe_ranks: Dict[Tuple[str, str], int], vocab: Dict[str, int], max_length: int = 77):
        self.max_length = max_length
        self.bpe_ranks = bpe_ranks
        self.vocab = vocab
        self.pat = regex.compile(self._PATTERN, regex.IGNORECASE)
        self._cache = {self.bos: [self.bos], self.eos: [self.eos]}

    bos = "<|startoftext|>"
    eos = "<|endoftext|>"

    @property
    def bos_token(self) -> int:
        return self.vocab[self.bos]

    @property
    def eos_token(self) -> int:
        return self.vocab[self.eos]

    def bpe(self, word: str) -> List[str]:
        """Greedy lowest-rank-first pair merging; the last symbol carries the end-of-word marker."""
        hit = self._cache.get(word)
        if hit is not None:
            return hit
        parts = list(word[:-1]) + [word[-1] + "</w>"]
        while len(parts) > 1:
            ranked = [(self.bpe_ranks.get(pair, None), i) for i, pair in enumerate(zip(parts, parts[1:]))]
            ranked = [(r, i) for r, i in ranked if r is not None]
            if not ranked:
                break
            best = min(ranked)[0]
            target = next(pair for pair in zip(parts, parts[1:]) if self.bpe_ranks.get(pair) == best)
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and (parts[i], parts[i + 1]) == target:
                    merged.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        self._cache[word] = parts
        return parts

    def tokenize(self, text, prepend_bos: bool = True, append_eos: bool = True):
        if isinstance(text, list):
            return [self.tokenize(t, prepend_bos, append_eos) for t in text]
        clean = regex.sub(r"\s+", " ", text.lower())
        ids = [self.vocab[piece] for word in regex.findall(self.pat, clean) for piece in self.bpe(word)]
        if prepend_bos:
            ids = [self.bos_token] + ids
        if append_eos:
            ids.append(self.eos_token)
        if len(ids) > self.max_length:
            ids = ids[: self.max_length]
            if append_eos:
                ids[-1] = self.eos_token
        return ids

    def encode(self, text) -> torch.Tensor:
        if not isinstance(text, list):
            return self.encode([text])
        rows = self.tokenize(text)
        n = max(len(r) for r in rows)
        return torch.tensor([r + [self.eos_token] * (n - len(r)) for r in rows], dtype=torch.int32)


class T5Tokenizer:
    def __init__(self, model_file: str, max_length: int = 512):
        from sentencepiece import SentencePieceProcessor
        self._tokenizer = SentencePieceProcessor(model_file)
        self.max_length = max_length

    def _piece(self, i):
        try:
            return self._tokenizer.id_to_piece(i)
        except IndexError:
            return None

    @property
    def pad_token(self) -> int:
        return self._tokenizer.pad_id()

    @property
    def bos_token(self) -> int:
        return self._tokenizer.bos_id()

    @property
    def eos_token(self) -> int:
        return self._tokenizer.eos_id()

    pad = property(lambda self: self._piece(self.pad_token))
    bos = property(lambda self: self._piece(self.bos_token))
    eos = property(lambda self: self._piece(self.eos_token))

    def tokenize(self, text, prepend_bos: bool = True, append_eos: bool = True, pad: bool = True):
        if isinstance(text, list):
            return [self.tokenize(t, prepend_bos, append_eos, pad) for t in text]
        ids = list(self._tokenizer.encode(text))
        if prepend_bos and self.bos_token >= 0:
            ids = [self.bos_token] + ids
        if append_eos and self.eos_token >= 0:
            ids.append(self.eos_token)
        if pad and len(ids) < self.max_length and self.pad_token >= 0:
            ids += [self.pad_token] * (self.max_length - len(ids))
        return ids

    def encode(self, text, pad: bool = True) -> torch.Tensor:
        if not isinstance(text, list):
            return self.encode([text], pad=pad)
        fill = self.pad_token if self.pad_token >= 0 else 0
        rows = self.tokenize(text, pad=pad)
        n = max(len(r) for r in rows)
        return torch.tensor([r + [fill] * (n - len(r)) for r in rows], dtype=torch.int32)

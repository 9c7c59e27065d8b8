"""BaselineMIPSModule on MI355X: brute-force maximum-inner-product search whose
[B, C] score matrix never touches HBM.

Mirrors ref:src/baseline_mips_module.py:10-72 (constructor, ``corpus_size`` /
``embedding_dim`` / ``corpus`` attributes, ``forward(query_embedding, num_items)
-> (indices, scores, embeddings)``).  Differences, all deliberate:
  * ties are ordered (score desc, index asc) -- ``torch.topk`` leaves them arbitrary;
  * ``corpus`` follows ``.to()`` / ``.cuda()`` (upstream leaves it on the CPU);
  * optional bf16 corpus storage (BASELINE config 5): ``use_bf16_storage()``.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn

from . import _native as N
from . import ops


class BaselineMIPSModule(nn.Module):
    def __init__(self, corpus_size: int, embedding_dim: int) -> None:
        super().__init__()
        self.corpus_size = corpus_size
        self.embedding_dim = embedding_dim
        # random corpus, plain tensor (not a Parameter / buffer: empty state_dict), ref :29-30
        self.corpus = torch.randn(corpus_size, embedding_dim)  # [C, DI]

    # `corpus` is a property only so that EVERY assignment -- set_corpus, use_bf16_storage, .to(), a caller's own
    # `module.corpus = ...` -- drops the split-fp16 copy derived from the previous one (a corpus freed and
    # re-allocated at the same address has the same (pointer, version, shape) key).  Writes INTO the tensor through raw
    # pointers cannot be seen here: call use_split_fp16_scoring() again after those, as its docstring says.
    @property
    def corpus(self) -> torch.Tensor:
        return self._corpus

    @corpus.setter
    def corpus(self, value: torch.Tensor) -> None:
        object.__setattr__(self, "_corpus", value)
        object.__setattr__(self, "_split16", None)
        object.__setattr__(self, "_split16_key", None)

    def _apply(self, fn, *a, **kw):
        super()._apply(fn, *a, **kw)
        moved = fn(self.corpus)
        # .to(dtype) requests must not silently change the storage format
        self.corpus = moved if moved.dtype == self.corpus.dtype else moved.to(self.corpus.dtype)
        return self

    def use_bf16_storage(self) -> "BaselineMIPSModule":
        """Store the corpus as bf16 (round-to-nearest-even); queries are rounded to bf16
        per call, products are exact, accumulation is fp32 on the bf16 MFMA path."""
        self.corpus = self.corpus.to(torch.bfloat16)
        return self

    def use_split_fp16_scoring(self, on: bool = True) -> "BaselineMIPSModule":
        """EXPLORATORY: keep the fp32 corpus and ALSO its two-term fp16 split (same size again); searches then score
        on the fp16 matrix pipe at fp32-grade accuracy (three fp16 MFMA products per fp32 product) -- same contract
        as the fp32 path, about 2.5x faster.  D = 128, fp32 corpus.  Call again after the corpus changes."""
        if on and (self.corpus.dtype != torch.float32 or self.corpus.shape[1] != 128):
            raise ValueError("split-fp16 scoring takes a float32 corpus with embedding_dim 128 "
                             f"(got {self.corpus.dtype}, D = {self.corpus.shape[1]})")
        self._split16_on = bool(on)
        self._split16 = self._split16_key = None  # built by the next search (and again whenever the corpus has changed)
        return self

    def _corpus_key(self):
        c = self.corpus
        return (c.data_ptr(), c._version, tuple(c.shape), c.dtype, c.device)

    def set_corpus(self, embeddings: torch.Tensor, bf16: bool = False) -> "BaselineMIPSModule":
        """Replace the random corpus (ref :29-30) by real item embeddings [C, DI]."""
        if embeddings.dim() != 2 or embeddings.shape[1] != self.embedding_dim:
            raise ValueError(f"corpus must be [C, {self.embedding_dim}]")
        self.corpus = embeddings.detach().to(torch.bfloat16 if bf16 else torch.float32).contiguous()
        self.corpus_size = self.corpus.shape[0]
        return self

    def search(self, query_embedding: torch.Tensor, num_items: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(indices int64 [B, K], scores fp32 [B, K]) without gathering the rows."""
        split = None
        if getattr(self, "_split16_on", False) and self.corpus.dtype == torch.float32 and self.corpus.shape[1] == 128 \
                and self.corpus.is_cuda:
            # the split belongs to ONE state of the corpus: storage, in-place version, shape (a corpus that was assigned,
            # refilled in place, converted or moved since gets a new one -- 2 ms at 10 M rows)
            key = self._corpus_key()
            if getattr(self, "_split16_key", None) != key:
                self._split16, self._split16_key = ops.mips_split_rows(self.corpus), key
            split = self._split16
        return ops.mips_topk(query_embedding, self.corpus, num_items, split16=split)

    def forward(
        self,
        query_embedding: torch.Tensor,  # [B, DI]
        num_items: int,  # (NI)
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """indices [B, NI] int64, mips_scores [B, NI], embeddings [B, NI, DI] (ref :32-72)."""
        indices, mips_scores = self.search(query_embedding, num_items)
        embeddings = ops.gather_corpus_rows(self.corpus, indices)
        return indices, mips_scores, embeddings

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

    def set_corpus(self, embeddings: torch.Tensor, bf16: bool = False) -> "BaselineMIPSModule":
        """Replace the random corpus (ref :29-30) by real item embeddings [C, DI]."""
        if embeddings.dim() != 2 or embeddings.shape[1] != self.embedding_dim:
            raise ValueError(f"corpus must be [C, {self.embedding_dim}]")
        self.corpus = embeddings.detach().to(torch.bfloat16 if bf16 else torch.float32).contiguous()
        self.corpus_size = self.corpus.shape[0]
        return self

    def search(self, query_embedding: torch.Tensor, num_items: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(indices int64 [B, K], scores fp32 [B, K]) without gathering the rows."""
        return ops.mips_topk(query_embedding, self.corpus, num_items)

    def forward(
        self,
        query_embedding: torch.Tensor,  # [B, DI]
        num_items: int,  # (NI)
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """indices [B, NI] int64, mips_scores [B, NI], embeddings [B, NI, DI] (ref :32-72)."""
        indices, mips_scores = self.search(query_embedding, num_items)
        embeddings = ops.gather_corpus_rows(self.corpus, indices)
        return indices, mips_scores, embeddings

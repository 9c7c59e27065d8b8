"""BaselineMIPSModule on MI355X: brute-force maximum-inner-product search whose
[B, C] score matrix never touches HBM.

Mirrors ref:src/baseline_mips_module.py:10-72 (constructor, ``corpus_size`` /
``embedding_dim`` / ``corpus`` attributes, ``forward(query_embedding, num_items)
-> (indices, scores, embeddings)``).  Differences, all deliberate:
  * ties are ordered (score desc, index asc) -- ``torch.topk`` leaves them arbitrary;
  * ``corpus`` follows ``.to()`` / ``.cuda()`` (upstream leaves it on the CPU);
  * optional bf16 corpus storage (BASELINE config 5): ``use_bf16_storage()``;
  * ROW-SHARDED over the GPUs of a node (BASELINE config 5, SURVEY.md 8e "MIPS corpus: row-sharded C/8 per GPU"): built
    under ``parallel.row_sharded()`` -- or cut down by ``shard_corpus_()`` / ``parallel.shard_model_(model)`` -- the
    module holds only this rank's contiguous row block of the corpus (``corpus_size`` stays the WHOLE corpus' size) and
    the SAME ``forward`` / ``search`` answer for the whole corpus: every rank brings its own B queries and receives its
    own queries' global top-K (int64 GLOBAL row numbers, scores, and -- ``forward`` only -- the [B, K, D] rows fetched
    from the ranks that own them).  ``model.forward()`` needs no other call.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn

from . import _native as N
from . import ops
from . import parallel


class BaselineMIPSModule(nn.Module):
    def __init__(self, corpus_size: int, embedding_dim: int) -> None:
        super().__init__()
        self.corpus_size = corpus_size
        self.embedding_dim = embedding_dim
        self._shard = None  # parallel.RowShard when `corpus` is this rank's row block of a corpus_size-row corpus
        if parallel._BUILDING[0]:  # under parallel.row_sharded(): the corpus is born as this rank's block, never whole
            world, rank = parallel._group()
            sh = self._shard = parallel.RowShard(corpus_size, embedding_dim, world, rank)
            probe = torch.empty(0)  # (device of the ambient `with torch.device(...)`, as torch.randn below would use)
            self.corpus = torch.randn(sh.n_local, embedding_dim,
                                      generator=parallel.block_generator(probe.device, sh.lo, corpus_size), device=probe.device)
            return
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

    def set_corpus(self, embeddings: torch.Tensor, bf16: bool = False, block: bool = False) -> "BaselineMIPSModule":
        """Replace the random corpus (ref :29-30) by real item embeddings [C, DI].  Row-sharded module (or
        `block=True`, which makes it one): `embeddings` is THIS RANK'S BLOCK -- rows [lo, hi) = parallel.block_range(C, rank, world)[1:] of the C-row corpus,
        C = the sum of the ranks' row counts -- and every rank must call (one small all-gather of the row counts)."""
        if embeddings.dim() != 2 or embeddings.shape[1] != self.embedding_dim:
            raise ValueError(f"corpus must be [C, {self.embedding_dim}]")
        if self._shard is not None or block:
            world, rank = parallel._group()
            mine = torch.zeros(world, dtype=torch.int64)
            mine[rank] = embeddings.shape[0]
            sizes = parallel.C.all_reduce_(mine.to(embeddings.device)).tolist()
            total = int(sum(sizes))
            want = [parallel.block_range(total, r, world) for r in range(world)]
            if any(int(n) != hi - lo for n, (_, lo, hi) in zip(sizes, want)):  # (every rank sees every count: all raise)
                raise ValueError(f"set_corpus on a row-sharded module: the ranks' blocks must be the contiguous row blocks "
                                 f"parallel.block_range({total}, rank, {world}) = {[hi - lo for _, lo, hi in want]} rows, got {sizes}")
            self._shard = parallel.RowShard(total, self.embedding_dim, world, rank)
            self.corpus = embeddings.detach().to(torch.bfloat16 if bf16 else torch.float32).contiguous()
            self.corpus_size = total
            return self
        self.corpus = embeddings.detach().to(torch.bfloat16 if bf16 else torch.float32).contiguous()
        self.corpus_size = self.corpus.shape[0]
        return self

    # ------------------------------------------------------------------ row-sharded corpus (parallel.py)
    def is_sharded(self) -> bool:
        return self._shard is not None

    def shard_corpus_(self) -> "BaselineMIPSModule":
        """Keep only this rank's contiguous row block of a corpus every rank holds whole (same rows on every rank, e.g. a
        seeded random corpus or one loaded from a file); `corpus_size` stays the whole corpus' size.  No-op when already
        sharded.  Needs torch.distributed (one process per GPU)."""
        if self._shard is None:
            world, rank = parallel._group()
            sh = parallel.RowShard(self.corpus.shape[0], self.embedding_dim, world, rank)
            self.corpus = self.corpus[sh.lo:sh.hi].clone()
            self.corpus_size = sh.n_rows
            self._shard = sh
        return self

    def search(self, query_embedding: torch.Tensor, num_items: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(indices int64 [B, K], scores fp32 [B, K]) without gathering the rows.  Row-sharded module: collective -- every
        rank calls with its own B queries (same B) and gets ITS queries' top-K over the whole corpus, global row numbers."""
        if self._shard is not None and not (0 < num_items <= self.corpus_size):
            raise RuntimeError("selected index k out of range")  # torch.topk's message, decided on the WHOLE corpus' size
        split = None
        if getattr(self, "_split16_on", False) and self.corpus.dtype == torch.float32 and self.corpus.shape[1] == 128 \
                and self.corpus.is_cuda:
            # the split belongs to ONE state of the corpus: storage, in-place version, shape (a corpus that was assigned,
            # refilled in place, converted or moved since gets a new one -- 2 ms at 10 M rows)
            key = self._corpus_key()
            if getattr(self, "_split16_key", None) != key:
                self._split16, self._split16_key = ops.mips_split_rows(self.corpus), key
            split = self._split16
        if self._shard is not None:
            topk = (lambda q, c, k: ops.mips_topk(q, c, k, split16=split)) if split is not None else None
            return parallel.sharded_topk(self.corpus, self._shard.lo, query_embedding.detach(), num_items, topk=topk)
        return ops.mips_topk(query_embedding, self.corpus, num_items, split16=split)

    def forward(
        self,
        query_embedding: torch.Tensor,  # [B, DI]
        num_items: int,  # (NI)
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """indices [B, NI] int64, mips_scores [B, NI], embeddings [B, NI, DI] (ref :32-72)."""
        indices, mips_scores = self.search(query_embedding, num_items)
        if self._shard is not None:  # rows live on their owners: ids out, rows back (the lookups' padded all-to-all)
            return indices, mips_scores, parallel.fetch_rows(self.corpus, self._shard, indices)
        embeddings = ops.gather_corpus_rows(self.corpus, indices)
        return indices, mips_scores, embeddings

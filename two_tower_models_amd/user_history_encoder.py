"""UserHistoryEncoder on MI355X (mirror of ref:src/user_history_encoder.py:11-124).

[B, H, DI] history embeddings -> [B, 2, DI]: (row 0 after L stacked, unmasked
multi-head self-attention layers with the flipped sinusoid table added,
mean over H of the raw embeddings).  Parameter names and initialisation match
``nn.MultiheadAttention`` inside an ``nn.ModuleList`` exactly
(``multihead_attn_layers.{i}.in_proj_weight`` ...), including the order of the
RNG draws, so the reference's seeded known-answer tests reproduce.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops


class _AttentionLayerParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's names, shapes and init."""

    def __init__(self, embed_dim: int, num_heads: int) -> None:
        super().__init__()
        if embed_dim % num_heads != 0:
            raise AssertionError("embed_dim must be divisible by num_heads")  # as nn.MultiheadAttention
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)  # default Linear init draws first ...
        nn.init.xavier_uniform_(self.in_proj_weight)  # ... then the in-projection
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def tensors(self):
        return self.in_proj_weight, self.in_proj_bias, self.out_proj.weight, self.out_proj.bias


class UserHistoryEncoder(nn.Module):
    def __init__(
        self,
        item_id_embedding_dim: int,
        history_len: int,
        num_attention_heads: int,
        num_attention_layers: int,
        use_positional_encoding: bool,
    ) -> None:
        super().__init__()
        self.item_id_embedding_dim = item_id_embedding_dim
        self.history_len = history_len
        self.num_attention_heads = num_attention_heads
        self.num_attention_layers = num_attention_layers
        self.use_positional_encoding = use_positional_encoding
        if self.use_positional_encoding:
            # newest item first => row 0 carries the encoding of position H-1 (ref :35-54)
            self.positional_embeddings = self.positional_encoding(
                seq_len=history_len, d_model=item_id_embedding_dim
            ).flip([0]).to(torch.empty(0).device)  # honours an enclosing `with torch.device(...)`
        self.multihead_attn_layers = nn.ModuleList(
            [_AttentionLayerParams(item_id_embedding_dim, num_attention_heads) for _ in range(num_attention_layers)]
        )

    def _apply(self, fn, *a, **kw):
        super()._apply(fn, *a, **kw)
        if self.use_positional_encoding:  # plain attribute upstream; here it follows .to()
            self.positional_embeddings = fn(self.positional_embeddings)
        return self

    def positional_encoding(self, seq_len: int, d_model: int) -> torch.Tensor:
        """The reference's table (ref :69-78): column c holds sin (c even) / cos (c odd) of
        pos / 10000^(2c/d_model), evaluated in Python floats."""
        table = torch.zeros(seq_len, d_model, device="cpu")  # filled element-wise: keep it on the host
        for pos in range(seq_len):
            for c in range(d_model):
                angle = pos / (10000 ** ((2 * c) / d_model))
                table[pos, c] = math.sin(angle) if c % 2 == 0 else math.cos(angle)
        return table

    def _layer_tensors(self):
        out = []
        for layer in self.multihead_attn_layers:
            out.extend(layer.tensors())
        return out

    def _pe(self):
        return self.positional_embeddings if self.use_positional_encoding else None

    def forward(self, user_history: torch.Tensor) -> torch.Tensor:
        """[B, H, DI] (newest item first) -> [B, 2, DI] (ref :80-121)."""
        return ops.HistoryEncoder.apply(user_history, None, self._pe(), self.num_attention_heads, *self._layer_tensors())

    def encode_ids(self, item_table: torch.Tensor, history_ids: torch.Tensor) -> torch.Tensor:
        """Same result as ``forward(item_table[history_ids])`` with the lookup fused in:
        each table row is read from HBM once for both the mean pool and the attention input."""
        return ops.HistoryEncoder.apply(item_table, history_ids, self._pe(), self.num_attention_heads,
                                        *self._layer_tensors())

    def get_output_dim(self) -> int:
        return self.item_id_embedding_dim * 2

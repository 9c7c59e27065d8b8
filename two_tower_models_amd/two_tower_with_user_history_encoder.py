"""TwoTowerWithUserHistoryEncoder on MI355X (mirror of
ref:src/two_tower_with_user_history_encoder.py:14-122): the user tower input gains
the [recent | mean] summary of the user's item history, looked up in the ITEM table."""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import ops
from .two_tower_base_retrieval import TwoTowerBaseRetrieval
from .user_history_encoder import UserHistoryEncoder


class TwoTowerWithUserHistoryEncoder(TwoTowerBaseRetrieval):
    # constructor keywords = ref :19-31 (callers pass them by name)
    def __init__(self, num_items: int, user_id_hash_size: int, user_id_embedding_dim: int, user_features_size: int,
                 user_history_seqlen: int, item_id_hash_size: int, item_id_embedding_dim: int,
                 item_features_size: int, user_value_weights: List[float], mips_module: nn.Module) -> None:
        base_kw = dict(num_items=num_items, user_id_hash_size=user_id_hash_size,
                       user_id_embedding_dim=user_id_embedding_dim, user_features_size=user_features_size,
                       item_id_hash_size=item_id_hash_size, item_id_embedding_dim=item_id_embedding_dim,
                       item_features_size=item_features_size, user_value_weights=user_value_weights,
                       mips_module=mips_module)
        super().__init__(**base_kw)
        DU, DI = user_id_embedding_dim, item_id_embedding_dim
        # upstream hard-codes 4 heads x 3 layers with positional encoding (ref :64-70)
        self.user_history_encoder = UserHistoryEncoder(DI, user_history_seqlen, num_attention_heads=4,
                                                       num_attention_layers=3, use_positional_encoding=True)
        # replaces the base tower: its input is now 2*DU + 2*DI wide (ref :81-83)
        self.user_tower_arch = nn.Linear(2 * DU + self.user_history_encoder.get_output_dim(), DI)

    def process_user_features(
        self, user_id: torch.Tensor, user_features: torch.Tensor, user_history: torch.Tensor
    ) -> torch.Tensor:
        """[id emb | feature MLP | recent | mean] -> [B, 2*DU + 2*DI] (ref :85-122)."""
        summary = self._history_summary(user_history)
        base = super().process_user_features(user_id=user_id, user_features=user_features, user_history=user_history)
        return torch.cat([base, summary], dim=1)

    def _history_summary(self, user_history: torch.Tensor) -> torch.Tensor:
        enc = self.user_history_encoder
        if isinstance(enc, UserHistoryEncoder):
            summary = enc.encode_ids(self.item_id_embedding_arch.weight, user_history)  # [B, 2, DI]
        else:  # a user-supplied encoder module: plain lookup, then the module's own forward
            summary = enc(ops.EmbeddingLookup.apply(self.item_id_embedding_arch.weight, user_history))
        return summary.view(summary.shape[0], -1)

    def compute_user_embedding(
        self, user_id: torch.Tensor, user_features: torch.Tensor, user_history: torch.Tensor
    ) -> torch.Tensor:
        """Query embedding [B, DI] (ref :85-122 + base :164-191).  With no hook overridden the id lookup, the feature
        MLP, the [id | MLP | recent | mean] cat and the Linear(4D -> D) run as ONE kernel per direction (K3 with a
        third input block, csrc/tower.hip); the encoder summary is that block."""
        cls = type(self)
        mlp, tower = self.user_features_arch, self.user_tower_arch
        if (cls.process_user_features is TwoTowerWithUserHistoryEncoder.process_user_features
                and cls.get_user_embedding is TwoTowerBaseRetrieval.get_user_embedding and user_id.is_cuda):
            ops.N.oob.poll(user_id.device)  # surfaces an out-of-range id seen by an earlier launch
            summary = self._history_summary(user_history)
            if ops.fused_tower_supported(self.user_id_embedding_arch.weight, user_features, mlp[0].weight, mlp[2].weight,
                                         tower.weight, extra_width=summary.shape[1]):
                return ops.FusedTower.apply(self.user_id_embedding_arch.weight, user_id, user_features, mlp[0].weight,
                                            mlp[0].bias, mlp[2].weight, mlp[2].bias, tower.weight, tower.bias, summary)
            base = TwoTowerBaseRetrieval.process_user_features(self, user_id=user_id, user_features=user_features,
                                                               user_history=user_history)
            return ops.Linear.apply(torch.cat([base, summary], dim=1), tower.weight, tower.bias)
        return super().compute_user_embedding(user_id, user_features, user_history)

    def _lookup_plan(self, user_id, user_history, item_id):
        # forward order: history rows (encoder), then the user row, then the item row; row-sharded tables: the item
        # tower runs first (TwoTowerBaseRetrieval.train_forward), so the item row is the item table's first lookup
        item_blocks = [item_id, user_history] if self._sharded() else [user_history, item_id]
        return {self.user_id_embedding_arch.weight: [user_id], self.item_id_embedding_arch.weight: item_blocks}

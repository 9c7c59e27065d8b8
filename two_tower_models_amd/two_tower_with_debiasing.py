"""TwoTowerWithDebiasing on MI355X (mirror of ref:src/two_tower_with_debiasing.py:17-129).

``forward`` (inference -> MIPS top-K, BASELINE config 5) is inherited unchanged and
runs entirely on the HIP path.  The debias head itself is O(B) work on [B]-sized
tensors; it is expressed with the reference's own tensor expressions so that its
gradient semantics (through the clamp, the division and ``torch.max``) and the
upstream [B,1]-vs-[B] ``mse_loss`` broadcast are reproduced literally, on top of the
fused in-batch-softmax kernel (SURVEY.md 8f item 2 lists fusing it as "next")."""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .two_tower_with_user_history_encoder import TwoTowerWithUserHistoryEncoder


class TwoTowerWithDebiasing(TwoTowerWithUserHistoryEncoder):
    # constructor keywords = ref :18-30
    def __init__(self, num_items: int, user_id_hash_size: int, user_id_embedding_dim: int, user_features_size: int,
                 user_history_seqlen: int, item_id_hash_size: int, item_id_embedding_dim: int,
                 item_features_size: int, user_value_weights: List[float], mips_module: nn.Module) -> None:
        super().__init__(num_items, user_id_hash_size, user_id_embedding_dim, user_features_size,
                         user_history_seqlen, item_id_hash_size, item_id_embedding_dim, item_features_size,
                         user_value_weights, mips_module)
        # position prior: one scalar per position bucket; user prior: Linear([user emb | position prior]) -> 1
        self.position_bias_net_user_value = nn.Embedding(100, 1)
        self.user_debias_net_user_value = nn.Sequential(nn.Linear(item_id_embedding_dim + 1, 1))

    def debias_net_user_value(self, net_user_value: torch.Tensor, position: torch.Tensor,
                              user_embedding: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(net_user_value / clamp(user prior), sum-MSE of both priors) -- ref :77-129, including
        upstream's [B,1]-vs-[B] broadcast inside the position loss."""
        pos_prior = self.position_bias_net_user_value(position)  # [B, 1]
        head_in = torch.cat([user_embedding, pos_prior], dim=-1)  # [B, DI + 1]
        user_prior = self.user_debias_net_user_value(head_in).squeeze(1)  # [B]
        aux = F.mse_loss(user_prior, net_user_value, reduction="sum") \
            + F.mse_loss(pos_prior, net_user_value, reduction="sum")
        return net_user_value / torch.clamp(user_prior, min=1e-3), aux

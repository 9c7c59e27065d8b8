"""TwoTowerWithDebiasing on MI355X (mirror of ref:src/two_tower_with_debiasing.py:17-129).

``forward`` (inference -> MIPS top-K, BASELINE config 5) is inherited unchanged and
runs entirely on the HIP path.  The training loss head is fused as well
(``ops.DebiasedWeightedLoss`` / ``tt_debias_loss_fwd``: position prior, user prior, both
sum-MSE terms with the upstream [B,1]-vs-[B] broadcast in closed form, the division by
the clamped prior, clamp and division by the batch maximum, weighted mean -- SURVEY.md 8f
item 2); ``debias_net_user_value`` keeps the reference's own tensor expressions for callers
of the hook and as the A/B partner of the fused head (TT_DEBIAS_NO_FUSED)."""
from __future__ import annotations

from typing import List, Tuple

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .two_tower_with_user_history_encoder import TwoTowerWithUserHistoryEncoder

_FUSED_HEAD = os.environ.get("TT_DEBIAS_NO_FUSED") is None  # A/B switch


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

    def _loss_head(self, row_ce: torch.Tensor, labels: torch.Tensor, position: torch.Tensor,
                   user_embedding: torch.Tensor) -> torch.Tensor:
        """The base loss head with this class's debias hook, fused: one op instead of ~15
        elementwise launches and a [B, B] temporary.  A subclass that overrides the hook, or labels the kernel does not
        take, get the reference's expressions (the base class's head).  Row-sharded training calls this on the
        gathered batch (TwoTowerBaseRetrieval._sharded_training_loss)."""
        hook_is_mine = type(self).debias_net_user_value is TwoTowerWithDebiasing.debias_net_user_value
        if not (_FUSED_HEAD and hook_is_mine and labels.dim() == 2 and labels.shape[1] == self.user_value_weights.numel()
                and ops.labels_fusable(labels) and user_embedding.is_cuda):
            return super()._loss_head(row_ce, labels, position, user_embedding)
        lin = self.user_debias_net_user_value[0]
        return ops.DebiasedWeightedLoss.apply(row_ce, labels, self.user_value_weights, position, user_embedding,
                                              self.position_bias_net_user_value.weight, lin.weight, lin.bias)

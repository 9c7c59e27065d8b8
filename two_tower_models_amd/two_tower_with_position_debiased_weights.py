"""TwoTowerWithPositionDebiasedWeights on MI355X (mirror of ref:src/two_tower_with_position_debiased_weights.py:16-113).

The history model with ONE extra parameter tensor -- a scalar prior of the net user value per position bucket,
`position_bias_net_user_value` Embedding(100, 1) (ref :72-74) -- and its `debias_net_user_value` hook (ref :76-113).
Lookups, towers, encoder, in-batch logits / CE and the optimiser are the inherited HIP path; the loss head -- hook,
clamp, division by the batch maximum, weighted mean -- is the fused debias kernel in its position-only mode
(csrc/debias.hip, TT_DEBIAS_POSITION).  A subclass that overrides the hook again, or labels the kernel does
not take, fall back to the hook's tensor expressions (general branch of TwoTowerBaseRetrieval.compute_training_loss)."""
from __future__ import annotations

from typing import List, Tuple

import os

import torch
import torch.nn as nn

from . import _native as N
from . import ops
from .two_tower_with_user_history_encoder import TwoTowerWithUserHistoryEncoder

_FUSED_HEAD = os.environ.get("TT_DEBIAS_NO_FUSED") is None  # A/B switch


class TwoTowerWithPositionDebiasedWeights(TwoTowerWithUserHistoryEncoder):
    # constructor keywords = ref :29-41
    def __init__(self, num_items: int, user_id_hash_size: int, user_id_embedding_dim: int, user_features_size: int,
                 user_history_seqlen: int, item_id_hash_size: int, item_id_embedding_dim: int,
                 item_features_size: int, user_value_weights: List[float], mips_module: nn.Module) -> None:
        super().__init__(num_items, user_id_hash_size, user_id_embedding_dim, user_features_size,
                         user_history_seqlen, item_id_hash_size, item_id_embedding_dim, item_features_size,
                         user_value_weights, mips_module)
        self.position_bias_net_user_value = nn.Embedding(100, 1)

    def debias_net_user_value(self, net_user_value: torch.Tensor, position: torch.Tensor,
                              user_embedding: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(value / max(position prior, 1e-3), sum of squared prior errors): the auxiliary loss sees the RAW prior,
        the division its clamped copy (ref :95-113)."""
        prior = self.position_bias_net_user_value.weight[position, 0]  # [B]
        aux = torch.sum((prior - net_user_value) ** 2)
        return net_user_value / prior.clamp(min=1e-3), aux

    def _loss_head(self, row_ce: torch.Tensor, labels: torch.Tensor, position: torch.Tensor,
                   user_embedding: torch.Tensor) -> torch.Tensor:
        hook_is_mine = type(self).debias_net_user_value is TwoTowerWithPositionDebiasedWeights.debias_net_user_value
        if not (_FUSED_HEAD and hook_is_mine and labels.dim() == 2 and labels.shape[1] == self.user_value_weights.numel()
                and ops.labels_fusable(labels) and user_embedding.is_cuda):
            return super()._loss_head(row_ce, labels, position, user_embedding)
        return ops.DebiasedWeightedLoss.apply(row_ce, labels, self.user_value_weights, position, user_embedding,
                                              self.position_bias_net_user_value.weight, None, None, N.TT_DEBIAS_POSITION)

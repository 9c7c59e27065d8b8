"""TwoTowerWithUserDebiasedWeights on MI355X (mirror of ref:src/two_tower_with_user_debiased_weights.py:41-135).

The history model plus a one-unit head on the user embedding, `user_debias_net_user_value` = Sequential(Linear(DI, 1))
(ref :96-100), that estimates how much value a user yields whatever is shown; the example weight is divided by it
(ref :102-135).  Everything up to the loss head is the inherited HIP path; the head -- hook (the Linear(DI, 1) prior with
its clamp FIRST), clamp, division by the batch maximum, weighted mean -- is the fused debias kernel in its user-only
mode (csrc/debias.hip, TT_DEBIAS_USER): no library gemv on the product path.  A subclass that overrides
the hook again, or labels the kernel does not take, fall back to the hook's tensor expressions."""
from __future__ import annotations

from typing import List, Tuple

import os

import torch
import torch.nn as nn

from . import _native as N
from . import ops
from .two_tower_with_user_history_encoder import TwoTowerWithUserHistoryEncoder

_FUSED_HEAD = os.environ.get("TT_DEBIAS_NO_FUSED") is None  # A/B switch


class TwoTowerWithUserDebiasedWeights(TwoTowerWithUserHistoryEncoder):
    # constructor keywords = ref :54-66
    def __init__(self, num_items: int, user_id_hash_size: int, user_id_embedding_dim: int, user_features_size: int,
                 user_history_seqlen: int, item_id_hash_size: int, item_id_embedding_dim: int,
                 item_features_size: int, user_value_weights: List[float], mips_module: nn.Module) -> None:
        super().__init__(num_items, user_id_hash_size, user_id_embedding_dim, user_features_size,
                         user_history_seqlen, item_id_hash_size, item_id_embedding_dim, item_features_size,
                         user_value_weights, mips_module)
        self.user_debias_net_user_value = nn.Sequential(nn.Linear(item_id_embedding_dim, 1))

    def debias_net_user_value(self, net_user_value: torch.Tensor, position: torch.Tensor,
                              user_embedding: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(value / prior, sum of squared prior errors) with prior = max(head(user embedding), 0.1): unlike the
        position variant the clamp comes FIRST here, so a clamped row contributes a constant to the auxiliary loss
        and no gradient to the head (ref :121-135)."""
        head = self.user_debias_net_user_value[0]
        prior = (user_embedding @ head.weight[0] + head.bias[0]).clamp(min=1e-1)  # [B]
        aux = torch.sum((prior - net_user_value) ** 2)
        return net_user_value / prior, aux

    def _loss_head(self, row_ce: torch.Tensor, labels: torch.Tensor, position: torch.Tensor,
                   user_embedding: torch.Tensor) -> torch.Tensor:
        hook_is_mine = type(self).debias_net_user_value is TwoTowerWithUserDebiasedWeights.debias_net_user_value
        if not (_FUSED_HEAD and hook_is_mine and labels.dim() == 2 and labels.shape[1] == self.user_value_weights.numel()
                and ops.labels_fusable(labels) and user_embedding.is_cuda):
            return super()._loss_head(row_ce, labels, position, user_embedding)
        head = self.user_debias_net_user_value[0]
        return ops.DebiasedWeightedLoss.apply(row_ce, labels, self.user_value_weights, position, user_embedding,
                                              None, head.weight, head.bias, N.TT_DEBIAS_USER)

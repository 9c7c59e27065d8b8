"""CPU oracle for the two-tower hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU and in explicit fp32 math, the algorithm of the
reference's hot path (gauravchak/two_tower_models @ 2025-02-11).  It is the
*checker* for the HIP kernels in ``two_tower_models_amd/csrc`` and the thing
timed as ``cpu_baseline`` (kind "port") in ``bench.py``.  Nothing under
``two_tower_models_amd/`` may import it: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg do.

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_golden.py`` against golden vectors produced by importing
the reference itself (``tests/golden/make_golden.py``, run in the build
container where ``/root/reference`` is mounted), including the reference's own
known-answer test (ref:tests/test_user_history_enc.py:48-124).

Style: functional.  Parameters live in a flat ``dict`` keyed by the
reference's ``state_dict`` names, e.g.
``user_features_arch.0.weight``.  Every op is written out (explicit
multi-head-attention algebra, explicit log-sum-exp cross entropy, explicit
Adam update) rather than delegated to ``nn.MultiheadAttention`` /
``F.cross_entropy`` / ``optim.Adam`` so that the restatement is independent of
the torch modules the reference calls.  Gradients come from torch autograd over
these explicit forward formulas (the reference also differentiates with
autograd, ref:train/train.py:124).
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# User-history encoder  (ref:src/user_history_encoder.py)
# --------------------------------------------------------------------------
def positional_table(history_len: int, dim: int, flipped: bool = True) -> torch.Tensor:
    """The reference's non-standard sinusoid table, computed in Python float64
    and stored as fp32 (ref:src/user_history_encoder.py:69-78).  Even column c
    holds sin(pos / 10000^(2c/dim)); odd column c holds cos(pos / 10000^(2c/dim))
    -- note the exponent uses 2*c for *both* parities (the reference steps ``i``
    by 2 and uses ``2*i`` / ``2*(i+1)``).  ``flipped`` applies the row reversal
    of ref:src/user_history_encoder.py:54 (newest item sits at index 0)."""
    table = torch.zeros(history_len, dim, dtype=torch.float32)
    for pos in range(history_len):
        for c in range(dim):
            angle = pos / (10000.0 ** ((2.0 * c) / dim))
            table[pos, c] = math.sin(angle) if c % 2 == 0 else math.cos(angle)
    if flipped:
        table = torch.flip(table, dims=[0])
    return table


def self_attention_layer(
    x: torch.Tensor,  # [B, H, D]
    w_in: torch.Tensor,  # [3D, D]  rows: Q | K | V
    b_in: torch.Tensor,  # [3D]
    w_out: torch.Tensor,  # [D, D]
    b_out: torch.Tensor,  # [D]
    heads: int,
) -> torch.Tensor:
    """One unmasked multi-head self-attention layer, no residual / norm / FFN /
    dropout: what ``nn.MultiheadAttention(D, heads)(x, x, x)[0]`` computes at
    ref:src/user_history_encoder.py:103-108 (SURVEY.md 3.3 for the algebra)."""
    B, H, D = x.shape
    dh = D // heads
    qkv = x.reshape(B * H, D) @ w_in.t() + b_in  # [BH, 3D]
    q, k, v = qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :]

    def split(t: torch.Tensor) -> torch.Tensor:  # [BH, D] -> [B, heads, H, dh]
        return t.reshape(B, H, heads, dh).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    scores = (q * (1.0 / math.sqrt(dh))) @ k.transpose(-1, -2)  # [B, heads, H, H]
    scores = scores - scores.max(dim=-1, keepdim=True).values
    prob = torch.exp(scores)
    prob = prob / prob.sum(dim=-1, keepdim=True)
    ctx = (prob @ v).permute(0, 2, 1, 3).reshape(B * H, D)  # concat heads
    return (ctx @ w_out.t() + b_out).reshape(B, H, D)


def history_encoder_forward(
    history_emb: torch.Tensor,  # [B, H, D]
    layers: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]],
    heads: int,
    pos_table: Optional[torch.Tensor],  # [H, D] (already flipped) or None
) -> torch.Tensor:
    """ref:src/user_history_encoder.py:80-121 -> [B, 2, D]: slot 0 = position-0
    row after the attention stack, slot 1 = mean over H of the RAW embeddings."""
    pooled = history_emb.sum(dim=1) / history_emb.shape[1]
    x = history_emb if pos_table is None else history_emb + pos_table.unsqueeze(0)
    for w_in, b_in, w_out, b_out in layers:
        x = self_attention_layer(x, w_in, b_in, w_out, b_out, heads)
    return torch.stack([x[:, 0, :], pooled], dim=1)


def encoder_layers_from_params(params: Params, prefix: str = "user_history_encoder.") -> List[Tuple]:
    out = []
    i = 0
    while f"{prefix}multihead_attn_layers.{i}.in_proj_weight" in params:
        base = f"{prefix}multihead_attn_layers.{i}."
        out.append(
            (
                params[base + "in_proj_weight"],
                params[base + "in_proj_bias"],
                params[base + "out_proj.weight"],
                params[base + "out_proj.bias"],
            )
        )
        i += 1
    return out


# --------------------------------------------------------------------------
# Towers  (ref:src/two_tower_base_retrieval.py:112-219,
#          ref:src/two_tower_with_user_history_encoder.py:85-122)
# --------------------------------------------------------------------------
def feature_mlp(x: torch.Tensor, params: Params, prefix: str) -> torch.Tensor:
    """Linear(F,256) -> ReLU -> Linear(256,D)  (ref:...base_retrieval.py:76-80)."""
    h = x @ params[prefix + "0.weight"].t() + params[prefix + "0.bias"]
    h = torch.clamp(h, min=0.0)
    return h @ params[prefix + "2.weight"].t() + params[prefix + "2.bias"]


def user_tower_input(
    params: Params,
    user_id: torch.Tensor,
    user_features: torch.Tensor,
    user_history: Optional[torch.Tensor],
    *,
    with_history: bool,
    heads: int = 4,
    pos_table: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """[id-embedding | feature-MLP (| recent | mean)]  -- base:
    ref:...base_retrieval.py:129-162; history variant appends the encoder
    summary computed from rows of the ITEM table
    (ref:...with_user_history_encoder.py:105-121)."""
    id_emb = params["user_id_embedding_arch.weight"][user_id]
    feat = feature_mlp(user_features, params, "user_features_arch.")
    pieces = [id_emb, feat]
    if with_history:
        hist_emb = params["item_id_embedding_arch.weight"][user_history]  # [B,H,DI]
        summary = history_encoder_forward(
            hist_emb, encoder_layers_from_params(params), heads, pos_table
        )
        pieces.append(summary.reshape(summary.shape[0], -1))
    return torch.cat(pieces, dim=1)


def user_embedding(params: Params, user_id, user_features, user_history, **kw) -> torch.Tensor:
    """ref:...base_retrieval.py:164-191."""
    tin = user_tower_input(params, user_id, user_features, user_history, **kw)
    return tin @ params["user_tower_arch.weight"].t() + params["user_tower_arch.bias"]


def item_embeddings(params: Params, item_id, item_features) -> torch.Tensor:
    """ref:...base_retrieval.py:193-219."""
    id_emb = params["item_id_embedding_arch.weight"][item_id]
    feat = feature_mlp(item_features, params, "item_features_arch.")
    tin = torch.cat([id_emb, feat], dim=1)
    return tin @ params["item_tower_arch.weight"].t() + params["item_tower_arch.bias"]


# --------------------------------------------------------------------------
# In-batch softmax loss  (ref:src/two_tower_base_retrieval.py:279-347)
# --------------------------------------------------------------------------
def inbatch_logits(user_emb: torch.Tensor, item_emb: torch.Tensor) -> torch.Tensor:
    return user_emb @ item_emb.t()  # ref :287


def inbatch_rowwise_ce(user_emb: torch.Tensor, item_emb: torch.Tensor, diag_offset: int = 0) -> torch.Tensor:
    """Per-row cross entropy against the diagonal (ref :301-312):
    ce_i = logsumexp_j S_ij - S_i,(i+diag_offset).  ``diag_offset`` (0 for the
    reference) supports the sharded layout where a rank's positives sit at
    columns rank*B + i of the all-gathered item block."""
    s = inbatch_logits(user_emb, item_emb)
    m = s.max(dim=1, keepdim=True).values
    lse = (m + torch.log(torch.exp(s - m).sum(dim=1, keepdim=True))).squeeze(1)
    idx = torch.arange(s.shape[0]) + diag_offset
    return lse - s[torch.arange(s.shape[0]), idx]


def net_user_value(labels: torch.Tensor, user_value_weights: torch.Tensor) -> torch.Tensor:
    """ref :322 -- literally ``sum(labels * weights, dim=-1)``; with train.py's
    1-D labels this collapses to a 0-d scalar."""
    return torch.sum(labels * user_value_weights, dim=-1)


def normalise_value_weights(nuv: torch.Tensor) -> torch.Tensor:
    """ref :334-339: clamp(min=1e-6) then divide by the batch max."""
    nuv = torch.clamp(nuv, min=0.000001)
    return nuv / torch.max(nuv)


def debias_identity(nuv, position, user_emb, params):
    """ref:...base_retrieval.py:251-277."""
    return nuv, 0


def debias_combined(nuv, position, user_emb, params: Params):
    """ref:src/two_tower_with_debiasing.py:77-129 (position embedding(100,1) +
    Linear(DI+1 -> 1); two sum-MSE auxiliary losses, the first with the
    reference's [B,1]-vs-[B] broadcast; divide by clamp(E, 1e-3))."""
    e_pos = params["position_bias_net_user_value.weight"][position]  # [B,1]
    w = params["user_debias_net_user_value.0.weight"]  # [1, DI+1]
    b = params["user_debias_net_user_value.0.bias"]
    e_user = (torch.cat([user_emb, e_pos], dim=-1) @ w.t() + b).squeeze(1)  # [B]
    pos_loss = ((e_pos - nuv) ** 2).sum()  # [B,1]-[B] -> [B,B] broadcast, as upstream
    user_loss = ((e_user - nuv) ** 2).sum()
    e_user = torch.clamp(e_user, min=1e-3)
    return nuv / e_user, user_loss + pos_loss


def debias_position(nuv, position, user_emb, params: Params):
    """ref:src/two_tower_with_position_debiased_weights.py:76-113: prior = Embedding(100, 1)[position]; sum-MSE of the
    raw prior; divide by clamp(prior, 1e-3)."""
    e_pos = params["position_bias_net_user_value.weight"][position].squeeze(1)  # [B]
    loss = ((e_pos - nuv) ** 2).sum()
    return nuv / torch.clamp(e_pos, min=1e-3), loss


def debias_user(nuv, position, user_emb, params: Params):
    """ref:src/two_tower_with_user_debiased_weights.py:102-135: prior = clamp(Linear(DI -> 1)(user_emb), 0.1) -- the
    clamp BEFORE the sum-MSE --, divide by it."""
    w = params["user_debias_net_user_value.0.weight"]  # [1, DI]
    b = params["user_debias_net_user_value.0.bias"]
    e_user = torch.clamp((user_emb @ w.t() + b).squeeze(1), min=1e-1)
    loss = ((e_user - nuv) ** 2).sum()
    return nuv / e_user, loss


def training_loss(
    user_emb: torch.Tensor,
    item_emb: torch.Tensor,
    position: torch.Tensor,
    labels: torch.Tensor,
    user_value_weights: torch.Tensor,
    params: Optional[Params] = None,
    debias=debias_identity,
) -> torch.Tensor:
    ce = inbatch_rowwise_ce(user_emb, item_emb)
    nuv = net_user_value(labels, user_value_weights)
    nuv, extra = debias(nuv, position, user_emb, params)
    w = normalise_value_weights(nuv)
    return (ce * w).sum() / ce.shape[0] + extra  # ref :342-346 (mean, then + aux)


def train_forward(
    params: Params,
    batch: Sequence[torch.Tensor],
    user_value_weights: torch.Tensor,
    *,
    with_history: bool = False,
    heads: int = 4,
    pos_table: Optional[torch.Tensor] = None,
    debias=debias_identity,
) -> torch.Tensor:
    """ref:...base_retrieval.py:349-394.  ``batch`` is the 7-tuple of
    ref:train/train.py:91-99."""
    user_id, user_features, user_history, item_id, item_features, position, labels = batch
    u = user_embedding(
        params, user_id, user_features, user_history,
        with_history=with_history, heads=heads, pos_table=pos_table,
    )
    it = item_embeddings(params, item_id, item_features)
    return training_loss(u, it, position, labels, user_value_weights, params, debias)


# --------------------------------------------------------------------------
# MIPS  (ref:src/baseline_mips_module.py:32-72)
# --------------------------------------------------------------------------
def mips_topk(query: torch.Tensor, corpus: torch.Tensor, k: int, chunk: int = 64):
    """Brute-force max-inner-product top-k, sorted by (score desc, index asc) --
    the total order the build defines where ``torch.topk`` leaves ties
    arbitrary (SURVEY.md section 7 hard part 1).  Returns (int64 idx [B,k],
    scores [B,k], rows [B,k,D]).  Queries are processed ``chunk`` at a time so
    the [B,C] score matrix the reference materialises (ref :57-61) stays small."""
    idx_out, sc_out = [], []
    for lo in range(0, query.shape[0], chunk):
        s = query[lo : lo + chunk] @ corpus.t()
        order = torch.sort(s, dim=1, descending=True, stable=True)  # stable => idx asc on ties
        idx_out.append(order.indices[:, :k].clone())
        sc_out.append(order.values[:, :k].clone())
    idx = torch.cat(idx_out)
    sc = torch.cat(sc_out)
    return idx, sc, corpus[idx]


def round_to_bf16(x: torch.Tensor) -> torch.Tensor:
    """bf16 storage model for the MIPS corpus / query (config 5): round to
    nearest-even bf16, then widen back to fp32 (products of two such values are
    exact in fp32)."""
    return x.to(torch.bfloat16).to(torch.float32)


# --------------------------------------------------------------------------
# Optimiser: dense Adam on every element  (ref:train/train.py:123-125,179)
# --------------------------------------------------------------------------
def adam_update(
    p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor,
    step: int, lr: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
) -> None:
    """In-place torch.optim.Adam semantics (no weight decay, no amsgrad) with the very tensor ops of torch's
    single-tensor path (torch/optim/adam.py `_single_tensor_adam`, the code behind ref:train/train.py:179):
    lerp_ for m, mul_ + addcmul_ for v, denom = sqrt(v) / sqrt(bc2) + eps, addcdiv_ for p.  One temporary
    (denom) per call -- the earlier spelled-out expressions allocated five table-sized ones, which made this
    port slower than the reference it restates."""
    m.lerp_(g, 1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt().div_(math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


class AdamState:
    def __init__(self, params: Params):
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.step = 0


def train_step(
    params: Params, state: AdamState, batch, user_value_weights: torch.Tensor,
    lr: float = 1e-3, **fwd_kw,
) -> float:
    """One iteration of the loop body ref:train/train.py:112-132: forward,
    (zero_grad,) backward, Adam on EVERY row of both tables, loss.item()."""
    leaves = {k: v.detach().requires_grad_(True) for k, v in params.items()}
    loss = train_forward(leaves, batch, user_value_weights, **fwd_kw)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    state.step += 1
    with torch.no_grad():
        for (name, p), g in zip(params.items(), grads):
            if g is None:
                continue  # torch.optim skips params whose .grad is None
            adam_update(p, g, state.m[name], state.v[name], state.step, lr)
    return loss.item()

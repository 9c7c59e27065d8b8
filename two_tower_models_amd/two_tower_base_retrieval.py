"""TwoTowerBaseRetrieval on MI355X.

Same constructor, methods, keyword names, Parameter names and initialisation
stream as ref:src/two_tower_base_retrieval.py:25-394, so a caller of the
reference (train loop, tests, a state_dict) can switch over unchanged.  The
submodules created in ``__init__`` (nn.Embedding / nn.Sequential / nn.Linear)
are parameter CONTAINERS only: they give the reference's state_dict keys and
its default initialisation (same RNG draws in the same order), and are never
called.  All arithmetic runs in libtt_hotpath.so through ``ops``.

Multi-GPU: the same class, built under ``parallel.row_sharded()`` or passed through
``parallel.shard_model_``, holds only this rank's row block of each table.  ``train_forward`` then computes the
reference's loss on the CONCATENATED batch of all ranks (routed lookups, global in-batch negatives, group-wide
value-weight maximum) and ``loss.backward()`` / ``DenseExactAdam.step()`` apply the reference's update; every hook
below stays overridable.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn

from . import _native as N
from . import ops
from . import parallel
from .baseline_mips_module import BaselineMIPSModule

_FEATURE_HIDDEN = 256  # ref:src/two_tower_base_retrieval.py:76-80


def _feature_mlp(in_features: int, out_features: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(in_features, _FEATURE_HIDDEN), nn.ReLU(), nn.Linear(_FEATURE_HIDDEN, out_features))


def mark_table(weight: nn.Parameter) -> None:
    """Tag an embedding table so DenseExactAdam consumes its gradient in row form."""
    weight._tt_is_table = True


class TwoTowerBaseRetrieval(nn.Module):
    """Two-tower candidate retrieval: id-embedding + feature-MLP towers, in-batch
    softmax loss weighted by net user value, MIPS top-K inference."""

    def __init__(
        self,
        num_items: int,
        user_id_hash_size: int,
        user_id_embedding_dim: int,
        user_features_size: int,
        item_id_hash_size: int,
        item_id_embedding_dim: int,
        item_features_size: int,
        user_value_weights: List[float],
        mips_module: BaselineMIPSModule,
    ) -> None:
        super().__init__()
        self.num_items = num_items
        # plain tensor attribute, like the reference (:62); unlike the reference it
        # follows .to()/.cuda() (see _apply) so the GPU path actually runs.
        self.user_value_weights = torch.tensor(user_value_weights)
        self.mips_module = mips_module
        # creation order == reference order (:70-110) so a seeded init is bit-identical
        self.user_id_embedding_arch = parallel.embedding(user_id_hash_size, user_id_embedding_dim)
        self.user_features_arch = _feature_mlp(user_features_size, user_id_embedding_dim)
        self.user_tower_arch = nn.Linear(2 * user_id_embedding_dim, item_id_embedding_dim)
        self.item_id_embedding_arch = parallel.embedding(item_id_hash_size, item_id_embedding_dim)
        self.item_features_arch = _feature_mlp(item_features_size, item_id_embedding_dim)
        self.item_tower_arch = nn.Linear(2 * item_id_embedding_dim, item_id_embedding_dim)
        mark_table(self.user_id_embedding_arch.weight)
        mark_table(self.item_id_embedding_arch.weight)

    # plain-tensor attributes travel with the module (the reference leaves them on the
    # CPU, ref "TODO add device input" :61, which makes its GPU path unusable)
    def _apply(self, fn, *a, **kw):
        super()._apply(fn, *a, **kw)
        self.user_value_weights = fn(self.user_value_weights)
        return self

    # ------------------------------------------------------------------ user tower
    def get_user_embedding(self, user_id: torch.Tensor, user_features: torch.Tensor) -> torch.Tensor:
        """[B] ids -> [B, DU] rows of the user table (ref :112-127)."""
        return ops.EmbeddingLookup.apply(self.user_id_embedding_arch.weight, user_id)

    def process_user_features(
        self, user_id: torch.Tensor, user_features: torch.Tensor, user_history: torch.Tensor
    ) -> torch.Tensor:
        """[id embedding | feature MLP] -> [B, 2*DU] (ref :129-162).  ``user_history`` is
        unused here, as upstream."""
        if type(self).get_user_embedding is not TwoTowerBaseRetrieval.get_user_embedding:
            # a subclass replaced the id representation: honour the hook, lose the fusion
            id_emb = self.get_user_embedding(user_id=user_id, user_features=user_features)
            mlp = self.user_features_arch
            feat = ops.FeatureMLP.apply(user_features, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)
            return torch.cat([id_emb, feat], dim=1)
        mlp = self.user_features_arch
        return ops.TowerInput.apply(
            self.user_id_embedding_arch.weight, user_id, user_features,
            mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias,
        )

    def compute_user_embedding(
        self, user_id: torch.Tensor, user_features: torch.Tensor, user_history: torch.Tensor
    ) -> torch.Tensor:
        """Query embedding [B, DI] (ref :164-191)."""
        if user_id.is_cuda:
            N.oob.poll(user_id.device)  # surfaces an out-of-range id seen by an earlier launch
        cls = type(self)
        mlp, tower = self.user_features_arch, self.user_tower_arch
        if (cls.process_user_features is TwoTowerBaseRetrieval.process_user_features
                and cls.get_user_embedding is TwoTowerBaseRetrieval.get_user_embedding
                and ops.fused_tower_supported(self.user_id_embedding_arch.weight, user_features, mlp[0].weight,
                                              mlp[2].weight, tower.weight)):
            # no hook overridden: lookup + feature MLP + cat + tower Linear as ONE kernel per direction (K3)
            return ops.FusedTower.apply(self.user_id_embedding_arch.weight, user_id, user_features, mlp[0].weight,
                                        mlp[0].bias, mlp[2].weight, mlp[2].bias, tower.weight, tower.bias)
        user_tower_input = self.process_user_features(
            user_id=user_id, user_features=user_features, user_history=user_history
        )
        return ops.Linear.apply(user_tower_input, self.user_tower_arch.weight, self.user_tower_arch.bias)

    # ------------------------------------------------------------------ item tower
    def compute_item_embeddings(self, item_id: torch.Tensor, item_features: torch.Tensor) -> torch.Tensor:
        """[B, DI] item embeddings (ref :193-219)."""
        mlp = self.item_features_arch
        if ops.fused_tower_supported(self.item_id_embedding_arch.weight, item_features, mlp[0].weight, mlp[2].weight,
                                     self.item_tower_arch.weight):
            return ops.FusedTower.apply(self.item_id_embedding_arch.weight, item_id, item_features, mlp[0].weight,
                                        mlp[0].bias, mlp[2].weight, mlp[2].bias, self.item_tower_arch.weight,
                                        self.item_tower_arch.bias)
        item_tower_input = ops.TowerInput.apply(
            self.item_id_embedding_arch.weight, item_id, item_features,
            mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias,
        )
        return ops.Linear.apply(item_tower_input, self.item_tower_arch.weight, self.item_tower_arch.bias)

    # ------------------------------------------------------------------ inference
    def forward(self, user_id: torch.Tensor, user_features: torch.Tensor, user_history: torch.Tensor) -> torch.Tensor:
        """Top ``num_items`` MIPS row indices per user, [B, num_items] int64 (ref :221-249)."""
        # (the result is an index tensor: nothing to differentiate.  Under no_grad the lookups are not recorded for the
        # optimiser's table step -- an inference call between two train steps, with or without the caller's own no_grad,
        # leaves the training bookkeeping alone, and a row-sharded model needs no optimiser to be served)
        with torch.no_grad():
            user_embedding = self.compute_user_embedding(user_id, user_features, user_history)
        if type(self.mips_module).forward is BaselineMIPSModule.forward:
            # this package's module: the reference discards the scores and the [B, K, DI] rows (ref :246-248) -- do not
            # gather (row-sharded corpus: do not exchange) half a gigabyte of rows at B = 1024, K = 1000 to drop them
            top_items, _ = self.mips_module.search(user_embedding, self.num_items)
        else:  # any other mips_module (a subclass, the caller's own): the reference's call, keyword for keyword
            top_items, _, _ = self.mips_module(query_embedding=user_embedding, num_items=self.num_items)
        N.oob.poll(user_embedding.device, blocking=True)
        return top_items

    # ------------------------------------------------------------------ loss
    @torch.no_grad()
    def index_corpus(self, item_id: torch.Tensor, item_features: torch.Tensor, chunk: int = 262144,
                     bf16: bool = False) -> None:
        """Serve what was trained (upstream searches a random corpus): run the item
        tower over the catalogue (item_id [C], item_features [C, II]) in chunks and install the
        embeddings as the MIPS corpus -- corpus row r is item item_id[r].

        Row-sharded model (collective: every rank calls): `item_id` / `item_features` are THIS RANK'S BLOCK of the
        catalogue -- rows [lo, hi) = parallel.block_range(C, rank, world)[1:], any item ids -- and the model's
        mips_module becomes (or stays) row-sharded with that block; `forward()` then searches all blocks together.  When
        every rank's ids are rows it owns itself (the usual catalogue: item r = row r of the item table) no row travels."""
        sharded = self._sharded()
        local_only = False
        if sharded:
            w = self.item_id_embedding_arch.weight
            sh = parallel.shard_of(w)
            foreign = torch.zeros(1, dtype=torch.int32, device=item_id.device)  # (MAX, not MIN: what tt_comm_* reduces)
            if item_id.numel():
                foreign = ((item_id < sh.lo) | (item_id >= sh.hi)).any().to(torch.int32).reshape(1)
            local_only = int(parallel.C.all_reduce_(foreign, op=parallel.dist.ReduceOp.MAX).item()) == 0  # one decision for the group
            # every rank runs the same number of chunks (routed lookups are collective); a rank whose block is shorter
            # looks up row 0 once per surplus chunk and drops the result
            per = torch.tensor([item_id.shape[0]], dtype=torch.int64, device=item_id.device)
            n_chunks = -(-int(parallel.C.all_reduce_(per, op=parallel.dist.ReduceOp.MAX).item()) // chunk)
        else:
            n_chunks = -(-item_id.shape[0] // chunk)
        out = torch.empty(item_id.shape[0], self.item_id_embedding_arch.weight.shape[1], dtype=torch.float32,
                          device=item_id.device)
        for c in range(n_chunks):
            lo, hi = min(c * chunk, item_id.shape[0]), min((c + 1) * chunk, item_id.shape[0])
            if local_only:
                if hi > lo:
                    with parallel.local_rows():
                        out[lo:hi] = self.compute_item_embeddings(item_id[lo:hi] - sh.lo, item_features[lo:hi])
            elif hi > lo:
                out[lo:hi] = self.compute_item_embeddings(item_id[lo:hi], item_features[lo:hi])
            else:
                self.compute_item_embeddings(item_id.new_zeros(1), item_features.new_zeros(1, item_features.shape[1]))
        if sharded:
            self.mips_module.set_corpus(out, bf16=bf16, block=True)  # becomes (or stays) row-sharded
        else:
            self.mips_module.set_corpus(out, bf16=bf16)

    def debias_net_user_value(
        self, net_user_value: torch.Tensor, position: torch.Tensor, user_embedding: torch.Tensor
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Identity hook (ref :251-277); subclasses return (debiased value, extra loss)."""
        return net_user_value, 0

    def compute_training_loss(
        self,
        user_embedding: torch.Tensor,  # [B, DI]
        item_embeddings: torch.Tensor,  # [B, DI]
        position: torch.Tensor,  # [B]
        labels: torch.Tensor,  # [B, T]
    ) -> torch.Tensor:
        """In-batch softmax loss weighted by normalised net user value (ref :279-347).
        The [B, B] logits are never materialised."""
        if self._sharded():
            return self._sharded_training_loss(user_embedding, item_embeddings, position, labels)
        hook_is_identity = type(self).debias_net_user_value is TwoTowerBaseRetrieval.debias_net_user_value
        T = self.user_value_weights.numel()
        B = user_embedding.shape[0]
        # train.py's [B] labels (ref:train/train.py:53-55,78): labels * weights sums to ONE scalar, which clamp and
        # / max turn into exactly 1.0 -- the plain mean of the row losses (labels = None below)
        plain_mean = labels.dim() == 1 and labels.shape[0] == B and T in (1, labels.shape[0])
        weighted = labels.dim() == 2 and labels.shape[1] == T and labels.shape[0] == B
        if hook_is_identity and ops.labels_fusable(labels) and (weighted or plain_mean):
            lab = labels if weighted else None
            if ops.fused_loss_supported(user_embedding, item_embeddings, lab, self.user_value_weights):
                # the loss head inside the launch that finishes the logits forward (one op, two launches fewer)
                return ops.InBatchSoftmaxWeightedLoss.apply(user_embedding, item_embeddings, lab, self.user_value_weights)
            row_ce = ops.InBatchSoftmaxCE.apply(user_embedding, item_embeddings, 0)  # [B]
            return ops.WeightedMeanLoss.apply(row_ce, lab, self.user_value_weights)
        row_ce = ops.InBatchSoftmaxCE.apply(user_embedding, item_embeddings, 0)  # [B]
        return self._loss_head(row_ce, labels, position, user_embedding)

    def _loss_head(self, row_ce, labels, position, user_embedding) -> torch.Tensor:
        """General loss head: exactly the reference's expressions on [B]-sized tensors (ref :322-345), so every
        broadcasting quirk (1-D labels collapsing to a scalar weight; debias heads that differentiate
        through the weights) behaves identically.  Subclasses with a fused head override this."""
        net_user_value = torch.sum(labels * self.user_value_weights, dim=-1)
        net_user_value, additional_loss = self.debias_net_user_value(
            net_user_value=net_user_value, position=position, user_embedding=user_embedding
        )
        net_user_value = torch.clamp(net_user_value, min=0.000001)
        net_user_value = net_user_value / torch.max(net_user_value)
        return torch.mean(row_ce * net_user_value) + additional_loss

    # ------------------------------------------------------------------ row-sharded tables (parallel.py)
    def _sharded(self) -> bool:
        return parallel.shard_of(self.item_id_embedding_arch.weight) is not None

    def _sharded_training_loss(self, user_embedding, item_embeddings, position, labels) -> torch.Tensor:
        """The same loss on the CONCATENATED batch of all W ranks (SURVEY.md 8e "parity definition"): every user row is
        scored against all W*B item embeddings (all-gather; its backward is the reduce-scatter of the partial dI), the
        positives sit at column rank*B + i.  Identity hook: the value weights' maximum and the mean are group-wide scalars
        (two all-reduces).  Any other head -- a fused debias head or an overridden hook -- is evaluated on the gathered
        head inputs by the single-device code, identically on every rank, and scaled 1/W in the backward."""
        world, rank = parallel.dist.get_world_size(), parallel.dist.get_rank()
        B = user_embedding.shape[0]
        single_use = getattr(self, "_tt_item_emb_single_use", False)  # set by the un-overridden train_forward
        started, self._tt_item_gather = getattr(self, "_tt_item_gather", None), None
        started = started[1] if (started is not None and started[0] is item_embeddings) else None
        items_all = parallel.AllGatherRows.apply(item_embeddings, "item_emb_allgather", single_use, started)
        row_ce = ops.InBatchSoftmaxCE.apply(user_embedding, items_all, rank * B)  # [B], this rank's users
        hook_is_identity = type(self).debias_net_user_value is TwoTowerBaseRetrieval.debias_net_user_value
        T = self.user_value_weights.numel()
        plain_mean = labels.dim() == 1 and labels.shape[0] == B and T == 1
        weighted = labels.dim() == 2 and labels.shape[1] == T and labels.shape[0] == B
        if hook_is_identity and ops.labels_fusable(labels) and (weighted or plain_mean):
            return parallel.GlobalWeightedMeanLoss.apply(row_ce, labels if weighted else None, self.user_value_weights)
        if world == 1:
            return self._loss_head(row_ce, labels, position, user_embedding)
        loss = self._loss_head(parallel.AllGatherRows.apply(row_ce, "head_row_ce_allgather"),
                               parallel.gather_no_grad(labels), parallel.gather_no_grad(position),
                               parallel.AllGatherRows.apply(user_embedding, "head_user_emb_allgather"))
        return parallel.ReplicatedLoss.apply(loss)

    def _tower_pair_args(self, user_id, user_features, item_id, item_features):
        """(user tower's FusedTower arguments, item tower's) when BOTH towers are this class's own fused form -- no hook
        overridden, no third input block, shapes the pair kernels take -- else None."""
        cls = type(self)
        if not (cls.process_user_features is TwoTowerBaseRetrieval.process_user_features
                and cls.get_user_embedding is TwoTowerBaseRetrieval.get_user_embedding
                and cls.compute_user_embedding is TwoTowerBaseRetrieval.compute_user_embedding
                and cls.compute_item_embeddings is TwoTowerBaseRetrieval.compute_item_embeddings):
            return None
        um, ut, im, it = self.user_features_arch, self.user_tower_arch, self.item_features_arch, self.item_tower_arch
        uw, iw = self.user_id_embedding_arch.weight, self.item_id_embedding_arch.weight
        if not (user_features.is_cuda and ops.fused_tower_supported(uw, user_features, um[0].weight, um[2].weight, ut.weight)):
            return None
        u = (uw, user_id, user_features, um[0].weight, um[0].bias, um[2].weight, um[2].bias, ut.weight, ut.bias)
        i = (iw, item_id, item_features, im[0].weight, im[0].bias, im[2].weight, im[2].bias, it.weight, it.bias)
        return (u, i) if ops.fused_tower_pair_supported(u, i) else None

    def _lookup_plan(self, user_id, user_history, item_id):
        """{table: [id blocks in the order this model's forward looks them up]}."""
        return {self.user_id_embedding_arch.weight: [user_id], self.item_id_embedding_arch.weight: [item_id]}

    def _announce_lookups(self, user_id, user_history, item_id) -> None:
        """DenseExactAdam(overlap_sweep="forward"): tell the optimiser which rows this step reads so its table sweep can
        start before the forward's gathers (optim.py).  Row-sharded tables: first start every lookup's exchange with the
        owning ranks (parallel.begin_lookups) -- the optimiser is then told the rows THIS rank serves."""
        if not user_id.is_cuda:
            return
        plan = self._lookup_plan(user_id, user_history, item_id)
        hold = False
        if self._sharded():
            # thin row blocks and W*B negatives per user: the logits kernels are the step, the sweep starts with the backward
            # one (optim._begin_overlapped); ~6 TB/s of sweep against ~125 TF/s of logits decide which regime this is
            B, tables = user_id.shape[0], [t for t in plan if parallel.shard_of(t) is not None]
            world = parallel.shard_of(tables[0]).world
            sweep_ms = sum(parallel.shard_of(t).n_local * t.shape[1] for t in tables) * 24.0 / 6.0e9
            hold = sweep_ms < 0.75 * (8.0 * B * B * world * self.item_id_embedding_arch.weight.shape[1] / 125.0e9)
            plan = parallel.begin_lookups(plan)
            if hold and torch.is_grad_enabled():
                ops.towers_wgrad_aside()
        ref = getattr(self.item_id_embedding_arch.weight, "_tt_optimizer", None)
        opt = ref() if ref is not None else None
        if opt is not None:
            opt.begin_step(plan, hold_sweep=hold)

    def train_forward(
        self,
        user_id: torch.Tensor,  # [B]
        user_features: torch.Tensor,  # [B, IU]
        user_history: torch.Tensor,  # [B, H]
        item_id: torch.Tensor,  # [B]
        item_features: torch.Tensor,  # [B, II]
        position: torch.Tensor,  # [B]
        labels: torch.Tensor,  # [B, T]
    ) -> torch.Tensor:
        """Scalar training loss with an autograd graph (ref :349-394)."""
        self._announce_lookups(user_id, user_history, item_id)
        if self._sharded():
            # the item tower FIRST (the towers are independent): autograd then runs the user tower's backward before the
            # item tower's, i.e. underneath the reduce-scatter of dI that the item tower's backward has to wait for
            # (deferred reduce-scatter of dI: only while every method that can see `item_embeddings` is this package's own --
            # a subclass's loss that ALSO uses them elsewhere would make autograd sum the in-flight result on the main stream)
            self._tt_item_emb_single_use = (
                type(self).compute_item_embeddings is TwoTowerBaseRetrieval.compute_item_embeddings
                and all(getattr(type(self), m).__module__.startswith(__package__ + ".")
                        for m in ("compute_training_loss", "_loss_head", "_sharded_training_loss")))
            item_embeddings = self.compute_item_embeddings(item_id, item_features)
            # ... and its all-gather (every rank scores against every rank's items) travels underneath the user tower
            self._tt_item_gather = (item_embeddings, parallel.start_all_gather(item_embeddings))
            user_embedding = self.compute_user_embedding(user_id, user_features, user_history)
        else:
            # the towers share nothing until the logits: the item tower's kernels (forward here, backward by autograd's
            # stream rule) go to the third stream and run NEXT TO the user tower's (ops.AuxFork)
            # (the tuned shapes only: there the item tower is one forward and three backward kernels; the generic forms --
            # any width, a dozen library launches -- stay on one stream)
            mlp = self.item_features_arch
            # ... and only this class's own item tower: a subclass's override may read per-batch tensors this method cannot
            # see (and therefore cannot record on the third stream: ops.AuxFork.uses)
            own = type(self).compute_item_embeddings is TwoTowerBaseRetrieval.compute_item_embeddings
            tuned = own and item_features.is_cuda and ops.fused_tower_supported(
                self.item_id_embedding_arch.weight, item_features, mlp[0].weight, mlp[2].weight, self.item_tower_arch.weight)
            # Two ways to run the towers next to each other: two streams (AuxFork) or both towers per launch
            # (ops.FusedTowerPair).  Next to the dense table sweep the fork wins (C2 1.10-1.12 vs 1.16-1.19 ms: 256 tower
            # workgroups at once keep the row plan's big-LDS workgroup waiting); with nothing else on the chip -- the deferred
            # optimiser schedule -- the pair does (P shape 1.07-1.09 vs 1.17-1.19 ms), and on ONE stream (a whole-step
            # hipGraph capture, batches too small for the fork to pay) it is the only way (graphed 1.13 -> 1.02 ms)
            ref = getattr(self.item_id_embedding_arch.weight, "_tt_optimizer", None)
            opt = ref() if ref is not None else None
            no_sweep = opt is not None and getattr(opt, "lazy", False)
            pair = self._tower_pair_args(user_id, user_features, item_id, item_features) if (tuned and no_sweep) else None
            fork = ops.AuxFork(user_id.device, rows=user_id.numel() if (tuned and pair is None) else 0)
            if tuned and pair is None and not fork.on:
                pair = self._tower_pair_args(user_id, user_features, item_id, item_features)
            if pair is not None:
                N.oob.poll(user_id.device)
                user_embedding, item_embeddings = ops.FusedTowerPair.apply(*pair[0], *pair[1])
            else:
                user_embedding = self.compute_user_embedding(user_id, user_features, user_history)
                with fork:
                    fork.uses(item_id, item_features)
                    item_embeddings = self.compute_item_embeddings(item_id, item_features)
                item_embeddings = fork.joined(item_embeddings)
        return self.compute_training_loss(
            user_embedding=user_embedding, item_embeddings=item_embeddings, position=position, labels=labels
        )

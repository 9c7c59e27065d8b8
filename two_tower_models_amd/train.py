#!/usr/bin/env python
"""Training driver: the loop of ref:train/train.py:85-183 on the HIP hot path.

Same CLI flags and defaults (ref:train/train.py:186-255), same ``train_one_epoch(model,
dataloader, optimizer, device)`` / ``main(args)`` entry points, same prints.  Two things
differ, both because the upstream pipeline caps at ~46 K samples/s:
  * ``DummyRecDataset`` keeps its tensors in HBM and ``DeviceBatches`` slices shuffled
    batches on the device (the reference's DataLoader does per-sample __getitem__ on the
    host); the fields and their distributions are the reference's (:47-65);
  * the optimiser is DenseExactAdam (same update as optim.Adam, row-form embedding grads).
``python -m two_tower_models_amd.train --model hist`` selects the history-encoder model
(upstream's script only ever builds the base model).

Multi-GPU: the SAME loop, one process per GPU --

    torchrun --nproc_per_node N --master-addr 127.0.0.1 -m two_tower_models_amd.train --world_size N ...

-- every rank builds the model under ``parallel.row_sharded()`` (each table's rows split N ways, replicated dense
parameters), draws its own ``num_samples / N`` records and runs ``train_one_epoch`` unchanged; ``--batch_size`` is per
rank and the loss is the reference's on the concatenated batch of N x batch_size rows (global in-batch negatives).
"""
from __future__ import annotations

import argparse
import contextlib
import os
import time
from typing import Optional

import torch

from . import BaselineMIPSModule, DenseExactAdam, TwoTowerBaseRetrieval, TwoTowerWithDebiasing, \
    TwoTowerWithUserHistoryEncoder, parallel


class DummyRecDataset:
    """Random records: user_ids, user_features, user_history, item_ids, item_features,
    positions, labels -- ref:train/train.py:20-79, generated once, resident on `device`."""

    def __init__(self, num_samples: int, num_users: int, num_items: int, feature_dim: int,
                 user_history_seqlen: int, device: torch.device = torch.device("cpu"), seed: Optional[int] = None):
        """Fields, dtypes, shapes and distributions of ref:train/train.py:47-65, drawn in the reference's order --
        but ON `device` (the device's own generator; `seed` pins it), so nothing crosses PCIe: a 10 M-sample
        dataset with H = 50 is 4 GB of ids that the reference would draw on the host and copy."""
        self.num_samples, self.num_users, self.num_items, self.feature_dim = num_samples, num_users, num_items, feature_dim
        gen = None
        if seed is not None:
            gen = torch.Generator(device=device).manual_seed(seed)
        kw = dict(device=device, generator=gen)
        self.user_ids = torch.randint(0, num_users, (num_samples,), **kw)      # int64 [n], 0 .. num_users-1
        self.item_ids = torch.randint(0, num_items, (num_samples,), **kw)      # int64 [n], 0 .. num_items-1
        self.labels = torch.randint(0, 2, (num_samples,), **kw).float()        # fp32 [n] (1-D!), 0 or 1
        self.user_features = torch.randn(num_samples, feature_dim, **kw)       # fp32 [n, F], N(0, 1)
        self.user_history = torch.randint(low=0, high=num_items, size=(num_samples, user_history_seqlen), **kw)
        self.item_features = torch.randn(num_samples, feature_dim, **kw)
        self.positions = torch.randint(0, 10, (num_samples,), **kw)            # int64 [n], 0 .. 9

    def __len__(self):
        return self.num_samples

    def fields(self):
        return (self.user_ids, self.user_features, self.user_history, self.item_ids, self.item_features,
                self.positions, self.labels)

    def __getitem__(self, idx):
        return tuple(f[idx] for f in self.fields())


class DeviceBatches:
    """DataLoader(dataset, batch_size, shuffle=True) semantics (ref:train/train.py:176) with
    the permutation and the batch slicing done on the device."""

    def __init__(self, dataset: DummyRecDataset, batch_size: int, shuffle: bool = True):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        dev = self.dataset.user_ids.device
        order = torch.randperm(n, device=dev) if self.shuffle else torch.arange(n, device=dev)
        for lo in range(0, n, self.batch_size):
            idx = order[lo:lo + self.batch_size]
            yield tuple(f[idx] for f in self.dataset.fields())


def train_one_epoch(model, dataloader, optimizer, device):
    """ref:train/train.py:85-135: forward -> zero_grad -> backward -> step per batch; returns
    the mean loss.  The per-step ``.item()`` of upstream is replaced by one at the end.
    Row-sharded model: the NEXT batch's lookups are announced underneath the current step (parallel.plan_ahead), which
    is what lets the host run ahead of the GPU; purely a scheduling hint."""
    model.train()
    total_loss = None
    sharded = parallel.is_sharded(model)
    batches = iter(dataloader)
    batch = next(batches, None)
    while batch is not None:
        user_ids, user_features, user_history, item_ids, item_features, positions, labels = (t.to(device) for t in batch)
        batch_loss = model.train_forward(user_ids, user_features, user_history, item_ids, item_features,
                                         positions, labels)
        batch = next(batches, None)
        if sharded and batch is not None:
            parallel.plan_ahead(model._lookup_plan(batch[0], batch[2], batch[3]))
        optimizer.zero_grad()
        batch_loss.backward()
        optimizer.step()
        total_loss = batch_loss.detach() if total_loss is None else total_loss + batch_loss.detach()
    return float(total_loss.item()) / len(dataloader)


MODELS = {"base": TwoTowerBaseRetrieval, "hist": TwoTowerWithUserHistoryEncoder, "debias": TwoTowerWithDebiasing}


def _init_distributed(world: int):
    """One process per GPU over RCCL (torch.distributed backend "nccl").  TT_DIST_BACKEND=gloo: test hook -- every rank
    on cuda:0, exchanging through gloo (RCCL refuses two ranks on one device)."""
    import torch.distributed as dist
    if "RANK" not in os.environ:
        raise SystemExit(f"--world_size {world}: launch one process per GPU, e.g.\n  torchrun --nproc_per_node {world} "
                         f"--master-addr 127.0.0.1 -m two_tower_models_amd.train --world_size {world} ...")
    if int(os.environ.get("WORLD_SIZE", "1")) != world:
        raise SystemExit(f"--world_size {world} but the launcher started {os.environ.get('WORLD_SIZE')} ranks")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
    backend = os.environ.get("TT_DIST_BACKEND", "nccl")
    device = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0')) if backend == 'nccl' else 0}")
    torch.cuda.set_device(device)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return device, dist.get_rank()


def main(args):
    if not torch.cuda.is_available():
        raise SystemExit("two_tower_models_amd.train needs an MI355X (ROCm) device; there is no CPU path")
    world = int(getattr(args, "world_size", 0) or os.environ.get("WORLD_SIZE", "1"))
    rank = 0
    if world > 1:
        device, rank = _init_distributed(world)
    else:
        device = torch.device("cuda")
    say = print if rank == 0 else (lambda *a, **k: None)
    say(f"Running on device: {device}" + (f" (x{world}, row-sharded tables)" if world > 1 else ""))
    # tables and the (random, upstream-style) MIPS corpus are initialised directly in HBM
    with torch.device(device):
        mips_module = BaselineMIPSModule(corpus_size=args.num_items, embedding_dim=args.embedding_dim)
    kw = dict(num_items=args.num_items_to_return, user_id_hash_size=args.user_id_hash_size,
              user_id_embedding_dim=args.embedding_dim, user_features_size=args.feature_dim,
              item_id_hash_size=args.item_id_hash_size, item_id_embedding_dim=args.embedding_dim,
              item_features_size=args.feature_dim, user_value_weights=[1.0], mips_module=mips_module)
    if args.model != "base":
        kw["user_history_seqlen"] = args.user_history_seqlen
    with torch.device(device), (parallel.row_sharded() if world > 1 else contextlib.nullcontext()):
        model = MODELS[args.model](**kw)
    model = model.to(device)
    if world > 1:
        parallel.shard_model_(model)
    # every rank draws its own records (a different seed per rank; the tables' row blocks are seeded per block by
    # parallel.embedding, so ranks whose default generators start identical still hold different rows)
    dataset = DummyRecDataset(num_samples=max(args.num_samples // world, 1), num_users=args.num_users, num_items=args.num_items,
                              feature_dim=args.feature_dim, user_history_seqlen=args.user_history_seqlen, device=device,
                              seed=(20240 + rank) if world > 1 else None)
    dataloader = DeviceBatches(dataset, batch_size=args.batch_size, shuffle=True)
    # the loop below is exactly train_forward -> zero_grad -> backward -> step, which is what the
    # forward-announced sweep start assumes (optim.py)
    optimizer = DenseExactAdam(model.parameters(), lr=args.learning_rate, overlap_sweep="forward",
                               lazy=getattr(args, "lazy_adam", False))
    stats = []
    for epoch in range(args.num_epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        avg_loss = train_one_epoch(model, dataloader, optimizer, device)  # ends with .item(): the device has drained
        dt = time.perf_counter() - t0
        say(f"Epoch [{epoch + 1}/{args.num_epochs}] - Loss: {avg_loss:.4f}")
        stats.append({"epoch": epoch + 1, "loss": avg_loss, "seconds": dt, "pairs_per_s": len(dataset) * world / dt})
        if getattr(args, "report_throughput", False):
            say(f"  {len(dataset) * world / dt:,.0f} user-item pairs/s end to end (shuffle + batch slicing + step), "
                f"{dt / len(dataloader) * 1e3:.3f} ms/step")
    optimizer.flush()  # deferred schedule: the tables are complete again from here on
    if getattr(args, "dtype", "fp32") == "bf16" and not getattr(args, "retrieve", False):
        model.mips_module.use_bf16_storage()
    if getattr(args, "retrieve", False):
        # serve what was trained: catalogue = the item table's rows; sharded: every rank indexes ITS rows
        n_cat = args.item_id_hash_size
        _, lo, hi = parallel.block_range(n_cat, rank, world) if world > 1 else (n_cat, 0, n_cat)
        feats = torch.randn(hi - lo, args.feature_dim, device=device,
                            generator=torch.Generator(device=device).manual_seed(777 + lo))
        model.eval()
        with torch.no_grad():
            model.index_corpus(torch.arange(lo, hi, device=device), feats, bf16=getattr(args, "dtype", "fp32") == "bf16")
            uid, uf, uh = (t.to(device) for t in next(iter(dataloader))[:3])
            model.num_items = min(model.num_items, model.mips_module.corpus_size)
            top = model(uid, uf, uh)
        say(f"Retrieved top-{top.shape[1]} of {model.mips_module.corpus_size} items for {top.shape[0]} users "
            f"({str(model.mips_module.corpus.dtype).replace('torch.', '')} corpus"
            + (f", {world} row blocks" if world > 1 else "") + f"); first row: {top[0, :5].tolist()}")
        stats.append({"retrieved": tuple(top.shape), "corpus_dtype": str(model.mips_module.corpus.dtype)})
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return stats


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Train two-tower retrieval model")
    for flag, typ, default, hlp in (
        ("--num_users", int, 100, "number of users in the dataset"),
        ("--num_items_to_return", int, 10, "number of items to return in the retrieval task"),
        ("--user_id_hash_size", int, 1024, "embedding table size for user_id"),
        ("--item_id_hash_size", int, 1024, "embedding table size for item_id"),
        ("--user_history_seqlen", int, 10, "length of user history sequence"),
        ("--num_items", int, 200, "number of items in the corpus/dataset"),
        ("--embedding_dim", int, 32, "Dimension of user/item embeddings"),
        ("--feature_dim", int, 8, "Dim of user_features, item_features, etc."),
        ("--num_samples", int, 1000, "Number of samples in the dataset"),
        ("--batch_size", int, 32, "Batch size in training loop"),
        ("--num_epochs", int, 5, "Number of epochs to train"),
        ("--learning_rate", float, 1e-3, "Learning rate"),
    ):
        p.add_argument(flag, type=typ, default=default, help=hlp)
    p.add_argument("--model", choices=sorted(MODELS), default="base", help="model variant (upstream: base only)")
    p.add_argument("--report_throughput", action="store_true", help="print end-to-end pairs/s per epoch")
    p.add_argument("--world_size", type=int, default=0,
                   help="row-shard the tables over this many GPUs (one process per GPU under torchrun; default: WORLD_SIZE "
                        "from the launcher, else 1); --batch_size and the printed loss are per rank / of the global batch")
    p.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32",
                   help="storage of the MIPS corpus the model serves from (BASELINE config 5: bf16).  Training arithmetic is "
                        "IEEE fp32 either way, like the reference's")
    p.add_argument("--retrieve", action="store_true",
                   help="after training: index the catalogue with the trained item tower (model.index_corpus; item r = row r of "
                        "the item table, random features) and run model.forward() on one batch -- row-sharded with --world_size N")
    p.add_argument("--lazy_adam", action="store_true",
                   help="value-exact deferred Adam: replay a row's zero-gradient steps when it is next needed")
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())

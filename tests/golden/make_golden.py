#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/*.npz by IMPORTING the
reference (gauravchak/two_tower_models, mounted read-only at /root/reference)
and running it on deterministic inputs.  Run in the build container only:

    python tests/golden/make_golden.py

The reference itself never travels to the GPU box; these .npz files (inputs,
weights the reference initialised, and the outputs it produced) do.  Each
fixture is listed in SURVEY.md section 8(c) (G1..G7).
"""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import fixture_gen as fg  # noqa: E402
from src.baseline_mips_module import BaselineMIPSModule  # noqa: E402
from src.two_tower_base_retrieval import TwoTowerBaseRetrieval  # noqa: E402
from src.two_tower_with_debiasing import TwoTowerWithDebiasing  # noqa: E402
from src.two_tower_with_position_debiased_weights import TwoTowerWithPositionDebiasedWeights  # noqa: E402
from src.two_tower_with_user_debiased_weights import TwoTowerWithUserDebiasedWeights  # noqa: E402
from src.two_tower_with_user_history_encoder import TwoTowerWithUserHistoryEncoder  # noqa: E402
from src.user_history_encoder import UserHistoryEncoder  # noqa: E402

torch.set_num_threads(8)


def t(x: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(x))


def sd_np(module, prefix="p."):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(module, prefix="g."):
    out = {}
    for k, p in module.named_parameters():
        if p.grad is not None:
            out[prefix + k] = p.grad.detach().cpu().numpy().copy()
    return out


def make_batch(B, n_users, n_items, iu, ii, H, T, seed, labels_1d=False):
    user_id = fg.uniform_ids((B,), n_users, seed + 1)
    item_id = fg.uniform_ids((B,), n_items, seed + 2)
    user_features = fg.gaussianish((B, iu), seed + 3)
    item_features = fg.gaussianish((B, ii), seed + 4)
    user_history = fg.uniform_ids((B, H), n_items, seed + 5)
    position = fg.uniform_ids((B,), 10, seed + 6)
    lab_shape = (B,) if labels_1d else (B, T)
    labels = (fg.hashed_u64(lab_shape, seed + 7) % np.uint64(2)).astype(np.float32)
    return dict(
        user_id=user_id, user_features=user_features, user_history=user_history,
        item_id=item_id, item_features=item_features, position=position, labels=labels,
    )


def batch_tensors(b):
    return [t(b[k]) for k in ("user_id", "user_features", "user_history", "item_id",
                              "item_features", "position", "labels")]


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1e6:.2f} MB  ({len(arrays)} arrays)")


# ---------------------------------------------------------------- G1 / G2
def base_model_case(name, *, n_users, du, iu, n_items, di, ii, T, uvw, B, H, steps=0):
    torch.manual_seed(0)
    mips = BaselineMIPSModule(corpus_size=64, embedding_dim=di)
    model = TwoTowerBaseRetrieval(
        num_items=10, user_id_hash_size=n_users, user_id_embedding_dim=du,
        user_features_size=iu, item_id_hash_size=n_items, item_id_embedding_dim=di,
        item_features_size=ii, user_value_weights=uvw, mips_module=mips,
    )
    out = sd_np(model)
    out["cfg"] = np.array([n_users, du, iu, n_items, di, ii, T, B, H], dtype=np.int64)
    out["uvw"] = np.array(uvw, dtype=np.float32)
    b = make_batch(B, n_users, n_items, iu, ii, H, T, seed=1234)
    out.update({"in." + k: v for k, v in b.items()})
    bt = batch_tensors(b)
    u = model.compute_user_embedding(bt[0], bt[1], bt[2])
    it = model.compute_item_embeddings(bt[3], bt[4])
    scores = u @ it.t()
    ce = torch.nn.functional.cross_entropy(scores, torch.arange(B), reduction="none")
    loss = model.train_forward(*bt)
    model.zero_grad()
    loss.backward()
    out.update(grads_np(model))
    out["user_emb"] = u.detach().numpy()
    out["item_emb"] = it.detach().numpy()
    out["scores"] = scores.detach().numpy()
    out["ce_rows"] = ce.detach().numpy()
    out["loss"] = np.array(loss.item(), dtype=np.float64)
    if T == 1:
        # train.py-style 1-D labels (ref:train/train.py:53-55): weighting degenerates
        b1 = make_batch(B, n_users, n_items, iu, ii, H, T, seed=1234, labels_1d=True)
        loss1 = model.train_forward(*batch_tensors(b1))
        out["in.labels_1d"] = b1["labels"]
        out["loss_labels_1d"] = np.array(loss1.item(), dtype=np.float64)
    if steps:
        # the train.py loop body (ref:train/train.py:112-132) with optim.Adam(lr=1e-3)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        traj = []
        for s in range(steps):
            bs = make_batch(B, n_users, n_items, iu, ii, H, T, seed=5000 + 100 * s)
            out.update({f"step{s}.in." + k: v for k, v in bs.items()})
            l = model.train_forward(*batch_tensors(bs))
            opt.zero_grad()
            l.backward()
            opt.step()
            traj.append(l.item())
        out["adam_losses"] = np.array(traj, dtype=np.float64)
        out.update(sd_np(model, prefix="after."))
    save(name, **out)


# ---------------------------------------------------------------- G3 / G7
def encoder_kat():
    """ref:tests/test_user_history_enc.py:48-124 replayed (seed 42)."""
    out = {}
    x = torch.tensor([[[1, 2], [3, 4], [-1, 0]]], dtype=torch.float32)
    for tag, pe in (("nope", False), ("pe", True)):
        torch.manual_seed(42)
        np.random.seed(42)
        enc = UserHistoryEncoder(2, 3, 1, 1, pe)
        out.update(sd_np(enc, prefix=f"{tag}.p."))
        out[f"{tag}.out"] = enc(x).detach().numpy()
        if pe:
            out["pe.table"] = enc.positional_embeddings.numpy()
    out["x"] = x.numpy()
    out["expected_nope"] = np.array([[[0.8240, 0.7119], [1.0, 2.0]]], dtype=np.float32)
    out["expected_pe"] = np.array([[[1.4978, 1.2425], [1.0, 2.0]]], dtype=np.float32)
    save("g3_encoder_kat", **out)


def encoder_case(name, D, H, heads, L, B, pe):
    torch.manual_seed(0)
    enc = UserHistoryEncoder(D, H, heads, L, pe)
    # biases are zero-initialised upstream; perturb so bias paths are exercised
    with torch.no_grad():
        for k, p in enc.named_parameters():
            if k.endswith("bias"):
                p.copy_(t(fg.gaussianish(tuple(p.shape), 77 + len(k)) * 0.1))
    x = t(fg.gaussianish((B, H, D), 31)).requires_grad_(True)
    cot = t(fg.gaussianish((B, 2, D), 32))
    y = enc(x)
    (y * cot).sum().backward()
    out = sd_np(enc)
    out.update(grads_np(enc))
    out["cfg"] = np.array([D, H, heads, L, B, int(pe)], dtype=np.int64)
    out["x"] = x.detach().numpy()
    out["cot"] = cot.numpy()
    out["y"] = y.detach().numpy()
    out["gx"] = x.grad.numpy()
    if pe:
        out["pe_table"] = enc.positional_embeddings.numpy()
    save(name, **out)


def pe_tables():
    out = {}
    for H, D in ((3, 2), (10, 32), (50, 128), (128, 64), (7, 5)):
        enc = UserHistoryEncoder(D, H, 1, 0, True)
        out[f"pe_{H}_{D}"] = enc.positional_embeddings.numpy()
    save("g7_pe_tables", **out)


# ---------------------------------------------------------------- G4 / G6
def hist_model_case(name, *, n_users, du, iu, n_items, di, ii, T, uvw, B, H, debias=False,
                    corpus=None, topk=10):
    torch.manual_seed(0)
    mips = BaselineMIPSModule(corpus_size=64 if corpus is None else corpus.shape[0], embedding_dim=di)
    if corpus is not None:
        mips.corpus = t(corpus)
    cls = {False: TwoTowerWithUserHistoryEncoder, True: TwoTowerWithDebiasing, "position": TwoTowerWithPositionDebiasedWeights,
           "user": TwoTowerWithUserDebiasedWeights}[debias]
    model = cls(
        num_items=topk, user_id_hash_size=n_users, user_id_embedding_dim=du,
        user_features_size=iu, user_history_seqlen=H, item_id_hash_size=n_items,
        item_id_embedding_dim=di, item_features_size=ii, user_value_weights=uvw,
        mips_module=mips,
    )
    out = sd_np(model)
    out["cfg"] = np.array([n_users, du, iu, n_items, di, ii, T, B, H], dtype=np.int64)
    out["uvw"] = np.array(uvw, dtype=np.float32)
    b = make_batch(B, n_users, n_items, iu, ii, H, T, seed=4321)
    out.update({"in." + k: v for k, v in b.items()})
    bt = batch_tensors(b)
    u = model.compute_user_embedding(bt[0], bt[1], bt[2])
    it = model.compute_item_embeddings(bt[3], bt[4])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # upstream's [B,1] vs [B] mse_loss broadcast warning
        loss = model.train_forward(*bt)
    model.zero_grad()
    loss.backward()
    out.update(grads_np(model))
    out["user_emb"] = u.detach().numpy()
    out["item_emb"] = it.detach().numpy()
    out["loss"] = np.array(loss.item(), dtype=np.float64)
    out["pe_table"] = model.user_history_encoder.positional_embeddings.numpy()
    if corpus is not None:
        with torch.no_grad():
            top = model(bt[0], bt[1], bt[2])
            sc = (u.detach() @ t(corpus).t())
        out["top_items"] = top.numpy()
        srt = torch.sort(sc, dim=1, descending=True).values
        out["topk_gap_min"] = (srt[:, :topk] - srt[:, 1 : topk + 1]).min(dim=1).values.numpy()
    save(name, **out)


# ---------------------------------------------------------------- G5
def mips_cases():
    out = {}
    for C in (4096, 65536):
        corpus = fg.exact_mips_corpus(C, 128)
        q = fg.exact_mips_queries(16, 128)
        m = BaselineMIPSModule(corpus_size=C, embedding_dim=128)
        m.corpus = t(corpus)
        for K in (10, 1000):
            idx, sc, emb = m(query_embedding=t(q), num_items=K)
            assert emb.shape == (16, K, 128)
            out[f"exact_C{C}_K{K}.idx"] = idx.numpy().astype(np.int32)
            out[f"exact_C{C}_K{K}.scores"] = sc.numpy()
        s = t(q) @ t(corpus).t()
        assert all(len(torch.unique(row)) == C for row in s), "scores must be pairwise distinct"
    # random (gaussian-ish) corpus, values already bf16-representable so the
    # fp32 and bf16-storage paths see the same numbers
    C, D, B = 4096, 128, 16
    corpus = fg.bf16_round(fg.gaussianish((C, D), 901))
    q = fg.bf16_round(fg.gaussianish((B, D), 902))
    m = BaselineMIPSModule(corpus_size=C, embedding_dim=D)
    m.corpus = t(corpus)
    for K in (10, 100):
        idx, sc, _ = m(query_embedding=t(q), num_items=K)
        out[f"rand_C{C}_K{K}.idx"] = idx.numpy().astype(np.int32)
        out[f"rand_C{C}_K{K}.scores"] = sc.numpy()
    s64 = torch.sort(t(q).double() @ t(corpus).double().t(), dim=1, descending=True).values
    out["rand_gap_min_K10"] = (s64[:, :10] - s64[:, 1:11]).min(dim=1).values.numpy()
    out["rand_gap_min_K100"] = (s64[:, :100] - s64[:, 1:101]).min(dim=1).values.numpy()
    save("g5_mips", **out)


def mips_wide_case():
    """D = 256 (> 128: the generic-width path of the build): exact-arithmetic corpus."""
    out = {}
    C, D = 4096, 256
    corpus = fg.exact_mips_corpus(C, D)
    q = fg.exact_mips_queries(16, D)
    m = BaselineMIPSModule(corpus_size=C, embedding_dim=D)
    m.corpus = t(corpus)
    for K in (10, 300):
        idx, sc, emb = m(query_embedding=t(q), num_items=K)
        assert emb.shape == (16, K, D)
        out[f"exact_C{C}_K{K}.idx"] = idx.numpy().astype(np.int32)
        out[f"exact_C{C}_K{K}.scores"] = sc.numpy()
    s = t(q) @ t(corpus).t()
    assert all(len(torch.unique(row)) == C for row in s), "scores must be pairwise distinct"
    save("g5_mips_d256", **out)


if __name__ == "__main__":
    if "--only-d256" in sys.argv:  # the round-2 additions alone (the others regenerate bit-identically anyway)
        base_model_case("g2_base_d256", n_users=96, du=256, iu=8, n_items=128, di=256, ii=8,
                        T=1, uvw=[1.0], B=160, H=4)
        mips_wide_case()
        sys.exit(0)
    if "--only-g8" in sys.argv:  # the round-3 additions alone
        for kind in ("position", "user"):
            hist_model_case(f"g8_debias_{kind}", n_users=120, du=64, iu=8, n_items=140, di=64, ii=8,
                            T=2, uvw=[0.6, 0.4], B=48, H=12, debias=kind)
        sys.exit(0)
    base_model_case("g1_base_tiny", n_users=100, du=50, iu=20, n_items=150, di=40, ii=30,
                    T=3, uvw=[0.1, 0.2, 0.3], B=32, H=8)
    base_model_case("g2_base_aligned", n_users=256, du=128, iu=8, n_items=256, di=128, ii=8,
                    T=1, uvw=[1.0], B=256, H=4, steps=3)
    encoder_kat()
    encoder_case("g3_encoder_d128", D=128, H=50, heads=4, L=3, B=8, pe=True)
    encoder_case("g3_encoder_d128_nope", D=128, H=50, heads=4, L=1, B=4, pe=False)
    encoder_case("g3_encoder_odd", D=40, H=7, heads=4, L=2, B=5, pe=True)
    pe_tables()
    hist_model_case("g4_hist_d128", n_users=256, du=128, iu=8, n_items=256, di=128, ii=8,
                    T=1, uvw=[1.0], B=64, H=50)
    hist_model_case("g4_hist_tiny", n_users=100, du=50, iu=20, n_items=150, di=40, ii=30,
                    T=3, uvw=[0.1, 0.2, 0.3], B=32, H=16)
    mips_cases()
    corpus = fg.bf16_round(fg.gaussianish((4096, 128), 901))
    hist_model_case("g6_debias_d128", n_users=256, du=128, iu=8, n_items=256, di=128, ii=8,
                    T=1, uvw=[1.0], B=64, H=50, debias=True, corpus=corpus, topk=10)
    # the two single-term debias heads (SURVEY 8f item 2's siblings): B = 48, positions 0..9, T = 2
    for kind in ("position", "user"):
        hist_model_case(f"g8_debias_{kind}", n_users=120, du=64, iu=8, n_items=140, di=64, ii=8,
                        T=2, uvw=[0.6, 0.4], B=48, H=12, debias=kind)
    base_model_case("g2_base_d256", n_users=96, du=256, iu=8, n_items=128, di=256, ii=8,
                    T=1, uvw=[1.0], B=160, H=4)
    mips_wide_case()

"""MIPS top-K parity on MI355X (`pytest -m gpu`): bit-exact indices AND scores on the
exact-arithmetic corpus the reference ranked (tests/golden/g5_mips.npz), fp32 and bf16
storage; random corpus vs the reference with sub-margin-swap accounting; ties, ragged
sizes, K == C; the reference's own unit-test shapes."""
import numpy as np
import pytest
import torch

import fixture_gen as fg
from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.fixture(scope="module")
def A():
    import two_tower_models_amd as A
    return A


def module_with(A, corpus, bf16=False):
    m = A.BaselineMIPSModule(corpus_size=corpus.shape[0], embedding_dim=corpus.shape[1])
    m.corpus = corpus.clone()
    m = m.to(DEV)
    return m.use_bf16_storage() if bf16 else m


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("C", [4096, 65536])
@pytest.mark.parametrize("K", [10, 1000])
def test_exact_corpus_bit_exact_vs_reference(A, golden, C, K, bf16):
    g = golden("g5_mips")
    m = module_with(A, T(fg.exact_mips_corpus(C, 128)), bf16)
    q = T(fg.exact_mips_queries(16, 128)).to(DEV)
    idx, sc, emb = m(query_embedding=q, num_items=K)
    assert idx.dtype == torch.int64 and idx.shape == (16, K) and sc.shape == (16, K) and emb.shape == (16, K, 128)
    assert np.array_equal(idx.cpu().numpy(), g[f"exact_C{C}_K{K}.idx"].astype(np.int64))
    assert np.array_equal(sc.cpu().numpy(), g[f"exact_C{C}_K{K}.scores"])
    assert torch.equal(emb.cpu(), T(fg.exact_mips_corpus(C, 128))[idx.cpu()])


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("K", [10, 300])
def test_exact_corpus_d256_bit_exact_vs_reference(A, golden, K, bf16):
    """D = 256 (> 128: the generic-width form, library GEMM + group epilogues) against what the reference ranked."""
    g = golden("g5_mips_d256")
    corpus = T(fg.exact_mips_corpus(4096, 256))
    m = module_with(A, corpus, bf16)
    idx, sc, emb = m(query_embedding=T(fg.exact_mips_queries(16, 256)).to(DEV), num_items=K)
    assert np.array_equal(idx.cpu().numpy(), g[f"exact_C4096_K{K}.idx"].astype(np.int64))
    assert np.array_equal(sc.cpu().numpy(), g[f"exact_C4096_K{K}.scores"])
    assert torch.equal(emb.cpu(), corpus[idx.cpu()])


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("B,C,D,K", [(40, 70_000, 200, 50), (3, 129, 130, 129), (1100, 3000, 384, 7)])
def test_wide_embeddings_match_oracle_order(A, B, C, D, K, bf16):
    """Widths above 128, ragged, more than one corpus slab (C > 65 536), more than one query batch (B > 1024),
    K == C (every item a candidate): exact-arithmetic data, so indices and scores equal the oracle's bit for bit."""
    from oracle import cpu_ref as R
    corpus, q = T(fg.exact_mips_corpus(C, D, seed=5)), T(fg.exact_mips_queries(B, D, seed=6))
    m = module_with(A, corpus, bf16)
    idx, sc = m.search(q.to(DEV), K)
    want_idx, want_sc, _ = R.mips_topk(q, corpus, K)
    assert torch.equal(sc.cpu(), want_sc)
    assert torch.equal(idx.cpu(), want_idx)


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("K", [10, 100])
def test_random_corpus_vs_reference(A, golden, K, bf16):
    g = golden("g5_mips")
    m = module_with(A, T(fg.bf16_round(fg.gaussianish((4096, 128), 901))), bf16)
    q = T(fg.bf16_round(fg.gaussianish((16, 128), 902))).to(DEV)
    idx, sc = m.search(q, K)
    want = g[f"rand_C4096_K{K}.idx"].astype(np.int64)
    gate = g[f"rand_gap_min_K{K}"] > 1e-4
    got = idx.cpu().numpy()
    assert np.array_equal(got[gate], want[gate])
    for r in np.nonzero(~gate)[0]:  # only sub-margin neighbour swaps may differ
        assert len(set(got[r]) ^ set(want[r])) <= 2
    assert np.allclose(sc.cpu().numpy(), g[f"rand_C4096_K{K}.scores"], atol=1e-4)


@pytest.mark.parametrize("B,C,D,K", [(32, 100, 50, 10), (32, 1001, 40, 10), (3, 129, 8, 129), (200, 5000, 64, 37),
                                     (1, 64, 128, 1), (130, 300, 2, 300), (50, 3000, 64, 20), (64, 5000, 128, 64),
                                     (33, 700, 32, 5)])
def test_ragged_shapes_match_oracle_order(A, B, C, D, K):
    """Includes the reference unit-test shapes (ref:tests/test_baseline_mips_module.py:16-36,
    ref:tests/test_two_tower_base_retrieval.py:51-71).  Integer-valued data: exact scores,
    many ties -> checks the (score desc, index asc) order bit for bit."""
    corpus = T((fg.hashed_u64((C, D), 5) % np.uint64(7)).astype(np.float32) - 3.0)
    q = T((fg.hashed_u64((B, D), 6) % np.uint64(5)).astype(np.float32) - 2.0)
    want_idx, want_sc, _ = R.mips_topk(q, corpus, K)
    for bf16 in (False, True):
        m = module_with(A, corpus, bf16)
        idx, sc, emb = m(query_embedding=q.to(DEV), num_items=K)
        assert torch.equal(idx.cpu(), want_idx), bf16
        assert torch.equal(sc.cpu(), want_sc)
        assert ((idx >= 0) & (idx < C)).all()
        assert torch.equal(emb.cpu(), corpus[want_idx])


@pytest.mark.parametrize("B,C,K", [(300, 256 * 23 + 100, 40), (700, 256 * 30 + 129, 45), (1024, 256 * 40 + 255, 33),
                                   (100, 128 * 41, 40)])
def test_bf16_128_row_groups_ties_and_tails(A, B, C, K):
    """bf16 storage, more than 64 queries, D = 128: pass 1 works on 256-row chunks (128-row groups), two / four query
    fragments per wave.  Corpus sizes that end inside the first / second half of a 256-row chunk (an empty last group,
    a one-row last group); integer-valued data = exact scores with many ties, so the (score desc, index asc) order is
    checked bit for bit, including which row of a 4-row quad pass 2 reports."""
    D = 128
    corpus = T((fg.hashed_u64((C, D), 15) % np.uint64(5)).astype(np.float32) - 2.0)
    q = T((fg.hashed_u64((B, D), 16) % np.uint64(3)).astype(np.float32) - 1.0)
    want_idx, want_sc, _ = R.mips_topk(q, corpus, K)
    m = module_with(A, corpus, bf16=True)
    idx, sc = m.search(q.to(DEV), K)
    assert torch.equal(idx.cpu(), want_idx)
    assert torch.equal(sc.cpu(), want_sc)


def test_group_size_and_pass2_switches_give_identical_results(tmp_path):
    """The A/B switches of the bf16 path (64- vs 128-row groups, every selected group re-scored instead of the
    best-quad shortcut, dense instead of sparse pass 2) are read once per process: run the same search in a child
    process per setting and compare indices and scores bit for bit with the default."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, 'tests/golden'); sys.path.insert(0, '.')\n"
        "import fixture_gen as fg, two_tower_models_amd as A\n"
        "C, D, B, K = 256 * 37 + 77, 128, 300, 60\n"
        "corpus = torch.from_numpy(fg.bf16_round(fg.gaussianish((C, D), 31)))\n"
        "q = torch.from_numpy(fg.bf16_round(fg.gaussianish((B, D), 32)))\n"
        "m = A.BaselineMIPSModule(corpus_size=C, embedding_dim=D); m.corpus = corpus; m = m.to('cuda:0').use_bf16_storage()\n"
        "idx, sc = m.search(q.to('cuda:0'), K)\n"
        "np.savez(sys.argv[1], idx=idx.cpu().numpy(), sc=sc.cpu().numpy())\n")
    outs = {}
    for name, var in (("default", None), ("g64", "TT_MIPS_NO_G128"), ("no_top2", "TT_MIPS_NO_TOP2"),
                      ("dense", "TT_MIPS_NO_SPARSE")):
        env = dict(os.environ)
        if var:
            env[var] = "1"
        out = str(tmp_path / f"{name}.npz")
        r = subprocess.run([sys.executable, "-c", script, out], cwd=root, env=env, capture_output=True, text=True,
                           timeout=600, stdin=subprocess.DEVNULL)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    for name in ("g64", "no_top2", "dense"):
        assert np.array_equal(outs[name]["idx"], outs["default"]["idx"]), name
        assert np.array_equal(outs[name]["sc"], outs["default"]["sc"]), name


def test_all_equal_scores_returns_lowest_indices(A):
    corpus = torch.ones(1000, 16)
    m = module_with(A, corpus)
    idx, sc = m.search(torch.ones(5, 16, device=DEV), 50)
    assert torch.equal(idx.cpu(), torch.arange(50).expand(5, 50))
    assert float(sc.min()) == 16.0 == float(sc.max())


def test_sorted_adversarial_corpus_hits_candidate_bound(A):
    """Scores strictly decreasing along the corpus: the K best groups are the first K
    groups, all 64*K of their items pass the threshold (the proven worst case)."""
    C, D, K = 20000, 8, 100
    corpus = torch.zeros(C, D)
    corpus[:, 0] = torch.arange(C, 0, -1).float()
    m = module_with(A, corpus)
    q = torch.zeros(2, D)
    q[:, 0] = 1.0
    idx, sc = m.search(q.to(DEV), K)
    assert torch.equal(idx.cpu(), torch.arange(K).expand(2, K))


def test_large_random_against_oracle_sets(A):
    C, D, B, K = 200_000, 128, 24, 1000
    corpus = T(fg.bf16_round(fg.gaussianish((C, D), 77)))
    q = T(fg.bf16_round(fg.gaussianish((B, D), 78)))
    want_idx, want_sc, _ = R.mips_topk(q, corpus, K)
    for bf16 in (False, True):
        m = module_with(A, corpus, bf16)
        idx, sc = m.search(q.to(DEV), K)
        got, want = idx.cpu().numpy(), want_idx.numpy()
        assert np.allclose(sc.cpu().numpy(), want_sc.numpy(), atol=2e-4)
        assert (np.diff(sc.cpu().numpy(), axis=1) <= 0).all()  # sorted descending
        exact_rows = sum(np.array_equal(got[r], want[r]) for r in range(B))
        for r in range(B):  # any disagreement is a swap of near-tied neighbours, not a wrong item
            assert len(set(got[r]) ^ set(want[r])) <= 4
        assert exact_rows >= B // 2


@pytest.mark.parametrize("bf16", [False, True])
def test_many_query_batches_reuse_the_workspace(A, bf16):
    """B = 2500 queries = three internal batches of <= 1024 sharing one workspace (group maxima,
    selected-group lists, select survivors, candidates): every batch must start from clean counters.
    Exact-arithmetic corpus -> indices and scores bit-exact against the oracle order."""
    C, D, B, K = 20_000, 64, 2500, 50
    corpus = T(fg.exact_mips_corpus(C, D))
    q = T(fg.exact_mips_queries(B, D))
    m = module_with(A, corpus, bf16)
    idx, sc = m.search(q.to(DEV), K)
    want_idx, want_sc, _ = R.mips_topk(q, corpus, K)
    assert torch.equal(idx.cpu(), want_idx) and torch.equal(sc.cpu(), want_sc)
    idx2, sc2 = m.search(q.to(DEV), K)  # and the next call as well
    assert torch.equal(idx2, idx) and torch.equal(sc2, sc)


def test_debias_model_forward_topk_vs_reference(A, golden):
    """BASELINE config 5's model: TwoTowerWithDebiasing.forward -> MIPS ids."""
    from test_gpu_models import make_model, batch_of
    g = golden("g6_debias_d128")
    corpus = T(fg.bf16_round(fg.gaussianish((4096, 128), 901)))
    for bf16 in (False, True):
        model = make_model("debias", g, corpus=corpus.clone())
        if bf16:
            model.mips_module.use_bf16_storage()
        b = batch_of(g)
        top = model(b[0], b[1], b[2])
        assert top.shape == (64, 10) and top.dtype == torch.int64
        gate = g["topk_gap_min"] > (2e-2 if bf16 else 1e-3)  # bf16 rounds the QUERY too
        assert gate.sum() >= 16
        assert np.array_equal(top.cpu().numpy()[gate], g["top_items"][gate])

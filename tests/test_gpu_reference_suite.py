"""The reference's own unit tests (ref:tests/*.py, 10 tests), re-run against the MI355X
modules: same constructor arguments, same shapes, same assertions.  Only the imports and the
device differ.  (The seed-42 known-answer tests live in test_gpu_models.py.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def A():
    import two_tower_models_amd as A
    return A


# ---- ref:tests/test_baseline_mips_module.py
def test_mips_output_shapes_and_values(A):
    corpus_size, embedding_dim, num_items, batch_size = 100, 50, 10, 32
    module = A.BaselineMIPSModule(corpus_size, embedding_dim).to(DEV)
    query = torch.randn(batch_size, embedding_dim, device=DEV)
    mips_ids, mips_scores, mips_embeddings = module(query, num_items)
    assert mips_ids.shape == (batch_size, num_items)
    assert mips_scores.shape == (batch_size, num_items)
    assert mips_embeddings.shape == (batch_size, num_items, embedding_dim)
    assert torch.all(mips_ids >= 0) and torch.all(mips_ids < corpus_size)
    assert mips_ids.dtype == torch.int64
    assert torch.all(mips_scores[:, :-1] >= mips_scores[:, 1:])  # torch.topk returns sorted scores


def _dims():
    return dict(num_items=10, user_id_hash_size=100, user_id_embedding_dim=50, user_features_size=20,
                item_id_hash_size=150, item_id_embedding_dim=40, item_features_size=30,
                user_value_weights=[0.1, 0.2, 0.3])


def _inputs(B=32, H=128):
    return (torch.randint(0, 100, (B,), device=DEV), torch.randn(B, 20, device=DEV),
            torch.randint(0, 150, (B, H), device=DEV), torch.randint(0, 150, (B,), device=DEV),
            torch.randn(B, 30, device=DEV), torch.randint(0, 10, (B,), device=DEV),
            torch.randint(0, 2, (B, 3), device=DEV).float())


# ---- ref:tests/test_two_tower_base_retrieval.py
def test_base_forward_pass_and_train_forward(A):
    mips = A.BaselineMIPSModule(corpus_size=1001, embedding_dim=40)
    model = A.TwoTowerBaseRetrieval(mips_module=mips, **_dims()).to(DEV)
    uid, uf, uh, iid, itf, pos, labels = _inputs()
    out = model(uid, uf, uh)
    assert out.shape == (32, 10)
    assert torch.all(out >= 0) and torch.all(out < mips.corpus_size)
    loss = model.train_forward(uid, uf, uh, iid, itf, pos, labels)
    assert isinstance(loss.item(), float)


# ---- ref:tests/test_two_tower_user_hist.py
def test_history_model_forward_pass(A):
    mips = A.BaselineMIPSModule(corpus_size=1001, embedding_dim=40)
    model = A.TwoTowerWithUserHistoryEncoder(mips_module=mips, user_history_seqlen=128, **_dims()).to(DEV)
    uid, uf, uh, *_ = _inputs()
    out = model(uid, uf, uh)
    assert out.shape == (32, 10)
    assert torch.all(out >= 0) and torch.all(out < mips.corpus_size)


# ---- ref:tests/test_two_tower_user_hist_position_debias.py (combined debias variant here)
def test_debias_model_forward_and_train_forward(A):
    import warnings
    mips = A.BaselineMIPSModule(corpus_size=1001, embedding_dim=40)
    model = A.TwoTowerWithDebiasing(mips_module=mips, user_history_seqlen=128, **_dims()).to(DEV)
    uid, uf, uh, iid, itf, pos, labels = _inputs()
    out = model(uid, uf, uh)
    assert out.shape == (32, 10) and torch.all(out >= 0) and torch.all(out < mips.corpus_size)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loss = model.train_forward(uid, uf, uh, iid, itf, pos, labels)
    assert isinstance(loss.item(), float)
    loss.backward()
    assert all(p.grad is not None for p in model.parameters())


# ---- ref:tests/test_user_history_enc.py::test_forward
def test_encoder_forward_shape(A):
    model = A.UserHistoryEncoder(item_id_embedding_dim=64, history_len=128, num_attention_heads=4,
                                 num_attention_layers=12, use_positional_encoding=True).to(DEV)
    output = model(torch.randn(32, 128, 64, device=DEV))
    assert output.shape == (32, model.get_output_dim() / 64, 64)


# ---- ref:train/train.py: the script runs end to end and the loss goes down
def test_train_script_default_flags(A, capsys):
    from two_tower_models_amd import train
    torch.manual_seed(0)
    train.main(train.build_parser().parse_args(["--num_epochs", "4"]))
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("Epoch")]
    assert len(lines) == 4
    first, last = (float(l.rsplit(" ", 1)[1]) for l in (lines[0], lines[-1]))
    assert last < first


def test_device_dataset_fields_match_the_reference_dataset(A):
    """DummyRecDataset generated ON the device: the seven fields of ref:train/train.py:47-65 with the reference's
    shapes, dtypes (1-D float labels!), value ranges and distributions (uniform ids: mean (n-1)/2, variance
    (n^2-1)/12; N(0,1) features; Bernoulli(1/2) labels; positions uniform on 0..9), plus DataLoader(shuffle=True)
    semantics of the on-device batcher: every sample exactly once per epoch, ragged last batch kept."""
    from two_tower_models_amd import train
    n, NU, NI, F, H = 200_000, 1000, 50_000, 8, 10
    ds = train.DummyRecDataset(n, NU, NI, F, H, device=torch.device(DEV), seed=3)
    uid, uf, hist, iid, itf, pos, lab = ds.fields()
    assert all(t.is_cuda for t in ds.fields()) and len(ds) == n
    assert (uid.dtype, iid.dtype, hist.dtype, pos.dtype) == (torch.int64,) * 4
    assert (uf.dtype, itf.dtype, lab.dtype) == (torch.float32,) * 3
    assert uid.shape == (n,) and iid.shape == (n,) and lab.shape == (n,) and pos.shape == (n,)
    assert uf.shape == (n, F) and itf.shape == (n, F) and hist.shape == (n, H)
    for t, hi in ((uid, NU), (iid, NI), (hist, NI), (pos, 10)):
        assert int(t.min()) >= 0 and int(t.max()) < hi
        m, v = t.double().mean().item(), t.double().var().item()
        assert abs(m - (hi - 1) / 2) < 6 * ((hi * hi - 1) / 12 / t.numel()) ** 0.5
        assert abs(v / ((hi * hi - 1) / 12) - 1) < 0.03
    assert int(uid.max()) == NU - 1 and int(uid.min()) == 0 and len(torch.unique(pos)) == 10
    assert set(torch.unique(lab).tolist()) == {0.0, 1.0} and abs(lab.mean().item() - 0.5) < 0.01
    for t in (uf, itf):
        assert abs(t.mean().item()) < 0.01 and abs(t.std().item() - 1) < 0.01
        assert abs((t ** 4).mean().item() - 3.0) < 0.15  # normal kurtosis
    same = train.DummyRecDataset(n, NU, NI, F, H, device=torch.device(DEV), seed=3)
    assert all(torch.equal(a, b) for a, b in zip(ds.fields(), same.fields()))
    # __getitem__ returns the reference's 7-tuple
    rec = ds[5]
    assert len(rec) == 7 and rec[2].shape == (H,) and rec[6].dim() == 0
    # one epoch of the device batcher = a permutation of the dataset
    tag = torch.arange(n, device=DEV)
    ds.positions = tag  # reuse a field as a sample tag
    seen = torch.cat([b[5] for b in train.DeviceBatches(ds, 8192, shuffle=True)])
    assert seen.numel() == n and torch.equal(torch.sort(seen).values, tag) and not torch.equal(seen, tag)
    sizes = [b[0].shape[0] for b in train.DeviceBatches(ds, 8192, shuffle=False)]
    assert sizes == [8192] * (n // 8192) + [n % 8192]


def test_index_corpus_serves_the_item_tower(A):
    """index_corpus installs item-tower outputs as the MIPS corpus: forward() then returns, for every
    user, the ids whose item embeddings have the largest inner products with the user embedding."""
    torch.manual_seed(5)
    mips = A.BaselineMIPSModule(corpus_size=8, embedding_dim=32)
    model = A.TwoTowerBaseRetrieval(num_items=7, user_id_hash_size=50, user_id_embedding_dim=32, user_features_size=6,
                                    item_id_hash_size=300, item_id_embedding_dim=32, item_features_size=5,
                                    user_value_weights=[1.0], mips_module=mips).to(DEV)
    C = 300
    item_id = torch.arange(C, device=DEV)
    item_features = torch.randn(C, 5, device=DEV)
    model.index_corpus(item_id, item_features, chunk=128)
    assert mips.corpus_size == C and tuple(mips.corpus.shape) == (C, 32)
    uid, uf = torch.randint(0, 50, (9,), device=DEV), torch.randn(9, 6, device=DEV)
    got = model(uid, uf, torch.zeros(9, 3, dtype=torch.long, device=DEV))
    with torch.no_grad():
        u = model.compute_user_embedding(uid, uf, torch.zeros(9, 3, dtype=torch.long, device=DEV))
        want = torch.topk(u.cpu().double() @ mips.corpus.cpu().double().t(), 7, dim=1).indices
    assert torch.equal(got.cpu().sort(dim=1).values, want.sort(dim=1).values)

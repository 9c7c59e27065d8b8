"""ShardedTrainer with the PRODUCT backend (HipBackend, libtt_hotpath.so) on one MI355X:
world_size 1 over RCCL must reproduce the oracle's train steps (loss trajectory, tables,
dense parameters).  The W > 1 routing logic is covered on CPU by tests/test_sharded_cpu.py;
this test covers the HIP arithmetic behind the same interface."""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("cfg", [dict(n_users=300, n_items=500, D=128, F=8, B=256, H=2),
                                 dict(n_users=53, n_items=71, D=40, F=20, B=32, H=2)])
def test_world1_hip_backend_matches_oracle(cfg):
    import torch.distributed as dist
    from oracle import cpu_ref as R
    from test_sharded_cpu import _dense_init
    from two_tower_models_amd import sharded
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    try:
        dense = _dense_init(cfg)
        tr = sharded.ShardedTrainer(cfg, dev, negatives="global", user_value_weights=(0.7,), dense_init=dense)
        assert isinstance(tr.be, sharded.HipBackend)
        params = dict(dense)
        params["user_id_embedding_arch.weight"] = tr.users.weight.cpu().clone()
        params["item_id_embedding_arch.weight"] = tr.items.weight.cpu().clone()
        state = R.AdamState(params)
        batches = tr.make_batches(3, seed=7)
        got, want = [], []
        for b in batches:
            got.append(float(tr.step(b)))
            want.append(R.train_step(params, state, [t.cpu() for t in b], torch.tensor([0.7])))
        assert np.allclose(got, want, atol=1e-4), (got, want)
        assert torch.allclose(tr.users.weight.cpu(), params["user_id_embedding_arch.weight"], atol=5e-6)
        assert torch.allclose(tr.items.weight.cpu(), params["item_id_embedding_arch.weight"], atol=5e-6)
        for k, v in tr.params.items():
            noise_only = k in ("item_tower_arch.bias", "item_features_arch.2.bias")
            assert torch.allclose(v.cpu(), params[k], atol=6.6e-3 if noise_only else 5e-6), k
    finally:
        dist.destroy_process_group()


def test_mips_merge_kernel_and_world1_sharded_mips():
    """tt_mips_merge on hand-made shard lists (ties across shards, padding) and ShardedMIPS with
    the product backend at world size 1."""
    import torch.distributed as dist
    import fixture_gen as fg
    from oracle import cpu_ref as R
    from two_tower_models_amd import ops, sharded
    dev = torch.device("cuda:0")
    # 3 "shards" x K=4 candidates for 2 queries; equal scores across shards must order by index
    sc = torch.tensor([[5., 3., 3., 1., 5., 4., 3., 0., 9., 3., 2., 0.],
                       [1., 1., 1., 1., 1., 1., 1., 0., 1., 1., 0., 0.]])
    ix = torch.tensor([[10, 11, 12, 13, 20, 21, 22, -1, 3, 30, 31, -1],
                       [7, 8, 9, 10, 1, 2, 3, -1, 4, 5, -1, -1]])
    oi, os_ = ops.mips_merge(sc.to(dev), ix.to(dev), 5)
    assert oi.cpu().tolist() == [[3, 10, 20, 21, 11], [1, 2, 3, 4, 5]]
    assert os_.cpu().tolist() == [[9., 5., 5., 4., 3.], [1., 1., 1., 1., 1.]]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    try:
        corpus = torch.from_numpy(fg.exact_mips_corpus(5000, 64))
        q = torch.from_numpy(fg.exact_mips_queries(9, 64))
        m = sharded.ShardedMIPS(corpus.to(dev), 0)
        idx, s = m.search(q.to(dev), 100)
        want_idx, want_sc, _ = R.mips_topk(q, corpus, 100)
        assert torch.equal(idx.cpu(), want_idx) and torch.equal(s.cpu(), want_sc)
    finally:
        dist.destroy_process_group()

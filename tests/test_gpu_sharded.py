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

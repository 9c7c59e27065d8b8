"""TEST DOUBLES for the kernels behind two_tower_models_amd.parallel: a torch (CPU) restatement of the four routing
kernels (csrc/route.hip + the owner-side gather) and of the two MIPS entry points, so that the exchange logic of
parallel.py -- who is sent what, in which slot, what comes back, what the owner's optimiser is told -- can run under gloo
on a box without a GPU.  Lives in tests/: the product never imports it."""
import torch

from oracle import cpu_ref as R


class OracleRouteKernels:
    """parallel._HipRouteKernels' interface: stable bucketing by owner, list order inside a bucket."""

    def route_plan(self, ids, n_rows, rows_per_rank, world, max_out):
        assert bool(((ids >= 0) & (ids < n_rows)).all()), "index out of range in self"
        owner = ids // rows_per_rank
        counts = torch.bincount(owner, minlength=world)
        max_out.copy_(torch.maximum(max_out, counts.max().to(torch.int32).reshape(1)))
        return ids, owner, counts

    def route_build(self, planned, rows_per_rank, world, cap):
        ids, owner, counts = planned
        n = ids.numel()
        order = torch.sort(owner, stable=True).indices  # positions grouped by owner, list order inside
        starts = torch.cumsum(counts, 0) - counts
        rank = torch.empty(n, dtype=torch.int64)
        rank[order] = torch.arange(n) - starts[owner[order]]
        assert bool((rank < cap).all())
        slot_of = owner * cap + rank
        send_ids = torch.full((world * cap,), -1, dtype=torch.int64)
        src_of = torch.full((world * cap,), -1, dtype=torch.int64)
        send_ids[slot_of] = ids
        src_of[slot_of] = torch.arange(n)
        return send_ids, slot_of, src_of

    def localize(self, ids, lo, n_local):
        r = ids - lo
        return torch.where((ids >= 0) & (r >= 0) & (r < n_local), r, torch.full_like(r, n_local))

    def gather_owned(self, table, local, n_local):
        out = torch.zeros(local.numel(), table.shape[1])
        mine = (local >= 0) & (local < n_local)
        out[mine] = table[local[mine]].float()
        return out


class OracleMipsKernels:
    def mips_topk(self, query, corpus, k):
        idx, sc, _ = R.mips_topk(query, corpus.float(), k)
        return idx, sc

    def mips_merge(self, scores, idx, k):
        out_i, out_s = [], []
        for s, i in zip(scores, idx):
            order = sorted(range(len(i)), key=lambda t: (i[t].item() < 0, -s[t].item(), i[t].item()))[:k]
            out_i.append(i[order])
            out_s.append(s[order])
        return torch.stack(out_i), torch.stack(out_s)

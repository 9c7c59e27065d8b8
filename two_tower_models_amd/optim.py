"""DenseExactAdam: torch.optim.Adam semantics for the two-tower trainer, on HIP.

``optim.Adam(model.parameters())`` in the reference loop (ref:train/train.py:179,
:123-125) updates EVERY row of both embedding tables every step, because
``nn.Embedding`` produces a dense gradient.  This optimiser is value-equivalent
(same update rule, same bias correction, every row stepped) but never builds the
dense gradient: the embedding backward leaves its row gradients with the table
(``weight._tt_rowgrads``), and ``step()`` runs

    plan (stable sort of the looked-up ids)  ->  Adam on the looked-up rows
    ->  zero-gradient Adam sweep over the whole table (the HBM-bound part)
    ->  write the looked-up rows back,

one multi-tensor launch for all dense parameters, and one 1-thread launch that
advances the step count / bias corrections in device memory (so a captured
hipGraph of the step replays correctly).

Use exactly like the reference uses ``optim.Adam``:
    opt = DenseExactAdam(model.parameters(), lr=1e-3)
    loss = model.train_forward(...); opt.zero_grad(); loss.backward(); opt.step()
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List

import torch

from . import _native as N
from . import ops


class DenseExactAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999),
                 eps: float = 1e-8) -> None:
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if len(self.param_groups) != 1:
            raise ValueError("DenseExactAdam supports a single parameter group")
        self._params: List[torch.nn.Parameter] = [p for p in self.param_groups[0]["params"]]
        self._tables = [p for p in self._params if getattr(p, "_tt_is_table", False)]
        self._dense = [p for p in self._params if not getattr(p, "_tt_is_table", False)]
        for p in self._tables:
            p._tt_rowgrads = []  # switches the embedding backward to row form
        self._hyper = None
        self._ready = False

    # state is created lazily, on the parameters' device
    def _init_state(self) -> None:
        dev = N.require_device(*self._params)
        g = self.param_groups[0]
        self._hyper = torch.tensor([g["lr"], g["betas"][0], g["betas"][1], g["eps"], 0.0, 0.0, 0.0, 0.0],
                                   dtype=torch.float64, device=dev)
        for p in self._params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("DenseExactAdam needs contiguous fp32 parameters")
            st = self.state[p]
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        self._ready = True

    @property
    def step_count(self) -> int:
        return 0 if self._hyper is None else int(self._hyper[4].item())

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self._tables:
            p._tt_rowgrads.clear()
        super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure is not supported")
        if not self._ready:
            self._init_state()
        lib = N.load()
        hyper = self._hyper.data_ptr()
        N.check(lib.tt_adam_advance(hyper, N.stream()), "tt_adam_advance")

        for p in self._tables:
            blocks = p._tt_rowgrads
            st = self.state[p]
            n_rows, dim = p.shape
            if p.grad is not None:
                raise RuntimeError("embedding table received a dense gradient while in row-gradient mode")
            if blocks:
                plan = ops.RowPlan(blocks, n_rows)
                wsp, wsn = ops._ws(p.device, lib.tt_adam_table_workspace_bytes(plan.n, dim), "adam_side")
                N.check(lib.tt_adam_table(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                          n_rows, dim, hyper, C.byref(plan.sources), plan.n,
                                          plan.sorted_ids.data_ptr(), plan.perm.data_ptr(),
                                          plan.seg_begin.data_ptr(), plan.n_unique.data_ptr(), wsp, wsn,
                                          N.stream()), "tt_adam_table")
            # a table with no lookups this step has grad None: torch.optim skips it too

        live = [p for p in self._dense if p.grad is not None]
        if live:
            descs = (N.AdamTensor * len(live))()
            keep = []
            for i, p in enumerate(live):
                gr = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(gr)
                st = self.state[p]
                descs[i].p, descs[i].g = p.data_ptr(), gr.data_ptr()
                descs[i].m, descs[i].v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                descs[i].n = p.numel()
            N.check(lib.tt_adam_dense(descs, len(live), hyper, N.stream()), "tt_adam_dense")
        return None

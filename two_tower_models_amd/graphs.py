"""Whole-step hipGraph capture: forward + zero_grad + backward + optimiser step replayed as ONE
graph launch.  The step is capturable because nothing in it depends on host-side state: sizes
come from device-side counts (row plans), the Adam step count and bias corrections live in
device memory (`tt_adam_advance`), out-of-range ids raise through a device flag, and the
workspaces are torch tensors (allocated from the graph's private pool during capture).

What it buys: the ~85 (base model) to ~200 (history model) kernel launches of a step cost
5-10 us of host time each; once the HBM-bound table sweep is out of the way (deferred Adam, or
small tables) the step is launch-bound -- C2 with the deferred schedule: 1.08 -> 0.65 ms/step.
`capture_overlap=True` with `DenseExactAdam(overlap_sweep="forward")` makes the capture multi-stream: the table sweep
stays on its side stream as a parallel branch of the graph (forked and joined through the optimiser's own events).
Replays are bit-identical to eager steps, but on ROCm 7.2 the two branches do not overlap the way two live streams do:
C2 2.87 ms per replay vs 1.34 ms eager, C3 7.8 vs 5.2, P 6.0 vs 5.5 -- measured, so it stays opt-in and the default
captures the single-stream schedules (serial sweep, or deferred).
"""
from __future__ import annotations

from typing import Sequence

import torch


class GraphedTrainStep:
    """``step = GraphedTrainStep(model, optimizer, example_batch); loss = step(*batch)``.

    `example_batch` fixes the shapes; every later batch is copied into the captured input
    buffers.  The returned loss tensor is the graph's static output (clone it to keep a value).
    Mirrors the body of ref:train/train.py:112-125.  The `warmup` eager steps are REAL optimiser
    steps on the example batch.  Build it before running eager backward passes on the default
    stream (torch's usual whole-network-capture rule: autograd nodes remember their stream)."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, example_batch: Sequence[torch.Tensor],
                 warmup: int = 3, capture_overlap: bool = False) -> None:
        self.model, self.optimizer = model, optimizer
        if (capture_overlap and getattr(optimizer, "overlap_sweep", False) == "forward"
                and not getattr(optimizer, "lazy", False)):
            # forward-announced overlapped schedule: the sweep's side stream joins the capture through the events the
            # optimiser already uses (fork after the stashes, join in step()), so the graph keeps the overlap
            optimizer.capture_overlap = True
        elif getattr(optimizer, "overlap_sweep", False):
            # zero_grad-started overlap decides on the host whether a sweep is pending: the captured step uses the
            # single-stream schedule; warm up (and stay) on that one so its workspaces exist before capture
            optimizer.overlap_sweep = False
        self.static_inputs = [t.clone() for t in example_batch]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        from . import ops
        forks, ops._CONCURRENT_TOWERS = ops._CONCURRENT_TOWERS, False  # warm up on the schedule that will be captured: one
        try:                                                           # stream, both towers per launch (ops.FusedTowerPair)
            with torch.cuda.stream(side):  # eager warm-up on a side stream: lazy state, workspaces, kernel attributes
                for _ in range(max(1, warmup)):
                    self._body()
        finally:
            ops._CONCURRENT_TOWERS = forks
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.warmup_steps = max(1, warmup)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body()

    def _body(self) -> torch.Tensor:
        loss = self.model.train_forward(*self.static_inputs)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return loss

    def __call__(self, *batch: torch.Tensor) -> torch.Tensor:
        if len(batch) != len(self.static_inputs):
            raise ValueError(f"expected {len(self.static_inputs)} input tensors")
        from . import _native as N
        descs = (N.AdamTensor * len(batch))()
        keep = []
        for i, (dst, src) in enumerate(zip(self.static_inputs, batch)):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError("batch shape / dtype differs from the captured example batch")
            if not src.is_cuda:  # (host tensors: torch's own H2D copy)
                dst.copy_(src, non_blocking=True)
                src = dst
            src = src if src.is_contiguous() else src.contiguous()
            keep.append(src)
            descs[i].p, descs[i].g, descs[i].n = dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size()
        # one launch for all inputs (seven copy launches were 35 us of a 1.1 ms deferred-Adam replay)
        N.check(N.load().tt_copy_buffers(descs, len(batch), N.stream()), "tt_copy_buffers")
        steps = getattr(self.optimizer, "_host_steps", None)
        if steps is not None:  # the captured step advances the device-side step count; keep the host's mirror in line
            self.optimizer._host_steps = steps + 1
        self.graph.replay()
        return self.loss

"""Whole training iterations in one hipGraph.

At the reference's example sizes (10^2..10^5 elements) an iteration -- ``zero_grad``, loss, ``backward``,
optimiser step -- is a dozen kernels of a few microseconds each, so the wall time is launch overhead and
Python.  ``GraphedTraining`` captures ``steps_per_replay`` complete iterations once and replays them: the GPU
then runs the kernels back to back (the CDNA playbook's "capture launch-bound inner loops in hipGraphs").

Requirements (the usual ones for stream capture): static shapes and addresses -- parameters, their ``.grad``
tensors (kept allocated: ``zero_grad(set_to_none=False)``) and every tensor the closure reads stay where they
are; no host synchronisation inside the closure (``.item()``, ``print`` of a loss, Python branches on tensor
values); an optimiser whose step count lives on the device (``FusedAdam(..., capturable=True)`` or
``torch.optim.Adam(..., capturable=True)``).  L-BFGS has host control flow and cannot be captured.
"""
from __future__ import annotations

from typing import Callable

import torch


class GraphedTraining:
    """``closure()`` must compute and return the loss (no ``backward``): one captured iteration is
    ``optimizer.zero_grad(set_to_none=False); loss = closure(); loss.backward(); optimizer.step()``."""

    def __init__(self, closure: Callable[[], torch.Tensor], optimizer: torch.optim.Optimizer, steps_per_replay: int = 1,
                 warmup: int = 3, direct: bool = False, begin: Callable[[], None] = None, end: Callable[[], None] = None,
                 check: Callable[[], None] = None):
        """``direct=True``: ``closure`` itself leaves the gradients in ``.grad`` (``EnergyLoss2D.value_and_grad_``):
        an iteration is ``loss = closure(); optimizer.step()`` -- no ``zero_grad``, no autograd.  ``check``: a callable that
        raises when the replayed work went wrong in a way only the device knows (``ShardedTri3Energy.check_exchange``: a
        peer-window get that timed out); ``synchronize()`` runs it, ``replay()`` stays asynchronous."""
        self.closure, self.optimizer, self.steps_per_replay = closure, optimizer, int(steps_per_replay)
        self.direct = direct
        self._check = check
        self._begin, self._end = begin or (lambda: None), end or (lambda: None)   # bracket every sequence of iterations
        if optimizer is None and not direct:
            raise ValueError("optimizer=None needs direct=True (a closure that also updates the parameters)")
        if self.steps_per_replay < 1:
            raise ValueError("steps_per_replay must be >= 1")
        self.steps_done = 0
        # optimiser state must exist BEFORE the capture: created inside it, its zero fills become graph nodes and
        # every replay resets the moments / step count
        if optimizer is not None:
            if hasattr(optimizer, "init_state"):
                optimizer.init_state()
            elif warmup < 1 and len(optimizer.state) == 0:
                raise ValueError("GraphedTraining: warmup=0 with an optimiser whose state is still empty would create "
                                 "that state inside the capture; use warmup >= 1")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # eager warm-up off the default stream (allocations,
            self._begin()                                  # optimiser state, autograd buffers) -- these count
            for _ in range(warmup):                        # as real training iterations
                self._one()
            self._end()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.steps_done += warmup
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._begin()
            for _ in range(self.steps_per_replay):
                self.loss = self._one()
            self._end()
        # capture does not execute: nothing to add to steps_done

    def _one(self):
        if self.direct:
            loss = self.closure()
            if self.optimizer is not None:       # None: the closure updates the parameters itself (EnergyAdamStep.step)
                self.optimizer.step()
            return loss
        self.optimizer.zero_grad(set_to_none=False)
        loss = self.closure()
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def replay(self, n: int = 1) -> torch.Tensor:
        """Run ``n * steps_per_replay`` iterations; returns the (static) loss tensor of the last one --
        the loss evaluated BEFORE that iteration's optimiser step, as in the eager loop."""
        for _ in range(n):
            self.graph.replay()
        self.steps_done += n * self.steps_per_replay
        return self.loss

    def synchronize(self) -> torch.Tensor:
        """Wait for the replays issued so far and run the ``check`` callable (if any); returns the loss tensor."""
        torch.cuda.synchronize()
        if self._check is not None:
            self._check()
        return self.loss

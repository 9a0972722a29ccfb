"""Element-sharded energy evaluation over the GPUs of one node.

The reference has no distributed code at all; this is the one parallel strategy the
hot path needs.  The energy is a plain sum over elements (``/root/reference/src/loss.py:85-88``)
and the gradients are sums of per-element contributions, so:

* every rank holds the full (replicated) parameters and the same tile plan;
* rank ``r`` evaluates the contiguous tile range ``plan.shard_range(r, world)`` -- tiles are
  Morton-ordered, so a range is a spatially compact strip -- and thereby produces the
  complete gradient rows of the nodes those tiles own, plus a partial scalar energy;
* ONE collective per evaluation: a sum all-reduce (RCCL over xGMI under the ``nccl``
  backend) of the packed buffer ``[gx_free | gu_free | loss]``, after which every rank
  holds the identical full gradient and can take the identical optimiser step.

Rows a rank does not own stay zero in its send buffer (the kernel never writes them),
so the reduction is exact: each row has exactly one non-zero contributor.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import torch
import torch.distributed as dist

from . import _lib

F64 = torch.float64


class ShardedTri3Energy:
    """``loss = sharded(model); loss.backward()`` with elements sharded over the ranks of
    ``group``.  ``evaluate`` is the per-rank evaluator ``(lo, hi, loss_view, gx_view, gu_view)``;
    it defaults to the HIP tiled kernel and exists so the host logic (range split, packing,
    collective) can be exercised by the multi-process CPU tests with the oracle standing in
    for the kernel."""

    def __init__(self, model, loss_fn, group=None, evaluate: Optional[Callable] = None, plan=None):
        self.model, self.loss_fn, self.group = model, loss_fn, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.plan = plan if plan is not None else model.tile_plan(loss_fn.tile_elems)
        self.lo, self.hi = self.plan.shard_range(self.rank, self.world)
        self._evaluate = evaluate or self._evaluate_hip
        nx, nu = model.node_coords_free.numel(), model.u_free.numel()
        self._nx, self._nu = nx, nu
        dev = model.node_coords_free.device
        self.send = torch.zeros(nx + nu + 1, dtype=F64, device=dev)      # non-owned rows stay 0 forever
        self.recv = torch.empty_like(self.send)

    # views into the packed buffers
    def _views(self, buf):
        nx, nu = self._nx, self._nu
        return buf[nx + nu:nx + nu + 1], buf[:nx].view(-1, 2), buf[nx:nx + nu].view(-1, 2)

    def _evaluate_hip(self, lo, hi, loss_v, gx_v, gu_v):
        m, lf = self.model, self.loss_fn
        dev = m.node_coords_free.device
        T_edge, Tconst = lf._traction(m, None)
        dv = lambda a: (C.c_double * len(a))(*a)
        xf = m.node_coords_free.detach()
        uf = m.u_free.detach()
        if xf.dtype != F64 or uf.dtype != F64:
            raise RuntimeError("sharded evaluation needs an fp64 model (model.double())")
        xfix, ufix = m.node_coords_fixed, m.u_fixed_rows()      # cached by the model: no per-step allocation
        rc = _lib.lib().hfem_tri3_energy_plan(
            self.plan.handle, xf.data_ptr(), xfix.data_ptr() if xfix.numel() else None, uf.data_ptr(),
            ufix.data_ptr() if ufix.numel() else None, dv(lf._mat), lf._W, dv(lf._body_table(None)), None,
            dv(Tconst), int(lo), int(hi), loss_v.data_ptr(), gx_v.data_ptr(), gu_v.data_ptr(), 0,
            _lib.stream_ptr(dev))
        _lib.check(rc, "hfem_tri3_energy_plan")

    def evaluate_local(self):
        """Kernel over this rank's tiles only (no communication): fills the send buffer."""
        loss_v, gx_v, gu_v = self._views(self.send)
        self._evaluate(self.lo, self.hi, loss_v, gx_v, gu_v)

    def exchange(self):
        """The single collective: recv = sum over ranks of send."""
        self.recv.copy_(self.send)
        if self.world > 1:
            dist.all_reduce(self.recv, op=dist.ReduceOp.SUM, group=self.group)
        return self._views(self.recv)

    def exchange_loss_only(self):
        """Owner-sharded mode: every rank keeps only the gradient rows its own tiles produced (they are
        complete -- owner-computes with halo recompute -- so no other rank contributes to them) and the
        ranks exchange just the scalar energy (8 bytes).  This is what a node-sharded optimiser needs;
        the parameters' interface rows are then exchanged after the optimiser step.
        Returns (global loss 0-d view, local gx view, local gu view) on the SEND buffer."""
        loss_v, gx_v, gu_v = self._views(self.send)
        self._loss_red = getattr(self, "_loss_red", None)
        if self._loss_red is None:
            self._loss_red = torch.empty(1, dtype=F64, device=self.send.device)
        self._loss_red.copy_(loss_v)
        if self.world > 1:
            dist.all_reduce(self._loss_red, op=dist.ReduceOp.SUM, group=self.group)
        return self._loss_red[0], gx_v, gu_v

    def value_and_grad(self):
        """One sharded fwd+bwd pass -> (loss, d/d node_coords_free, d/d u_free), identical on all ranks."""
        self.evaluate_local()
        loss_v, gx_v, gu_v = self.exchange()
        return loss_v[0], gx_v, gu_v

    def __call__(self, model=None):
        return _ShardedFn.apply(self.model.node_coords_free, self.model.u_free, self)


class _ShardedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_free, u_free, sharded):
        loss, gx, gu = sharded.value_and_grad()
        ctx.unit = (gx.clone(), gu.clone())
        return loss.clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        gx, gu = ctx.unit
        return gx * g, gu * g, None

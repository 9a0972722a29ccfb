"""Fused optimiser steps.

``FusedAdam`` is a drop-in for ``torch.optim.Adam(params, lr=...)`` as the reference's examples use it
(``/root/reference/examples/example1.py:31``, ``example2.py:37``, ``example3.py:89``: default betas and
eps, no weight decay, no amsgrad): one HIP launch per parameter tensor instead of torch's chain of
element-wise kernels, same arithmetic operation for operation.  ROCm tensors only (no CPU fallback).
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, ptr, require_gpu_tensor, stream_ptr, dev_index


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                require_gpu_tensor(p.data, "parameter", dtype=None)
                if p.dtype not in (torch.float64, torch.float32):
                    raise RuntimeError("FusedAdam supports fp64 / fp32 parameters")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if g.dtype != p.dtype:
                    g = g.to(p.dtype)
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                check(L.hfem_adam_step(dev_index(p.device), ptr(p.data), ptr(g), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]),
                                       p.numel(), 0 if p.dtype == torch.float64 else 1, float(group["lr"]), float(b1),
                                       float(b2), float(group["eps"]), int(st["step"]), stream_ptr(p.device)),
                      "hfem_adam_step")
        return loss

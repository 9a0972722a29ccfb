"""Post-processing on the device: the compute cores of the reference's plotting helpers
(``/root/reference/src/plots.py``), without matplotlib and without per-element Python loops.

* ``von_mises(model, E, nu)``  -- ``plots.plot_von_mises`` (plots.py:177-198) up to the array it colours the
  triangles with: centroid ``grad_u`` -> strain -> plane-stress stress -> von Mises, one fused kernel.
* ``compute_du_dx_per_element(model)`` -- plots.py:5-27 (one ``autograd.grad`` call per element in a Python
  loop there): the per-element slope of the 1D piecewise-linear field, one kernel.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, require_gpu_tensor

F64 = torch.float64


def von_mises(model, E: float = 10e9, nu: float = 0.3, return_grad_u: bool = False):
    """Per-element von Mises stress [Ne] (and optionally the centroid ``grad_u`` [Ne,2,2]) of a triangular
    model; dtype of ``model.coords``."""
    if getattr(model, "nodes_per_element", 3) != 3:
        raise NotImplementedError("von_mises: TRI3 models (the reference's plot_von_mises is triangular)")
    with torch.no_grad():
        X, U = model.coords.detach(), model.u_full.detach()
        require_gpu_tensor(X.contiguous(), "model.coords", dtype=None)
        dt = X.dtype
        X64, U64 = X.to(F64).contiguous(), U.to(F64).contiguous()
        ne = model.Nelems
        vm = torch.empty(ne, dtype=F64, device=X.device)
        gu = torch.empty(ne, 2, 2, dtype=F64, device=X.device) if return_grad_u else None
        check(_lib.lib().hfem_tri3_von_mises(_lib.dev_index(X.device), ptr(X64), ptr(U64), ptr(model._conn32), ne, float(E),
                                             float(nu), ptr(vm), ptr(gu), stream_ptr(X.device)), "hfem_tri3_von_mises")
        vm = vm.to(dt)
        return (vm, gu.to(dt)) if return_grad_u else vm


def compute_du_dx_per_element(model) -> torch.Tensor:
    """du/dx on every element of a 1D model ([Ne] for scalar u, [Ne, dim_u] otherwise), returned on the CPU
    like the reference's helper."""
    with torch.no_grad():
        grid, u = model.grid.detach(), model.u_full.detach()
        require_gpu_tensor(grid.contiguous(), "model.grid", dtype=None)
        g64 = grid.to(F64).contiguous()
        u64 = u.to(F64).reshape(grid.shape[0], -1).contiguous()
        dim_u = u64.shape[1]
        out = torch.empty(grid.shape[0] - 1, dim_u, dtype=F64, device=grid.device)
        check(_lib.lib().hfem_line2_slopes(_lib.dev_index(grid.device), ptr(g64), ptr(u64), grid.shape[0], dim_u, ptr(out),
                                           stream_ptr(grid.device)), "hfem_line2_slopes")
        out = out.to(grid.dtype).cpu()
        return out[:, 0] if dim_u == 1 else out

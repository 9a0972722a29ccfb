"""Energy / L2 losses with the reference's API over the fused HIP kernels.

``EnergyLoss2D`` keeps the constructor, attributes and call signature of
``/root/reference/src/loss.py:6-116``; ``loss_fn(model)`` is ONE tiled kernel launch
(+ a tiny reduction) that yields the scalar and the gradients w.r.t.
``model.node_coords_free`` and ``model.u_free``.  The inline losses of examples 1-3
(``examples/example1.py:38``, ``example2.py:46``, ``example3.py:27-70``) are provided as
library functions with fused kernels.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import ops
from .utils import interval_gauss_points, triangle_gauss_points

HFEM_FLAG_NO_EDGES = 4
HFEM_FLAG_PHYSICAL_GRAD = 64
HFEM_FLAG_DETERMINISTIC = 128
HFEM_FLAG_FP32_MATH = 1024


class EnergyLoss2D:
    def __init__(self, E: float = 10e9, nu: float = 0.3, length: float = 1.0, height: float = 1.0,
                 gauss_order: int = 4, gauss_order_1d: int = 2, device: Optional[torch.device] = None,
                 dtype: torch.dtype = torch.float32, tile_elems: int = 0, grad_convention: Optional[str] = None,
                 deterministic: bool = False, arithmetic: str = "auto"):
        """Reference signature (loss.py:7-17) plus opt-in switches:
        ``arithmetic`` -- what fp32 MODELS (the reference's default dtype, loss.py:16) compute in: ``"fp32"`` = the
        reference's own arithmetic (packed-fp32 element math, float accumulators; results inside the band the reference's
        fp32 run occupies around exact arithmetic, tests/test_gpu_tri3_f32.py; TRI3 meshes whose plan has paired slots),
        ``"fp64"`` = float rows widened on load, fp64 arithmetic, ONE rounding on store (the accurate option, ~1.4x slower),
        ``"auto"`` (default) = fp32 where that kernel exists, fp64 elsewhere (unpaired plans, QUAD4, physical convention,
        deterministic).  fp64 models always compute in fp64.
        ``grad_convention``: ``"reference"`` (``dN_dx = Jinv * dN_dxi`` exactly as models.py:351, the parity contract),
        ``"physical"`` (``Jinv^T``: exact for linear fields, invariant to an element's local node order) or
        ``None`` = whatever the model says (``model.grad_convention``, ``"reference"`` unless set);
        ``deterministic``: fixed-order accumulation -- loss and gradients bit-identical run to run (node-centric
        cross-check kernel, ~3-4x slower); ``tile_elems``: home elements per tile (0 = library default)."""
        if grad_convention not in (None, "reference", "physical"):
            raise ValueError("grad_convention must be 'reference', 'physical' or None")
        if arithmetic not in ("auto", "fp32", "fp64"):
            raise ValueError("arithmetic must be 'auto', 'fp32' or 'fp64'")
        self.arithmetic = arithmetic
        self.grad_convention, self.deterministic = grad_convention, bool(deterministic)
        self.E, self.nu = E, nu
        self.length, self.height = length, height          # stored, never read (as upstream, F9)
        self.gauss_order, self.gauss_order_1d = gauss_order, gauss_order_1d
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.dtype = dtype
        self.tile_elems = tile_elems
        self.quad4_planless = False       # True: QUAD4 energy through the planless atomic kernel (cross-check)
        factor = E / (1 - nu ** 2)
        self.C = torch.tensor([[1.0, nu, 0.0], [nu, 1.0, 0.0], [0.0, 0.0, (1.0 - nu) / 2.0]],
                              dtype=dtype, device=self.device) * factor          # loss.py:29-32
        self.xg, self.wg = triangle_gauss_points(order=gauss_order, device=self.device, dtype=dtype)
        self.ng = self.xg.shape[0]
        self.xg_1d, self.wg_1d = interval_gauss_points(order=gauss_order_1d, device=self.device, dtype=dtype)
        self.ng1 = self.xg_1d.shape[0]
        # host copies of the constants the kernel takes by value (no D2H sync per call)
        Cc = self.C.detach().cpu().double()
        self._mat = [Cc[0, 0].item(), Cc[0, 1].item(), Cc[1, 1].item(), Cc[2, 2].item()]
        self._W = float(self.wg.detach().cpu().double().sum())
        xg1, wg1 = self.xg_1d.detach().cpu().double(), self.wg_1d.detach().cpu().double()
        self._ci, self._cj = float((wg1 * (1.0 - xg1)).sum()), float((wg1 * xg1).sum())

    # ---- default forces (loss.py:43-51)
    def uniform_body_force(self, x: torch.Tensor) -> torch.Tensor:
        return torch.zeros_like(x)

    def uniform_edge_force(self, x: torch.Tensor, L: float = 1.0, F_total: float = 100e3) -> torch.Tensor:
        t_x = torch.full((x.shape[0],), F_total / L, device=x.device, dtype=x.dtype)
        return torch.stack([t_x, torch.zeros_like(t_x)], dim=1)

    # ---- tables handed to the kernel instead of Python callables
    def _body_table(self, b_force):
        """B_k = sum_q w_q N_k(xi_q) b(xi_q) with b at the REFERENCE points (loss.py:60,80; F6)."""
        if b_force is None:
            return [0.0] * 6
        xg = self.xg
        b = b_force(xg).to(torch.float64)
        w, x = self.wg.to(torch.float64), xg.to(torch.float64)
        N = torch.stack([x[:, 0], x[:, 1], 1.0 - x[:, 0] - x[:, 1]], dim=1)
        return torch.einsum("q,qk,qi->ki", w, N, b).reshape(-1).cpu().tolist()

    def _traction(self, model, t_force):
        """Constant table {c_i t, c_j t} for the default traction, else per-edge
        {T_i, T_j} = sum_q w_q {(1-xi_q), xi_q} t(x_q) with raw Legendre xi (loss.py:96-106; F3)."""
        if t_force is None:
            t = [100e3 / 1.0, 0.0]                                     # uniform_edge_force defaults
            return None, [self._ci * t[0], self._ci * t[1], self._cj * t[0], self._cj * t[1]]
        with torch.no_grad():
            x_i, x_j = model.nm_edges[:]
            xg1 = self.xg_1d.to(x_i.dtype)
            xq = (1.0 - xg1[None, :, None]) * x_i[:, None, :] + xg1[None, :, None] * x_j[:, None, :]
            tq = t_force(xq.reshape(-1, 2)).reshape(x_i.shape[0], self.ng1, 2).to(torch.float64)
            w1, x1 = self.wg_1d.to(torch.float64), self.xg_1d.to(torch.float64)
            Ti = torch.einsum("q,eqi->ei", w1 * (1.0 - x1), tq)
            Tj = torch.einsum("q,eqi->ei", w1 * x1, tq)
            return torch.cat([Ti, Tj], dim=1).contiguous(), None

    def _edge_nodes_free(self, model):
        if model.neumann_edges is None or model.N_edges == 0:
            return False
        return bool(model.free_mask[model.neumann_edges.reshape(-1)].any().item())

    def _mode_flags(self, model) -> int:
        conv = self.grad_convention or getattr(model, "grad_convention", "reference")
        if conv not in ("reference", "physical"):
            raise ValueError(f"unknown grad_convention {conv!r}")
        return (HFEM_FLAG_PHYSICAL_GRAD if conv == "physical" else 0) | (HFEM_FLAG_DETERMINISTIC if self.deterministic else 0)

    def _f32_math(self, model, plan, flags) -> bool:
        """fp32 arithmetic for this launch?  Only fp32 TRI3 models on paired-slot plans in the reference convention."""
        if self.arithmetic == "fp64" or model.node_coords_free.dtype != torch.float32:
            return False
        ok = bool(plan.stats["paired"]) and not (flags & (HFEM_FLAG_PHYSICAL_GRAD | HFEM_FLAG_DETERMINISTIC))
        if self.arithmetic == "fp32" and not ok:
            raise RuntimeError("arithmetic='fp32': the fp32-arithmetic kernel needs a TRI3 plan with paired slots, the reference "
                               "gradient convention and atomic accumulation; use arithmetic='auto' or 'fp64'")
        return ok

    def _fused(self, model, b_force, T_edge, Tconst, flags, tile_range=(0, -1)):
        flags |= self._mode_flags(model)
        plan = model.tile_plan(self.tile_elems)
        if self._f32_math(model, plan, flags):
            flags |= HFEM_FLAG_FP32_MATH
        return ops.Tri3EnergyFn.apply(model.node_coords_free, model.u_free,
                                      model.node_coords_fixed.to(model.dtype), model.u_fixed_rows(), plan,
                                      self._mat, self._W, self._body_table(b_force), T_edge, Tconst,
                                      tile_range, flags)

    # ---- reference API
    def domain_energy(self, model, b_force: Optional[Callable] = None) -> torch.Tensor:
        """loss.py:55-88 (strain energy minus body work), fused."""
        return self._fused(model, b_force, None, [0.0] * 4, HFEM_FLAG_NO_EDGES)

    def edge_energy(self, model, t_force: Optional[Callable] = None) -> torch.Tensor:
        """loss.py:91-110 through the unfused edge forward; differentiable in everything,
        including a position-dependent traction (edges are O(sqrt(Ne)): negligible)."""
        x_i, x_j = model.nm_edges[:]
        n_edges, dev = model.N_edges, x_i.device
        xg1, wg1 = self.xg_1d.to(device=dev, dtype=x_i.dtype), self.wg_1d.to(device=dev, dtype=x_i.dtype)
        xq = (1.0 - xg1[None, :, None]) * x_i[:, None, :] + xg1[None, :, None] * x_j[:, None, :]
        xq_flat = xq.reshape(-1, 2)
        wq_flat = wg1[None, :].expand(n_edges, self.ng1).reshape(-1)
        x_eval = xg1[None, :].expand(n_edges, self.ng1).reshape(-1, 1).contiguous()
        edge_id = torch.repeat_interleave(torch.arange(n_edges, device=dev), repeats=self.ng1)
        u_edge, ds = model(x_eval, edge_id, edge=True)
        t_edge = t_force(xq_flat) if t_force is not None else self.uniform_edge_force(xq_flat)
        return torch.sum((u_edge * t_edge).sum(dim=1) * (wq_flat * ds))

    def _quad4_body(self, b_force):
        """b at the 2x2 Gauss points of the reference square (order (-,-) (+,-) (-,+) (+,+)): as on triangles the body
        force receives REFERENCE coordinates (loss.py:60,80)."""
        if b_force is None:
            return None
        g = 1.0 / 3.0 ** 0.5
        pts = torch.tensor([[-g, -g], [g, -g], [-g, g], [g, g]], dtype=self.dtype, device=self.device)
        return b_force(pts).to(torch.float64).reshape(-1).cpu().tolist()

    def _quad4(self, model, b_force, t_force):
        """QUAD4 extension: 2x2 Gauss (``gauss_order`` is a triangle-rule setting and does not apply); body force at the
        reference Gauss points, traction as on triangles (constant table, per-edge table, or -- when the traction
        depends on points that move with free nodes -- autograd through ``t_force`` on the unfused edge path)."""
        mode = self._mode_flags(model)            # physical convention / deterministic: instances of the tiled QUAD4 path
        Bq = self._quad4_body(b_force)
        no_edges = model.neumann_edges is None or model.N_edges == 0
        if self.quad4_planless:                                    # cross-check path: fp64 global atomics
            if mode:
                raise NotImplementedError("QUAD4 planless cross-check kernel: reference convention, atomic accumulation")
            if b_force is not None or t_force is not None:
                raise NotImplementedError("QUAD4 planless cross-check kernel: default forces only")
            _, Tconst = self._traction(model, None)
            edges = model._edges32 if model.N_edges else None
            return ops.Quad4EnergyFn.apply(model.coords, model.u_full, model._conn32, edges, self._mat, Tconst)
        plan = model.tile_plan(self.tile_elems)
        args = (model.node_coords_free, model.u_free, model.node_coords_fixed.to(model.dtype), model.u_fixed_rows(), plan,
                self._mat)
        if no_edges:
            return ops.Quad4PlanEnergyFn.apply(*args, [0.0] * 4, Bq, None, HFEM_FLAG_NO_EDGES | mode)
        if t_force is not None and model.node_coords_free.requires_grad and self._edge_nodes_free(model):
            return ops.Quad4PlanEnergyFn.apply(*args, [0.0] * 4, Bq, None, HFEM_FLAG_NO_EDGES | mode) - self.edge_energy(model, t_force)
        T_edge, Tconst = self._traction(model, t_force)
        return ops.Quad4PlanEnergyFn.apply(*args, Tconst, Bq, T_edge, mode)

    def value_and_grad_(self, model) -> torch.Tensor:
        """Autograd-free fast path: ONE launch writes the total potential's gradients straight into
        ``model.node_coords_free.grad`` and ``model.u_free.grad`` (overwritten -- every free row is owned by exactly
        one tile, so no ``zero_grad`` is needed) and returns the loss (0-d, fp64).  Default forces only (zero body
        force, constant traction); TRI3 and QUAD4 models, fp64 or fp32 rows.  An optimiser can step right after it; under
        ``GraphedTraining(..., direct=True)`` a whole iteration is this launch plus the optimiser's."""
        import ctypes as C
        from . import _lib
        quad = getattr(model, "nodes_per_element", 3) == 4
        xf, uf = model.node_coords_free, model.u_free
        if xf.dtype != uf.dtype or xf.dtype not in (torch.float64, torch.float32):
            raise RuntimeError("value_and_grad_: parameters must both be fp64 or both fp32")
        for p in (xf, uf):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        plan = model.tile_plan(self.tile_elems)
        f32 = xf.dtype == torch.float32
        if f32 and (plan.stats["max_tile_nodes"] > 1024 or plan.stats["max_tile_elems"] > 2048):
            raise RuntimeError("value_and_grad_: this tile plan has no fp32-storage kernel; use model.double()")
        # cached per (model, dtype, device): only host-side constants and the static loss tensor.  The fixed rows are
        # re-derived on every call -- `u_fixed_rows()` tracks `u_fixed._version`, `.to()` is a no-op when nothing
        # changed -- so `model.to(device)` or an in-place edit of `u_fixed` / `node_coords_fixed` is never stale.
        import weakref
        cache = getattr(self, "_direct_cache", None)
        key = (xf.dtype, xf.device)
        if cache is None or cache[0][0]() is not model or cache[0][1:] != key:     # weak reference: an id() can be reused
            key = (weakref.ref(model),) + key
            _, Tconst = self._traction(model, None)
            dv = lambda v: (C.c_double * len(v))(*v)
            cache = self._direct_cache = (key, dv(self._mat), dv([0.0] * 6), dv(Tconst),
                                          torch.zeros((), dtype=torch.float64, device=xf.device))
        _, mat, Bk, Tc, loss = cache
        xfix = model.node_coords_fixed.to(device=xf.device, dtype=xf.dtype).contiguous()
        ufix = model.u_fixed_rows().to(device=xf.device, dtype=xf.dtype).contiguous()
        flags = (0 if model.N_edges else HFEM_FLAG_NO_EDGES) | self._mode_flags(model)
        if f32 and not quad and self._f32_math(model, plan, flags):
            flags |= HFEM_FLAG_FP32_MATH
        if flags & HFEM_FLAG_DETERMINISTIC and f32:
            raise NotImplementedError("deterministic: fp64 models (model.double())")
        if flags & HFEM_FLAG_PHYSICAL_GRAD and f32 and not quad:
            raise NotImplementedError("grad_convention='physical' on fp32 rows: QUAD4 models, or model.double()")
        if quad:                                     # QUAD4-iso extension: same contract, tiled QUAD4 kernel
            _lib.check(_lib.lib().hfem_quad4_energy_plan_ex(
                plan.handle, 1 if f32 else 0, xf.data_ptr(), xfix.data_ptr() if xfix.numel() else None, uf.data_ptr(),
                ufix.data_ptr() if ufix.numel() else None, mat, None, None, Tc, 0, -1, loss.data_ptr(), xf.grad.data_ptr(),
                uf.grad.data_ptr(), flags, _lib.stream_ptr(xf.device)), "hfem_quad4_energy_plan")
            return loss
        fn = _lib.lib().hfem_tri3_energy_plan_f32 if f32 else _lib.lib().hfem_tri3_energy_plan
        _lib.check(fn(plan.handle, xf.data_ptr(), xfix.data_ptr() if xfix.numel() else None, uf.data_ptr(),
                      ufix.data_ptr() if ufix.numel() else None, mat, self._W, Bk, None, Tc, 0, -1, loss.data_ptr(),
                      xf.grad.data_ptr(), uf.grad.data_ptr(), flags, _lib.stream_ptr(xf.device)), "hfem_tri3_energy_plan")
        return loss

    def __call__(self, model, b_force=None, t_force=None) -> torch.Tensor:
        """Total potential = domain - edge (loss.py:113-116), one fused launch."""
        if getattr(model, "nodes_per_element", 3) == 4:
            return self._quad4(model, b_force, t_force)
        if model.neumann_edges is None or model.N_edges == 0:
            return self._fused(model, b_force, None, [0.0] * 4, HFEM_FLAG_NO_EDGES)
        if t_force is not None and model.node_coords_free.requires_grad and self._edge_nodes_free(model):
            # traction depends on points that move with free nodes: keep autograd through t_force
            return self.domain_energy(model, b_force) - self.edge_energy(model, t_force)
        T_edge, Tconst = self._traction(model, t_force)
        return self._fused(model, b_force, T_edge, Tconst, 0)


# ---------------------------------------------------------------- inline losses of examples 1-3
def l2_projection_loss(model, x_eval, target):
    """``((model(x) - target)**2).mean()`` of examples/example1.py:38 as one fused launch
    (1D model) or examples/example2.py:46 (structured 2D model)."""
    if hasattr(model, "Nx"):
        gx, gy = model.grid
        return ops.RectQ4MseFn.apply(gx, gy, model.u_full, x_eval, target)
    return ops.Line2MseFn.apply(model.grid, model.u_full, x_eval, target)


def bar_energy_loss(model, xi, wi, b_force, E, L=None):
    """1D bar total potential of examples/example3.py:27-70:
    ``sum_q wq (E/2 (du/dx)^2 - b(xq) u(xq))`` with quadrature points and weights built
    from the *detached* grid (the reference computes them under ``no_grad``)."""
    grid = model.grid
    with torch.no_grad():
        g = grid.detach()
        x_i, x_j = g[:-1].unsqueeze(1), g[1:].unsqueeze(1)
        xq = 0.5 * (x_j - x_i) * xi + 0.5 * (x_j + x_i)
        wq = 0.5 * (x_j - x_i) * wi
        bq = b_force(xq)
    return ops.BarEnergyFn.apply(grid, model.u_full, xq, wq, bq, float(E))

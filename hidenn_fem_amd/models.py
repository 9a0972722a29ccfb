"""Shape-function models with the reference's construction API, backed by HIP kernels.

Public names follow ``/root/reference/src/models.py``:

* ``PiecewiseLinearShapeNN``            1D, ``models.py:6-90``
* ``PiecewiseLinearShapeNN2D(...)``     dispatches on its arguments (the reference
  defines the name twice and the second definition shadows the first):
  ``grid_x=, grid_y=`` -> ``StructuredShapeNN2D`` (``models.py:93-212``);
  ``node_coords, connectivity`` -> ``TriangularShapeNN2D`` (``models.py:241-376``).

Parameter / buffer names are the reference's, so ``state_dict()`` round-trips with it.
Every ``forward`` launches gfx950 kernels through ``hidenn_fem_amd.ops``; CPU tensors raise.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .plan import TilePlan, row_maps


def _np_mask(t):
    return t.detach().cpu().numpy().astype(bool)


# ============================================================================ 1D
class PiecewiseLinearShapeNN(nn.Module):
    """Nodes on a line, hat functions, optional r-adaptivity through positive increments
    (``softplus -> clamp(1e-6) -> cumsum -> renormalise``), optional Dirichlet end values."""

    def __init__(self, node_coords, r_adapt=False, u0=None, uN=None):
        super().__init__()
        self.N = len(node_coords)
        self.r_adapt = r_adapt
        self.register_buffer("x0", node_coords[0:1].clone())
        self.register_buffer("xN", node_coords[-1:].clone())
        if self.r_adapt and self.N > 2:
            self.x_increments = nn.Parameter(node_coords[1:] - node_coords[:-1])
        else:
            self.register_buffer("x_inner", node_coords[1:-1].clone())
        # end values are float32 buffers upstream (models.py:25,30): .double() widens the
        # float32-rounded value, and parity needs the same number
        if u0 is not None:
            self.register_buffer("u0_fixed", torch.tensor([u0], dtype=torch.float32))
        else:
            self.u0_fixed = None
        if uN is not None:
            self.register_buffer("uN_fixed", torch.tensor([uN], dtype=torch.float32))
        else:
            self.uN_fixed = None
        self._ends = (float(node_coords[0]), float(node_coords[-1]))   # host copies: no D2H sync per call
        n_fixed = (u0 is not None) + (uN is not None)
        self.u = nn.Parameter(torch.zeros(self.N - n_fixed))
        self.epsilon = 1e-10

    @property
    def grid(self):
        if self.r_adapt and self.N > 2:
            return ops.GridParamFn.apply(self.x_increments, self._ends[0], self._ends[1], None, None)
        return torch.cat([self.x0, self.x_inner, self.xN], dim=0)

    @property
    def u_full(self):
        parts = [p for p in (self.u0_fixed, self.u.view(-1), self.uN_fixed) if p is not None]
        return torch.cat([p.to(self.u.dtype) for p in parts]) if len(parts) > 1 else parts[0]

    def forward(self, x_eval):
        return ops.Line2EvalFn.apply(self.grid, self.u_full, x_eval)[0]

    def forward_with_derivative(self, x_eval):
        """``(u, du/dx)`` -- both differentiable w.r.t. the parameters (first order)."""
        return ops.Line2EvalFn.apply(self.grid, self.u_full, x_eval)


# ============================================================================ one name, two element families
def _pick_family_2d(args, kwargs):
    """Which concrete class a ``PiecewiseLinearShapeNN2D(...)`` call means: the reference module defines the
    name twice -- structured ``(grid_x, grid_y, ...)`` (models.py:93-212, shadowed) and triangular
    ``(node_coords, connectivity, ...)`` (models.py:241-376)."""
    structured_kw = {"grid_x", "grid_y", "boundary_mask_x", "boundary_mask_y", "r_adapt"}
    tri_kw = {"node_coords", "connectivity", "boundary_mask", "dirichlet_mask", "neumann_edges"}
    if structured_kw & kwargs.keys():
        return StructuredShapeNN2D
    conn = kwargs.get("connectivity", args[1] if len(args) >= 2 else None)
    unstructured = bool(tri_kw & kwargs.keys()) or (
        len(args) >= 2 and torch.is_tensor(args[1]) and args[1].dim() == 2 and not torch.is_floating_point(args[1]))
    if unstructured:                                         # (node_coords [N,2], connectivity [Ne,3|4] int)
        if conn is not None and conn.dim() == 2 and conn.shape[1] == 4:
            return QuadShapeNN2D                             # extension element
        return TriangularShapeNN2D
    return StructuredShapeNN2D                               # (grid_x [Nx], grid_y [Ny])


class PiecewiseLinearShapeNN2D(nn.Module):
    """One name, two element families, like the reference module.

    ``PiecewiseLinearShapeNN2D(grid_x=..., grid_y=..., ...)`` (examples/example2.py:31-36) builds the structured
    model; ``PiecewiseLinearShapeNN2D(node_coords, connectivity, ...)`` (examples/example4.py:40-46) the triangular
    one (``connectivity [Ne,4]``: the QUAD4 extension).  It is a real class: every model it builds is an instance
    (``isinstance(m, PiecewiseLinearShapeNN2D)``), and a user subclass ``class My(PiecewiseLinearShapeNN2D)`` gets
    the family its constructor arguments select mixed in behind it (``My.__init__`` may call
    ``super().__init__(*args, **kwargs)`` as it would against the reference class)."""

    def __new__(cls, *args, **kwargs):
        if cls is PiecewiseLinearShapeNN2D:
            cls = _pick_family_2d(args, kwargs)
        elif not any(issubclass(cls, f) for f in (StructuredShapeNN2D, TriangularShapeNN2D)):
            fam = _pick_family_2d(args, kwargs)              # user subclass of the dispatching name
            cls = type(cls.__name__, (cls, fam), {"__module__": cls.__module__, "__qualname__": cls.__qualname__})
        return super().__new__(cls)


# ============================================================================ structured 2D
class StructuredShapeNN2D(PiecewiseLinearShapeNN2D):
    """Tensor-product rectilinear grid, bilinear interpolation (the reference's first,
    shadowed ``PiecewiseLinearShapeNN2D``, ``models.py:93-212``)."""

    def __init__(self, grid_x, grid_y, boundary_mask_x=None, boundary_mask_y=None, r_adapt=False, u_fixed=None):
        super().__init__()
        self.Nx, self.Ny = grid_x.numel(), grid_y.numel()
        self.r_adapt = r_adapt
        self.register_buffer("initial_x_grid", grid_x.clone())
        self.register_buffer("initial_y_grid", grid_y.clone())
        self.register_buffer("x0", grid_x.flatten()[0:1].clone())
        self.register_buffer("xN", grid_x.flatten()[-1:].clone())
        self.register_buffer("y0", grid_y.flatten()[0:1].clone())
        self.register_buffer("yN", grid_y.flatten()[-1:].clone())
        if self._adaptive:
            self.increments_x = nn.Parameter(grid_x[1:] - grid_x[:-1])
            self.increments_y = nn.Parameter(grid_y[1:] - grid_y[:-1])
        else:
            self.register_buffer("x_grid_inner", grid_x[1:-1].clone())
            self.register_buffer("y_grid_inner", grid_y[1:-1].clone())

        def ends(n, dev):
            m = torch.zeros(n, dtype=torch.bool, device=dev)
            m[0] = m[-1] = True
            return m

        if boundary_mask_x is None:
            boundary_mask_x = ends(self.Nx, grid_x.device)
        if boundary_mask_y is None:
            boundary_mask_y = ends(self.Ny, grid_y.device)
        self.register_buffer("boundary_mask_x", boundary_mask_x)
        self.register_buffer("boundary_mask_y", boundary_mask_y)
        self.register_buffer("node_mask", boundary_mask_x[:, None] | boundary_mask_y[None, :])
        if u_fixed is not None:   # float32 buffer upstream (models.py:137)
            self.register_buffer("u_fixed", torch.tensor([u_fixed], dtype=torch.float32))
        else:
            self.u_fixed = None
        self.u = nn.Parameter(torch.randn(self.Nx, self.Ny))
        self.epsilon = 1e-10
        self.register_buffer("_mask_x_u8", boundary_mask_x.to(torch.uint8), persistent=False)
        self.register_buffer("_mask_y_u8", boundary_mask_y.to(torch.uint8), persistent=False)
        gx_, gy_ = grid_x.flatten(), grid_y.flatten()
        self._ends = (float(gx_[0]), float(gx_[-1]), float(gy_[0]), float(gy_[-1]))

    @property
    def _adaptive(self):
        return self.r_adapt and max(self.Nx, self.Ny) > 2

    @property
    def grid(self):
        if self._adaptive:
            e = self._ends
            gx = ops.GridParamFn.apply(self.increments_x, e[0], e[1], self._mask_x_u8, self.initial_x_grid)
            gy = ops.GridParamFn.apply(self.increments_y, e[2], e[3], self._mask_y_u8, self.initial_y_grid)
            return gx, gy
        gx = torch.cat([self.x0, self.x_grid_inner, self.xN], dim=0)
        gy = torch.cat([self.y0, self.y_grid_inner, self.yN], dim=0)
        return (torch.where(self.boundary_mask_x, self.initial_x_grid, gx),
                torch.where(self.boundary_mask_y, self.initial_y_grid, gy))

    @property
    def u_full(self):
        if self.u_fixed is not None:
            return torch.where(self.node_mask, self.u_fixed.to(self.u.dtype), self.u)
        return self.u

    def forward(self, x_eval):
        gx, gy = self.grid
        return ops.RectQ4EvalFn.apply(gx, gy, self.u_full, x_eval)


# ============================================================================ triangular 2D
class NeumannEdgesWrapper:
    """``(x_i, x_j)`` node coordinates of Neumann edges (reference ``models.py:214-226``)."""

    def __init__(self, coords, edges):
        self.coords, self.edges = coords, edges

    def __getitem__(self, idx):
        return self.coords[self.edges[idx, 0]], self.coords[self.edges[idx, 1]]

    def __len__(self):
        return self.edges.shape[0]


class ConnectivityWrapper:
    """``coords[connectivity[idx]] -> [M,3,2]`` (reference ``models.py:228-238``)."""

    def __init__(self, coords, connectivity):
        self.coords, self.connectivity = coords, connectivity

    def __getitem__(self, idx):
        return self.coords[self.connectivity[idx]]

    def __len__(self):
        return self.connectivity.shape[0]


class TriangularShapeNN2D(PiecewiseLinearShapeNN2D):
    """Unstructured P1 triangles, vector field u in R^2, free node coordinates
    (r-adaptivity) and free nodal values as parameters (reference ``models.py:241-376``)."""

    def __init__(self, node_coords, connectivity, boundary_mask=None, dirichlet_mask=None, u_fixed=None,
                 neumann_edges=None):
        super().__init__()
        self.scale = 1e-5
        self.dim_u = 2
        # opt-in: "reference" = dN_dx = Jinv * dN_dxi exactly as models.py:351; "physical" = Jinv^T
        self.grad_convention = "reference"
        self.register_buffer("initial_node_coords", node_coords.clone())
        self.Nnodes = node_coords.shape[0]
        self.register_buffer("connectivity", connectivity.long().clone())
        self.Nelems = connectivity.shape[0]
        if boundary_mask is None:
            boundary_mask = torch.zeros(self.Nnodes, dtype=torch.bool, device=node_coords.device)
        if dirichlet_mask is None:
            dirichlet_mask = torch.zeros(self.Nnodes, dtype=torch.bool, device=node_coords.device)
        self.register_buffer("boundary_mask", boundary_mask.clone())
        free_mask = ~boundary_mask
        self.node_coords_free = nn.Parameter(node_coords[free_mask].clone())
        self.register_buffer("node_coords_fixed", node_coords[boundary_mask].clone())
        self.register_buffer("free_mask", free_mask)
        self.register_buffer("dirichlet_mask", dirichlet_mask.clone())
        u_free_mask = ~dirichlet_mask
        self.register_buffer("u_free_mask", u_free_mask)
        # same RNG call as upstream (models.py:274): identical u_free for an identical seed
        u0 = self.scale * torch.randn(int(u_free_mask.sum().item()), self.dim_u)
        self.u_free = nn.Parameter(u0.to(device=node_coords.device, dtype=node_coords.dtype))
        if u_fixed is not None:
            self.register_buffer("u_fixed", torch.as_tensor(u_fixed).to(node_coords.device))
        else:
            self.u_fixed = None
        if neumann_edges is not None:
            self.register_buffer("neumann_edges", neumann_edges.long().clone())
            self.N_edges = neumann_edges.shape[0]
        else:
            self.neumann_edges = None
            self.N_edges = 0
        # int32 index lists / maps (derived from the masks; not part of the state dict)
        fm, um = _np_mask(free_mask), _np_mask(u_free_mask)
        self._x_src, self._u_src = row_maps(fm), row_maps(um)
        i32 = dict(dtype=torch.int32)
        self.register_buffer("_idx_free", torch.tensor(np.nonzero(fm)[0], **i32), persistent=False)
        self.register_buffer("_idx_fixed", torch.tensor(np.nonzero(~fm)[0], **i32), persistent=False)
        self.register_buffer("_idx_ufree", torch.tensor(np.nonzero(um)[0], **i32), persistent=False)
        self.register_buffer("_idx_udir", torch.tensor(np.nonzero(~um)[0], **i32), persistent=False)
        self.register_buffer("_conn32", connectivity.to(torch.int32).contiguous(), persistent=False)
        e32 = (neumann_edges if neumann_edges is not None else torch.zeros((0, 2), dtype=torch.long))
        self.register_buffer("_edges32", e32.to(torch.int32).contiguous(), persistent=False)
        self._plans = {}

    # -- copy / pickle: the tile plans wrap ctypes handles (not picklable) and the Dirichlet-row cache is derived;
    #    both are rebuilt lazily, so copy.deepcopy(model) / torch.save(model) work as for the reference's plain Module
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_plans"] = {}
        state.pop("_ufix_cache", None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault("_plans", {})

    # -- reference attribute surface ------------------------------------------------------
    @property
    def device(self):
        return self.node_coords_free.device

    @property
    def dtype(self):
        return self.node_coords_free.dtype

    def u_fixed_rows(self):
        """Dirichlet rows ``[Ndir, 2]`` (``u[dirichlet_mask] = u_fixed`` broadcast, models.py:303-304)."""
        key = (str(self.device), self.dtype)
        hit = getattr(self, "_ufix_cache", None)
        if hit is not None and hit[0] == key and (self.u_fixed is None or hit[2] == self.u_fixed._version):
            return hit[1]
        nd = int(self._idx_udir.shape[0])
        if self.u_fixed is None or nd == 0:
            rows = torch.zeros((nd, self.dim_u), dtype=self.dtype, device=self.device)
        else:
            rows = torch.broadcast_to(self.u_fixed.to(device=self.device, dtype=self.dtype),
                                      (nd, self.dim_u)).contiguous()
        self._ufix_cache = (key, rows, None if self.u_fixed is None else self.u_fixed._version)
        return rows

    @property
    def coords(self):
        return ops.AssembleRowsFn.apply(self.node_coords_free, self.node_coords_fixed.to(self.dtype),
                                        self._idx_free, self._idx_fixed, self.Nnodes)

    @property
    def u_full(self):
        return ops.AssembleRowsFn.apply(self.u_free, self.u_fixed_rows(), self._idx_ufree, self._idx_udir,
                                        self.Nnodes)

    @property
    def domain_elements(self):
        return ConnectivityWrapper(self.coords, self.connectivity)

    @property
    def nm_edges(self):
        if self.neumann_edges is None:
            raise AttributeError("model was built without neumann_edges")
        return NeumannEdgesWrapper(self.coords, self.neumann_edges)

    # -- tile plan for the fused energy ----------------------------------------------------
    def tile_plan(self, tile_elems: int = 0, shards: int = 1) -> TilePlan:
        """The owner-computes tile plan of this mesh on this device (built once, cached).  ``shards``: ranks the tiles
        will be split over (``hidenn_fem_amd.sharded``): tiles sized for the elements per rank, boundary tiles first."""
        if self.device.type != "cuda":
            raise RuntimeError(f"hidenn_fem_amd: model is on {self.device}; the fused energy kernel needs a ROCm "
                               "device. There is no CPU fallback -- call model.to('cuda').")
        key = (str(self.device), int(tile_elems)) if shards == 1 else (str(self.device), int(tile_elems), int(shards))
        if key not in self._plans:
            self._plans[key] = TilePlan(self.connectivity, self.Nnodes, coords_hint=self.initial_node_coords,
                                        x_src=self._x_src, u_src=self._u_src, edges=self.neumann_edges,
                                        tile_elems=tile_elems, device=self.device,
                                        nodes_per_elem=getattr(self, "nodes_per_element", 3), shards=shards)
        return self._plans[key]

    # -- the (x_ref, element_id) forward contract -------------------------------------------
    def forward(self, x_eval, elem_id, edge=False):
        if not edge:
            return ops.Tri3EvalFn.apply(self.coords, self.u_full, self._conn32, x_eval, elem_id,
                                        1 if self.grad_convention == "physical" else 0)
        if self.neumann_edges is None:
            raise AttributeError("model was built without neumann_edges")
        return ops.Edge2EvalFn.apply(self.coords, self.u_full, self._edges32, x_eval[:, 0], elem_id)


class QuadShapeNN2D(TriangularShapeNN2D):
    """QUAD4-iso extension element: same construction API, parameters, buffers and
    attribute surface as the triangular model, ``connectivity [Ne,4]`` (local nodes CCW), reference
    square ``[-1,1]^2``.  ``forward(x_eval, elem_id)`` keeps the ``(x_ref, element_id)`` contract.  The
    fused energy runs on the same owner-computes tile plan as TRI3; there is no reference counterpart."""

    nodes_per_element = 4

    def forward(self, x_eval, elem_id, edge=False):
        if not edge:
            return ops.Quad4EvalFn.apply(self.coords, self.u_full, self._conn32, x_eval, elem_id)
        return super().forward(x_eval, elem_id, edge=True)

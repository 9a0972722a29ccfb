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


ROW_REORDER_MIN_NODES = 4096  # reorder="auto": meshes of this many nodes or more store their parameter rows TILE-MAJOR (smaller ones
                              # live in the caches and keep the reference's layout).  Measured against the numbering as given
                              # (us per launch, same buffers / rotating sets beyond the Infinity Cache): row-major structured
                              # T1M 9.12 / 11.8 -> 8.96 / 10.8, T2M 19.65 -> 18.75, random numbering (cfg5) 128 -> 34.6,
                              # Delaunay 40.6 -> 38.9; rows along the Hilbert curve ("hilbert"): T1M 9.38, cfg5 36.3


def row_line_factor(rows_along_curve: np.ndarray, chunk: int = 512, rows_per_line: int = 8) -> float:
    """Locality figure of a row numbering: walk the rows in the order of the locality curve (the order in which tiles
    gather them) in chunks of ``chunk`` and count the distinct 128-byte lines (8 rows of 16 bytes) a chunk touches,
    relative to the minimum ``chunk / 8``.  1.0 = rows stored along the curve; ~1.4 = row-major structured numbering;
    8.0 = random numbering (every 16-byte row gather pulls its own 128-byte line)."""
    n = (len(rows_along_curve) // chunk) * chunk
    if n == 0:
        return 1.0
    lines = np.sort((rows_along_curve[:n] // rows_per_line).reshape(-1, chunk), axis=1)
    distinct = 1 + (np.diff(lines, axis=1) != 0).sum(axis=1)
    return float(distinct.mean() / (chunk / rows_per_line))


def ordered_row_maps(mask: np.ndarray, idx_free: np.ndarray):
    """int32 map node -> row: ``idx_free[j]`` (node id) lives in row ``j`` of the free array; masked-out nodes map to
    ``-1 - k`` with ``k`` their rank among the masked-out nodes in the caller's numbering (``plan.row_maps`` when
    ``idx_free`` is ascending)."""
    src = np.empty(mask.shape[0], dtype=np.int32)
    src[idx_free] = np.arange(len(idx_free), dtype=np.int32)
    src[~mask] = -1 - np.arange(int((~mask).sum()), dtype=np.int32)
    return src


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
                 neumann_edges=None, reorder="auto"):
        """Reference signature (models.py:242-251) plus ``reorder``: how the rows of the two parameter tensors are
        STORED.  The reference hands over whatever numbering the mesher produced (mesh.py:136-144); a tile gathers
        16-byte rows and writes 16-byte gradient rows, so a numbering without locality costs up to 8x the read traffic
        and even a row-major one makes most of a tile's gradient stores partial 128-byte lines.  ``"auto"`` (default):
        meshes of >= 4096 nodes store ``node_coords_free`` / ``u_free`` TILE-MAJOR -- the rows of the nodes a tile of the
        default plan owns are one contiguous run (``"tile"`` forces it; costs one host-side plan build at construction);
        ``"hilbert"``: rows along the Hilbert curve of the initial coordinates; ``"off"``: the reference's layout, bit
        for bit.  ``row_line_factor`` reports the locality of the numbering as given (1 = along the curve, 1.4-1.5
        row-major structured, 8 = random).  Node numbering, ``connectivity``, masks, ``coords``, ``u_full``, ``forward`` and
        ``state_dict()`` stay in the CALLER's numbering either way; only the raw parameter tensors (and hence
        ``.grad`` and optimiser state) are in storage order -- ``to_caller_order`` / ``from_caller_order`` convert."""
        super().__init__()
        if reorder not in ("auto", "hilbert", "tile", "off"):
            raise ValueError("reorder must be 'auto', 'tile', 'hilbert' or 'off'")
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
        u_free_mask = ~dirichlet_mask
        fm, um = _np_mask(free_mask), _np_mask(u_free_mask)
        # ---- storage order of the parameter rows (see the docstring): node ids of the free rows, in row order
        idx_free, idx_ufree = np.nonzero(fm)[0], np.nonzero(um)[0]
        self.row_order, self.row_line_factor, self.plan_cache = "as given", None, None
        if reorder != "off" and self.Nnodes >= 2:
            from .mesh import _hilbert_keys
            curve = np.argsort(_hilbert_keys(node_coords.detach().cpu().double().numpy()), kind="stable")   # node ids along the curve
            cf, cu = curve[fm[curve]], curve[um[curve]]
            caller_row_x = np.cumsum(fm) - 1                     # row of node n in the caller's numbering
            self.row_line_factor = row_line_factor(caller_row_x[cf])
            if reorder == "hilbert":
                idx_free, idx_ufree = cf, cu
                self.row_order = "hilbert"
            elif reorder == "tile" or (reorder == "auto" and self.Nnodes >= ROW_REORDER_MIN_NODES):
                # TILE-MAJOR: the rows of the nodes a tile owns are one contiguous run, in the tile's local order (host-only
                # plan of the default tiling; hfem_plan_export 11): every gradient store covers whole 128-byte lines and the
                # gather of the owned rows is perfectly coalesced
                hp = TilePlan(connectivity, self.Nnodes, coords_hint=node_coords, edges=neumann_edges, device=None,
                              nodes_per_elem=connectivity.shape[1])
                tm = hp.export("owned_node_ids").astype(np.int64)
                self.plan_cache = hp.cache                    # "hit" / "miss" under $HFEM_PLAN_CACHE, else None
                hp.close()
                idx_free, idx_ufree = tm[fm[tm]], tm[um[tm]]
                self.row_order = "tile"
        t_free = torch.from_numpy(idx_free).to(node_coords.device)
        self.node_coords_free = nn.Parameter(node_coords[t_free].clone())
        self.register_buffer("node_coords_fixed", node_coords[boundary_mask].clone())
        self.register_buffer("free_mask", free_mask)
        self.register_buffer("dirichlet_mask", dirichlet_mask.clone())
        self.register_buffer("u_free_mask", u_free_mask)
        # same RNG call as upstream (models.py:274): identical u_free for an identical seed -- drawn in the caller's row
        # order, then stored in storage order
        u0 = self.scale * torch.randn(int(u_free_mask.sum().item()), self.dim_u)
        if self.row_order != "as given":
            u0 = u0[torch.from_numpy((np.cumsum(um) - 1)[idx_ufree])]
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
        # int32 index lists / maps (derived from the masks and the storage order; not part of the state dict)
        self._x_src, self._u_src = ordered_row_maps(fm, idx_free), ordered_row_maps(um, idx_ufree)
        i32 = dict(dtype=torch.int32)
        self.register_buffer("_idx_free", torch.tensor(idx_free, **i32), persistent=False)
        self.register_buffer("_idx_fixed", torch.tensor(np.nonzero(~fm)[0], **i32), persistent=False)
        self.register_buffer("_idx_ufree", torch.tensor(idx_ufree, **i32), persistent=False)
        self.register_buffer("_idx_udir", torch.tensor(np.nonzero(~um)[0], **i32), persistent=False)
        # storage row j <-> caller row perm[j] (None: the same order); state_dict() speaks the caller's order
        if self.row_order != "as given":
            self.register_buffer("_perm_x", torch.from_numpy((np.cumsum(fm) - 1)[idx_free]), persistent=False)
            self.register_buffer("_perm_u", torch.from_numpy((np.cumsum(um) - 1)[idx_ufree]), persistent=False)
            self.register_state_dict_post_hook(TriangularShapeNN2D._state_to_caller_order)
            self.register_load_state_dict_pre_hook(TriangularShapeNN2D._state_from_caller_order)
        else:
            self._perm_x = self._perm_u = None
        self._tag_parameters()
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
        self._tag_parameters()

    def _tag_parameters(self):
        """The row permutation travels WITH the parameter objects, so that an optimiser can convert its state without knowing
        the model (``hidenn_fem_amd.optim.caller_order_hooks``)."""
        self.node_coords_free._hfem_caller_perm = self._perm_x
        self.u_free._hfem_caller_perm = self._perm_u

    def attach_optimizer(self, optimizer):
        """``optimizer.state_dict()`` / ``load_state_dict()`` then speak the reference's row order (Adam moments, L-BFGS
        history): REQUIRED for any ``torch.optim`` optimiser whose checkpoints cross row orders -- written by the reference,
        by a ``reorder="off"`` model, or by a build with other tile defaults -- since a state dict in the wrong order loads
        without an error (same shapes).  ``FusedAdam`` / ``FusedLBFGS`` do it themselves.  Returns the optimiser."""
        from .optim import caller_order_hooks
        return caller_order_hooks(optimizer)

    def grad_in_caller_order(self):
        """``{"node_coords_free": grad, "u_free": grad}`` with rows as the reference indexes them
        (``node_coords[free_mask]`` order, models.py:260); ``None`` where no gradient exists.  The raw ``param.grad`` tensors
        are in STORAGE order (tile-major for meshes of >= 4096 nodes)."""
        out = {}
        for name, which in (("node_coords_free", "x"), ("u_free", "u")):
            g = getattr(self, name).grad
            out[name] = None if g is None else self.to_caller_order(g, which)
        return out

    # -- storage order of the parameter rows <-> the caller's numbering ----------------------
    def to_caller_order(self, t: torch.Tensor, which: str = "x") -> torch.Tensor:
        """A ``[rows, ...]`` tensor in the storage order of ``node_coords_free`` (``which="x"``) or ``u_free`` (``"u"``)
        -- the parameter itself, its ``.grad``, an optimiser moment -- as the reference would index it: row ``k`` = the
        ``k``-th free node in the caller's numbering (``node_coords[free_mask]`` order, models.py:260)."""
        p = self._perm_x if which == "x" else self._perm_u
        if p is None:
            return t
        out = torch.empty_like(t)
        out[p.to(t.device)] = t
        return out

    def from_caller_order(self, t: torch.Tensor, which: str = "x") -> torch.Tensor:
        """Inverse of ``to_caller_order``: rows in the caller's order -> storage order."""
        p = self._perm_x if which == "x" else self._perm_u
        return t if p is None else t[p.to(t.device)]

    @staticmethod
    def _state_to_caller_order(module, state_dict, prefix, local_metadata):
        for name, which in (("node_coords_free", "x"), ("u_free", "u")):
            if prefix + name in state_dict:
                state_dict[prefix + name] = module.to_caller_order(state_dict[prefix + name], which)

    @staticmethod
    def _state_from_caller_order(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for name, which in (("node_coords_free", "x"), ("u_free", "u")):
            key = prefix + name
            if key in state_dict and state_dict[key].shape == getattr(module, name).shape:
                state_dict[key] = module.from_caller_order(state_dict[key], which)

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

"""Synthetic mesh generators producing the reference's 6-tuple.

The reference builds its meshes with third-party meshers (``gmsh``, ``meshzoo``;
``/root/reference/src/mesh.py:8-153,155-276``) that are host-side, run once, and
are out of scope for the hot path.  What the hot path needs
is only their *output contract* (``mesh.py:261-276``)::

    node_coords [N,2] float, connectivity [Ne,3] int64, geom_boundary_mask [N] bool,
    bc_mask [N] bool (Dirichlet), mn_mask [N] bool (Neumann), neumann_edges [E,2] int64

with Neumann edges = index-sorted unique element edges whose two nodes are both
in ``mn_mask`` (``mesh.py:125-134,249-259``).  The generators below reproduce
that contract with numpy only (vectorised; no per-cell Python loop).  Mesh
*geometry* parity with gmsh/meshzoo is unpinned -- inputs to the hot path are
tensors.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

_DEFAULT_BOUNDARIES = {"up": 0, "down": 0, "right": 2, "left": 1}


def _neumann_edges(cells: np.ndarray, mn_mask: np.ndarray) -> np.ndarray:
    """Sorted unique element edges with both nodes Neumann (rule of mesh.py:249-259)."""
    if cells.shape[0] == 0 or not mn_mask.any():
        return np.zeros((0, 2), dtype=np.int64)
    e = np.vstack([cells[:, [0, 1]], cells[:, [1, 2]], cells[:, [2, 0]]])
    e = e[mn_mask[e].all(axis=1)]            # filter first: cheap at 10^6 elements
    e = np.unique(np.sort(e, axis=1), axis=0)
    return e.astype(np.int64)


def _face_masks(pts, length, height, boundaries, tol=1e-6):
    bc = np.zeros(len(pts), dtype=bool)
    mn = np.zeros(len(pts), dtype=bool)
    faces = {
        "up": np.abs(pts[:, 1] - height) < tol,
        "down": np.abs(pts[:, 1]) < tol,
        "left": np.abs(pts[:, 0]) < tol,
        "right": np.abs(pts[:, 0] - length) < tol,
    }
    for face, cond in boundaries.items():
        if face not in faces:
            continue
        if cond == 1:
            bc |= faces[face]
        elif cond == 2:
            mn |= faces[face]
    return bc, mn


def structured_tri_mesh(
    nx: int,
    ny: int,
    length: float = 2.0,
    height: float = 1.0,
    jitter: float = 0.0,
    seed: int = 0,
    diagonal: str = "fixed",
    boundaries: Optional[Dict[str, int]] = None,
    permute: bool = False,
    flip_fraction: float = 0.0,
    dtype: torch.dtype = torch.float32,
):
    """``nx x ny`` nodes on ``[0,length]x[0,height]``, every cell split into two
    CCW triangles -> ``2 (nx-1)(ny-1)`` TRI3 elements.

    Node id is ``i*ny + j`` (``indexing="ij"``); cell ``(i,j)`` has corners
    ``a=(i,j) b=(i+1,j) c=(i+1,j+1) d=(i,j+1)`` and is split ``(a,b,c),(a,c,d)``
    (``diagonal="fixed"``), alternately ``/`` and ``\\``
    (``"zigzag"``, what meshzoo's variant of that name produces, mesh.py:187) or
    at random per cell (``"random"``, SURVEY cfg5).

    ``jitter``: interior nodes moved by U(-jitter*h, jitter*h) per axis (h = cell
    size) -- keep < 0.5 so no element inverts.  ``permute``: random element
    permutation + random global node renumbering (worst-case locality, cfg5).
    ``flip_fraction``: that fraction of elements gets local order [1,0,2]
    (det < 0; exercises the ``abs(detJ)`` / sign terms).
    """
    rng = np.random.default_rng(seed)
    boundaries = dict(_DEFAULT_BOUNDARIES if boundaries is None else boundaries)
    xs = np.linspace(0.0, length, nx)
    ys = np.linspace(0.0, height, ny)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel()], axis=1)
    idx = np.arange(nx * ny, dtype=np.int64).reshape(nx, ny)

    geom = np.zeros((nx, ny), dtype=bool)
    geom[0, :] = geom[-1, :] = geom[:, 0] = geom[:, -1] = True
    geom = geom.ravel()
    if jitter > 0.0:
        hx, hy = length / (nx - 1), height / (ny - 1)
        d = rng.uniform(-jitter, jitter, size=pts.shape) * np.array([hx, hy])
        pts = pts + d * (~geom)[:, None]

    a, b = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel()
    c, d_ = idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    if diagonal == "fixed":
        alt = np.zeros(a.shape[0], dtype=bool)
    elif diagonal == "zigzag":
        ii, jj = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), indexing="ij")
        alt = ((ii + jj) % 2 == 1).ravel()
    elif diagonal == "random":
        alt = rng.random(a.shape[0]) < 0.5
    else:
        raise ValueError("diagonal must be 'fixed', 'zigzag' or 'random'")
    # diagonal a-c: (a,b,c),(a,c,d)   |   diagonal b-d: (a,b,d),(b,c,d)   (both CCW)
    t1 = np.where(alt[:, None], np.stack([a, b, d_], 1), np.stack([a, b, c], 1))
    t2 = np.where(alt[:, None], np.stack([b, c, d_], 1), np.stack([a, c, d_], 1))
    cells = np.empty((2 * a.shape[0], 3), dtype=np.int64)
    cells[0::2], cells[1::2] = t1, t2

    if flip_fraction > 0.0:
        flip = rng.random(cells.shape[0]) < flip_fraction
        cells[flip] = cells[flip][:, [1, 0, 2]]

    bc, mn = _face_masks(pts if jitter == 0.0 else np.stack([X.ravel(), Y.ravel()], 1),
                         length, height, boundaries)

    if permute:
        cells = cells[rng.permutation(cells.shape[0])]
        new_of_old = rng.permutation(pts.shape[0])
        old_of_new = np.argsort(new_of_old)
        pts, geom, bc, mn = pts[old_of_new], geom[old_of_new], bc[old_of_new], mn[old_of_new]
        cells = new_of_old[cells]

    edges = _neumann_edges(cells, mn)
    return (
        torch.tensor(pts, dtype=dtype),
        torch.tensor(cells, dtype=torch.long),
        torch.tensor(geom),
        torch.tensor(bc),
        torch.tensor(mn),
        torch.tensor(edges, dtype=torch.long),
    )


def structured_quad_mesh(nx: int, ny: int, length: float = 2.0, height: float = 1.0, jitter: float = 0.0,
                         seed: int = 0, boundaries: Optional[Dict[str, int]] = None,
                         dtype: torch.dtype = torch.float32):
    """``nx x ny`` nodes -> ``(nx-1)(ny-1)`` QUAD4 cells, local nodes CCW ``(a, b, c, d)`` with
    ``a=(i,j) b=(i+1,j) c=(i+1,j+1) d=(i,j+1)`` (node id ``i*ny + j``).  Same 6-tuple as the
    triangular generators, with ``connectivity [Ne,4]``.  QUAD4 is this build's extension element
    (the reference has none)."""
    rng = np.random.default_rng(seed)
    boundaries = dict(_DEFAULT_BOUNDARIES if boundaries is None else boundaries)
    xs, ys = np.linspace(0.0, length, nx), np.linspace(0.0, height, ny)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    pts0 = np.stack([X.ravel(), Y.ravel()], axis=1)
    idx = np.arange(nx * ny, dtype=np.int64).reshape(nx, ny)
    geom = np.zeros((nx, ny), dtype=bool)
    geom[0, :] = geom[-1, :] = geom[:, 0] = geom[:, -1] = True
    geom = geom.ravel()
    pts = pts0.copy()
    if jitter > 0.0:
        h = np.array([length / (nx - 1), height / (ny - 1)])
        pts = pts + rng.uniform(-jitter, jitter, size=pts.shape) * h * (~geom)[:, None]
    cells = np.stack([idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()], axis=1)
    bc, mn = _face_masks(pts0, length, height, boundaries)
    e = np.vstack([cells[:, [0, 1]], cells[:, [1, 2]], cells[:, [2, 3]], cells[:, [3, 0]]])
    e = e[mn[e].all(axis=1)]
    edges = np.unique(np.sort(e, axis=1), axis=0).astype(np.int64) if len(e) else np.zeros((0, 2), dtype=np.int64)
    return (torch.tensor(pts, dtype=dtype), torch.tensor(cells, dtype=torch.long), torch.tensor(geom),
            torch.tensor(bc), torch.tensor(mn), torch.tensor(edges, dtype=torch.long))


def generate_mesh(
    length: float = 2.0,
    height: float = 1.0,
    holes: List[Tuple[float, float, float]] = ((0.5, 0.7, 0.12), (1.0, 0.3, 0.15), (1.4, 0.6, 0.1)),
    boundaries: Dict[str, int] = _DEFAULT_BOUNDARIES,
    nx: int = 100,
    ny: int = 50,
):
    """Same call signature and output contract as the reference's structured
    mesher (``mesh.py:155-276``): zig-zag triangulated rectangle, nodes inside
    the circular ``holes`` removed, triangles touching a removed node dropped
    and their surviving nodes flagged as geometric boundary.  The kept-cell scan
    is vectorised (the reference loops over all cells in Python, mesh.py:208-215).
    """
    pts_t, cells_t, _, _, _, _ = structured_tri_mesh(
        nx, ny, length, height, diagonal="zigzag", boundaries={}, dtype=torch.float64)
    pts, cells = pts_t.numpy(), cells_t.numpy()
    keep = np.ones(len(pts), dtype=bool)
    for cx, cy, r in holes:
        keep &= (pts[:, 0] - cx) ** 2 + (pts[:, 1] - cy) ** 2 > r ** 2
    new_id = -np.ones(len(pts), dtype=np.int64)
    new_id[keep] = np.arange(int(keep.sum()))
    whole = keep[cells].all(axis=1)
    pts_k = pts[keep]
    geom = np.zeros(len(pts_k), dtype=bool)
    cut_nodes = cells[~whole].ravel()
    cut_nodes = cut_nodes[keep[cut_nodes]]
    geom[new_id[cut_nodes]] = True
    cells_k = new_id[cells[whole]]
    tol = 1e-6
    geom |= ((np.abs(pts_k[:, 0]) < tol) | (np.abs(pts_k[:, 0] - length) < tol)
             | (np.abs(pts_k[:, 1]) < tol) | (np.abs(pts_k[:, 1] - height) < tol))
    bc, mn = _face_masks(pts_k, length, height, boundaries, tol)
    edges = _neumann_edges(cells_k, mn)
    return (
        torch.tensor(pts_k, dtype=torch.float32),
        torch.tensor(cells_k, dtype=torch.long),
        torch.tensor(geom),
        torch.tensor(bc),
        torch.tensor(mn),
        torch.tensor(edges, dtype=torch.long),
    )


def generate_mesh_gmsh(*args, **kwargs):
    """The reference's gmsh-based mesher (``mesh.py:8-153``) is an optional
    third-party front-end outside the hot path; this build does not wrap gmsh."""
    raise ImportError(
        "generate_mesh_gmsh needs the gmsh python module, which this build does not "
        "bundle; use generate_mesh(...) or structured_tri_mesh(...) (same 6-tuple)."
    )

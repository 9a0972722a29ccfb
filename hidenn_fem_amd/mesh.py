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


def _hilbert_keys(pts: np.ndarray, bits: int = 16) -> np.ndarray:
    """Hilbert-curve position of every point (vectorised; one isotropic scale, as the tile planner uses)."""
    lo = pts.min(axis=0)
    span = max(float((pts.max(axis=0) - lo).max()), 1e-300)
    q = np.clip(((pts - lo) * ((2 ** bits - 1) / span)).astype(np.int64), 0, 2 ** bits - 1)
    x, y = q[:, 0].copy(), q[:, 1].copy()
    d = np.zeros(len(pts), dtype=np.int64)
    s_ = 1 << (bits - 1)
    full = 2 ** bits - 1
    while s_ > 0:
        rx = ((x & s_) > 0).astype(np.int64)
        ry = ((y & s_) > 0).astype(np.int64)
        d += s_ * s_ * ((3 * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, full - x, x)
        y = np.where(flip, full - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s_ >>= 1
    return d


def reorder_for_locality(mesh):
    """Locality renumbering of a mesh 6-tuple ("locality reordering ... decides LDS-tile hit
    rate"): nodes are renumbered along a Hilbert curve of their coordinates and the elements sorted along the curve
    of their centroids, so that the nodes of a tile are neighbours in memory (a tile's 16-byte row gathers then share
    cache lines instead of touching one line per row).  The energy and its gradients do not depend on the global
    numbering (only on each element's LOCAL node order, which is kept -- SURVEY F4).  Works for TRI3 and QUAD4
    connectivities.  Returns ``(mesh6_reordered, new_of_old)`` with ``new_of_old[n]`` = new id of old node ``n``:
    ``coords_new[new_of_old] == coords_old``."""
    coords, conn, geom, bc, mn, edges = mesh
    pts = coords.detach().cpu().double().numpy()
    order = np.argsort(_hilbert_keys(pts), kind="stable")            # old ids in new order
    new_of_old = np.empty(len(pts), dtype=np.int64)
    new_of_old[order] = np.arange(len(pts))
    cn = new_of_old[conn.cpu().numpy()]
    cen = pts[conn.cpu().numpy()].mean(axis=1)
    cn = cn[np.argsort(_hilbert_keys(cen), kind="stable")]
    ed = new_of_old[edges.cpu().numpy()] if edges is not None and edges.numel() else np.zeros((0, 2), dtype=np.int64)
    if len(ed):
        ed = np.unique(np.sort(ed, axis=1), axis=0)                   # index-sorted unique, the rule of mesh.py:125-134
    idx = torch.from_numpy(order)
    out = (coords[idx], torch.from_numpy(cn), geom[idx], bc[idx], mn[idx], torch.from_numpy(ed.astype(np.int64)))
    return out, torch.from_numpy(new_of_old)


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


def unstructured_tri_mesh(
    n_points: int = 20000,
    length: float = 2.0,
    height: float = 1.0,
    holes: List[Tuple[float, float, float]] = ((0.5, 0.7, 0.12), (1.0, 0.3, 0.15), (1.4, 0.6, 0.1)),
    boundaries: Optional[Dict[str, int]] = None,
    grading: float = 0.4,
    grading_reach: float = 2.0,
    smooth: int = 0,
    seed: int = 0,
    dtype: torch.dtype = torch.float32,
):
    """Genuinely unstructured triangulation of the rectangle minus circular holes -- the geometry of the
    reference's gmsh mesher (``mesh.py:8-153``: OCC rectangle cut by disks) -- as a Delaunay triangulation
    (scipy / Qhull) of graded random points:

    * boundary points on the four sides (spacing h) and on every hole circle (spacing ``grading * h``);
    * interior points drawn with density 1 / h(x)^2, where the size field h(x) grows linearly from
      ``grading * h`` on a hole boundary to ``h`` at ``grading_reach`` hole radii from it (rejection sampling
      of a jittered fine grid; ``smooth`` Laplacian passes optionally relax them);
    * triangles whose centroid lies in a hole are dropped, every triangle is made CCW, unused points removed.

    Variable node valence (typically 3..11), slivers and graded element sizes -- what a structured split never
    shows the tile planner.  Same 6-tuple and BC-mask rules as ``mesh.py:97-134``: geometric boundary = outer
    rectangle + hole circles, ``boundaries`` faces 1 -> Dirichlet, 2 -> Neumann, Neumann edges = sorted unique
    element edges with both nodes Neumann.  ``n_points`` is a target (the result is within a few per cent)."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    boundaries = dict(_DEFAULT_BOUNDARIES if boundaries is None else boundaries)
    holes = [tuple(float(v) for v in hl) for hl in holes]
    g = float(min(max(grading, 0.05), 1.0))

    def size_ratio(p):                       # h(x) / h  in [g, 1]
        r_ = np.ones(len(p))
        for cx, cy, rad in holes:
            d = np.hypot(p[:, 0] - cx, p[:, 1] - cy) - rad
            r_ = np.minimum(r_, g + (1.0 - g) * np.clip(d / (grading_reach * rad), 0.0, 1.0))
        return r_

    def inside_hole(p, margin=0.0):
        m = np.zeros(len(p), dtype=bool)
        for cx, cy, rad in holes:
            m |= np.hypot(p[:, 0] - cx, p[:, 1] - cy) < rad + margin
        return m

    # mean density factor E[1/ratio^2] over the domain (Monte Carlo) fixes h for the requested point count
    probe = rng.random((20000, 2)) * [length, height]
    probe = probe[~inside_hole(probe)]
    area = length * height * len(probe) / 20000.0
    dens = float(np.mean(1.0 / size_ratio(probe) ** 2))
    h = np.sqrt(area * dens / max(n_points, 16))          # the thinned jittered grid leaves one point per h(x)^2

    # ---- boundary points
    def side(n):
        return np.linspace(0.0, 1.0, max(int(round(n)), 2) + 1)[:-1]
    nxs, nys = max(int(round(length / h)), 2), max(int(round(height / h)), 2)
    tx, ty = side(nxs), side(nys)
    outer = np.concatenate([
        np.stack([tx * length, np.zeros_like(tx)], 1), np.stack([np.full_like(ty, length), ty * height], 1),
        np.stack([length - tx * length, np.full_like(tx, height)], 1), np.stack([np.zeros_like(ty), height - ty * height], 1)])
    rings = []
    for cx, cy, rad in holes:
        nh = max(int(round(2.0 * np.pi * rad / (g * h))), 8)
        a = 2.0 * np.pi * (np.arange(nh) + rng.random()) / nh
        rings.append(np.stack([cx + rad * np.cos(a), cy + rad * np.sin(a)], 1))
    # ---- interior points: jittered grid at the finest spacing, thinned to the size field
    hf = g * h
    gx, gy = np.arange(0.5 * hf, length, hf), np.arange(0.5 * hf, height, hf)
    X, Y = np.meshgrid(gx, gy, indexing="ij")
    cand = np.stack([X.ravel(), Y.ravel()], 1) + rng.uniform(-0.35, 0.35, size=(X.size, 2)) * hf
    ratio = size_ratio(cand)
    keep = rng.random(len(cand)) < (g / ratio) ** 2
    keep &= ~inside_hole(cand, margin=0.6 * g * h)                            # clear of the hole rings
    keep &= (cand[:, 0] > 0.6 * h * ratio) & (cand[:, 0] < length - 0.6 * h * ratio)
    keep &= (cand[:, 1] > 0.6 * h * ratio) & (cand[:, 1] < height - 0.6 * h * ratio)
    inner = cand[keep]
    pts = np.concatenate([outer] + rings + [inner])
    n_fixed = len(pts) - len(inner)

    def triangulate(p):
        tri = Delaunay(p).simplices.astype(np.int64)
        cen = p[tri].mean(axis=1)
        tri = tri[~inside_hole(cen)]
        a_, b_, c_ = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
        area2 = (b_[:, 0] - a_[:, 0]) * (c_[:, 1] - a_[:, 1]) - (b_[:, 1] - a_[:, 1]) * (c_[:, 0] - a_[:, 0])
        tri = tri[np.abs(area2) > 1e-12 * h * h]                              # flat triangles on collinear boundary points
        area2 = area2[np.abs(area2) > 1e-12 * h * h]
        neg = area2 < 0
        tri[neg] = tri[neg][:, [0, 2, 1]]                                     # counter-clockwise
        return tri

    cells = triangulate(pts)
    for _ in range(int(smooth)):                                              # Laplacian relaxation of the interior points
        acc = np.zeros_like(pts)
        cnt = np.zeros(len(pts))
        for a_, b_ in ((0, 1), (1, 2), (2, 0)):
            np.add.at(acc, cells[:, a_], pts[cells[:, b_]]); np.add.at(cnt, cells[:, a_], 1.0)
            np.add.at(acc, cells[:, b_], pts[cells[:, a_]]); np.add.at(cnt, cells[:, b_], 1.0)
        mv = np.arange(len(pts)) >= n_fixed
        mv &= cnt > 0
        pts[mv] = 0.5 * pts[mv] + 0.5 * acc[mv] / cnt[mv, None]
        cells = triangulate(pts)
    used = np.zeros(len(pts), dtype=bool)
    used[cells.ravel()] = True
    if not used.all():
        new_id = -np.ones(len(pts), dtype=np.int64)
        new_id[used] = np.arange(int(used.sum()))
        pts, cells = pts[used], new_id[cells]
    tol = 1e-6
    geom = ((np.abs(pts[:, 0]) < tol) | (np.abs(pts[:, 0] - length) < tol)
            | (np.abs(pts[:, 1]) < tol) | (np.abs(pts[:, 1] - height) < tol))
    for cx, cy, rad in holes:                                                 # mesh.py:90-95
        geom |= np.abs(np.hypot(pts[:, 0] - cx, pts[:, 1] - cy) - rad) < tol
    bc, mn = _face_masks(pts, length, height, boundaries, tol)
    edges = _neumann_edges(cells, mn)
    return (torch.tensor(pts, dtype=dtype), torch.tensor(cells, dtype=torch.long), torch.tensor(geom),
            torch.tensor(bc), torch.tensor(mn), torch.tensor(edges, dtype=torch.long))


def generate_mesh_gmsh(length: float = 2.0, height: float = 1.0,
                       holes: List[Tuple[float, float, float]] = ((0.5, 0.7, 0.12), (1.0, 0.3, 0.15), (1.4, 0.6, 0.1)),
                       boundaries: Dict[str, int] = _DEFAULT_BOUNDARIES, lc: float = 0.02):
    """Call signature and output contract of the reference's gmsh front-end (``mesh.py:8-153``,
    ``examples/example4.py:26``): rectangle minus disks, characteristic length ``lc``, unstructured triangles.
    gmsh itself is a third-party mesher outside the hot path and is not bundled; the triangulation comes from
    ``unstructured_tri_mesh`` (Delaunay of graded points, element size ~ ``lc``, finer at the holes) -- the same
    6-tuple and BC-mask rules, not gmsh's node positions (mesh parity unpinned)."""
    area = length * height - sum(np.pi * r * r for _, _, r in holes)
    n_points = int(area / (np.sqrt(3.0) / 2.0 * lc * lc))
    return unstructured_tri_mesh(n_points, length, height, holes, boundaries, grading=0.5, smooth=2, seed=0,
                                 dtype=torch.float32)

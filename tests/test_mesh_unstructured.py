"""The genuinely unstructured mesher (hidenn_fem_amd.mesh.unstructured_tri_mesh: Delaunay of graded random points in
the rectangle minus disks -- the geometry of the reference's gmsh front-end, /root/reference/src/mesh.py:8-153) and
what it shows the tile planner that structured splits never do: valence >= 10 fans, slivers, graded element sizes,
and the kMaxLocal (1024 local nodes) retry path.  CPU only."""
import numpy as np
import pytest
import torch

from hidenn_fem_amd.mesh import generate_mesh_gmsh, unstructured_tri_mesh
from hidenn_fem_amd.plan import TilePlan, row_maps
from oracle import closed_form as CF
from test_plan_host import check_invariants, emulate

HOLES = ((0.5, 0.7, 0.12), (1.0, 0.3, 0.15), (1.4, 0.6, 0.1))       # examples/example4.py:17


def _geometry(pts, cells):
    a, b, c = pts[cells[:, 0]], pts[cells[:, 1]], pts[cells[:, 2]]
    area2 = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])

    def ang(u, v, w):
        e1, e2 = v - u, w - u
        return np.arccos(np.clip((e1 * e2).sum(1) / np.linalg.norm(e1, axis=1) / np.linalg.norm(e2, axis=1), -1, 1))
    min_angle = np.degrees(np.minimum(np.minimum(ang(a, b, c), ang(b, c, a)), ang(c, a, b)))
    return area2, min_angle


def test_unstructured_mesh_contract_and_shape():
    pts_t, cells_t, geom, bc, mn, edges = unstructured_tri_mesh(20000, seed=1, dtype=torch.float64)
    pts, cells = pts_t.numpy(), cells_t.numpy()
    assert cells_t.dtype == torch.long and edges.dtype == torch.long and geom.dtype == torch.bool
    assert abs(len(pts) - 20000) < 0.08 * 20000
    assert len(np.unique(cells)) == len(pts)                                   # no unused node
    area2, min_angle = _geometry(pts, cells)
    assert (area2 > 0).all()                                                   # every element counter-clockwise
    want_area = 2.0 - np.pi * sum(r * r for _, _, r in HOLES)
    assert abs(area2.sum() / 2 - want_area) < 2e-3 * want_area                 # polygonal holes: slightly less than pi r^2 removed
    e = np.unique(np.sort(np.vstack([cells[:, [0, 1]], cells[:, [1, 2]], cells[:, [2, 0]]]), axis=1), axis=0)
    assert len(pts) - len(e) + len(cells) == 1 - len(HOLES)                    # Euler: a plate with three holes
    valence = np.bincount(cells.ravel(), minlength=len(pts))
    assert valence.max() >= 10 and valence.min() >= 1                          # fans the structured split never has (<= 8)
    assert min_angle.min() < 10.0                                              # slivers
    assert np.sqrt(area2.max() / np.percentile(area2, 1)) > 4.0                # graded sizes (finer at the holes)
    # BC masks, rules of mesh.py:97-134
    g = geom.numpy()
    on_rect = (np.abs(pts[:, 0]) < 1e-6) | (np.abs(pts[:, 0] - 2.0) < 1e-6) | (np.abs(pts[:, 1]) < 1e-6) | (np.abs(pts[:, 1] - 1.0) < 1e-6)
    on_hole = np.zeros(len(pts), dtype=bool)
    for cx, cy, r in HOLES:
        on_hole |= np.abs(np.hypot(pts[:, 0] - cx, pts[:, 1] - cy) - r) < 1e-6
    assert np.array_equal(g, on_rect | on_hole) and on_hole.sum() > 50
    assert np.array_equal(bc.numpy(), np.abs(pts[:, 0]) < 1e-6) and np.array_equal(mn.numpy(), np.abs(pts[:, 0] - 2.0) < 1e-6)
    ed = edges.numpy()
    assert (ed[:, 0] < ed[:, 1]).all() and mn.numpy()[ed].all() and len(np.unique(ed, axis=0)) == len(ed)
    assert len(ed) == int(mn.sum()) - 1                                        # the right side is one chain of edges
    # boundary nodes are only touched by boundary-adjacent elements: interior nodes are closed fans
    interior = ~g
    edge_count = np.bincount(e.ravel(), minlength=len(pts))
    assert (edge_count[interior] == valence[interior]).all()                   # closed fan: #edges == #elements at the node


def test_gmsh_signature_front_end():
    out = generate_mesh_gmsh(2.0, 1.0, list(HOLES), {"up": 0, "down": 0, "right": 2, "left": 1}, lc=0.04)
    pts, cells, geom, bc, mn, edges = out
    assert pts.dtype == torch.float32 and cells.shape[1] == 3 and edges.shape[1] == 2       # mesh.py:139-151
    area2, min_angle = _geometry(pts.double().numpy(), cells.numpy())
    assert (area2 > 0).all() and min_angle.min() > 5.0                          # two smoothing passes: no slivers left
    h = np.sqrt(np.median(area2) / 2 * 4 / np.sqrt(3))
    assert 0.4 * 0.04 < h < 1.6 * 0.04


@pytest.mark.parametrize("seed,tile_elems", [(0, 0), (3, 600), (5, 4000)])
def test_tile_plan_on_unstructured_mesh(seed, tile_elems):
    """Plan invariants + a numpy emulation of the tiled kernel against the C closed form on a graded Delaunay mesh
    with free / fixed row maps.  tile_elems = 4000 cannot fit 1024 local nodes: the planner must retry smaller."""
    pts_t, cells_t, geom, bc, mn, edges_t = unstructured_tri_mesh(9000, seed=seed, dtype=torch.float64)
    X, conn, edges = pts_t.numpy(), cells_t.numpy(), edges_t.numpy()
    nn = X.shape[0]
    plan = TilePlan(conn, nn, coords_hint=X, edges=edges, tile_elems=tile_elems, device=None)
    st = plan.stats
    assert st["max_tile_nodes"] <= 1024
    if tile_elems == 4000:
        assert st["tile_elems"] < 4000                                          # the kMaxLocal retry shrank the tiles
    a = check_invariants(conn, edges, nn, plan)
    U = 1e-5 * np.random.default_rng(seed).standard_normal(X.shape)
    mat, W, Tc = CF.plane_stress(), 0.25, np.array([2e5, 0.0, 0.0, 0.0])
    e_ref, gX_ref, gU_ref = CF.tri3_energy(X, U, conn, mat, W)
    e_ref -= CF.edge2_energy(X, U, edges, Tconst=Tc, gX=gX_ref, gU=gU_ref)
    loss, gX, gU = emulate(a, X, U, mat, W, None, Tc)
    assert abs(loss - e_ref) <= 1e-12 * abs(e_ref)
    assert np.abs(gX - gX_ref).max() <= 1e-10 * np.abs(gX_ref).max()
    assert np.abs(gU - gU_ref).max() <= 1e-10 * np.abs(gU_ref).max()
    # halo factors stay moderate on a graded mesh (the planner's locality curve works on centroids, not on a grid)
    assert st["tile_elem_total"] < 1.6 * conn.shape[0] and st["tile_node_total"] < 1.9 * nn
    plan.close()


def test_reorder_for_locality_keeps_the_mesh_and_makes_tiles_contiguous():
    """mesh.reorder_for_locality: a pure renumbering (same geometry, same local node order per element, same BC sets,
    same energy by the C closed form), after which a tile's owned nodes are (nearly) one contiguous id range."""
    from hidenn_fem_amd.mesh import reorder_for_locality, structured_tri_mesh
    mesh = structured_tri_mesh(61, 47, jitter=0.3, seed=5, diagonal="random", permute=True, flip_fraction=0.2, dtype=torch.float64)
    (c2, cn2, g2, b2, m2, e2), new_of_old = reorder_for_locality(mesh)
    c, cn, g, b, m, e = mesh
    assert torch.equal(c2[new_of_old], c) and torch.equal(g2[new_of_old], g) and torch.equal(b2[new_of_old], b)
    # same elements with the same LOCAL order (as a multiset of coordinate triples in order)
    key = lambda cc, co: np.sort(np.ascontiguousarray(cc.numpy()[co.numpy()].reshape(len(co), -1)).view([("", "f8")] * 6).ravel())
    assert np.array_equal(key(c, cn), key(c2, cn2))
    ed2 = e2.numpy()
    assert (ed2[:, 0] < ed2[:, 1]).all() and m2.numpy()[ed2].all() and len(ed2) == len(e)
    U = 1e-4 * np.random.default_rng(0).standard_normal(c.shape)
    U2 = np.empty_like(U)
    U2[new_of_old.numpy()] = U
    mat = CF.plane_stress()
    e_a, gXa, gUa = CF.tri3_energy(c.numpy(), U, cn.numpy(), mat, 0.25)
    e_b, gXb, gUb = CF.tri3_energy(c2.numpy(), U2, cn2.numpy(), mat, 0.25)
    assert abs(e_a - e_b) <= 1e-12 * abs(e_a)
    assert np.abs(gXb[new_of_old.numpy()] - gXa).max() <= 1e-10 * np.abs(gXa).max()
    # locality: span of owned node ids per tile
    def spans(cc, co):
        p = TilePlan(co, cc.shape[0], coords_hint=cc, tile_elems=600, device=None)
        td, ns = p.export("tile_desc"), p.export("node_src")
        out = [np.ptp(ns[no:no + nown, 0]) / max(nown, 1) for (_, _, no, _, nown, _, _, _) in td if nown > 8]
        p.close()
        return float(np.median(out))
    s_after, s_before = spans(c2, cn2), spans(c, cn)      # id span per owned node of a tile
    assert s_after < 6.0 and s_before > 2.0 * s_after, (s_after, s_before)

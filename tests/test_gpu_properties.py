"""Size-independent properties of the fused energy kernel on MI355X (run with -m gpu):
invariance to element order and to global node renumbering, NON-invariance to the local node
order of an element (the reference's J^-1 . D_N convention), finite-difference
gradient check, additivity over tile ranges at full size, and the unstructured 'cfg5-like'
4M-element mesh (random diagonals, random element permutation + node renumbering) against the
plain-C oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _energy(coords, conn, geom, bc, edges, u_free=None, seed=0, tile_elems=0):
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = torch.device("cuda:0")
    torch.manual_seed(seed)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                 neumann_edges=edges).to(d)
    if u_free is not None:
        with torch.no_grad():
            m.u_free.copy_(u_free.to(d))
    lf = EnergyLoss2D(device=d, dtype=F64, tile_elems=tile_elems)
    loss = lf(m)
    loss.backward()
    return m, loss.item()


def _full(m, which):
    """Scatter a free-row gradient back to full node numbering (zeros on fixed rows)."""
    g = torch.zeros(m.Nnodes, 2, dtype=F64)
    if which == "x":
        g[m.free_mask.cpu()] = m.node_coords_free.grad.cpu()
    else:
        g[m.u_free_mask.cpu()] = m.u_free.grad.cpu()
    return g.numpy()


def test_invariance_to_element_order_and_node_renumbering():
    from hidenn_fem_amd.mesh import structured_tri_mesh
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(61, 37, jitter=0.25, seed=7, dtype=F64)
    rng = np.random.default_rng(0)
    m0, l0 = _energy(coords, conn, geom, bc, edges)
    u_full = torch.zeros(coords.shape[0], 2, dtype=F64)
    u_full[~bc] = m0.u_free.detach().cpu()
    # (1) permute the elements
    perm = torch.from_numpy(rng.permutation(conn.shape[0]))
    m1, l1 = _energy(coords, conn[perm], geom, bc, edges, u_free=u_full[~bc])
    assert abs(l1 - l0) <= 1e-12 * abs(l0)
    assert np.abs(_full(m1, "x") - _full(m0, "x")).max() <= 1e-10 * np.abs(_full(m0, "x")).max()
    # (2) renumber the nodes globally (connectivity, masks, edges, coords and u follow)
    new_of_old = torch.from_numpy(rng.permutation(coords.shape[0]))
    old_of_new = torch.argsort(new_of_old)
    e2 = new_of_old[edges]
    e2 = torch.sort(e2, dim=1).values                      # edges stay index-sorted (mesh.py:130,255) ...
    # ... but sorting swaps the (i, j) roles of an edge, and the reference's raw-Legendre xi (F3) is
    # not symmetric in i<->j: only compare when the traction work is symmetric -> drop the edges here
    none = torch.zeros((0, 2), dtype=torch.long)
    m0b, l0b = _energy(coords, conn, geom, bc, none, u_free=u_full[~bc])
    m2, l2 = _energy(coords[old_of_new], new_of_old[conn], geom[old_of_new], bc[old_of_new], none,
                     u_free=u_full[old_of_new][~bc[old_of_new]])
    assert abs(l2 - l0b) <= 1e-12 * abs(l0b)
    gx0, gx2 = _full(m0b, "x"), _full(m2, "x")
    assert np.abs(gx2[new_of_old.numpy()] - gx0).max() <= 1e-10 * np.abs(gx0).max()
    gu0, gu2 = _full(m0b, "u"), _full(m2, "u")
    assert np.abs(gu2[new_of_old.numpy()] - gu0).max() <= 1e-10 * np.abs(gu0).max()
    # (3) the local node order of an element is NOT a symmetry of the reference energy (F4)
    flipped = conn.clone()
    flipped[::2] = flipped[::2][:, [1, 2, 0]]              # cyclic shift keeps det > 0
    m3, l3 = _energy(coords, flipped, geom, bc, none, u_free=u_full[~bc])
    assert abs(l3 - l0b) > 1e-6 * abs(l0b)


def test_finite_difference_gradients():
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.loss import EnergyLoss2D
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(13, 9, jitter=0.2, seed=2, dtype=F64)
    m, l0 = _energy(coords, conn, geom, bc, edges)
    with torch.no_grad():
        m.u_free.mul_(1e3)                                   # make strain energy and traction work comparable
    m.zero_grad()
    lf = EnergyLoss2D(device=torch.device("cuda:0"), dtype=F64)
    lf(m).backward()
    rng = np.random.default_rng(1)
    for p, h in ((m.u_free, 1e-9), (m.node_coords_free, 1e-7)):
        g = p.grad.clone()
        for _ in range(6):
            i, j = int(rng.integers(p.shape[0])), int(rng.integers(2))
            with torch.no_grad():
                p[i, j] += h
                lp = lf(m).item()
                p[i, j] -= 2 * h
                lm = lf(m).item()
                p[i, j] += h
            fd = (lp - lm) / (2 * h)
            assert abs(fd - g[i, j].item()) <= 1e-5 * max(abs(g[i, j].item()), g.abs().max().item() * 1e-3)


def test_cfg5_like_unstructured_4m_elements_vs_oracle():
    """BASELINE config 5 shape: 2001x1001 nodes -> 4,000,000 TRI3, jitter 0.3, random diagonals, random
    element permutation + global node renumbering (worst-case locality); C-ABI called directly."""
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.plan import TilePlan
    from oracle import closed_form as CF
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(2001, 1001, jitter=0.3, seed=11, diagonal="random",
                                                            permute=True, dtype=F64)
    X = coords.numpy()
    U = 1e-5 * np.random.default_rng(3).standard_normal(X.shape)
    mat, W, Tc = CF.plane_stress(), 0.25, np.array([2e5, 0.0, 0.0, 0.0])
    e_ref, gX_ref, gU_ref = CF.tri3_energy(X, U, conn.numpy(), mat, W)
    e_ref -= CF.edge2_energy(X, U, edges.numpy(), Tconst=Tc, gX=gX_ref, gU=gU_ref)
    plan = TilePlan(conn, X.shape[0], coords_hint=X, edges=edges, device=d)
    assert plan.stats["max_tile_nodes"] <= 1024
    L = _lib.lib()
    dv = lambda a: (C.c_double * len(a))(*a)
    Xd, Ud = torch.from_numpy(X).to(d), torch.from_numpy(U).to(d)
    s = _lib.stream_ptr(d)
    loss = torch.full((), 3.0, dtype=F64, device=d)
    gX, gU = torch.full_like(Xd, float("nan")), torch.full_like(Ud, float("nan"))
    _lib.check(L.hfem_tri3_energy_plan(plan.handle, Xd.data_ptr(), None, Ud.data_ptr(), None, dv(mat), W, dv([0.0] * 6),
                                       None, dv(Tc), 0, -1, loss.data_ptr(), gX.data_ptr(), gU.data_ptr(), 0, s))
    assert abs(loss.item() - e_ref) <= 1e-12 * abs(e_ref)
    assert (gX.cpu().numpy() - gX_ref).__abs__().max() <= 1e-10 * np.abs(gX_ref).max()
    assert (gU.cpu().numpy() - gU_ref).__abs__().max() <= 1e-10 * np.abs(gU_ref).max()
    # additivity over 8 tile ranges (the 8-GPU element sharding), rows written exactly once overall
    acc = 0.0
    cover = torch.zeros(X.shape[0], dtype=torch.int32, device=d)
    for r in range(8):
        lo, hi = plan.shard_range(r, 8)
        l = torch.zeros((), dtype=F64, device=d)
        gx, gu = torch.full_like(Xd, float("nan")), torch.full_like(Ud, float("nan"))
        _lib.check(L.hfem_tri3_energy_plan(plan.handle, Xd.data_ptr(), None, Ud.data_ptr(), None, dv(mat), W,
                                           dv([0.0] * 6), None, dv(Tc), lo, hi, l.data_ptr(), gx.data_ptr(),
                                           gu.data_ptr(), 0, s))
        acc += l.item()
        cover += (~torch.isnan(gx[:, 0])).int()
    assert abs(acc - e_ref) <= 1e-12 * abs(e_ref)
    assert int(cover.min()) == 1 and int(cover.max()) == 1
    plan.close()


@pytest.mark.gpu
def test_fp32_model_takes_the_float_storage_kernel_and_matches_fp64_arithmetic():
    """An fp32 model (the reference's default dtype) runs the float-row instance of the tiled kernel: rows are widened on
    load, all arithmetic is fp64, gradients are rounded once on store -- so it must equal the fp64 path evaluated on the
    same float values, rounded to fp32."""
    import copy
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(201, 151, jitter=0.25, seed=8, flip_fraction=0.2, dtype=torch.float32)
    torch.manual_seed(4)
    m32 = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.3, neumann_edges=edges).to(d)
    with torch.no_grad():
        m32.u_free.mul_(50.0)
    assert m32.node_coords_free.dtype == torch.float32
    m64 = copy.deepcopy(m32).double()
    lf32, lf64 = EnergyLoss2D(device=d, dtype=torch.float32, arithmetic="fp64"), EnergyLoss2D(device=d, dtype=torch.float64)
    # an fp32 loss object carries the reference's fp32-rounded constants (C, quadrature sums); same numbers for both
    lf64._mat, lf64._W, lf64._ci, lf64._cj = lf32._mat, lf32._W, lf32._ci, lf32._cj
    l32 = lf32(m32)
    l64 = lf64(m64)
    l32.backward()
    l64.backward()
    assert l32.dtype == torch.float32 and m32.u_free.grad.dtype == torch.float32
    assert l32.item() == l64.float().item()
    for a, b in ((m32.node_coords_free.grad, m64.node_coords_free.grad), (m32.u_free.grad, m64.u_free.grad)):
        want = b.float()
        ulp = torch.finfo(torch.float32).eps * want.abs().clamp_min(1e-30)
        assert ((a - want).abs() <= 1.01 * ulp).all()
        assert (a == want).float().mean().item() > 0.99          # differences only where fp64 atomics order flips a rounding
    # scaled upstream gradient and a tile sub-range go through the same entry point
    m32.zero_grad()
    (3.0 * lf32(m32)).backward()
    want = 3.0 * m64.u_free.grad.float()
    assert ((m32.u_free.grad - want).abs() <= 4 * torch.finfo(torch.float32).eps * want.abs().clamp_min(1e-30)).all()


@pytest.mark.gpu
def test_cfg5_genuinely_unstructured_4m_elements_vs_oracle():
    """BASELINE config 5 ("4 M-element unstructured triangular mesh") on a REAL unstructured mesh: Delaunay of ~2 M
    graded random points in the rectangle minus three disks (the reference's example-4 geometry, mesh.py:8-153 /
    example4.py:17-26): valence up to ~13, slivers, element sizes graded 10:1.  Through the model API (free / fixed
    row maps: outer rectangle + hole rings fixed, left edge Dirichlet, right edge Neumann) against the C closed
    forms on the assembled arrays."""
    from hidenn_fem_amd.mesh import unstructured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from oracle import closed_form as CF
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = unstructured_tri_mesh(2_050_000, seed=2, dtype=F64)
    ne, nn = conn.shape[0], coords.shape[0]
    assert 3.9e6 < ne < 4.3e6
    valence = np.bincount(conn.numpy().ravel(), minlength=nn)
    assert valence.max() >= 11
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                 neumann_edges=edges).to(d)
    lf = EnergyLoss2D(E=10e9, nu=0.3, device=d, dtype=F64)
    loss = lf.value_and_grad_(m)
    st = m.tile_plan(lf.tile_elems).stats
    assert st["max_tile_nodes"] <= 1024
    # oracle on the assembled arrays
    X = coords.numpy()
    U = np.zeros_like(X)
    U[~bc.numpy()] = m.to_caller_order(m.u_free.detach(), "u").cpu().numpy()      # the model may store its rows along the curve
    mat, W = CF.plane_stress(), lf._W
    Tc = np.array([lf._ci * 1e5, 0.0, lf._cj * 1e5, 0.0])
    e_ref, gX_ref, gU_ref = CF.tri3_energy(X, U, conn.numpy(), mat, W)
    e_ref -= CF.edge2_energy(X, U, edges.numpy(), Tconst=Tc, gX=gX_ref, gU=gU_ref)
    assert abs(loss.item() - e_ref) <= 1e-12 * abs(e_ref)
    gx = m.to_caller_order(m.node_coords_free.grad, "x").cpu().numpy()
    gu = m.to_caller_order(m.u_free.grad, "u").cpu().numpy()
    assert np.abs(gx - gX_ref[~geom.numpy()]).max() <= 1e-10 * np.abs(gX_ref).max()
    assert np.abs(gu - gU_ref[~bc.numpy()]).max() <= 1e-10 * np.abs(gU_ref).max()
    print(f"[unstructured 4M] elements {ne} nodes {nn} tiles {st['n_tiles']} halo elems x{st['tile_elem_total'] / ne:.3f} "
          f"nodes x{st['tile_node_total'] / nn:.3f} max valence {valence.max()}")


@pytest.mark.gpu
def test_cfg5_as_specified_through_the_plain_model_api_reorders_rows_and_matches_oracle():
    """BASELINE config 5 exactly as SURVEY specifies it -- 4 x 10^6 TRI3, random diagonals, random element permutation AND
    random global node renumbering -- through the reference's construction API, nothing else: the model notices the
    numbering (row_line_factor ~ 8) and stores its parameter rows along the locality curve (reorder="auto").  Loss and
    every gradient row against the C oracle on the CALLER's numbering; coords / u_full / state_dict in the caller's
    numbering; same numbers as the same model with reorder="off"."""
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from oracle import closed_form as CF
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(2001, 1001, jitter=0.3, seed=11, diagonal="random",
                                                            permute=True, dtype=F64)
    kw = dict(boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges)
    torch.manual_seed(7)
    m = PiecewiseLinearShapeNN2D(coords, conn, **kw).to(d)
    assert m.row_order == "tile" and m.row_line_factor > 6.0
    lf = EnergyLoss2D(device=d, dtype=F64)
    loss = lf(m)
    loss.backward()
    # oracle on the caller's numbering
    g_np, b_np = geom.numpy(), bc.numpy()
    X = coords.numpy().copy()
    U = np.zeros_like(X)
    U[~b_np] = m.to_caller_order(m.u_free.detach(), "u").cpu().numpy()
    mat, W = CF.plane_stress(), 0.25
    Tc = np.array([lf._ci * 1e5, 0.0, lf._cj * 1e5, 0.0])
    e_ref, gX_ref, gU_ref = CF.tri3_energy(X, U, conn.numpy(), mat, W)
    e_ref -= CF.edge2_energy(X, U, edges.numpy(), Tconst=Tc, gX=gX_ref, gU=gU_ref)
    assert abs(loss.item() - e_ref) <= 1e-12 * abs(e_ref)
    gx = m.to_caller_order(m.node_coords_free.grad, "x").cpu().numpy()
    gu = m.to_caller_order(m.u_free.grad, "u").cpu().numpy()
    assert np.abs(gx - gX_ref[~g_np]).max() <= 1e-10 * np.abs(gX_ref).max()
    assert np.abs(gu - gU_ref[~b_np]).max() <= 1e-10 * np.abs(gU_ref).max()
    # the assembled fields are in the caller's numbering
    assert torch.equal(m.coords.detach().cpu(), coords)
    assert torch.equal(m.u_full.detach().cpu()[~bc], torch.from_numpy(U[~b_np]))
    sd = m.state_dict()
    assert torch.equal(sd["node_coords_free"].cpu(), coords[~geom])
    # same model, rows stored as given: the same energy (the locality of the row gathers is all that differs)
    torch.manual_seed(7)
    m_off = PiecewiseLinearShapeNN2D(coords, conn, reorder="off", **kw).to(d)
    assert m_off.row_order == "as given" and torch.equal(m_off.state_dict()["u_free"], sd["u_free"])
    l_off = lf(m_off)
    assert abs(l_off.item() - loss.item()) <= 1e-13 * abs(loss.item())
    # the plan's row maps of the reordered model are local: a tile's rows touch few 128-byte lines
    ns, td = m.tile_plan(0).export("node_src"), m.tile_plan(0).export("tile_desc")
    ns_off = m_off.tile_plan(0).export("node_src")

    def lines_per_tile(ns_, t):
        rows = ns_[td[t, 2]:td[t, 2] + td[t, 3], 0]
        return len(np.unique(rows[rows >= 0] // 8))
    mid = td.shape[0] // 2
    assert lines_per_tile(ns, mid) * 4 < lines_per_tile(ns_off, mid)

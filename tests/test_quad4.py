"""QUAD4-iso EXTENSION (no reference counterpart, SURVEY F11: "parity unpinned by reference").
CPU: sanity of the test-side oracle (oracle/quad4.py).  GPU (-m gpu): kernels vs that oracle."""
import numpy as np
import pytest
import torch

F64 = torch.float64


def test_quad4_oracle_sanity_on_rectilinear_cells():
    """For axis-aligned rectangles J is diagonal, so the reference convention (Jinv * D_N) and the textbook
    one coincide: a linear field must give its exact gradient, and detJ = cell area / 4."""
    from oracle import quad4 as Q
    from hidenn_fem_amd.mesh import structured_quad_mesh
    coords, conn, *_ = structured_quad_mesh(5, 4, length=2.0, height=1.5, dtype=F64)
    A = torch.tensor([[2.0, 3.0], [-1.0, 5.0]], dtype=F64)
    U = coords @ A.T
    g = torch.Generator().manual_seed(0)
    x_eval = torch.rand(conn.shape[0], 2, generator=g, dtype=F64) * 2 - 1
    elem_id = torch.arange(conn.shape[0])
    u_h, detJ, grad_u = Q.quad4_forward(coords, U, conn, x_eval, elem_id)
    assert torch.allclose(grad_u, A.expand_as(grad_u), atol=1e-12)
    assert torch.allclose(detJ, torch.full_like(detJ, (2.0 / 4) * (1.5 / 3) / 4), atol=1e-14)
    # interpolation reproduces the linear field at the mapped physical point
    xk, ek = Q.XI.to(F64)[None, :], Q.ETA.to(F64)[None, :]
    N = 0.25 * (1 + xk * x_eval[:, 0:1]) * (1 + ek * x_eval[:, 1:2])
    xp = torch.sum(N.unsqueeze(2) * coords[conn], dim=1)
    assert torch.allclose(u_h, xp @ A.T, atol=1e-12)


def test_quad4_c_closed_form_matches_the_autograd_restatement():
    """oracle/hfem_oracle.c:oracle_quad4_energy (node-by-node D_N form, hand backward) against oracle/quad4.py (autograd
    through the op chain) with and without a body force -- two independent statements of the SURVEY section 8a spec.
    PARITY UNPINNED BY THE REFERENCE (it has no QUAD4 element): these two are each other's only check."""
    from oracle import quad4 as Q, ref_chain as R, closed_form as CF
    from hidenn_fem_amd.mesh import structured_quad_mesh
    from conftest import b_force_fn
    c, cn, *_ = structured_quad_mesh(23, 17, jitter=0.3, seed=3, dtype=F64)
    U = 1e-3 * torch.randn(c.shape, dtype=F64, generator=torch.Generator().manual_seed(1))
    for bf in (None, b_force_fn):
        X, Uu = c.clone().requires_grad_(True), U.clone().requires_grad_(True)
        e = Q.quad4_domain_energy(X, Uu, cn, R.plane_stress_C(), bf)
        e.backward()
        Bq = None if bf is None else bf(Q.gauss_2x2()).numpy()
        e2, gX, gU = CF.quad4_energy(c.numpy(), U.numpy(), cn.numpy(), CF.plane_stress(), Bq)
        assert abs(e.item() - e2) <= 1e-13 * abs(e2)
        assert np.abs(gX - X.grad.numpy()).max() <= 1e-12 * np.abs(gX).max()
        assert np.abs(gU - Uu.grad.numpy()).max() <= 1e-12 * np.abs(gU).max()


@pytest.mark.gpu
def test_quad4_body_force_and_traction_function():
    """The QUAD4 path takes a body force (at the reference Gauss points, as the triangle path does, F6) and a
    position-dependent traction (per-edge table when the Neumann nodes are fixed, autograd through ``t_force`` when
    they move) -- what EnergyLoss2D.__call__(model, b_force, t_force) offers for triangles (loss.py:80,106)."""
    from oracle import quad4 as Q, ref_chain as R
    from hidenn_fem_amd.mesh import structured_quad_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from conftest import b_force_fn, t_force_fn
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_quad_mesh(23, 17, jitter=0.25, seed=4, dtype=F64)
    for fixed_boundary in (True, False):                   # False: the Neumann nodes are free -> autograd through t_force
        bmask = geom if fixed_boundary else torch.zeros_like(geom)
        torch.manual_seed(1)
        m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=bmask, dirichlet_mask=bc, u_fixed=0.0,
                                     neumann_edges=edges).to(d)
        with torch.no_grad():
            m.u_free.mul_(50.0)
        lf = EnergyLoss2D(device=d, dtype=F64)
        loss = lf(m, b_force=lambda x: b_force_fn(x.cpu()).to(d), t_force=lambda x: t_force_fn(x.cpu()).to(d))
        loss.backward()
        xf = m.node_coords_free.detach().cpu().clone().requires_grad_(True)
        uf = m.u_free.detach().cpu().clone().requires_grad_(True)
        X = R.assemble_coords(coords.shape[0], ~bmask, xf, bmask, coords[bmask])
        U = R.assemble_u(coords.shape[0], ~bc, uf, bc, torch.tensor(0.0, dtype=F64))
        ref = Q.quad4_domain_energy(X, U, conn, R.plane_stress_C(), b_force_fn) - \
            R.edge_energy(X, U, edges, *R.interval_gauss(2), t_force=t_force_fn)
        ref.backward()
        assert abs(loss.item() - ref.item()) <= 1e-12 * abs(ref.item()), fixed_boundary
        gx, gu = m.node_coords_free.grad.cpu().numpy(), m.u_free.grad.cpu().numpy()
        assert np.abs(gx - xf.grad.numpy()).max() <= 1e-10 * np.abs(xf.grad.numpy()).max()
        assert np.abs(gu - uf.grad.numpy()).max() <= 1e-10 * np.abs(uf.grad.numpy()).max()


@pytest.mark.gpu
def test_quad4_kernels_match_the_autograd_oracle():
    from oracle import quad4 as Q, ref_chain as R
    from hidenn_fem_amd.mesh import structured_quad_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D, QuadShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_quad_mesh(23, 17, jitter=0.25, seed=4, dtype=F64)
    torch.manual_seed(1)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                 neumann_edges=edges).to(d)
    assert isinstance(m, QuadShapeNN2D)
    with torch.no_grad():
        m.u_free.mul_(50.0)
    lf = EnergyLoss2D(device=d, dtype=F64)
    loss = lf(m)
    loss.backward()
    # oracle: same assembly, autograd through the op chain (domain) + reference edge work
    xf = m.node_coords_free.detach().cpu().clone().requires_grad_(True)
    uf = m.u_free.detach().cpu().clone().requires_grad_(True)
    X = R.assemble_coords(coords.shape[0], ~geom, xf, geom, coords[geom])
    U = R.assemble_u(coords.shape[0], ~bc, uf, bc, torch.tensor(0.0, dtype=F64))
    ref = Q.quad4_domain_energy(X, U, conn, R.plane_stress_C()) - R.edge_energy(X, U, edges, *R.interval_gauss(2))
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-12 * abs(ref.item())
    gx, gu = m.node_coords_free.grad.cpu().numpy(), m.u_free.grad.cpu().numpy()
    assert np.abs(gx - xf.grad.numpy()).max() <= 1e-10 * np.abs(xf.grad.numpy()).max()
    assert np.abs(gu - uf.grad.numpy()).max() <= 1e-10 * np.abs(uf.grad.numpy()).max()
    # the planless (global fp64 atomics) kernel is kept as a cross-check of the tiled one
    m.zero_grad()
    lf.quad4_planless = True
    loss_a = lf(m)
    loss_a.backward()
    lf.quad4_planless = False
    assert abs(loss_a.item() - ref.item()) <= 1e-12 * abs(ref.item())
    assert np.abs(m.node_coords_free.grad.cpu().numpy() - gx).max() <= 1e-10 * np.abs(gx).max()
    assert np.abs(m.u_free.grad.cpu().numpy() - gu).max() <= 1e-10 * np.abs(gu).max()
    # autograd-free form: gradients straight into .grad
    with torch.no_grad():
        for p_ in m.parameters():
            p_.grad.fill_(3.0)
    loss_d = lf.value_and_grad_(m)
    assert abs(loss_d.item() - ref.item()) <= 1e-12 * abs(ref.item())
    assert np.abs(m.node_coords_free.grad.cpu().numpy() - gx).max() <= 1e-10 * np.abs(gx).max()
    assert np.abs(m.u_free.grad.cpu().numpy() - gu).max() <= 1e-10 * np.abs(gu).max()
    # per-point forward / backward with the (x_ref, element_id) contract
    g = torch.Generator().manual_seed(5)
    M = 700
    x_eval = torch.rand(M, 2, generator=g, dtype=F64) * 2 - 1
    elem_id = torch.randint(0, conn.shape[0], (M,), generator=g)
    cu, cd, cg = (torch.randn(s, generator=g, dtype=F64) for s in ((M, 2), (M,), (M, 2, 2)))
    m.zero_grad()
    u_h, detJ, grad_u = m(x_eval.to(d), elem_id.to(d))
    ((u_h * cu.to(d)).sum() + (detJ * cd.to(d)).sum() + (grad_u * cg.to(d)).sum()).backward()
    xf.grad = None
    uf.grad = None
    X = R.assemble_coords(coords.shape[0], ~geom, xf, geom, coords[geom])
    U = R.assemble_u(coords.shape[0], ~bc, uf, bc, torch.tensor(0.0, dtype=F64))
    ru, rd, rg = Q.quad4_forward(X, U, conn, x_eval, elem_id)
    ((ru * cu).sum() + (rd * cd).sum() + (rg * cg).sum()).backward()
    np.testing.assert_allclose(u_h.detach().cpu().numpy(), ru.detach().numpy(), rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(detJ.detach().cpu().numpy(), rd.detach().numpy(), rtol=1e-12)
    np.testing.assert_allclose(grad_u.detach().cpu().numpy(), rg.detach().numpy(), rtol=1e-10, atol=1e-14)
    gx, gu = m.node_coords_free.grad.cpu().numpy(), m.u_free.grad.cpu().numpy()
    assert np.abs(gx - xf.grad.numpy()).max() <= 1e-10 * np.abs(xf.grad.numpy()).max()
    assert np.abs(gu - uf.grad.numpy()).max() <= 1e-10 * np.abs(uf.grad.numpy()).max()


@pytest.mark.gpu
def test_quad4_one_million_elements_runs_and_matches_sampled_oracle():
    """cfg4-Q shape (1001x1001 nodes -> 10^6 QUAD4): full-size run; oracle on a 60k-element sub-mesh."""
    from oracle import quad4 as Q, ref_chain as R
    from hidenn_fem_amd.mesh import structured_quad_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_quad_mesh(1001, 1001, length=2.0, height=2.0, jitter=0.2, seed=0, dtype=F64)
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    lf = EnergyLoss2D(device=d, dtype=F64)
    loss = lf(m)
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(m.u_free.grad).all() and torch.isfinite(m.node_coords_free.grad).all()
    # domain energy of the first 60 columns of cells == oracle on that sub-mesh (energy is additive over elements)
    sub = conn[: 60 * 1000]
    X, U = m.coords.detach().cpu(), m.u_full.detach().cpu()
    e_sub = Q.quad4_domain_energy(X, U, sub, R.plane_stress_C()).item()
    from hidenn_fem_amd import _lib
    import ctypes as C
    acc = torch.zeros((), dtype=F64, device=d)
    Xd, Ud = m.coords.detach(), m.u_full.detach()
    _lib.check(_lib.lib().hfem_quad4_energy_atomic(0, Xd.data_ptr(), Ud.data_ptr(), m._conn32.data_ptr(), 0, sub.shape[0],
                                                   Xd.shape[0], (C.c_double * 4)(*lf._mat), acc.data_ptr(), None, None,
                                                   _lib.stream_ptr(d)))
    assert abs(acc.item() - e_sub) <= 1e-12 * abs(e_sub)
    # tiled (default) vs planless at full size: same loss and gradients up to fp64 summation order
    gx, gu = m.node_coords_free.grad.clone(), m.u_free.grad.clone()
    m.zero_grad()
    lf.quad4_planless = True
    loss_a = lf(m)
    loss_a.backward()
    assert abs(loss_a.item() - loss.item()) <= 1e-12 * abs(loss.item())
    assert (m.node_coords_free.grad - gx).abs().max() <= 1e-10 * gx.abs().max()
    assert (m.u_free.grad - gu).abs().max() <= 1e-10 * gu.abs().max()
    st = m.tile_plan().stats
    assert st["n_elems"] == 10 ** 6 and st["lds_bytes"] <= 40 * 1024
    # FULL comparison at 10^6 elements against the C closed form (oracle/hfem_oracle.c:oracle_quad4_energy -- parity
    # unpinned by the reference: it has no QUAD4 element), loss and every gradient row, with the Neumann edge work
    from oracle import closed_form as CF
    Xn, Un = X.numpy(), U.numpy()
    e_ref, gX_ref, gU_ref = CF.quad4_energy(Xn, Un, conn.numpy(), CF.plane_stress())
    e_ref -= CF.edge2_energy(Xn, Un, edges.numpy(), Tconst=np.array([lf._ci * 1e5, 0.0, lf._cj * 1e5, 0.0]), gX=gX_ref, gU=gU_ref)
    assert abs(loss.item() - e_ref) <= 1e-12 * abs(e_ref)
    gx_c, gu_c = m.to_caller_order(gx, "x"), m.to_caller_order(gu, "u")     # big meshes store their rows tile-major
    assert np.abs(gx_c.cpu().numpy() - gX_ref[~geom.numpy()]).max() <= 1e-10 * np.abs(gX_ref).max()
    assert np.abs(gu_c.cpu().numpy() - gU_ref[~bc.numpy()]).max() <= 1e-10 * np.abs(gU_ref).max()


@pytest.mark.gpu
def test_quad4_fp32_rows_physical_convention_and_deterministic_instances():
    """The QUAD4 extension at TRI3's feature level: float-row instance for fp32 models (the reference's
    default dtype, /root/reference/src/loss.py:16: rows widened on load, gradients rounded once, fp64 arithmetic),
    grad_convention="physical" (G Jinv) and deterministic=True (node-centric fixed-order kernel, bit-identical run to run),
    each with and without a body force, against the autograd restatement oracle/quad4.py.
    PARITY UNPINNED BY THE REFERENCE (it has no QUAD4 element)."""
    import copy
    from oracle import quad4 as Q, ref_chain as R
    from hidenn_fem_amd.mesh import structured_quad_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from conftest import b_force_fn
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_quad_mesh(61, 45, jitter=0.25, seed=4, dtype=F64)
    torch.manual_seed(1)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    with torch.no_grad():
        m.u_free.mul_(50.0)
    bf_gpu = lambda x: b_force_fn(x.cpu()).to(d)

    def oracle(convention, bf):
        xf = m.node_coords_free.detach().cpu().clone().requires_grad_(True)
        uf = m.u_free.detach().cpu().clone().requires_grad_(True)
        X = R.assemble_coords(coords.shape[0], ~geom, xf, geom, coords[geom])
        U = R.assemble_u(coords.shape[0], ~bc, uf, bc, torch.tensor(0.0, dtype=F64))
        ref = Q.quad4_domain_energy(X, U, conn, R.plane_stress_C(), bf, convention) - \
            R.edge_energy(X, U, edges, *R.interval_gauss(2))
        ref.backward()
        return ref.item(), xf.grad, uf.grad

    def run(lf, model, bf):
        model.zero_grad()
        loss = lf(model, b_force=bf)
        loss.backward()
        return loss.item(), model.node_coords_free.grad.detach().cpu().double(), model.u_free.grad.detach().cpu().double()
    refs = {}
    for conv in ("reference", "physical"):
        for has_b in (False, True):
            refs[(conv, has_b)] = oracle(conv, b_force_fn if has_b else None)
            e_ref, gx_ref, gu_ref = refs[(conv, has_b)]
            for det in (False, True):
                lf = EnergyLoss2D(device=d, dtype=F64, grad_convention=conv, deterministic=det)
                e, gx, gu = run(lf, m, bf_gpu if has_b else None)
                assert abs(e - e_ref) <= 1e-12 * abs(e_ref), (conv, has_b, det)
                assert (gx - gx_ref).abs().max() <= 1e-10 * gx_ref.abs().max(), (conv, has_b, det)
                assert (gu - gu_ref).abs().max() <= 1e-10 * gu_ref.abs().max(), (conv, has_b, det)
                if det:                                     # fixed order: the same bits again
                    e2, gx2, gu2 = run(lf, m, bf_gpu if has_b else None)
                    assert e2 == e and torch.equal(gx2, gx) and torch.equal(gu2, gu)
    assert abs(refs[("physical", False)][0] - refs[("reference", False)][0]) > 1e-6 * abs(refs[("reference", False)][0]), \
        "on a skewed mesh the two conventions differ"
    # value_and_grad_ (the autograd-free path) takes the switches too
    lf = EnergyLoss2D(device=d, dtype=F64, grad_convention="physical", deterministic=True)
    assert abs(lf.value_and_grad_(m).item() - refs[("physical", False)][0]) <= 1e-12 * abs(refs[("physical", False)][0])
    assert (m.u_free.grad.cpu() - refs[("physical", False)][2]).abs().max() <= 1e-10 * refs[("physical", False)][2].abs().max()
    # ---- fp32 rows: the same values as the fp64 path on the same float numbers, rounded once
    torch.manual_seed(1)
    m32 = PiecewiseLinearShapeNN2D(coords.float(), conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                   neumann_edges=edges).to(d)
    with torch.no_grad():
        m32.u_free.mul_(50.0)
    m64 = copy.deepcopy(m32).double()
    for conv in ("reference", "physical"):
        for has_b in (False, True):
            lf32 = EnergyLoss2D(device=d, dtype=torch.float32, grad_convention=conv)
            lf64 = EnergyLoss2D(device=d, dtype=F64, grad_convention=conv)
            lf64._mat, lf64._W, lf64._ci, lf64._cj = lf32._mat, lf32._W, lf32._ci, lf32._cj   # the fp32 object's rounded constants
            bf = (lambda x: b_force_fn(x.cpu().double()).to(d)) if has_b else None
            l32, gx32, gu32 = run(lf32, m32, bf)
            l64, gx64, gu64 = run(lf64, m64, bf)
            assert m32.u_free.grad.dtype == torch.float32
            assert abs(l32 - l64) <= 2e-7 * abs(l64)
            for a, b in ((gx32, gx64), (gu32, gu64)):
                want = b.float().double()
                ulp = torch.finfo(torch.float32).eps * want.abs().clamp_min(1e-30)
                assert ((a - want).abs() <= 1.01 * ulp).all(), (conv, has_b)
    # autograd-free fp32 path
    lf32 = EnergyLoss2D(device=d, dtype=torch.float32)
    lv = lf32.value_and_grad_(m32)
    assert m32.u_free.grad.dtype == torch.float32 and abs(lv.item() - run(lf32, copy.deepcopy(m32), None)[0]) <= 2e-7 * abs(lv.item())

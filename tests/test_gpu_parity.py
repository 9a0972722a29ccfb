"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the
C ABI / host mirror, against (a) the golden vectors generated from the imported reference
and (b) the oracle on larger seeded meshes.

Tolerances (fp64): loss rel <= 1e-12, gradients
max-abs <= 1e-10 * max|g|.  Index outputs and pure copies are bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import tri_mesh_dict, tri_case_forces

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-12
GRAD_RTOL = 1e-10
F64 = torch.float64


def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return torch.device("cuda:0")


def assert_grad_close(got, want, what="", rtol=GRAD_RTOL):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    scale = max(np.abs(want).max(), 1e-300)
    err = np.abs(got - want).max()
    assert err <= rtol * scale, f"{what}: max-abs err {err:.3e} vs scale {scale:.3e}"


def tri_model_from_golden(g, case, device):
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    p = case + "/"
    nc = g.t(p + "node_coords")
    bmask = g.t(p + "boundary_mask")
    model = PiecewiseLinearShapeNN2D(nc, g.t(p + "conn"), boundary_mask=bmask if bmask.any() else None,
                                     dirichlet_mask=g.t(p + "dirichlet_mask"), u_fixed=0.0,
                                     neumann_edges=g.t(p + "edges")).double().to(device)
    with torch.no_grad():
        model.u_free.copy_(g.t(p + "u_free").to(device))
    return model


def test_library_loaded_and_native():
    from hidenn_fem_amd import _lib
    L = _lib.lib()
    assert L.hfem_version() == 114
    assert L.hfem_device_count() >= 1


def test_cpu_tensors_fail_loudly():
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN
    m = PiecewiseLinearShapeNN(torch.linspace(0, 1, 5, dtype=F64)).double()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(3, dtype=F64))


@pytest.mark.parametrize("tile_elems", [0, 32])
def test_fused_energy_matches_reference_golden(g_tri, tile_elems):
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = dev()
    for case in g_tri.cases():
        model = tri_model_from_golden(g_tri, case, d)
        go, go1 = (int(v) for v in g_tri[case + "/gauss_order"])
        b, t = tri_case_forces(case)
        loss_fn = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=go, gauss_order_1d=go1, device=d, dtype=F64,
                               tile_elems=tile_elems)
        loss = loss_fn(model, b_force=b, t_force=t)
        loss.backward()
        want = g_tri[case + "/loss"].item()
        assert abs(loss.item() - want) <= LOSS_RTOL * abs(want), (case, loss.item(), want)
        assert_grad_close(model.u_free.grad, g_tri[case + "/g_u_free"], case + " gU")
        assert_grad_close(model.node_coords_free.grad, g_tri[case + "/g_coords_free"], case + " gX")
        # the two halves of the reference API
        dom = loss_fn.domain_energy(model, b)
        edg = loss_fn.edge_energy(model, t)
        wd, we = g_tri[case + "/domain"].item(), g_tri[case + "/edge"].item()
        assert abs(dom.item() - wd) <= LOSS_RTOL * abs(wd)
        assert abs(edg.item() - we) <= LOSS_RTOL * max(abs(we), 1e-300)


def test_unfused_forward_backward_matches_golden(g_tri):
    d = dev()
    case = "order4"
    model = tri_model_from_golden(g_tri, case, d)
    p = case + "/pp_"
    x_eval, elem_id = g_tri.t(p + "x_eval").to(d), g_tri.t(p + "elem_id").to(d)
    u_h, detJ, grad_u = model(x_eval, elem_id)
    np.testing.assert_allclose(u_h.detach().cpu().numpy(), g_tri[p + "u_h"], rtol=1e-13, atol=1e-20)
    np.testing.assert_allclose(detJ.detach().cpu().numpy(), g_tri[p + "detJ"], rtol=1e-13)
    np.testing.assert_allclose(grad_u.detach().cpu().numpy(), g_tri[p + "grad_u"], rtol=1e-11, atol=1e-16)
    ((u_h * g_tri.t(p + "cu").to(d)).sum() + (detJ * g_tri.t(p + "cd").to(d)).sum()
     + (grad_u * g_tri.t(p + "cg").to(d)).sum()).backward()
    assert_grad_close(model.u_free.grad, g_tri[p + "g_u_free"], "unfused gU")
    assert_grad_close(model.node_coords_free.grad, g_tri[p + "g_coords_free"], "unfused gX")
    q = case + "/pe_"
    ue, ds = model(g_tri.t(q + "x_eval").to(d), g_tri.t(q + "edge_id").to(d), edge=True)
    np.testing.assert_allclose(ue.detach().cpu().numpy(), g_tri[q + "u_h"], rtol=1e-13, atol=1e-20)
    np.testing.assert_allclose(ds.detach().cpu().numpy(), g_tri[q + "ds"], rtol=1e-14)
    # assembly is a pure copy: bit-exact
    X = model.coords.detach().cpu().numpy()
    assert np.array_equal(X, g_tri[case + "/node_coords"])


def _random_problem(nx, ny, seed, **kw):
    from hidenn_fem_amd.mesh import structured_tri_mesh
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(nx, ny, jitter=0.2, seed=seed, dtype=F64, **kw)
    rng = np.random.default_rng(seed)
    U = 1e-5 * rng.standard_normal((coords.shape[0], 2))
    return coords.numpy(), U, conn.numpy(), edges.numpy()


@pytest.mark.parametrize("nx,ny,kw", [(64, 33, {}), (301, 151, dict(diagonal="random", permute=True)),
                                      (120, 90, dict(flip_fraction=0.3))])
def test_atomic_and_tiled_kernels_vs_oracle(nx, ny, kw):
    """C ABI called directly: planless atomic kernels and the tiled kernel (identity row maps)
    against the plain-C closed forms on seeded meshes."""
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.plan import TilePlan
    from oracle import closed_form as CF
    d = dev()
    X, U, conn, edges = _random_problem(nx, ny, seed=nx, **kw)
    mat, W = CF.plane_stress(), 0.25
    rng = np.random.default_rng(1)
    Bk = rng.standard_normal(6) * 1e4
    Tc = np.array([2e5, 0.0, 0.0, 1e4])
    e_ref, gX_ref, gU_ref = CF.tri3_energy(X, U, conn, mat, W, Bk)
    e_ref -= CF.edge2_energy(X, U, edges, Tconst=Tc, gX=gX_ref, gU=gU_ref)

    L = _lib.lib()
    s = _lib.stream_ptr(d)
    Xd, Ud = torch.from_numpy(X).to(d), torch.from_numpy(U).to(d)
    conn32 = torch.from_numpy(conn.astype(np.int32)).to(d)
    edges32 = torch.from_numpy(edges.astype(np.int32)).to(d)
    dv = lambda a: (C.c_double * len(a))(*a)
    # --- planless atomics
    loss = torch.zeros((), dtype=F64, device=d)
    gX, gU = torch.zeros_like(Xd), torch.zeros_like(Ud)
    _lib.check(L.hfem_tri3_energy_atomic(0, Xd.data_ptr(), Ud.data_ptr(), conn32.data_ptr(), 0, conn.shape[0],
                                         X.shape[0], dv(mat), W, dv(Bk), loss.data_ptr(), gX.data_ptr(),
                                         gU.data_ptr(), s))
    _lib.check(L.hfem_edge2_energy_atomic(0, Xd.data_ptr(), Ud.data_ptr(), edges32.data_ptr(), edges.shape[0],
                                          None, dv(Tc), loss.data_ptr(), gX.data_ptr(), gU.data_ptr(), s))
    assert abs(loss.item() - e_ref) <= LOSS_RTOL * abs(e_ref)
    assert_grad_close(gX, gX_ref, "atomic gX")
    assert_grad_close(gU, gU_ref, "atomic gU")
    # --- tiled, identity maps, several tile sizes; outputs are OVERWRITTEN (poison first)
    for T in (0, 200):
        plan = TilePlan(conn, X.shape[0], coords_hint=X, edges=edges, tile_elems=T, device=d)
        loss2 = torch.full((), 7.0, dtype=F64, device=d)
        gX2, gU2 = torch.full_like(Xd, float("nan")), torch.full_like(Ud, float("nan"))
        _lib.check(L.hfem_tri3_energy_plan(plan.handle, Xd.data_ptr(), None, Ud.data_ptr(), None, dv(mat), W, dv(Bk),
                                           None, dv(Tc), 0, -1, loss2.data_ptr(), gX2.data_ptr(), gU2.data_ptr(),
                                           0, s))
        assert abs(loss2.item() - e_ref) <= LOSS_RTOL * abs(e_ref)
        assert_grad_close(gX2, gX_ref, "tiled gX")
        assert_grad_close(gU2, gU_ref, "tiled gU")
        # element sharding: tile ranges are additive and write disjoint rows
        acc_l, accX, accU = 0.0, torch.zeros_like(Xd), torch.zeros_like(Ud)
        for r in range(3):
            lo, hi = plan.shard_range(r, 3)
            l3 = torch.zeros((), dtype=F64, device=d)
            gX3, gU3 = torch.zeros_like(Xd), torch.zeros_like(Ud)
            _lib.check(L.hfem_tri3_energy_plan(plan.handle, Xd.data_ptr(), None, Ud.data_ptr(), None, dv(mat), W,
                                               dv(Bk), None, dv(Tc), lo, hi, l3.data_ptr(), gX3.data_ptr(),
                                               gU3.data_ptr(), 0, s))
            assert int(((gX3 != 0).any(dim=1) & (accX != 0).any(dim=1)).sum()) == 0
            acc_l += l3.item()
            accX += gX3
            accU += gU3
        assert abs(acc_l - e_ref) <= LOSS_RTOL * abs(e_ref)
        assert_grad_close(accX, gX_ref, "sharded gX")
        assert_grad_close(accU, gU_ref, "sharded gU")
        plan.close()


@pytest.mark.parametrize("world,nx,ny,kw", [(8, 401, 301, {}), (2, 401, 301, {}), (3, 301, 151, dict(diagonal="random", permute=True)),
                                            (8, 161, 121, dict(flip_fraction=0.3))])
def test_sharded_plan_ranges_vs_oracle(world, nx, ny, kw):
    """A plan prepared for `world` ranks (shard-aware tile policy: 512-thread / one-row tiles at these sizes; every rank's
    BOUNDARY tiles first in its range) evaluated range by range through the C ABI -- boundary part and interior part as two
    launches, the way the overlapped step runs them -- against the plain-C closed forms: energies add up, every gradient row is
    written by exactly one rank and equals the oracle's (no test held a shards=N plan against the oracle)."""
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.plan import TilePlan
    from oracle import closed_form as CF
    d = dev()
    X, U, conn, edges = _random_problem(nx, ny, seed=nx + world, **kw)
    mat, W = CF.plane_stress(), 0.25
    Tc = np.array([2e5, 0.0, 0.0, 1e4])
    e_ref, gX_ref, gU_ref = CF.tri3_energy(X, U, conn, mat, W, np.zeros(6))
    e_ref -= CF.edge2_energy(X, U, edges, Tconst=Tc, gX=gX_ref, gU=gU_ref)
    L = _lib.lib()
    s = _lib.stream_ptr(d)
    Xd, Ud = torch.from_numpy(X).to(d), torch.from_numpy(U).to(d)
    dv = lambda a: (C.c_double * len(a))(*a)
    plan = TilePlan(conn, X.shape[0], coords_hint=X, edges=edges, device=d, shards=world)
    st = plan.stats
    assert st["shards"] == world
    if (nx, ny) == (401, 301):               # 240 k elements: 120 k per rank -> 512-thread one-row tiles; 30 k per rank -> 256-thread ones
        assert st["paired"] == 1 and st["threads_per_tile"] == (512 if world == 2 else 256) and st["slot_rows"] == 1, st
    acc_l, accX, accU = 0.0, torch.zeros_like(Xd), torch.zeros_like(Ud)
    written = torch.zeros(X.shape[0], dtype=torch.int32, device=d)
    n_bnd = 0
    for r in range(world):
        lo, mid, hi = plan.shard_parts(r, world)
        assert lo <= mid <= hi
        n_bnd += mid - lo
        for a, b in ((mid, hi), (lo, mid)):                  # interior first, boundary second: the overlapped step's order
            if b <= a:
                continue
            l3 = torch.zeros((), dtype=F64, device=d)
            gX3, gU3 = torch.full_like(Xd, float("nan")), torch.full_like(Ud, float("nan"))
            _lib.check(L.hfem_tri3_energy_plan(plan.handle, Xd.data_ptr(), None, Ud.data_ptr(), None, dv(mat), W, dv([0.0] * 6), None,
                                               dv(Tc), a, b, l3.data_ptr(), gX3.data_ptr(), gU3.data_ptr(), 0, s))
            rows = ~torch.isnan(gX3[:, 0])
            assert bool((rows == ~torch.isnan(gU3[:, 0])).all())
            written += rows.int()
            acc_l += l3.item()
            accX += torch.nan_to_num(gX3)
            accU += torch.nan_to_num(gU3)
    assert 0 < n_bnd < plan.n_tiles
    assert int(written.min()) == 1 and int(written.max()) == 1, "every node row is written by exactly one tile range"
    assert abs(acc_l - e_ref) <= LOSS_RTOL * abs(e_ref)
    assert_grad_close(accX, gX_ref, f"shards={world} gX")
    assert_grad_close(accU, gU_ref, f"shards={world} gU")
    plan.close()


def test_full_size_1m_elements_vs_oracle():
    """BASELINE config 'Example 4 / T1M': 1001x501 nodes -> 1,000,000 TRI3, through the model API."""
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from oracle import closed_form as CF
    d = dev()
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(1001, 501, jitter=0.2, seed=0, dtype=F64)
    torch.manual_seed(0)
    model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                     neumann_edges=edges).to(d)
    loss_fn = EnergyLoss2D(device=d, dtype=F64)
    loss = loss_fn(model)
    loss.backward()
    X = model.coords.detach().cpu().numpy()
    U = model.u_full.detach().cpu().numpy()
    e_ref, gX_ref, gU_ref = CF.tri3_energy(X, U, conn.numpy(), CF.plane_stress(), 0.25)
    e_ref -= CF.edge2_energy(X, U, edges.numpy(), Tconst=np.array([2e5, 0, 0, 0]), gX=gX_ref, gU=gU_ref)
    assert abs(loss.item() - e_ref) <= LOSS_RTOL * abs(e_ref)
    assert model.row_order == "tile"                 # meshes of this size store their parameter rows tile-major (reorder="auto")
    assert_grad_close(model.to_caller_order(model.node_coords_free.grad, "x"), gX_ref[~geom.numpy()], "1M gX")
    assert_grad_close(model.to_caller_order(model.u_free.grad, "u"), gU_ref[~bc.numpy()], "1M gU")
    # idempotence: a second evaluation overwrites, never accumulates
    model.zero_grad()
    loss2 = loss_fn(model)
    loss2.backward()
    assert loss2.item() == loss.item() or abs(loss2.item() - loss.item()) <= 1e-13 * abs(loss.item())


# ----------------------------------------------------------------------------- 1D
def _line_model(g_line, p, device, r_adapt=True):
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN
    bc = g_line[p + "bc"] if g_line.has(p + "bc") else np.array([np.nan, np.nan])
    u0 = None if np.isnan(bc[0]) else float(bc[0])
    uN = None if np.isnan(bc[1]) else float(bc[1])
    m = PiecewiseLinearShapeNN(g_line.t(p + "x_grid"), r_adapt=r_adapt, u0=u0, uN=uN).double().to(device)
    with torch.no_grad():
        m.u.copy_(g_line.t(p + "u").to(device))
        if r_adapt:
            m.x_increments.copy_(g_line.t(p + "incr").to(device))
    return m


def test_line2_forward_backward_matches_golden(g_line):
    d = dev()
    for name in ("free", "dir0", "dirN", "dir"):
        p = f"line_{name}/"
        m = _line_model(g_line, p, d)
        np.testing.assert_allclose(m.grid.detach().cpu().numpy(), g_line[p + "grid"], rtol=1e-14, atol=1e-15)
        xe = g_line.t(p + "x_eval").to(d).requires_grad_(True)
        pred = m(xe)
        np.testing.assert_allclose(pred.detach().cpu().numpy(), g_line[p + "pred"], rtol=1e-12, atol=1e-16)
        (pred * g_line.t(p + "cot").to(d)).sum().backward()
        assert_grad_close(m.u.grad, g_line[p + "g_u"], name + " gu")
        assert_grad_close(m.x_increments.grad, g_line[p + "g_incr"], name + " gincr")
        assert_grad_close(xe.grad, g_line[p + "g_x_eval"], name + " gx_eval")


def test_line2_fixed_nodes_float32_bc_quirk(g_line):
    d = dev()
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN
    p = "line_fixed/"
    m = PiecewiseLinearShapeNN(g_line.t(p + "x_grid"), r_adapt=False, u0=0.1).double().to(d)
    with torch.no_grad():
        m.u.copy_(g_line.t(p + "u").to(d))
    xe = g_line.t(p + "x_eval").to(d).requires_grad_(True)
    pred = m(xe)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g_line[p + "pred"], rtol=1e-12, atol=1e-16)
    (pred * g_line.t(p + "cot").to(d)).sum().backward()
    assert_grad_close(m.u.grad, g_line[p + "g_u"], "fixed gu")
    # du/dx at points exactly on nodes: searchsorted(right=False)-1 puts them in the LEFT element
    assert_grad_close(xe.grad, g_line[p + "g_x_eval"], "fixed gx_eval (on-node rule)")


def test_example1_adam_trajectory_fused_mse(g_line):
    """examples/example1.py:25-42 with the fused L2 kernel: first 20 Adam losses, fp64."""
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN
    from hidenn_fem_amd.loss import l2_projection_loss
    d = dev()
    for r_adapt in (True, False):
        xg = torch.linspace(0, 1, 100, dtype=F64, device=d)
        xt = torch.linspace(0, 1, 1000, dtype=F64, device=d)
        ut = torch.sin(2 * torch.pi * xt)
        m = PiecewiseLinearShapeNN(xg, r_adapt=r_adapt).double().to(d)
        opt = torch.optim.Adam(m.parameters(), lr=0.005)
        got = []
        for _ in range(20):
            opt.zero_grad()
            loss = l2_projection_loss(m, xt, ut)
            loss.backward()
            opt.step()
            got.append(loss.item())
        # Adam's first steps are lr*sign(g)-like (m/sqrt(v)): 1e-16 noise in near-zero gradient entries
        # is amplified to O(1e-8) in the trajectory; the per-evaluation parity is pinned elsewhere
        np.testing.assert_allclose(got, g_line[f"ex1_f64_r{int(r_adapt)}/adam_losses"], rtol=1e-6)
        # unfused form of the same loss agrees with the fused kernel
        l1 = ((m(xt) - ut) ** 2).mean()
        l2 = l2_projection_loss(m, xt, ut)
        assert abs(l1.item() - l2.item()) <= 1e-13 * abs(l1.item())


def test_example3_bar_energy_fused_and_autograd_form(g_line):
    from hidenn_fem_amd.loss import bar_energy_loss
    from hidenn_fem_amd.utils import gauss_legendre_points_weights
    from oracle import ref_chain as R
    d = dev()
    xi, wi = gauss_legendre_points_weights(2, device=d, dtype=F64)
    for tag in ("n89", "n1001"):
        p = f"ex3_{tag}/"
        from hidenn_fem_amd.models import PiecewiseLinearShapeNN
        m = PiecewiseLinearShapeNN(g_line.t(p + "x_grid"), r_adapt=True, u0=0.0, uN=0.0).double().to(d)
        with torch.no_grad():
            m.u.copy_(g_line.t(p + "u").to(d))
            m.x_increments.copy_(g_line.t(p + "incr").to(d))
        loss = bar_energy_loss(m, xi, wi, R.example3_body_force, 175.0)
        loss.backward()
        want = g_line[p + "loss"].item()
        assert abs(loss.item() - want) <= LOSS_RTOL * abs(want)
        assert_grad_close(m.u.grad, g_line[p + "g_u"], tag + " gu")
        assert_grad_close(m.x_increments.grad, g_line[p + "g_incr"], tag + " gincr")
        # the reference's own formulation (autograd.grad(..., create_graph=True), example3.py:52-68)
        m.zero_grad()
        with torch.no_grad():
            g = m.grid
            x_i, x_j = g[:-1].unsqueeze(1), g[1:].unsqueeze(1)
            xq = 0.5 * (x_j - x_i) * xi + 0.5 * (x_j + x_i)
            wq = 0.5 * (x_j - x_i) * wi
        xq.requires_grad_(True)
        u = m(xq)
        du = torch.autograd.grad(u, xq, grad_outputs=torch.ones_like(u), create_graph=True)[0]
        loss2 = torch.sum(wq * (0.5 * 175.0 * du ** 2 - R.example3_body_force(xq) * u))
        loss2.backward()
        assert abs(loss2.item() - want) <= LOSS_RTOL * abs(want)
        assert_grad_close(m.u.grad, g_line[p + "g_u"], tag + " gu (autograd form)")
        assert_grad_close(m.x_increments.grad, g_line[p + "g_incr"], tag + " gincr (autograd form)")


# ----------------------------------------------------------------------------- structured 2D
def test_rectq4_forward_backward_matches_golden(g_rect):
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import l2_projection_loss
    d = dev()
    for case in g_rect.cases():
        p = case + "/"
        uf = g_rect[p + "u_fixed"][0]
        r_adapt = bool(g_rect[p + "r_adapt"][0])
        m = PiecewiseLinearShapeNN2D(grid_x=g_rect.t(p + "grid_x"), grid_y=g_rect.t(p + "grid_y"),
                                     boundary_mask_x=g_rect.t(p + "mask_x"), boundary_mask_y=g_rect.t(p + "mask_y"),
                                     r_adapt=r_adapt, u_fixed=None if np.isnan(uf) else float(uf)).double().to(d)
        with torch.no_grad():
            m.u.copy_(g_rect.t(p + "u").to(d))
            if r_adapt:
                m.increments_x.copy_(g_rect.t(p + "incr_x").to(d))
                m.increments_y.copy_(g_rect.t(p + "incr_y").to(d))
        gx, gy = m.grid
        np.testing.assert_allclose(gx.detach().cpu().numpy(), g_rect[p + "gx_full"], rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(gy.detach().cpu().numpy(), g_rect[p + "gy_full"], rtol=1e-14, atol=1e-15)
        xe = g_rect.t(p + "x_eval").to(d).requires_grad_(True)
        pred = m(xe)
        np.testing.assert_allclose(pred.detach().cpu().numpy(), g_rect[p + "pred"], rtol=1e-12, atol=1e-15)
        cot = g_rect.t(p + "cot").to(d)
        (pred * cot).sum().backward()
        assert_grad_close(m.u.grad, g_rect[p + "g_u"], case + " gu")
        assert_grad_close(xe.grad, g_rect[p + "g_x_eval"], case + " gx_eval")
        if r_adapt:
            assert_grad_close(m.increments_x.grad, g_rect[p + "g_incr_x"], case + " gincr_x")
            assert_grad_close(m.increments_y.grad, g_rect[p + "g_incr_y"], case + " gincr_y")
        # fused MSE == unfused MSE (value and gradients)
        target = torch.sin(xe.detach()[:, 0]) * torch.cos(xe.detach()[:, 1])
        m.zero_grad()
        la = ((m(xe.detach()) - target) ** 2).mean()
        la.backward()
        ga = m.u.grad.clone()
        m.zero_grad()
        lb = l2_projection_loss(m, xe.detach(), target)
        lb.backward()
        assert abs(la.item() - lb.item()) <= 1e-13 * abs(la.item())
        assert_grad_close(m.u.grad, ga.cpu().numpy(), case + " fused mse gu")


# ----------------------------------------------------------------------------- examples (acceptance)
def test_examples_run_against_the_drop_in_api(g_lbfgs):
    """examples/ keep the reference's problem set-ups; short runs + the reference's LBFGS trace (fp64)."""
    import examples.example1 as e1
    import examples.example2 as e2
    import examples.example3 as e3
    import examples.example4 as e4
    _, hist = e1.run(epochs=101, log_every=100)
    # reference run prints loss=0.499500 at epoch 0 and 0.101897 at epoch 100 (fp32)
    assert abs(hist[0][1] - 0.4995) < 1e-4 and abs(hist[1][1] - 0.101897) < 2e-3
    _, l2 = e2.run(epochs=60, log_every=1000)
    assert l2 < 1.5
    _, l3, _ = e3.run(epochs=30, log_every=1000)
    _, l3r, _ = e3.run(epochs=30, reference_form=True, log_every=1000)
    assert abs(l3 - l3r) <= 1e-4 * abs(l3r) + 1e-7       # fused kernel == the reference's autograd form (fp32 run)
    _, l4 = e4.run(nx=40, ny=20, steps=2, log_every=100)
    assert l4 < 0.0
    # example 4 in fp64 against the reference's op chain on the CPU, driven by the same torch.optim.LBFGS (2 outer steps = 40
    # closure calls): same mesh, same initial u_free (same seed, same RNG call) -> the same final energy
    from src.mesh import generate_mesh
    from oracle import ref_chain as R
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D as P2D
    torch.manual_seed(17)
    m4, l4_64 = e4.run(nx=40, ny=20, steps=2, dtype=F64, log_every=100)
    nodes, conn4, geom4, bc4, mn4, edges4 = generate_mesh(2.0, 1.0, [(0.5, 0.7, 0.12), (1.0, 0.3, 0.15), (1.4, 0.6, 0.1)],
                                                          {"up": 0, "down": 0, "right": 2, "left": 1}, 40, 20)
    torch.manual_seed(17)
    cpu = P2D(nodes.to(F64), conn4, boundary_mask=geom4, dirichlet_mask=bc4, u_fixed=0.0, neumann_edges=edges4)   # construction only
    mesh4 = dict(n_nodes=nodes.shape[0], conn=conn4, free_mask=~geom4, boundary_mask=geom4, coords_fixed=nodes.to(F64)[geom4],
                 u_free_mask=~bc4, dirichlet_mask=bc4, u_fixed=torch.tensor(0.0, dtype=F64), edges=edges4)
    xf = cpu.node_coords_free.detach().clone().requires_grad_(True)
    uf = cpu.u_free.detach().clone().requires_grad_(True)
    opt_ref = torch.optim.LBFGS([xf, uf])

    def closure_ref():
        opt_ref.zero_grad()
        v = R.total_energy(xf, uf, mesh4)
        v.backward()
        return v
    for _ in range(2):
        l4_ref = opt_ref.step(closure_ref).item()
    assert abs(l4_64 - l4_ref) <= 1e-8 * abs(l4_ref), (l4_64, l4_ref)
    assert (m4.to_caller_order(m4.u_free.detach(), "u").cpu() - uf.detach()).abs().max().item() <= 1e-6 * uf.detach().abs().max().item()
    # LBFGS on the reference's own mini-mesh, fp64: same closure-loss sequence as the reference
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = dev()
    g = g_lbfgs
    model = PiecewiseLinearShapeNN2D(g.t("lbfgs/node_coords"), g.t("lbfgs/conn"), boundary_mask=g.t("lbfgs/boundary_mask"),
                                     dirichlet_mask=g.t("lbfgs/dirichlet_mask"), u_fixed=0.0,
                                     neumann_edges=g.t("lbfgs/edges")).double().to(d)
    with torch.no_grad():
        model.u_free.copy_(g.t("lbfgs/u_free0").to(d))
    loss_fn = EnergyLoss2D(E=10e9, nu=0.3, device=d, dtype=F64)
    opt = torch.optim.LBFGS(model.parameters())
    trace = []

    def closure():
        opt.zero_grad()
        v = loss_fn(model)
        v.backward()
        trace.append(v.item())
        return v

    for _ in range(2):
        opt.step(closure)
    want = g["lbfgs/closure_losses"]
    n = min(len(trace), len(want), 12)
    # LBFGS amplifies rounding differences along the trajectory: the first evaluations are tight,
    # later ones agree to line-search accuracy
    np.testing.assert_allclose(trace[:3], want[:3], rtol=1e-10)
    np.testing.assert_allclose(trace[:n], want[:n], rtol=1e-5)


def test_kernel_variants_agree(g_tri):
    """Every product launch variant of the tiled energy (register-prefetched 'fast' kernel with sc1 or plain stores,
    compile-time or runtime strides, generic loop kernel at 256/1024 threads) gives the same numbers on a mid-size
    mesh with free boundary nodes, a body force and Neumann edges.  Options are defaults captured by the NEXT plan:
    every variant builds its own model (hence its own plan)."""
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from conftest import b_force_fn
    d = dev()
    L = _lib.lib()
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(181, 95, jitter=0.25, seed=9, dtype=F64)
    results = {}
    settings = {
        "fast_sc1": dict(tiled_fast=1, store_policy=16, tiled_block=512, fast_const_caps=1),
        "fast_runtime_strides": dict(tiled_fast=1, store_policy=16, tiled_block=512, fast_const_caps=0),
        "fast_plain": dict(tiled_fast=1, store_policy=0, tiled_block=512),
        "fast_256": dict(tiled_fast=1, store_policy=16, tiled_block=256),
        "loop_256": dict(tiled_fast=0, store_policy=16, tiled_block=256),
        "loop_1024": dict(tiled_fast=0, store_policy=16, tiled_block=1024),
    }
    try:
        for body in (None, b_force_fn):
            for name, opts in settings.items():
                for k, v in opts.items():
                    _lib.check(L.hfem_set_option(k.encode(), v))
                torch.manual_seed(4)
                m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=None, dirichlet_mask=bc, u_fixed=0.0,
                                             neumann_edges=edges).to(d)
                with torch.no_grad():
                    m.u_free.mul_(30.0)
                lf = EnergyLoss2D(device=d, dtype=F64)
                loss = lf(m, b_force=body)
                loss.backward()
                results[(body is None, name)] = (loss.item(), m.node_coords_free.grad.cpu().numpy(),
                                                 m.u_free.grad.cpu().numpy())
            ref = results[(body is None, "loop_256")]
            for name in settings:
                got = results[(body is None, name)]
                assert abs(got[0] - ref[0]) <= 1e-13 * abs(ref[0]), name
                assert np.abs(got[1] - ref[1]).max() <= 1e-12 * np.abs(ref[1]).max(), name
                assert np.abs(got[2] - ref[2]).max() <= 1e-12 * np.abs(ref[2]).max(), name
    finally:
        for k, v in dict(tiled_fast=1, store_policy=16, tiled_block=512, fast_const_caps=1).items():
            L.hfem_set_option(k.encode(), v)


def test_example3_converges_to_the_analytic_bar_solution():
    """Acceptance: examples/example3.py at full length (4000 Adam epochs, fp32 model as the reference runs it):
    the reference reaches loss -0.03138 and max |u_h - u_exact| = 1.9e-4."""
    import examples.example3 as e3
    _, loss, err = e3.run(epochs=4000, log_every=100000)
    assert abs(loss - (-0.03138)) < 2e-4
    assert err < 4e-4
    _, loss_f, err_f = e3.run(epochs=4000, log_every=100000, fused_adam=True)     # fused Adam: same trajectory
    assert abs(loss_f - loss) < 2e-5 and err_f < 4e-4
    _, loss_g, err_g = e3.run(epochs=4000, graphed=True)                          # 40 hipGraph replays of 100 iterations
    assert abs(loss_g - loss) < 2e-5 and err_g < 4e-4


def test_fused_adam_matches_torch_adam():
    from hidenn_fem_amd.optim import FusedAdam
    d = dev()
    for dt, tol in ((F64, 1e-14), (torch.float32, 2e-6)):
        g = torch.Generator().manual_seed(2)
        p0 = [torch.randn(1000, 2, generator=g, dtype=dt), torch.randn(333, generator=g, dtype=dt)]
        grads = [[torch.randn(t.shape, generator=g, dtype=dt) * 10 ** float(k % 5 - 2) for t in p0] for k in range(25)]
        pa = [t.clone().to(d).requires_grad_(True) for t in p0]
        pb = [t.clone().to(d).requires_grad_(True) for t in p0]
        oa, ob = torch.optim.Adam(pa, lr=3e-3), FusedAdam(pb, lr=3e-3)
        for gs in grads:
            for a, b, gi in zip(pa, pb, gs):
                a.grad, b.grad = gi.to(d).clone(), gi.to(d).clone()
            oa.step()
            ob.step()
        for a, b in zip(pa, pb):
            err = (a - b).abs().max().item() / a.abs().max().item()
            assert err <= tol, (dt, err)


# ----------------------------------------------------------------------------- BASELINE configs 2 and 3 at FULL size
def test_cfg2_bar_10k_elements_r_adapt_vs_reference_chain():
    """BASELINE config 2: example 3 at 10 001 nodes on [0, 10], 2 Gauss points per element,
    E = 175, b_force of examples/example3.py:16-24, u0 = uN = 0, r-adaptivity on, u ~ 1e-2 N(0,1), increments
    perturbed 5 %: fused bar energy (one launch: loss + d/du + d/d increments) against the reference's op chain
    (oracle/ref_chain.py: grid_param -> bar_energy with autograd.grad(create_graph=True), example3.py:27-70)."""
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN
    from hidenn_fem_amd.loss import bar_energy_loss
    from hidenn_fem_amd.utils import gauss_legendre_points_weights
    from oracle import ref_chain as R
    d = dev()
    n = 10001
    g = torch.Generator().manual_seed(0)
    grid0 = torch.linspace(0.0, 10.0, n, dtype=F64)
    m = PiecewiseLinearShapeNN(grid0, r_adapt=True, u0=0.0, uN=0.0).double()
    with torch.no_grad():
        m.u.copy_(1e-2 * torch.randn(m.u.shape, generator=g, dtype=F64))
        m.x_increments.mul_(1.0 + 0.05 * (2.0 * torch.rand(m.x_increments.shape, generator=g, dtype=F64) - 1.0))
    u_cpu, inc_cpu = m.u.detach().clone(), m.x_increments.detach().clone()
    m = m.to(d)
    xi, wi = gauss_legendre_points_weights(2, device=d, dtype=F64)
    loss = bar_energy_loss(m, xi, wi, R.example3_body_force, 175.0)
    loss.backward()
    # oracle: the reference chain on the CPU, same parameters
    inc = inc_cpu.clone().requires_grad_(True)
    u = u_cpu.clone().requires_grad_(True)
    grid = R.grid_param(inc, grid0[:1], grid0[-1:])
    u_full = torch.cat([torch.zeros(1, dtype=F64), u, torch.zeros(1, dtype=F64)])      # models.py:58-67 (u0 = uN = 0)
    xi_c, wi_c = R.interval_gauss(2)
    want = R.bar_energy(grid, u_full, xi_c, wi_c, R.example3_body_force, 175.0)
    want.backward()
    assert m.u.shape == u.shape and m.x_increments.shape == inc.shape
    np.testing.assert_allclose(m.grid.detach().cpu().numpy(), grid.detach().numpy(), rtol=1e-13, atol=1e-14)
    assert abs(loss.item() - want.item()) <= 1e-11 * abs(want.item())      # 20 000 terms of mixed sign
    assert_grad_close(m.u.grad, u.grad.numpy(), "cfg2 gu")
    assert_grad_close(m.x_increments.grad, inc.grad.numpy(), "cfg2 gincr")


def test_cfg3_structured_256x256_l2_projection_vs_reference_chain():
    """BASELINE config 3 (SURVEY cfg3): example 2 at 257 x 257 nodes (256 x 256 cells) on [0,1]^2, fixed nodes,
    target sin 2 pi x cos 2 pi y, evaluated at 2 x 2 Gauss points per cell (M = 262 144): fused L2 loss + backward
    against the reference's structured forward (oracle/ref_chain.rectq4_forward = src/models.py:180-212) + MSE
    (examples/example2.py:46).  r-adaptivity on as well: gradients of both increment vectors."""
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import l2_projection_loss
    from oracle import ref_chain as R
    d = dev()
    n = 257
    gx0 = torch.linspace(0.0, 1.0, n, dtype=F64)
    gq = torch.tensor([-1.0, 1.0], dtype=F64) / np.sqrt(3.0)
    mid, half = 0.5 * (gx0[1:] + gx0[:-1]), 0.5 * (gx0[1:] - gx0[:-1])
    q1 = (mid[:, None] + half[:, None] * gq[None, :]).reshape(-1)                      # 512 points per axis
    X, Y = torch.meshgrid(q1, q1, indexing="ij")
    pts = torch.stack([X.reshape(-1), Y.reshape(-1)], dim=1).contiguous()              # [262144, 2]
    target = torch.sin(2 * torch.pi * pts[:, 0]) * torch.cos(2 * torch.pi * pts[:, 1])
    assert pts.shape[0] == 4 * 256 * 256
    for r_adapt in (False, True):
        torch.manual_seed(0)
        m = PiecewiseLinearShapeNN2D(grid_x=gx0, grid_y=gx0, r_adapt=r_adapt).double()
        u_cpu = m.u.detach().clone()
        m = m.to(d)
        loss = l2_projection_loss(m, pts.to(d), target.to(d))
        loss.backward()
        u = u_cpu.clone().requires_grad_(True)
        if r_adapt:
            ix = m.increments_x.detach().cpu().clone().requires_grad_(True)
            iy = m.increments_y.detach().cpu().clone().requires_grad_(True)
            ends = torch.zeros(n, dtype=torch.bool)
            ends[0] = ends[-1] = True                                   # default boundary masks, models.py:118-131
            gx = R.masked_grid(R.grid_param(ix, gx0[:1], gx0[-1:]), ends, gx0)
            gy = R.masked_grid(R.grid_param(iy, gx0[:1], gx0[-1:]), ends, gx0)
        else:
            gx = gy = gx0
        want = R.mse_loss(R.rectq4_forward(gx, gy, u, pts), target)
        want.backward()
        assert abs(loss.item() - want.item()) <= LOSS_RTOL * abs(want.item()), (r_adapt, loss.item(), want.item())
        assert_grad_close(m.u.grad, u.grad.numpy(), f"cfg3 gu r_adapt={r_adapt}")
        if r_adapt:
            # each entry sums 131 072 point contributions of size O(1e-5) that cancel to O(1e-4 .. 1e-9) and then runs
            # through a 256-term reverse cumsum: fp64 summation order (atomics here, autograd's tree there) leaves
            # ~1e-16 * sqrt(N) * cancellation ~ 5e-10 of max|g|; 1e-8 is the stated tolerance for these two vectors
            assert_grad_close(m.increments_x.grad, ix.grad.numpy(), "cfg3 g increments_x", rtol=1e-8)
            assert_grad_close(m.increments_y.grad, iy.grad.numpy(), "cfg3 g increments_y", rtol=1e-8)


@pytest.mark.parametrize("order", [3, 5])
def test_golden_tri3_cases_in_both_element_orders(g_tri, order):
    """The reference's golden TRI3 cases (all gauss orders, body force, traction function, flipped elements, permuted mesh)
    through BOTH production element orders: 3 = one element per slot (tri3_energy_fast_kernel / generic loop kernel),
    5 = paired slots (tri3_pair.hip: two fan-adjacent elements per slot, shared-node contributions added in registers).
    (The strip order 6 -- pairs chained along a row, the carrying slot loop -- is lab-only since round 4.)
    The auto policy picks between them by pairing coverage; parity must not depend on the pick.  Also the fp32-row and
    physical-convention instances of the pair kernel against the one-element-per-slot path."""
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = dev()
    L = _lib.lib()
    prev = L.hfem_get_option(b"plan_elem_order")
    _lib.check(L.hfem_set_option(b"plan_elem_order", order))
    try:
        for case in g_tri.cases():
            go, go1 = (int(v) for v in g_tri[case + "/gauss_order"])
            b, t = tri_case_forces(case)
            m = tri_model_from_golden(g_tri, case, d)
            lf = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=go, gauss_order_1d=go1, device=d, dtype=F64)
            assert m.tile_plan(lf.tile_elems).is_paired() == (order >= 5)
            loss = lf(m, b_force=(lambda x: b(x.cpu()).to(d)) if b else None, t_force=(lambda x: t(x.cpu()).to(d)) if t else None)
            loss.backward()
            want = g_tri[case + "/loss"].item()
            assert abs(loss.item() - want) <= LOSS_RTOL * abs(want), (order, case)
            assert_grad_close(m.u_free.grad, g_tri[case + "/g_u_free"], f"{case} gu (order {order})")
            if m.node_coords_free.numel():
                assert_grad_close(m.node_coords_free.grad, g_tri[case + "/g_coords_free"], f"{case} gx (order {order})")
    finally:
        L.hfem_set_option(b"plan_elem_order", prev)


def test_pair_kernel_variants_match_the_single_slot_kernels():
    """fp32 rows, physical convention, fused Adam and the lagged loss sum on a paired plan (a split-quad mesh: 98 % pairs)
    against the same paths on a one-element-per-slot plan."""
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import EnergyAdamStep
    d = dev()
    L = _lib.lib()
    prev = L.hfem_get_option(b"plan_elem_order")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(201, 151, jitter=0.25, seed=8, flip_fraction=0.1, dtype=F64)
    out = {}
    try:
        _lib.check(L.hfem_set_option(b"plan_elem_order", 6))
        with pytest.raises(RuntimeError, match="lab build only"):      # the strip order (chained pairs): the planner keeps it for
            PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.3,      # host-only plans,
                                     neumann_edges=edges, reorder="off").to(d).tile_plan(0)                    # device plans are lab-only
        for order in (3, 5):
            _lib.check(L.hfem_set_option(b"plan_elem_order", order))
            res = {}
            for tag, dt, conv in (("f64", F64, None), ("phys", F64, "physical"), ("f32", torch.float32, None)):
                torch.manual_seed(4)
                m = PiecewiseLinearShapeNN2D(coords.to(dt), conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.3,
                                             neumann_edges=edges).to(d)
                with torch.no_grad():
                    m.u_free.mul_(50.0)
                lf = EnergyLoss2D(device=d, dtype=dt, grad_convention=conv)
                assert m.tile_plan(lf.tile_elems).is_paired() == (order >= 5)
                if order == 6:                                   # the strip order really chains pairs on this mesh
                    assert ((m.tile_plan(lf.tile_elems).export("elem_pack_hi") >> 12) & 1).sum() > 1000
                v = lf.value_and_grad_(m) if conv is None else None
                if conv is not None:
                    v = lf(m)
                    v.backward()
                res[tag] = (v.item(), m.node_coords_free.grad.double().cpu().numpy(), m.u_free.grad.double().cpu().numpy())
            torch.manual_seed(4)
            m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.3, neumann_edges=edges).to(d)
            tr = EnergyAdamStep(m, EnergyLoss2D(device=d, dtype=F64), lr_x=1e-6, lr_u=1e-8)
            losses = [tr.step().item() for _ in range(4)]
            res["adam"] = (losses, m.node_coords_free.detach().cpu().numpy().copy(), m.u_free.detach().cpu().numpy().copy())
            out[order] = res
    finally:
        L.hfem_set_option(b"plan_elem_order", prev)
    for o in (5,):
        for tag in ("f64", "phys", "f32"):
            a, b = out[o][tag], out[3][tag]
            rt = 1e-12 if tag != "f32" else 1e-6
            assert abs(a[0] - b[0]) <= rt * abs(b[0]), (o, tag)
            assert np.abs(a[1] - b[1]).max() <= (1e-11 if tag != "f32" else 2e-6) * np.abs(b[1]).max(), (o, tag)
            assert np.abs(a[2] - b[2]).max() <= (1e-11 if tag != "f32" else 2e-6) * np.abs(b[2]).max(), (o, tag)
        np.testing.assert_allclose(out[o]["adam"][0], out[3]["adam"][0], rtol=1e-12)
        assert np.abs(out[o]["adam"][1] - out[3]["adam"][1]).max() <= 1e-12 * np.abs(out[3]["adam"][1]).max()
        assert np.abs(out[o]["adam"][2] - out[3]["adam"][2]).max() <= 1e-10 * np.abs(out[3]["adam"][2]).max()

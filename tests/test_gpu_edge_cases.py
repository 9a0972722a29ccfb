"""Edge cases of the newer entry points through the C ABI: empty and minimal inputs, argument errors that must come
back as error codes (never a fault), tiny optimiser problems."""
import ctypes as C

import numpy as np
import pytest
import torch

F64 = torch.float64


@pytest.mark.gpu
def test_empty_and_minimal_inputs_are_handled_by_the_abi():
    from hidenn_fem_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    z = torch.zeros(8, dtype=F64, device=d)
    i32 = torch.zeros(4, dtype=torch.int32, device=d)
    # zero elements / zero rows / a single-node grid: success, nothing written
    assert L.hfem_tri3_von_mises(0, None, None, None, 0, 1e9, 0.3, None, None, st) == 0
    assert L.hfem_line2_slopes(0, z.data_ptr(), z.data_ptr(), 1, 1, z.data_ptr(), st) == 0
    assert L.hfem_iface_pack(0, None, None, None, 0, 0, None, st) == 0
    assert L.hfem_lbfgs_apply(None, None, 0, 0, st) != 0                     # null handle: an error code, not a crash
    assert b"null" in L.hfem_last_error()
    # unpack with no rows still sums the partial energies of `world` payloads
    recv = torch.tensor([[1.5, 0.0], [2.25, 0.0], [4.0, 0.0]], dtype=F64, device=d)      # stride 1: only loss slots
    out = torch.zeros(1, dtype=F64, device=d)
    assert L.hfem_iface_unpack(0, recv.data_ptr(), None, None, 0, 0, None, None, 3, 1, 0, out.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert out.item() == 7.75
    # argument errors
    assert L.hfem_iface_unpack(0, recv.data_ptr(), None, None, 0, 0, None, None, 3, 1, 1, out.data_ptr(), st) != 0   # slot >= stride
    h = C.c_void_p()
    assert L.hfem_lbfgs_create(0, 0, 10, 0, C.byref(h)) != 0 and not h.value
    assert L.hfem_lbfgs_create(0, 5, 0, 0, C.byref(h)) != 0
    assert L.hfem_lbfgs_create(0, 5, 4, 7, C.byref(h)) != 0
    assert L.hfem_grid_param_ws_elems(0) >= 8 and L.hfem_grid_param_ws_elems(5000) >= 5000
    assert L.hfem_adam_prep(0, None, 0.9, 0.999, None, st) != 0
    # one-increment grid through the workspace kernels: grid = {x0, xN}, gradient 0 (renormalisation kills it)
    p = torch.tensor([0.3], dtype=F64, device=d)
    grid, cum = torch.zeros(2, dtype=F64, device=d), torch.zeros(1, dtype=F64, device=d)
    ws = torch.zeros(L.hfem_grid_param_ws_elems(1), dtype=F64, device=d)
    assert L.hfem_grid_param_fwd_ws(0, p.data_ptr(), 1, 2.0, 5.0, None, None, grid.data_ptr(), cum.data_ptr(), ws.data_ptr(), st) == 0
    gg, gp = torch.tensor([1.0, -2.0], dtype=F64, device=d), torch.zeros(1, dtype=F64, device=d)
    assert L.hfem_grid_param_bwd_ws(0, p.data_ptr(), 1, 2.0, 5.0, None, gg.data_ptr(), cum.data_ptr(), gp.data_ptr(), ws.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert grid.tolist() == [2.0, 5.0] and abs(gp.item()) <= 1e-15


@pytest.mark.gpu
def test_lbfgs_on_tiny_problems():
    """n = 1 and n = 3 parameters (fewer than one wave, history longer than the problem is wide): same iterates as torch."""
    from hidenn_fem_amd.optim import FusedLBFGS
    d = torch.device("cuda:0")
    for n in (1, 3):
        A = torch.diag(torch.linspace(1.0, 4.0, n, dtype=F64)).to(d)
        b = torch.arange(1, n + 1, dtype=F64, device=d)

        def run(cls):
            p = torch.nn.Parameter(torch.full((n,), 2.0, dtype=F64, device=d))
            opt = cls([p])
            out = []

            def closure():
                opt.zero_grad()
                loss = 0.5 * p @ (A @ p) - b @ p + 0.1 * (p ** 4).sum()
                loss.backward()
                return loss

            for _ in range(3):
                out.append(opt.step(closure).item())
            return out, p.detach().cpu().numpy()

        ref_l, ref_p = run(torch.optim.LBFGS)
        got_l, got_p = run(FusedLBFGS)
        np.testing.assert_allclose(got_l, ref_l, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(got_p, ref_p, rtol=1e-8, atol=1e-10)


@pytest.mark.gpu
def test_planned_energy_on_degenerate_meshes():
    """hfem_tri3_energy_plan on plans that are mostly padding: nodes without any element (orphan-node tiles only: zero
    energy, zero gradient rows), one single element, and one element plus 700 unreferenced nodes.  The kernels load their
    index arrays unguarded from uniform strides, so these sizes exercise the stride floors and the tail padding."""
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.plan import TilePlan
    from oracle import closed_form as CF
    L = _lib.lib()
    d = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    dv = lambda a: (C.c_double * len(a))(*a)
    mat, W = CF.plane_stress(), 0.25
    rng = np.random.default_rng(0)
    cases = {
        "no elements": (np.zeros((0, 3), dtype=np.int64), 40),
        "one element": (np.array([[0, 1, 2]], dtype=np.int64), 3),
        "one element + orphans": (np.array([[5, 9, 2]], dtype=np.int64), 703),
    }
    for name, (conn, nn) in cases.items():
        X = rng.random((nn, 2))
        U = 1e-3 * rng.standard_normal((nn, 2))
        for order in (3, 5):
            plan = TilePlan(conn, nn, coords_hint=X, edges=np.zeros((0, 2), dtype=np.int64), device=d, elem_order=order)
            Xd, Ud = torch.from_numpy(X).to(d), torch.from_numpy(U).to(d)
            loss = torch.full((), 7.0, dtype=F64, device=d)
            gX, gU = torch.full_like(Xd, float("nan")), torch.full_like(Ud, float("nan"))
            _lib.check(L.hfem_tri3_energy_plan(plan.handle, Xd.data_ptr(), None, Ud.data_ptr(), None, dv(mat), W, dv([0.0] * 6),
                                               None, dv([0.0] * 4), 0, -1, loss.data_ptr(), gX.data_ptr(), gU.data_ptr(), 0, st))
            torch.cuda.synchronize()
            if conn.shape[0]:
                e_ref, gX_ref, gU_ref = CF.tri3_energy(X, U, conn, mat, W)
            else:
                e_ref, gX_ref, gU_ref = 0.0, np.zeros_like(X), np.zeros_like(U)
            assert abs(loss.item() - e_ref) <= 1e-12 * max(abs(e_ref), 1e-300), (name, order)
            assert not torch.isnan(gX).any() and not torch.isnan(gU).any(), (name, order)      # every row written
            assert np.abs(gX.cpu().numpy() - gX_ref).max() <= 1e-10 * max(np.abs(gX_ref).max(), 1e-300), (name, order)
            assert np.abs(gU.cpu().numpy() - gU_ref).max() <= 1e-10 * max(np.abs(gU_ref).max(), 1e-300), (name, order)
            plan.close()


@pytest.mark.gpu
def test_tiny_models_through_every_planned_entry_point():
    """One-quad / two-triangle models through the host mirror: QUAD4 plan kernel, fp32-row TRI3 kernel, fused energy+Adam
    step -- the planned kernels on tiles that are almost entirely stride padding."""
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import EnergyAdamStep
    d = torch.device("cuda:0")
    coords = torch.tensor([[0.0, 0.0], [1.0, 0.0], [1.1, 0.9], [0.0, 1.0], [2.0, 0.1], [2.1, 1.2]], dtype=F64)
    bc = torch.tensor([True, False, False, True, False, False])
    for conn in (torch.tensor([[0, 1, 2, 3], [1, 4, 5, 2]]), torch.tensor([[0, 1, 2], [0, 2, 3], [1, 4, 5], [1, 5, 2]])):
        for dt in (F64, torch.float32):
            torch.manual_seed(3)
            m = PiecewiseLinearShapeNN2D(coords.to(dt), conn, boundary_mask=None, dirichlet_mask=bc, u_fixed=0.0).to(d)
            with torch.no_grad():
                m.u_free.mul_(1e-3)
            lf = EnergyLoss2D(device=d, dtype=dt)
            loss = lf(m)
            loss.backward()
            assert torch.isfinite(loss) and torch.isfinite(m.u_free.grad).all() and torch.isfinite(m.node_coords_free.grad).all()
            # oracle on the same numbers
            X = m.coords.detach().double().cpu()
            U = m.u_full.detach().double().cpu()
            if conn.shape[1] == 4:
                from oracle import closed_form as CF
                want = CF.quad4_energy(X.numpy(), U.numpy(), conn.numpy(), CF.plane_stress())[0]
            else:
                from oracle import closed_form as CF
                want = CF.tri3_energy(X.numpy(), U.numpy(), conn.numpy(), CF.plane_stress(), lf._W)[0]
            tol = 1e-11 if dt == F64 else 2e-5
            assert abs(loss.item() - want) <= tol * abs(want), (conn.shape, dt, loss.item(), want)
    # fused energy + Adam step on the four-triangle model == energy launch + FusedAdam launch on a copy, step by step
    from hidenn_fem_amd.optim import FusedAdam
    tri = torch.tensor([[0, 1, 2], [0, 2, 3], [1, 4, 5], [1, 5, 2]])
    models = []
    for _ in range(2):
        torch.manual_seed(5)
        m = PiecewiseLinearShapeNN2D(coords, tri, boundary_mask=None, dirichlet_mask=bc, u_fixed=0.0).double().to(d)
        with torch.no_grad():
            m.u_free.mul_(1e-3)
        models.append(m)
    lf = EnergyLoss2D(device=d, dtype=F64)
    tr = EnergyAdamStep(models[0], lf, lr_x=1e-7, lr_u=1e-9)
    opt = FusedAdam([dict(params=[models[1].node_coords_free], lr=1e-7), dict(params=[models[1].u_free], lr=1e-9)])
    for _ in range(5):
        l0 = tr.step().item()
        l1 = lf.value_and_grad_(models[1]).item()
        opt.step()
        assert abs(l0 - l1) <= 1e-12 * abs(l1)
    assert torch.allclose(models[0].u_free, models[1].u_free, rtol=1e-12, atol=0)
    assert torch.allclose(models[0].node_coords_free, models[1].node_coords_free, rtol=1e-13, atol=0)


@pytest.mark.gpu
def test_fp32_arithmetic_kernel_on_degenerate_paired_plans():
    """hfem_tri3_energy_plan_f32 + HFEM_FLAG_FP32_MATH (csrc/tri3_pair_f32.hip) on paired plans that are almost entirely stride
    padding: no element at all (zero energy, every row written with zeros), one unpaired element, one split quad (one full
    pair), and the pair among 700 unreferenced nodes -- against the C closed forms in fp64 on the same float values."""
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.plan import TilePlan
    from oracle import closed_form as CF
    L = _lib.lib()
    d = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    dv = lambda a: (C.c_double * len(a))(*a)
    mat, W = CF.plane_stress(), 0.25
    rng = np.random.default_rng(1)
    cases = {
        "no elements": (np.zeros((0, 3), dtype=np.int64), 40),
        "one element": (np.array([[0, 1, 2]], dtype=np.int64), 3),
        "one pair": (np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int64), 4),
        "one pair + orphans": (np.array([[5, 9, 2], [5, 2, 700]], dtype=np.int64), 703),
    }
    for name, (conn, nn) in cases.items():
        X = rng.random((nn, 2)).astype(np.float32)
        quad = np.array([[0.0, 0.0], [1.0, 0.1], [1.1, 1.0], [0.1, 0.9]], dtype=np.float32)      # a proper quad for the element cases
        X[:min(nn, 4)] = quad[:min(nn, 4)]
        if nn > 700:
            X[[5, 9, 2, 700]] = quad
        U = (1e-3 * rng.standard_normal((nn, 2))).astype(np.float32)
        plan = TilePlan(conn, nn, coords_hint=X.astype(np.float64), edges=np.zeros((0, 2), dtype=np.int64), device=d, elem_order=5)
        assert plan.stats["paired"] == 1, name
        Xd, Ud = torch.from_numpy(X).to(d), torch.from_numpy(U).to(d)
        loss = torch.full((), 7.0, dtype=F64, device=d)
        gX, gU = torch.full_like(Xd, float("nan")), torch.full_like(Ud, float("nan"))
        _lib.check(L.hfem_tri3_energy_plan_f32(plan.handle, Xd.data_ptr(), None, Ud.data_ptr(), None, dv(mat), W, dv([0.0] * 6),
                                               None, dv([0.0] * 4), 0, -1, loss.data_ptr(), gX.data_ptr(), gU.data_ptr(), 1024, st))
        torch.cuda.synchronize()
        if conn.shape[0]:
            e_ref, gX_ref, gU_ref = CF.tri3_energy(X.astype(np.float64), U.astype(np.float64), conn, mat, W)
        else:
            e_ref, gX_ref, gU_ref = 0.0, np.zeros((nn, 2)), np.zeros((nn, 2))
        assert abs(loss.item() - e_ref) <= 2e-6 * max(abs(e_ref), 1e-300), (name, loss.item(), e_ref)
        assert not torch.isnan(gX).any() and not torch.isnan(gU).any(), name                        # every row written
        assert np.abs(gX.double().cpu().numpy() - gX_ref).max() <= 4e-6 * max(np.abs(gX_ref).max(), 1e-300), name
        assert np.abs(gU.double().cpu().numpy() - gU_ref).max() <= 4e-6 * max(np.abs(gU_ref).max(), 1e-300), name
        plan.close()


@pytest.mark.gpu
def test_sharded_lbfgs_on_a_tiny_model_matches_fused_lbfgs():
    """ShardedLBFGS (one rank, eager and with the steady-state iteration as a hipGraph) on a four-triangle model -- 8 owned
    parameters, history far longer than the problem is wide, tiles that are all padding: same losses and parameters as
    FusedLBFGS on the same energy."""
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.optim import FusedLBFGS, ShardedLBFGS
    from hidenn_fem_amd.sharded import ShardedTri3Energy
    d = torch.device("cuda:0")
    coords = torch.tensor([[0.0, 0.0], [1.0, 0.0], [1.1, 0.9], [0.0, 1.0], [2.0, 0.1], [2.1, 1.2]], dtype=F64)
    bc = torch.tensor([True, False, False, True, False, False])
    tri = torch.tensor([[0, 1, 2], [0, 2, 3], [1, 4, 5], [1, 5, 2]])
    edges = torch.tensor([[4, 5]])

    def model():
        torch.manual_seed(5)
        m = PiecewiseLinearShapeNN2D(coords, tri, boundary_mask=None, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).double().to(d)
        with torch.no_grad():
            m.u_free.mul_(1e-3)
        return m

    lf = EnergyLoss2D(E=10.0, nu=0.3, device=d, dtype=F64)
    ref_m = model()
    ref = FusedLBFGS(ref_m.parameters())
    ref_l = [ref.step(lambda: lf.value_and_grad_(ref_m)).item() for _ in range(3)]
    for graph in (False, True):
        m = model()
        opt = ShardedLBFGS(ShardedTri3Energy(m, lf).setup_interfaces(), graph=graph)
        got = [opt.step().item() for _ in range(3)]
        opt.finish()
        np.testing.assert_allclose(got, ref_l, rtol=1e-9, atol=1e-14, err_msg=f"graph={graph}")
        assert opt.state["func_evals"] == ref.state[ref._params[0]]["func_evals"], graph
        assert torch.allclose(m.u_free, ref_m.u_free, rtol=1e-7, atol=1e-12), graph
        assert torch.allclose(m.node_coords_free, ref_m.node_coords_free, rtol=1e-9, atol=1e-12), graph

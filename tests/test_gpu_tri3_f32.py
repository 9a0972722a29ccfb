"""The reference's DEFAULT dtype on the GPU: fp32 models (``/root/reference/src/loss.py:16``, ``src/models.py:274``; examples 1-4
as shipped) against vectors the reference itself produced IN fp32 (tests/golden/g7_tri3_f32.npz, ex1_f32_* in g3_line.npz --
make_golden.py runs the reference unmodified, no ``.double()``).

Tolerances.  The reference's own fp32 arithmetic differs from exact arithmetic on the same float inputs by up to 6e-7 x
max|g| (gradients) and 2e-7 relative (loss) on these meshes (fixture keys ``*64`` = the reference run in fp64 on the same float
values; tests/test_oracle_golden.py asserts the band).  So:
* vs the reference's fp32 output: loss rel <= 2e-6, gradients max-abs <= 4e-6 x max|g| -- both sides carry fp32 rounding;
* fp64-arithmetic instances (float rows widened on load, one rounding on store) vs the ``*64`` values: loss rel and gradients
  <= 2.5e-7 (ONE fp32 rounding of the result, plus the fp32 loss object's fp32-rounded constants C, W -- the ``*64`` run used
  fp64 constants);
* fp32-arithmetic instances vs the ``*64`` values: gradients <= 4e-6 x max|g| -- no worse than the band the reference itself
  occupies."""
import numpy as np
import pytest
import torch

from conftest import tri_case_forces_f32

pytestmark = pytest.mark.gpu
F32, F64 = torch.float32, torch.float64
EPS32 = float(torch.finfo(F32).eps)


def _model(g, case, d):
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    p = case + "/"
    nc = g.t(p + "node_coords")
    assert nc.dtype == F32
    m = PiecewiseLinearShapeNN2D(nc, g.t(p + "conn"), boundary_mask=g.t(p + "boundary_mask"), dirichlet_mask=g.t(p + "dirichlet_mask"),
                                 u_fixed=0.0, neumann_edges=g.t(p + "edges")).to(d)
    with torch.no_grad():
        m.u_free.copy_(m.from_caller_order(g.t(p + "u_free").to(d), "u"))
    assert m.node_coords_free.dtype == F32 and m.u_free.dtype == F32
    return m


def _err(got, want):
    want = np.asarray(want, dtype=np.float64)
    return float(np.abs(got.detach().double().cpu().numpy() - want).max() / np.abs(want).max())


@pytest.mark.parametrize("arithmetic", ["fp64", "auto"])
def test_fp32_models_against_the_reference_run_in_fp32(g_tri_f32, arithmetic):
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = torch.device("cuda:0")
    worst = dict(loss=0.0, g=0.0, loss64=0.0, g64=0.0)
    ran_fp32 = []
    for case in g_tri_f32.cases():
        m = _model(g_tri_f32, case, d)
        go, go1 = (int(v) for v in g_tri_f32[case + "/gauss_order"])
        b, t = tri_case_forces_f32(case)
        lf = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=go, gauss_order_1d=go1, device=d, dtype=F32, arithmetic=arithmetic)
        loss = lf(m, b_force=b, t_force=t)
        loss.backward()
        if arithmetic == "auto" and m.tile_plan(0).stats["paired"]:      # "auto": fp32 arithmetic wherever the plan has paired slots
            ran_fp32.append(case)
        assert loss.dtype == F32 and m.u_free.grad.dtype == F32
        gx = m.to_caller_order(m.node_coords_free.grad, "x")
        gu = m.to_caller_order(m.u_free.grad, "u")
        p = case + "/"
        e_l = abs(loss.item() - g_tri_f32[p + "loss"].item()) / abs(g_tri_f32[p + "loss"].item())
        e_g = max(_err(gx, g_tri_f32[p + "g_coords_free"]), _err(gu, g_tri_f32[p + "g_u_free"]))
        e_l64 = abs(loss.item() - g_tri_f32[p + "loss64"].item()) / abs(g_tri_f32[p + "loss64"].item())
        e_g64 = max(_err(gx, g_tri_f32[p + "g_coords_free64"]), _err(gu, g_tri_f32[p + "g_u_free64"]))
        assert e_l <= 2e-6 and e_g <= 4e-6, (case, arithmetic, e_l, e_g)
        if arithmetic == "fp64":
            assert e_l64 <= 2.5e-7 and e_g64 <= 2.5e-7, (case, e_l64, e_g64)
        else:
            assert e_l64 <= 2e-6 and e_g64 <= 4e-6, (case, e_l64, e_g64)
        for k, v in (("loss", e_l), ("g", e_g), ("loss64", e_l64), ("g64", e_g64)):
            worst[k] = max(worst[k], v)
        if b is None and t is None:              # the autograd-free form (default forces) leaves the same rows in .grad
            m.zero_grad()
            l2 = lf.value_and_grad_(m)
            assert abs(l2.item() - loss.item()) <= 4 * EPS32 * abs(loss.item())
            assert _err(m.to_caller_order(m.u_free.grad, "u"), gu.double().cpu().numpy()) <= (4 * EPS32 if arithmetic == "fp64" else 4e-6)
    if arithmetic == "auto":
        assert len(ran_fp32) >= 6, ran_fp32                      # most golden meshes pair up: the fp32 kernel really ran
        with pytest.raises(RuntimeError):                        # ... and where it cannot, asking for it by name fails loudly
            m_flip = _model(g_tri_f32, "flipped", d)
            if m_flip.tile_plan(0).stats["paired"]:
                raise RuntimeError("paired after all")
            EnergyLoss2D(device=d, dtype=F32, arithmetic="fp32")(m_flip)
    print(f"[fp32 goldens, arithmetic={arithmetic}, fp32 kernel on {len(ran_fp32)} cases] worst vs reference-fp32: loss {worst['loss']:.2e} grad {worst['g']:.2e}; "
          f"vs reference-fp64-on-the-same-floats: loss {worst['loss64']:.2e} grad {worst['g64']:.2e}")


def test_example1_as_shipped_fp32_adam_trajectory(g_line):
    """examples/example1.py:25-42 exactly as the reference ships it -- fp32 model, fp32 data, torch.optim.Adam -- first 20
    losses against the reference's own fp32 run (ex1_f32_r*: generated since round 1, consumed from round 4 on).  Adam's
    early steps are lr * sign(g)-like, which amplifies fp32 rounding of near-zero gradient entries: 2e-4 relative."""
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN
    from hidenn_fem_amd.loss import l2_projection_loss
    d = torch.device("cuda:0")
    for r_adapt in (True, False):
        xg = torch.linspace(0, 1, 100, dtype=F32)
        xt = torch.linspace(0, 1, 1000, dtype=F32).to(d)
        ut = torch.sin(2 * torch.pi * xt)
        m = PiecewiseLinearShapeNN(xg, r_adapt=r_adapt).to(d)
        assert m.u.dtype == F32
        opt = torch.optim.Adam(m.parameters(), lr=0.005)
        got = []
        for _ in range(20):
            opt.zero_grad()
            loss = l2_projection_loss(m, xt, ut)
            assert loss.dtype == F32
            loss.backward()
            opt.step()
            got.append(loss.item())
        np.testing.assert_allclose(got, g_line[f"ex1_f32_r{int(r_adapt)}/adam_losses"], rtol=2e-4)


def test_fp32_arithmetic_entry_point_ranges_lagged_sum_and_body_force():
    """hfem_tri3_energy_plan_f32 with HFEM_FLAG_FP32_MATH through the C ABI on a 60 k-element mesh: tile ranges are additive and
    write disjoint rows, the lagged loss sum (NO_LOSS_SUM -> SUM_PREVIOUS -> hfem_plan_loss_sum) delivers the same energies as
    separate reductions, a body force is taken (the fp64-arithmetic float-row instance refuses one) -- all against the C closed
    forms evaluated in fp64 on the same float values (fp32 band: loss 2e-6, gradients 4e-6 x max|g|)."""
    import ctypes as C
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.plan import TilePlan
    from oracle import closed_form as CF
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(201, 151, jitter=0.2, seed=5, dtype=F64)
    rng = np.random.default_rng(5)
    X = coords.numpy().astype(np.float32)
    U = (1e-5 * rng.standard_normal(X.shape)).astype(np.float32)
    mat, W = CF.plane_stress(), 0.25
    Bk = (rng.standard_normal(6) * 1e4).astype(np.float32).astype(np.float64)
    Tc = np.array([2e5, 0.0, 0.0, 1e4])
    L = _lib.lib()
    s = _lib.stream_ptr(d)
    dv = lambda a: (C.c_double * len(a))(*a)
    Xd, Ud = torch.from_numpy(X).to(d), torch.from_numpy(U).to(d)
    plan = TilePlan(conn, X.shape[0], coords_hint=coords, edges=edges, device=d)
    assert plan.stats["paired"] == 1

    def launch(lo, hi, flags, bk, loss, gx, gu):
        _lib.check(L.hfem_tri3_energy_plan_f32(plan.handle, Xd.data_ptr(), None, Ud.data_ptr(), None, dv(mat), W, dv(bk), None, dv(Tc),
                                               lo, hi, loss.data_ptr(), gx.data_ptr(), gu.data_ptr(), flags | 1024, s), "f32 math")

    for bk in (np.zeros(6), Bk):
        e_ref, gX_ref, gU_ref = CF.tri3_energy(X.astype(np.float64), U.astype(np.float64), conn.numpy(), mat, W, bk)
        e_ref -= CF.edge2_energy(X.astype(np.float64), U.astype(np.float64), edges.numpy(), Tconst=Tc, gX=gX_ref, gU=gU_ref)
        loss = torch.zeros((), dtype=F64, device=d)
        gx, gu = torch.full_like(Xd, float("nan")), torch.full_like(Ud, float("nan"))
        launch(0, -1, 0, bk, loss, gx, gu)
        assert abs(loss.item() - e_ref) <= 2e-6 * abs(e_ref)
        assert _err(gx, gX_ref) <= 4e-6 and _err(gu, gU_ref) <= 4e-6
        # three tile ranges: additive energies, every row written by exactly one range
        acc, cover = 0.0, torch.zeros(X.shape[0], dtype=torch.int32, device=d)
        gxa, gua = torch.zeros_like(Xd), torch.zeros_like(Ud)
        for r in range(3):
            lo, hi = plan.shard_range(r, 3)
            l3 = torch.zeros((), dtype=F64, device=d)
            g3x, g3u = torch.full_like(Xd, float("nan")), torch.full_like(Ud, float("nan"))
            launch(lo, hi, 0, bk, l3, g3x, g3u)
            acc += l3.item()
            cover += (~torch.isnan(g3x[:, 0])).int()
            gxa += torch.nan_to_num(g3x)
            gua += torch.nan_to_num(g3u)
        assert int(cover.min()) == 1 and int(cover.max()) == 1
        assert abs(acc - loss.item()) <= 1e-12 * abs(acc)
        assert torch.equal(gxa, gx) and torch.equal(gua, gu) or (_err(gxa, gx.double().cpu().numpy()) <= 2 * EPS32)
    # lagged loss sum (zero body force): launch k delivers the energy of launch k - 1, the flush the last one
    got = torch.zeros(3, dtype=F64, device=d)
    launch(0, -1, 8, np.zeros(6), got[0:1], gx, gu)                   # NO_LOSS_SUM: nothing delivered
    Ud2 = (Ud * 1.5).contiguous()
    _lib.check(L.hfem_tri3_energy_plan_f32(plan.handle, Xd.data_ptr(), None, Ud2.data_ptr(), None, dv(mat), W, dv(np.zeros(6)), None, dv(Tc),
                                           0, -1, got[0:1].data_ptr(), gx.data_ptr(), gu.data_ptr(), 8 | 32 | 1024, s), "lagged")
    _lib.check(L.hfem_plan_loss_sum(plan.handle, 0, -1, got[1:2].data_ptr(), s), "flush")
    l_a, l_b = torch.zeros((), dtype=F64, device=d), torch.zeros((), dtype=F64, device=d)
    launch(0, -1, 0, np.zeros(6), l_a, gx, gu)
    _lib.check(L.hfem_tri3_energy_plan_f32(plan.handle, Xd.data_ptr(), None, Ud2.data_ptr(), None, dv(mat), W, dv(np.zeros(6)), None, dv(Tc),
                                           0, -1, l_b.data_ptr(), gx.data_ptr(), gu.data_ptr(), 1024, s), "plain")
    assert got[0].item() == l_a.item() and got[1].item() == l_b.item(), (got.tolist(), l_a.item(), l_b.item())
    # the fp64-arithmetic float-row instance refuses a body force (the caller widens instead); with the flag it is taken
    with pytest.raises(RuntimeError):
        _lib.check(L.hfem_tri3_energy_plan_f32(plan.handle, Xd.data_ptr(), None, Ud.data_ptr(), None, dv(mat), W, dv(Bk), None, dv(Tc),
                                               0, -1, loss.data_ptr(), gx.data_ptr(), gu.data_ptr(), 0, s), "no flag")
    plan.close()

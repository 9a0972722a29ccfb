"""FusedLBFGS (csrc/lbfgs.hip) against torch.optim.LBFGS -- the optimiser the reference's example 4 runs
(/root/reference/examples/example4.py:68-78) -- on the same GPU, same closure, same defaults.  The two evaluate the
same recursion in a different order (coefficient space vs vector space), so trajectories agree to rounding, not bits;
the tolerances below are what fp64 / fp32 rounding through tens of quasi-Newton iterations leaves."""
import copy

import numpy as np
import pytest
import torch

F64 = torch.float64


def _quadratic(d, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    n1, n2 = 37, 50
    n = n1 + n2
    Q = torch.randn(n, n, generator=g, dtype=F64)
    A = (Q @ Q.T / n + torch.diag(torch.linspace(0.5, 20.0, n, dtype=F64))).to(dtype).to(d)
    b = torch.randn(n, generator=g, dtype=F64).to(dtype).to(d)
    p1 = torch.nn.Parameter(torch.randn(n1, generator=g, dtype=F64).to(dtype).to(d))
    p2 = torch.nn.Parameter(torch.randn(n2 // 2, 2, generator=g, dtype=F64).to(dtype).to(d))

    def f(a1, a2):
        x = torch.cat([a1, a2.reshape(-1)])
        return 0.5 * x @ (A @ x) - b @ x

    return [p1, p2], f


def _run(opt_cls, params, f, n_steps, **kw):
    params = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt = opt_cls(params, **kw)
    losses = []

    def closure():
        opt.zero_grad()
        loss = f(*params)
        loss.backward()
        return loss

    for _ in range(n_steps):
        losses.append(opt.step(closure).item())
    return losses, params, opt


@pytest.mark.gpu
@pytest.mark.parametrize("history", [100, 5])
def test_fused_lbfgs_matches_torch_on_a_quadratic(history):
    """history 5 makes the ring wrap many times; two parameter tensors exercise the segment offsets."""
    from hidenn_fem_amd.optim import FusedLBFGS
    d = torch.device("cuda:0")
    params, f = _quadratic(d, F64)
    ref_l, ref_p, ref_o = _run(torch.optim.LBFGS, params, f, 4, history_size=history)
    got_l, got_p, got_o = _run(FusedLBFGS, params, f, 4, history_size=history)
    assert ref_l[0] == got_l[0]
    np.testing.assert_allclose(got_l, ref_l, rtol=1e-9, atol=1e-9)
    for a, b in zip(got_p, ref_p):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-7, atol=1e-9)
    assert got_o.state[got_p[0]]["n_iter"] == ref_o.state[ref_p[0]]["n_iter"]
    assert got_o.state[got_p[0]]["func_evals"] == ref_o.state[ref_p[0]]["func_evals"]
    assert ref_l[-1] < ref_l[0] - 1.0                      # the optimiser really moved


@pytest.mark.gpu
def test_fused_lbfgs_fp32_and_stopping_rules():
    from hidenn_fem_amd.optim import FusedLBFGS
    d = torch.device("cuda:0")
    params, f = _quadratic(d, torch.float32, seed=1)
    ref_l, ref_p, _ = _run(torch.optim.LBFGS, params, f, 2)
    got_l, got_p, _ = _run(FusedLBFGS, params, f, 2)
    np.testing.assert_allclose(got_l, ref_l, rtol=2e-4, atol=1e-3)
    # converged problem: both stop on the gradient tolerance at the first evaluation of a later step
    params, f = _quadratic(d, F64, seed=2)
    ref_l, ref_p, ref_o = _run(torch.optim.LBFGS, params, f, 12, tolerance_grad=1e-6)
    got_l, got_p, got_o = _run(FusedLBFGS, params, f, 12, tolerance_grad=1e-6)
    assert abs(got_l[-1] - ref_l[-1]) <= 1e-9 * abs(ref_l[-1])
    assert got_o.state[got_p[0]]["n_iter"] == ref_o.state[ref_p[0]]["n_iter"]
    with pytest.raises(NotImplementedError):
        FusedLBFGS(params, line_search_fn="strong_wolfe")
    with pytest.raises(RuntimeError):
        FusedLBFGS([torch.nn.Parameter(torch.zeros(3))])   # CPU tensors fail loudly


@pytest.mark.gpu
def test_fused_lbfgs_on_example4_energy(g_lbfgs):
    """Mini example 4 (the G6 golden case: 41 x 21 nodes, fp64): the closure-loss sequence follows torch.optim.LBFGS
    on the same fused energy closure, and the trace the reference code itself produced (golden vectors)."""
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import FusedLBFGS
    d = torch.device("cuda:0")
    g = g_lbfgs
    base = PiecewiseLinearShapeNN2D(g.t("lbfgs/node_coords"), g.t("lbfgs/conn"), boundary_mask=g.t("lbfgs/boundary_mask"),
                                    dirichlet_mask=g.t("lbfgs/dirichlet_mask"), u_fixed=0.0,
                                    neumann_edges=g.t("lbfgs/edges")).double().to(d)
    with torch.no_grad():
        base.u_free.copy_(g.t("lbfgs/u_free0").to(d))
    lf = EnergyLoss2D(E=10e9, nu=0.3, device=d, dtype=F64)

    def run(opt_cls, n_steps):
        m = copy.deepcopy(base)
        opt = opt_cls(m.parameters())
        trace = []

        def closure():
            opt.zero_grad()
            loss = lf(m)
            loss.backward()
            trace.append(loss.item())
            return loss

        for _ in range(n_steps):
            opt.step(closure)
        return trace, m

    ref, m_ref = run(torch.optim.LBFGS, 2)
    got, m_got = run(FusedLBFGS, 2)
    assert len(got) == len(ref) == 40
    np.testing.assert_allclose(got[:3], ref[:3], rtol=1e-12)
    np.testing.assert_allclose(got, ref, rtol=1e-6)
    want = g["lbfgs/closure_losses"]                              # produced by the reference (tests/golden/make_golden.py)
    np.testing.assert_allclose(got[:3], want[:3], rtol=1e-10)
    np.testing.assert_allclose(got[:12], want[:12], rtol=1e-5)
    du = (m_got.u_free - m_ref.u_free).abs().max().item()
    assert du <= 1e-6 * m_ref.u_free.abs().max().item()


@pytest.mark.gpu
def test_fused_lbfgs_step_returns_the_first_closure_value_with_a_static_loss_tensor():
    """``EnergyLoss2D.value_and_grad_`` returns ONE static loss tensor that every later closure call overwrites;
    ``FusedLBFGS.step`` must still return the loss of the step's FIRST evaluation, as torch.optim.LBFGS does
    (example 4 with ``fused_lbfgs=True`` prints it)."""
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import FusedLBFGS
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(41, 21, jitter=0.1, seed=3, dtype=F64)
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                 neumann_edges=edges).double().to(d)
    lf = EnergyLoss2D(E=10e9, nu=0.3, device=d, dtype=F64)
    opt = FusedLBFGS(m.parameters())
    seen = []

    def closure():
        v = lf.value_and_grad_(m)
        seen.append(v.item())
        return v

    for _ in range(2):
        n0 = len(seen)
        ret = opt.step(closure)
        assert len(seen) - n0 > 1                       # the step really re-evaluated (and overwrote the static tensor)
        assert ret.item() == seen[n0] != seen[-1]
    import examples.example4 as e4                      # and the example's fused path runs end to end
    _, final = e4.run(nx=40, ny=20, steps=2, dtype=F64, log_every=1, fused_lbfgs=True)
    assert np.isfinite(final)

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
@pytest.mark.parametrize("history", [100, 5, 64, 65, 128, 129, 300])
def test_fused_lbfgs_matches_torch_on_a_quadratic(history):
    """history 5 makes the ring wrap many times; two parameter tensors exercise the segment offsets.  The other sizes sit on the
    recursion kernels' boundaries (csrc/lbfgs.hip): 64 / 65 = one / two rows per lane of the one-wavefront form, 128 = its
    largest history (129 KB of LDS for S^T Y), 129 = the 256-thread form, 300 = the reduction form."""
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


# ---------------------------------------------------------------------------------------------------------------------
# Node-sharded L-BFGS (optim.ShardedLBFGS, hfem_lbfgs_shard_*): example 4's optimiser owner-sharded end to end
def _e4_model(d, dtype=F64, nx=121, ny=81):
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(nx, ny, jitter=0.1, seed=3, dtype=dtype)
    torch.manual_seed(0)
    return PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)


def _fused_reference(d, dtype, n_steps, history):
    """The unsharded trajectory: FusedLBFGS on value_and_grad_ (pinned against torch.optim.LBFGS and the reference's own
    trace above)."""
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import FusedLBFGS
    m = _e4_model(d, dtype)
    lf = EnergyLoss2D(E=10e9, nu=0.3, device=d, dtype=dtype, arithmetic="fp64")
    opt = FusedLBFGS(m.parameters(), history_size=history)
    rets = [opt.step(lambda: lf.value_and_grad_(m)).item() for _ in range(n_steps)]
    return rets, m, opt


def _sharded_run(d, dtype, n_steps, history, group=None):
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import ShardedLBFGS
    from hidenn_fem_amd.sharded import ShardedTri3Energy
    m = _e4_model(d, dtype)
    sh = ShardedTri3Energy(m, EnergyLoss2D(E=10e9, nu=0.3, device=d, dtype=dtype), group=group).setup_interfaces()
    opt = ShardedLBFGS(sh, history_size=history)
    rets = [opt.step().item() for _ in range(n_steps)]
    opt.finish()
    torch.cuda.synchronize()
    return rets, m, opt, sh


@pytest.mark.gpu
@pytest.mark.parametrize("history,dtype", [(100, F64), (7, F64), (100, torch.float32)])
def test_sharded_lbfgs_one_rank_follows_fused_lbfgs(history, dtype):
    """world = 1: the sharded flow (speculative pair, payload, finish) is the unsharded algorithm -- same returned losses,
    same n_iter / func_evals, same parameters, to the tolerance FusedLBFGS holds against torch.optim.LBFGS (history 7: the ring
    wraps and drops its oldest pair on nearly every iteration)."""
    d = torch.device("cuda:0")
    n = 3
    ref, m_ref, o_ref = _fused_reference(d, dtype, n, history)
    got, m_got, o_got, sh = _sharded_run(d, dtype, n, history)
    tol = 1e-8 if dtype == F64 else 2e-4
    np.testing.assert_allclose(got, ref, rtol=tol)
    assert got[0] == pytest.approx(ref[0], rel=1e-14 if dtype == F64 else 1e-6)
    st = o_ref.state[o_ref._params[0]]
    assert (o_got.state["n_iter"], o_got.state["func_evals"]) == (st["n_iter"], st["func_evals"])
    for a, b in zip(m_got.parameters(), m_ref.parameters()):
        scale = b.detach().abs().max().item()
        assert (a.detach() - b.detach()).abs().max().item() <= (1e-7 if dtype == F64 else 1e-3) * scale
    assert ref[-1] < ref[0]                                # the optimiser really moved


@pytest.mark.gpu
def test_sharded_lbfgs_several_iterations_per_graph_stop_where_the_eager_loop_stops():
    """graph_iterations = k replays k inner iterations per hipGraph and reads the status only afterwards: an iteration behind the
    one that ended the step (break tests of torch/optim/lbfgs.py, provoked here by loose tolerances; the evaluation limit
    max_eval is the host's) must do nothing on the device.  Same losses, same n_iter / func_evals / flags, same parameters as the
    eager loop (graph=False), for k = 1, 3, 4 and the default -- with steps that run all 20 iterations, steps that end on their
    first, and steps that end somewhere inside a batch."""
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import ShardedLBFGS
    from hidenn_fem_amd.sharded import ShardedTri3Energy
    d = torch.device("cuda:0")

    def run(opts, **kw):
        m = _e4_model(d, F64, nx=61, ny=41)
        sh = ShardedTri3Energy(m, EnergyLoss2D(E=10e9, nu=0.3, device=d, dtype=F64)).setup_interfaces()
        opt = ShardedLBFGS(sh, history_size=10, **opts, **kw)
        rets, counts = [], []
        for _ in range(4):
            rets.append(opt.step().item())
            counts.append((opt.state["n_iter"], opt.state["func_evals"], int(opt.status()[1]) & 15))
        opt.finish()
        return rets, counts, m

    def per_step(counts):
        return np.diff([0] + [c[0] for c in counts])

    scenarios = [dict(), dict(max_eval=7), dict(max_iter=6)]
    # a tolerance_change that ends some step strictly inside it (the loss flattens step by step: found by bisection on the eager loop)
    mid = None
    for tol in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7):
        it = per_step(run(dict(tolerance_change=tol), graph=False)[1])
        if ((it > 1) & (it < 20)).any():
            mid = tol
            break
    assert mid is not None, "no tolerance_change ends a step in the middle: the test would not exercise the device-side halt"
    scenarios += [dict(tolerance_change=mid), dict(tolerance_change=3e-3)]
    for opts in scenarios:
        ref_l, ref_c, ref_m = run(opts, graph=False)
        print(opts, "inner iterations per step:", per_step(ref_c).tolist(), "flags:", [c[2] for c in ref_c])
        for kw in (dict(graph_iterations=1), dict(graph_iterations=3), dict(graph_iterations=4), dict()):
            got_l, got_c, got_m = run(opts, **kw)
            assert got_c == ref_c, (opts, kw, got_c, ref_c)
            np.testing.assert_allclose(got_l, ref_l, rtol=1e-9, err_msg=str((opts, kw)))
            for a, b in zip(got_m.parameters(), ref_m.parameters()):
                assert (a.detach() - b.detach()).abs().max().item() <= 1e-8 * b.detach().abs().max().item(), (opts, kw)


def _worker_sharded_lbfgs(rank, world, port, q):
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = torch.device("cuda:0")                         # all ranks share the box's one GPU; gloo is the transport
        torch.cuda.set_device(d)
        out = {}
        for history in (100, 6):
            got, m, opt, sh = _sharded_run(d, F64, 2, history)
            xr, ur = sh.owned_rows()
            # a rank's copy of the parameters is complete on the rows it OWNS and the interface rows it reads (finish());
            # rows deep inside another rank's patch are never sent to it -- that is the point of the owner-sharded mode
            out[history] = dict(losses=got, n_iter=opt.state["n_iter"], evals=opt.state["func_evals"], status=opt.status(),
                                xr=xr.cpu().numpy(), ur=ur.cpu().numpy(),
                                x=m.node_coords_free.detach()[xr].cpu().numpy(), u=m.u_free.detach()[ur].cpu().numpy(),
                                seen_x=sh._need_dst[:sh._need_n[0]].long().cpu().numpy(),
                                x_seen=m.node_coords_free.detach()[sh._need_dst[:sh._need_n[0]].long()].cpu().numpy(),
                                n_local=opt._n, tiles=(sh.lo, sh.hi))
            dist.barrier()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_lbfgs_processes_follow_the_unsharded_trajectory(world):
    """2 and 3 rank processes (real exchanges: gloo all_gather of the interface rows and of the L-BFGS payload): every rank
    reports the SAME losses, iteration counts and status record -- decisions are taken on rank-ordered sums --, the trajectory
    is the unsharded FusedLBFGS one, and after finish() every rank holds the complete parameters."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sharded_lbfgs, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = torch.device("cuda:0")
    for history in (100, 6):
        ref, m_ref, o_ref = _fused_reference(d, F64, 2, history)
        st = o_ref.state[o_ref._params[0]]
        x_ref, u_ref = m_ref.node_coords_free.detach().cpu().numpy(), m_ref.u_free.detach().cpu().numpy()      # storage order: the same
        assert sum(res[r][history]["n_local"] for r in range(world)) == 2 * (x_ref.shape[0] + u_ref.shape[0])      # permutation in every process
        x_all, u_all = np.full_like(x_ref, np.nan), np.full_like(u_ref, np.nan)
        for r in range(world):                                 # the ranks' owned rows partition the parameters
            o = res[r][history]
            assert np.isnan(x_all[o["xr"]]).all() and np.isnan(u_all[o["ur"]]).all(), "a row is owned by two ranks"
            x_all[o["xr"]], u_all[o["ur"]] = o["x"], o["u"]
        assert not np.isnan(x_all).any() and not np.isnan(u_all).any()
        # 40 quasi-Newton iterations on an ill-conditioned r-adaptive energy amplify the summation-order difference between
        # the sharded and the unsharded dots ("why FusedLBFGS and torch.optim.LBFGS end at different losses")
        assert np.abs(x_all - x_ref).max() <= 2e-6 * np.abs(x_ref).max()
        assert np.abs(u_all - u_ref).max() <= 2e-6 * np.abs(u_ref).max()
        for r in range(world):
            o = res[r][history]
            assert o["losses"] == res[0][history]["losses"] and o["status"] == res[0][history]["status"], "ranks disagree"
            np.testing.assert_allclose(o["losses"], ref, rtol=1e-8)
            assert (o["n_iter"], o["evals"]) == (st["n_iter"], st["func_evals"])
            if len(o["seen_x"]):                                # finish(): the interface rows this rank reads are the owners' final values
                assert np.abs(o["x_seen"] - x_all[o["seen_x"]]).max() == 0.0


@pytest.mark.gpu
def test_example4_runs_owner_sharded_end_to_end():
    """examples/example4.py --sharded (one process here): the reference's Example 4 loop with the sharded energy and the
    node-sharded L-BFGS -- the same optimisation as the FusedLBFGS path of the example (an fp64 run: same closure count, losses
    equal to the tolerance of the trajectory tests)."""
    import examples.example4 as e4
    torch.manual_seed(0)                                         # the sharded run seeds 0 itself (every rank must draw the same u_free)
    _, fused = e4.run(nx=60, ny=30, steps=3, dtype=F64, log_every=1, fused_lbfgs=True)
    _, shard = e4.run(nx=60, ny=30, steps=3, dtype=F64, log_every=1, sharded=True)
    assert np.isfinite(shard) and abs(shard - fused) <= 1e-7 * abs(fused), (shard, fused)

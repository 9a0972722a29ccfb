"""GraphedTraining: whole iterations (zero_grad, fused loss, backward, capturable FusedAdam) captured in one hipGraph
must follow the eager loop -- examples 1 and 3 of the reference (1D L2 projection, 1D bar energy with r-adaptivity)."""
import numpy as np
import pytest
import torch

F64 = torch.float64


def _example1(d):
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN
    from hidenn_fem_amd.loss import l2_projection_loss
    nodes = torch.linspace(0, 1, 100, dtype=F64, device=d)
    xs = torch.linspace(0, 1, 1000, dtype=F64, device=d)
    target = torch.sin(2 * torch.pi * xs)
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN(nodes, r_adapt=True).to(d)
    return m, (lambda: l2_projection_loss(m, xs, target)), 5e-3


def _example3(d):
    import examples.example3 as e3
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN
    from hidenn_fem_amd.loss import bar_energy_loss
    from hidenn_fem_amd.utils import gauss_legendre_points_weights
    grid = torch.linspace(0, e3.LENGTH, 2001, dtype=F64, device=d)
    xi, wi = gauss_legendre_points_weights(2, device=d, dtype=F64)
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN(grid, r_adapt=True, u0=0.0, uN=0.0).to(d)
    return m, (lambda: bar_energy_loss(m, xi, wi, e3.body_force, E=e3.E_MOD)), 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["example1", "example3"])
def test_graphed_training_follows_the_eager_loop(case):
    from hidenn_fem_amd.optim import FusedAdam
    from hidenn_fem_amd.graphed import GraphedTraining
    d = torch.device("cuda:0")
    make = _example1 if case == "example1" else _example3
    n_replays, per = 6, 5
    def eager(opt_cls, **kw):                                  # 3 + 30 iterations, the plain loop
        m_e, closure_e, lr_e = make(d)
        opt = opt_cls(m_e.parameters(), lr=lr_e, **kw)
        out = []
        for _ in range(3 + n_replays * per):
            opt.zero_grad()
            loss = closure_e()
            loss.backward()
            opt.step()
            out.append(loss.item())
        return m_e, out

    m_ref, losses_ref = eager(FusedAdam, capturable=True)      # same optimiser, eager: isolates the capture
    _, losses_torch = eager(torch.optim.Adam)                  # torch's Adam: same trajectory up to Adam's own
    # graphed: FusedAdam with the step count on the device
    m, closure, lr = make(d)
    gt = GraphedTraining(closure, FusedAdam(m.parameters(), lr=lr, capturable=True), steps_per_replay=per, warmup=3)
    got = []
    for _ in range(n_replays):
        got.append(gt.replay().item())
    assert gt.steps_done == 3 + n_replays * per
    pick = [3 + (r + 1) * per - 1 for r in range(n_replays)]
    want = [losses_ref[i] for i in pick]
    np.testing.assert_allclose(got, want, rtol=1e-10)          # fp64 atomics order is the only freedom
    # vs torch's Adam: same arithmetic, but the r-adaptive 1D problems amplify last-bit differences of the fp32
    # parameter `u` step by step (example 3: 1e-9 after two steps, 1e-2 after a dozen)
    np.testing.assert_allclose(got[:2], [losses_torch[i] for i in pick[:2]], rtol=1e-5 if case == "example1" else 3e-2)
    for a, b in zip(m.parameters(), m_ref.parameters()):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-8, atol=1e-12)
    assert want[-1] < want[0] or case == "example3"


@pytest.mark.gpu
@pytest.mark.parametrize("warmup", [0, 1])
def test_graphed_training_without_warmup_keeps_the_optimiser_state(warmup):
    """FusedAdam's moments and step counter must not be (re)created inside the capture: with warmup=0 they used to be,
    so every replay restarted Adam.  Parameters after several replays == the eager loop, to rounding."""
    from hidenn_fem_amd.optim import FusedAdam
    from hidenn_fem_amd.graphed import GraphedTraining
    d = torch.device("cuda:0")
    per, n_replays = 4, 3
    m_e, closure_e, lr = _example1(d)
    opt_e = FusedAdam(m_e.parameters(), lr=lr, capturable=True)
    for _ in range(warmup + per * n_replays):
        opt_e.zero_grad()
        closure_e().backward()
        opt_e.step()
    m, closure, lr = _example1(d)
    opt = FusedAdam(m.parameters(), lr=lr, capturable=True)
    gt = GraphedTraining(closure, opt, steps_per_replay=per, warmup=warmup)
    gt.replay(n_replays)
    torch.cuda.synchronize()
    assert int(opt.state[next(iter(m.parameters()))]["step"].item()) == warmup + per * n_replays
    for a, b in zip(m.parameters(), m_e.parameters()):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-9, atol=1e-13)
    # the device step counter travels with state_dict(): a resumed optimiser continues, it does not restart
    opt2 = FusedAdam(m.parameters(), lr=lr, capturable=True)
    opt2.load_state_dict(opt.state_dict())
    assert int(opt2._step_dev.item()) == warmup + per * n_replays
    # a stock optimiser whose state does not exist yet must be refused with warmup=0
    m3, closure3, lr3 = _example1(d)
    with pytest.raises(ValueError):
        GraphedTraining(closure3, torch.optim.Adam(m3.parameters(), lr=lr3, capturable=True), warmup=0)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_direct_value_and_grad_matches_autograd_and_trains_in_a_graph(dtype):
    """EnergyLoss2D.value_and_grad_: one launch, gradients straight into .grad -- same numbers as loss.backward();
    with GraphedTraining(direct=True) + capturable FusedAdam it follows the eager autograd loop."""
    import copy
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import FusedAdam
    from hidenn_fem_amd.graphed import GraphedTraining
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(121, 81, jitter=0.2, seed=2, dtype=dtype)
    torch.manual_seed(1)
    base = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    lf = EnergyLoss2D(device=d, dtype=dtype)
    tol = 1e-12 if dtype == F64 else 2e-6
    a, b = copy.deepcopy(base), copy.deepcopy(base)
    la = lf(a)
    la.backward()
    with torch.no_grad():                                    # stale values in .grad must be overwritten, not accumulated
        for p in b.parameters():
            p.grad = torch.full_like(p, 7.0)
    lb = lf.value_and_grad_(b)
    assert abs(lb.item() - la.item()) <= tol * abs(la.item())
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert (pa.grad - pb.grad).abs().max().item() <= tol * pa.grad.abs().max().item()
    # training: eager autograd loop vs graphed direct loop, 3 + 20 iterations
    n_rep, per = 4, 5
    oa = FusedAdam(a.parameters(), lr=1e-7, capturable=True)
    for _ in range(3 + n_rep * per):
        oa.zero_grad()
        lf(a).backward()
        oa.step()
    b = copy.deepcopy(base)
    gt = GraphedTraining(lambda: lf.value_and_grad_(b), FusedAdam(b.parameters(), lr=1e-7, capturable=True),
                         steps_per_replay=per, warmup=3, direct=True)
    gt.replay(n_rep)
    torch.cuda.synchronize()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert (pa - pb).abs().max().item() <= (1e-10 if dtype == F64 else 1e-5) * pa.abs().max().item()
    assert (a.u_free - base.u_free).abs().max().item() > 0


@pytest.mark.gpu
def test_fused_energy_adam_step_matches_the_two_launch_loop():
    """EnergyAdamStep: the tile that owns a row applies Adam's update at write-out (one launch per iteration, ping-pong
    parameter buffers).  Must follow value_and_grad_ + FusedAdam with the same per-tensor learning rates, eagerly and
    inside a hipGraph."""
    import copy
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import FusedAdam, EnergyAdamStep
    from hidenn_fem_amd.graphed import GraphedTraining
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(161, 101, jitter=0.2, seed=3, dtype=F64)
    torch.manual_seed(2)
    base = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    lf = EnergyLoss2D(device=d, dtype=F64)
    lr_x, lr_u, n_steps = 1e-8, 1e-7, 12
    a = copy.deepcopy(base)
    opt = FusedAdam([dict(params=[a.node_coords_free], lr=lr_x), dict(params=[a.u_free], lr=lr_u)], capturable=True)
    ref_losses = []
    for _ in range(n_steps):
        ref_losses.append(lf.value_and_grad_(a).item())
        opt.step()
    b = copy.deepcopy(base)
    tr = EnergyAdamStep(b, lf, lr_x=lr_x, lr_u=lr_u)
    got = [tr.step().item() for _ in range(n_steps)]
    np.testing.assert_allclose(got, ref_losses, rtol=1e-11)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert (pa - pb).abs().max().item() <= 1e-11 * pa.abs().max().item()
    assert (b.u_free - base.u_free).abs().max().item() > 0 and int(tr.state["step"].item()) == n_steps
    # graphed: 2 warm-up + 5 replays of 2 iterations == 12 iterations
    c = copy.deepcopy(base)
    tr2 = EnergyAdamStep(c, lf, lr_x=lr_x, lr_u=lr_u)
    gt = GraphedTraining(tr2.step, None, steps_per_replay=2, warmup=2, direct=True)
    last = gt.replay(5)
    torch.cuda.synchronize()
    assert abs(last.item() - ref_losses[-1]) <= 1e-11 * abs(ref_losses[-1])
    for pa, pc in zip(a.parameters(), c.parameters()):
        assert (pa - pc).abs().max().item() <= 1e-11 * pa.abs().max().item()
    # lagged loss sum inside the graph: same trajectory, last loss delivered by the trailing flush
    e = copy.deepcopy(base)
    tr3 = EnergyAdamStep(e, lf, lr_x=lr_x, lr_u=lr_u)
    gt3 = GraphedTraining(tr3.step_lagged, None, steps_per_replay=4, warmup=2, direct=True, begin=tr3.begin_lagged,
                          end=tr3.flush_loss)
    gt3.replay(2)
    torch.cuda.synchronize()
    assert abs(tr3.loss.item() - ref_losses[9]) <= 1e-11 * abs(ref_losses[9])      # 2 + 8 iterations: loss of the 10th
    tr3.step(); tr3.step()
    for pa, pe in zip(a.parameters(), e.parameters()):
        assert (pa - pe).abs().max().item() <= 1e-11 * pa.abs().max().item()


@pytest.mark.gpu
def test_multi_tensor_fused_adam_one_launch_matches_torch_and_survives_a_checkpoint_restore():
    """FusedAdam updates ALL parameter tensors in one launch (hfem_adam_multi_dev: device pointer table, the step counter
    bumped by the last block to finish).  Mixed dtypes, odd lengths, per-group learning rates, eager (host count and
    device count) and captured -- against torch.optim.Adam.  load_state_dict keeps the device counter's address, so a
    graph captured before a restore goes on with the restored count."""
    from hidenn_fem_amd.optim import FusedAdam
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    shapes = [((70001, 2), F64), ((333,), torch.float32), ((4097, 2), F64), ((5,), F64)]
    p0 = [torch.randn(s, generator=g, dtype=dt) for s, dt in shapes]
    n_steps = 12
    grads = [[torch.randn(t.shape, generator=g, dtype=t.dtype) * 10 ** float(k % 5 - 2) for t in p0] for k in range(n_steps)]
    lrs = [3e-3, 1e-2, 5e-4, 1e-3]

    def params():
        return [t.clone().to(d).requires_grad_(True) for t in p0]

    def groups(ps):
        return [dict(params=[p], lr=lr) for p, lr in zip(ps, lrs)]
    pa = params()
    oa = torch.optim.Adam(groups(pa))
    for gs in grads:
        for a, gi in zip(pa, gs):
            a.grad = gi.to(d).clone()
        oa.step()

    def close(ps, what):
        for a, b in zip(pa, ps):
            tol = 1e-13 if a.dtype == F64 else 3e-6
            err = (a - b).abs().max().item() / a.abs().max().item()
            assert err <= tol, (what, a.dtype, err)
    for capturable in (False, True):
        pb = params()
        ob = FusedAdam(groups(pb), capturable=capturable)
        for gs in grads:
            for b, gi in zip(pb, gs):
                b.grad = gi.to(d).clone()
            ob.step()
        close(pb, f"eager capturable={capturable}")
        assert int(ob.state[pb[0]]["step"]) == n_steps
    # captured: static gradient buffers refilled between replays; checkpoint restore in the middle
    pc = params()
    for c in pc:
        c.grad = torch.zeros_like(c)
    oc = FusedAdam(groups(pc), capturable=True).init_state()

    def load(k):
        for c, gi in zip(pc, grads[k]):
            c.grad.copy_(gi.to(d))
    load(0)
    oc.step()                                                # eager warm-up step = step 1
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        oc.step()
    counter_addr = oc._step_dev.data_ptr()
    for k in range(1, 6):
        load(k)
        graph.replay()
    torch.cuda.synchronize()
    assert int(oc._step_dev.item()) == 6
    saved = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in ((i, t.detach().clone()) for i, t in enumerate(pc))}
    ck = oc.state_dict()
    ck = dict(state={k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in ck["state"].items()},
              param_groups=ck["param_groups"])
    for k in range(6, 9):                                    # go on, then rewind to the checkpoint
        load(k)
        graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        for i, c in enumerate(pc):
            c.copy_(saved[i])
    moments_before = [oc.state[c]["exp_avg"].data_ptr() for c in pc]
    oc.load_state_dict(ck)
    assert oc._step_dev.data_ptr() == counter_addr and int(oc._step_dev.item()) == 6
    # the moments are new tensors after a load (as in torch): re-capture, the counter is still the same tensor
    assert [oc.state[c]["exp_avg"].data_ptr() for c in pc] != moments_before
    graph2 = torch.cuda.CUDAGraph()
    load(6)
    oc.step()                                                # eager step 7 (rebuilds the pointer table outside the capture)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph2):
        oc.step()
    for k in range(7, n_steps):
        load(k)
        graph2.replay()
    torch.cuda.synchronize()
    assert int(oc._step_dev.item()) == n_steps
    close(pc, "captured + restored")


@pytest.mark.gpu
@pytest.mark.parametrize("mesh_kw", [dict(), dict(diagonal="zigzag")], ids=["paired", "one_per_slot"])
def test_fused_energy_adam_step_fp32_rows_and_body_force(mesh_kw):
    """EnergyAdamStep instances beyond fp64 / zero body force: an fp32 model -- the reference's default
    dtype, /root/reference/src/loss.py:16, src/models.py:274 -- keeps fp32 parameters and moments and trains in one launch
    per iteration; a body force enters through the B_k table.  Each against the two-launch loop on the same kernels'
    gradient path (value_and_grad_ / autograd + FusedAdam), on a paired plan and on a one-element-per-slot plan."""
    import copy
    from conftest import b_force_fn
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import FusedAdam, EnergyAdamStep
    d = torch.device("cuda:0")
    lr_x, lr_u, n_steps = 1e-8, 1e-7, 8
    # ---- fp64 + body force
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(121, 81, jitter=0.2, seed=3, dtype=F64, **mesh_kw)
    torch.manual_seed(2)
    base = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    lf = EnergyLoss2D(device=d, dtype=F64)
    assert base.tile_plan(0).is_paired() == (not mesh_kw)
    a = copy.deepcopy(base)
    opt = FusedAdam([dict(params=[a.node_coords_free], lr=lr_x), dict(params=[a.u_free], lr=lr_u)])
    ref = []
    for _ in range(n_steps):
        opt.zero_grad()
        loss = lf(a, b_force=b_force_fn)
        loss.backward()
        ref.append(loss.item())
        opt.step()
    b = copy.deepcopy(base)
    tr = EnergyAdamStep(b, lf, lr_x=lr_x, lr_u=lr_u, b_force=b_force_fn)
    got = [tr.step().item() for _ in range(n_steps)]
    np.testing.assert_allclose(got, ref, rtol=1e-11)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert (pa - pb).abs().max().item() <= 1e-11 * pa.abs().max().item()
    l0 = lf(base).item()
    assert abs(got[0] - l0) > 1e-6 * abs(l0), "the body force changes the energy"
    # ---- fp32 rows (zero body force, then with one)
    coords32 = coords.float()
    torch.manual_seed(2)
    base32 = PiecewiseLinearShapeNN2D(coords32, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    assert base32.u_free.dtype == torch.float32
    lf32 = EnergyLoss2D(device=d, dtype=torch.float32, arithmetic="fp64")      # the fused step computes in fp64 on float rows
    for bf in (None, b_force_fn):
        a32 = copy.deepcopy(base32)
        o32 = FusedAdam([dict(params=[a32.node_coords_free], lr=lr_x), dict(params=[a32.u_free], lr=lr_u)])
        ref32 = []
        for _ in range(n_steps):
            o32.zero_grad()
            loss = lf32(a32, b_force=bf)
            loss.backward()
            ref32.append(loss.double().item())
            o32.step()
        b32 = copy.deepcopy(base32)
        tr32 = EnergyAdamStep(b32, lf32, lr_x=lr_x, lr_u=lr_u, b_force=bf)
        got32 = [tr32.step().item() for _ in range(n_steps)]
        assert b32.u_free.dtype == torch.float32 and tr32.state["exp_avg_u"].dtype == torch.float32
        np.testing.assert_allclose(got32, ref32, rtol=2e-6)              # the reference loop rounds its loss to fp32
        for pa, pb in zip(a32.parameters(), b32.parameters()):
            # same fp32 update on the same once-rounded gradient; a last-bit flip of an fp64 LDS sum can move a rounding
            assert (pa - pb).abs().max().item() <= 4 * torch.finfo(torch.float32).eps * pa.abs().max().item()
            assert (pa == pb).float().mean().item() > 0.98
        assert (b32.u_free - base32.u_free).abs().max().item() > 0
    # ---- fp32 rows AND fp32 arithmetic (round 4: HFEM_FLAG_FP32_MATH in the fused step, csrc/tri3_pair_f32.hip ADAM instances):
    #      what EnergyLoss2D(arithmetic="auto") selects for an fp32 model on a paired plan -- against the two-launch loop on the
    #      same arithmetic (the fp32-arithmetic energy kernel + FusedAdam), and close to the fp64-arithmetic trajectory above
    lf32a = EnergyLoss2D(device=d, dtype=torch.float32)
    for bf in (None, b_force_fn):
        b32 = copy.deepcopy(base32)
        tr32 = EnergyAdamStep(b32, lf32a, lr_x=lr_x, lr_u=lr_u, b_force=bf)
        assert bool(tr32._flags & 1024) == (not mesh_kw), "fp32 arithmetic exactly when the plan has paired slots"
        a32 = copy.deepcopy(base32)
        o32 = FusedAdam([dict(params=[a32.node_coords_free], lr=lr_x), dict(params=[a32.u_free], lr=lr_u)])
        ref32 = []
        for _ in range(n_steps):
            o32.zero_grad()
            loss = lf32a(a32, b_force=bf)
            loss.backward()
            ref32.append(loss.double().item())
            o32.step()
        got32 = [tr32.step().item() for _ in range(n_steps)]
        np.testing.assert_allclose(got32, ref32, rtol=2e-6)
        for pa, pb in zip(a32.parameters(), b32.parameters()):
            assert (pa - pb).abs().max().item() <= 4 * torch.finfo(torch.float32).eps * pa.abs().max().item()
            assert (pa == pb).float().mean().item() > 0.98
        # captured: K iterations per hipGraph continue the same trajectory
        from hidenn_fem_amd.graphed import GraphedTraining
        gt = GraphedTraining(tr32.step, None, steps_per_replay=4, direct=True, warmup=2)      # 2 eager iterations, then 4 per replay
        gt.replay()
        torch.cuda.synchronize()
        for _ in range(6):
            o32.zero_grad()
            loss = lf32a(a32, b_force=bf)
            loss.backward()
            o32.step()
        for pa, pb in zip(a32.parameters(), b32.parameters()):
            assert (pa - pb).abs().max().item() <= 8 * torch.finfo(torch.float32).eps * pa.abs().max().item()


@pytest.mark.gpu
def test_fused_adam_pointer_tables_with_an_lr_scheduler_and_a_captured_graph():
    """(i) a captured hipGraph replays the host-to-device copy of ITS pointer-table slot, so that slot must never be
    rewritten -- after more than four other (parameter, gradient, lr) sets have been seen the graph still steps the right
    tensors; (ii) an LR scheduler changes the key every step: the steps stay equal to torch.optim.Adam's with the same
    schedule (and no longer synchronise the stream every four steps)."""
    from hidenn_fem_amd.optim import FusedAdam
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    w0 = [torch.randn(1000, 2, dtype=torch.float64, device=d), torch.randn(777, dtype=torch.float64, device=d)]
    tgt = [torch.randn_like(w) for w in w0]

    def make(cls, **kw):
        ps = [torch.nn.Parameter(w.clone()) for w in w0]
        return ps, cls(ps, lr=1e-2, **kw)

    def grads(ps):
        for p, t in zip(ps, tgt):
            p.grad = (p.detach() - t) * 2.0 if p.grad is None else p.grad.copy_((p.detach() - t) * 2.0)

    # (ii) scheduler
    pa, oa = make(torch.optim.Adam)
    pb, ob = make(FusedAdam)
    sa = torch.optim.lr_scheduler.ExponentialLR(oa, gamma=0.9)
    sb = torch.optim.lr_scheduler.ExponentialLR(ob, gamma=0.9)
    for _ in range(12):
        grads(pa); grads(pb)
        oa.step(); ob.step()
        sa.step(); sb.step()
    for a, b in zip(pa, pb):
        assert (a - b).abs().max().item() <= 1e-13 * a.abs().max().item()
    assert len(ob._tabs["slots"]) == FusedAdam._TAB_SLOTS          # the ring was walked, not grown
    # (i) a graph captured on one set of tensors, then many other keys through the same optimiser, then the graph again
    pc, oc = make(FusedAdam, capturable=True)
    oc.init_state()
    grads(pc)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        oc.step()
    pr, orf = make(torch.optim.Adam)
    for k in range(3):
        grads(pr); orf.step()
        grads(pc); g.replay()
    held = [sl for sl in oc._tabs["slots"] if sl["captured"]]
    assert len(held) == 1
    for lr in (3e-3, 4e-3, 5e-3, 6e-3, 7e-3, 8e-3):              # six more keys: eager steps with changing lr (and undone below)
        for grp in oc.param_groups:
            grp["lr"] = lr
        keep = [p.detach().clone() for p in pc]
        st = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in oc.state[p].items() if k != "step"} for p in pc}
        step_before = oc._step_dev.clone()
        grads(pc); oc.step()
        with torch.no_grad():                                     # undo: parameters, moments, the device step counter
            for p, kp in zip(pc, keep):
                p.copy_(kp)
                for k, v in st[id(p)].items():
                    oc.state[p][k].copy_(v)
            oc._step_dev.copy_(step_before)
    assert [sl for sl in oc._tabs["slots"] if sl["captured"]] == held and held[0]["key"] is not None
    for grp in oc.param_groups:
        grp["lr"] = 1e-2
    for k in range(3):
        grads(pr); orf.step()
        grads(pc); g.replay()
    torch.cuda.synchronize()
    for a, b in zip(pr, pc):
        assert (a - b).abs().max().item() <= 1e-13 * a.abs().max().item(), "the captured graph stepped with a recycled pointer table"

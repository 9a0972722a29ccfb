"""Peer-window interface exchange (csrc/peer.hip, sharded.PeerWindows): the owner-sharded steps with the interface rows
stored by the pack launch into every rank's receive window instead of an all_gather; the ``*_overlapped`` steps run the get
INSIDE their one energy launch (HFEM_FLAG_PEER_GET: service workgroups + boundary tiles that wait in the kernel).

* one rank (the window is the rank's own): every ``owner_*`` step, eager and captured, against the same step over the
  in-library RCCL communicator -- the same parameters and energies;
* TWO and THREE PROCESSES sharing the one GPU of the box (IPC-mapped windows, gloo only for the handle exchange and as the
  reference transport): world-2 and world-3 trajectories over peer windows == over all_gather, plain / overlapped / fused, eager and
  captured; and the bounded wait -- a rank whose peer never puts gets the sticky status bit, not a hang.
  (Stores into a window on ANOTHER GPU go over xGMI; that leg needs the driver's multi-GPU node.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F64 = torch.float64
LR_X, LR_U = 1e-6, 1e-8


def _model(d, nx=201, ny=151):
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(nx, ny, jitter=0.2, seed=8, dtype=F64)
    torch.manual_seed(4)
    return PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                    neumann_edges=edges).to(d)


def _model_delaunay(d, n_nodes=40000):
    from hidenn_fem_amd.mesh import unstructured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    coords, conn, geom, bc, mn, edges = unstructured_tri_mesh(n_nodes, seed=3, dtype=F64)
    torch.manual_seed(4)
    return PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                    neumann_edges=edges).to(d)


def _trainer(d, peer, comm=None, group=None, fused=False, split=None, timeout_s=5.0, inkernel=None, delaunay=False, f32=False):
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.sharded import ShardedTri3Energy
    m = _model_delaunay(d) if delaunay else _model(d)
    if f32:
        m = m.float()
    sh = ShardedTri3Energy(m, EnergyLoss2D(device=d, dtype=torch.float32 if f32 else F64), comm=comm, group=group)
    sh.setup_interfaces()
    sh.init_owner_adam(LR_X, LR_U, fused=fused)
    if split is not None:
        sh.mid = split(sh)
    if peer:
        sh.enable_peer_exchange(timeout_s=timeout_s, inkernel_get=inkernel)
        assert sh.inkernel_get == (True if inkernel is None else inkernel)      # these plans' kernels have the in-launch get
    return sh


def _close(a, b):
    """Same trajectory: the energy kernel's LDS atomics are order-dependent, so two runs agree to rounding, not to the bit
    (the tolerance of tests/test_gpu_sharded.py).  NaN entries of ``a`` = energies that step form never delivers."""
    a, b = torch.as_tensor(a, dtype=F64), torch.as_tensor(b, dtype=F64)
    if a.shape != b.shape:
        return False
    seen = ~torch.isnan(a)
    return bool(seen.any()) and bool((a[seen] - b[seen]).abs().max().item() <= 1e-12 * b[seen].abs().max().item())


def _lagged(sh, name):
    """The one-launch step (the put inside the fused energy + Adam launch) publishes the energy of the step BEFORE its own."""
    return name == "owner_train_step_fused_overlapped" and getattr(sh, "inkernel_put", False) and sh.mid > sh.lo


def _run(sh, name, n, overlapped):
    """Energies of the n steps as the caller sees them: serial steps deliver step k's after step k, overlapped steps one step
    later, the one-launch step two steps later (finish_overlapped() delivers the last one; the one before it is never seen)."""
    step = getattr(sh, name)
    lag = 2 if _lagged(sh, name) else (1 if overlapped else 0)
    out = [float("nan")] * n
    for k in range(n):
        step()
        if k - lag >= 0:
            out[k - lag] = sh.loss_global.item()
    if overlapped:
        out[n - 1] = sh.finish_overlapped().item()
    torch.cuda.synchronize()
    return out


def _puts(sh, name, steps, finishes):
    """Completed puts after `steps` steps and `finishes` finish_overlapped() calls (the one-launch step's finish flushes the
    last energy with one more put)."""
    return steps + (finishes if _lagged(sh, name) else 0)


STEPS = (("owner_train_step", False, False), ("owner_train_step_overlapped", False, True),
         ("owner_train_step_fused", True, False), ("owner_train_step_fused_overlapped", True, True))


@pytest.mark.gpu
def test_peer_windows_one_rank_every_step_matches_the_collective_path():
    from hidenn_fem_amd.sharded import LibraryComm
    d = torch.device("cuda:0")
    comm = LibraryComm(d)
    third = lambda sh: sh.plan.n_tiles // 3
    n = 8
    for name, fused, over in STEPS:
        ref = _trainer(d, False, comm=comm, fused=fused, split=third if over else None)
        l_ref = _run(ref, name, n, over)
        got = _trainer(d, True, fused=fused, split=third if over else None)
        assert got.peer is not None and got.comm is None
        l_got = _run(got, name, n, over)
        assert _close(l_got, l_ref), (name, l_got, l_ref)
        for a, b in zip(got.model.parameters(), ref.model.parameters()):
            assert _close(a.detach(), b.detach()), name
        assert got.peer.status() == (0, _puts(got, name, n, 1))
        assert int(got._adam["step"].item()) == n
        if name == "owner_train_step_fused_overlapped":
            assert got.inkernel_put and _lagged(got, name), "paired plan + fused state: the put runs inside the energy launch"
        # captured: 4 steps per hipGraph, two replays, on top of the 8 eager ones -> 16 steps; the reference goes on eagerly
        g = torch.cuda.CUDAGraph()
        step = getattr(got, name)
        with torch.cuda.graph(g):
            for _ in range(4):
                step()
            if over:
                got.finish_overlapped()
        g.replay(); g.replay()
        torch.cuda.synchronize()
        # capture itself executes nothing: 8 eager + 2 x 4 replayed
        l_more = _run(ref, name, 8, over)
        assert got.peer.status() == (0, _puts(got, name, 16, 3 if over else 0))
        assert int(got._adam["step"].item()) == 16
        assert _close(got.loss_global.item(), l_more[-1]), name
        for a, b in zip(got.model.parameters(), ref.model.parameters()):
            assert _close(a.detach(), b.detach()), name + " (captured)"
        got.close_peer_exchange()
    # the overlapped steps with the get as a launch of its own (what plans without paired slots use)
    for name, fused, over in STEPS:
        if not over:
            continue
        ref = _trainer(d, False, comm=comm, fused=fused, split=third)
        got = _trainer(d, True, fused=fused, split=third, inkernel=False)
        assert not got.inkernel_put
        assert _close(_run(got, name, n, True), _run(ref, name, n, True)), name
        for a, b in zip(got.model.parameters(), ref.model.parameters()):
            assert _close(a.detach(), b.detach()), name
        assert got.peer.status() == (0, n)
        got.close_peer_exchange()
    # a Delaunay mesh: its plan has one element per slot (no paired records) -- that kernel's in-launch get
    for name, fused, over in STEPS:
        if not over:
            continue
        ref = _trainer(d, False, comm=comm, fused=fused, split=third, delaunay=True)
        got = _trainer(d, True, fused=fused, split=third, delaunay=True)
        assert not got.plan.is_paired() and got.inkernel_get and not got.inkernel_put
        assert _close(_run(got, name, n, True), _run(ref, name, n, True)), name
        for a, b in zip(got.model.parameters(), ref.model.parameters()):
            assert _close(a.detach(), b.detach()), name
        assert got.peer.status() == (0, n)
        got.close_peer_exchange()
    # evaluation-only steps
    ref = _trainer(d, False, comm=comm, split=third)
    got = _trainer(d, True, split=third)
    l0, gx0, gu0 = ref.owner_step()
    l1, gx1, gu1 = got.owner_step()
    assert l0.item() == l1.item() and _close(gx1, gx0) and _close(gu1, gu0)
    for _ in range(3):
        got.owner_step_overlapped()
    assert got.finish_overlapped().item() == l0.item()          # the energy itself is summed in a fixed order
    assert got.peer.status() == (0, 4) and int(got._adam["step"].item()) == 0
    # life cycle: back to the collective path (the plan lets go of the windows), on again with fresh windows
    got.close_peer_exchange()
    assert got.peer is None and not got.inkernel_get
    l2 = got.owner_step()[0].item()
    assert l2 == l0.item()
    got.enable_peer_exchange()
    assert got.peer.status() == (0, 0)
    for _ in range(2):
        got.owner_step_overlapped()
    assert got.finish_overlapped().item() == l0.item() and got.peer.status() == (0, 2)
    # recovery path: new windows in place of the old ones (what a caller does after a timed-out get), counters start again
    got.reset_peer_exchange()
    assert got.peer is not None and got.peer.status() == (0, 0) and got.inkernel_get
    got.owner_step_overlapped()
    assert got.finish_overlapped().item() == l0.item() and got.peer.status() == (0, 1)
    got.check_exchange()
    got.close_peer_exchange()
    comm.close()


def _close32(a, b, tol=2e-5):
    a, b = torch.as_tensor(a, dtype=F64), torch.as_tensor(b, dtype=F64)
    if a.shape != b.shape:
        return False
    seen = ~torch.isnan(a)                                  # NaN: an energy that step form never delivers (_run)
    return bool(seen.any()) and bool((a[seen] - b[seen]).abs().max().item() <= tol * b[seen].abs().max().item())


@pytest.mark.gpu
def test_fp32_model_owner_sharded_steps_one_rank():
    """An fp32 model (the reference's default dtype) in the owner-sharded steps: float parameter rows, double2 payload
    (pack_f32 / unpack_f32, put_f32 / get_f32), Adam inside the energy launch on float rows.  Collective path and peer windows,
    plain and overlapped, against the unsharded one-launch EnergyAdamStep of the same model; and the evaluation-only step."""
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import EnergyAdamStep
    from hidenn_fem_amd.sharded import LibraryComm
    d = torch.device("cuda:0")
    comm = LibraryComm(d)
    third = lambda sh: sh.plan.n_tiles // 3
    n = 8
    m0 = _model(d).float()
    # the sharded steps compute in fp64 on float rows: so does the unsharded step they are held against (arithmetic="fp64")
    tr = EnergyAdamStep(m0, EnergyLoss2D(device=d, dtype=torch.float32, arithmetic="fp64"), lr_x=LR_X, lr_u=LR_U)
    l0 = [tr.step().item() for _ in range(n)]
    for name, over in (("owner_train_step_fused", False), ("owner_train_step_fused_overlapped", True)):
        for peer in (False, True):
            sh = _trainer(d, peer, comm=None if peer else comm, fused=True, split=third if over else None, f32=True)
            assert sh.model.node_coords_free.dtype == torch.float32 and sh.send.dtype == torch.float32
            got = _run(sh, name, n, over)
            assert _close32(got, l0, 1e-6), (name, peer, got, l0)       # energies: fp64 sums of the same float rows
            for a, b in zip(sh.model.parameters(), m0.parameters()):
                assert _close32(a.detach(), b.detach()), (name, peer)
            if peer:
                assert sh.peer.status() == (0, _puts(sh, name, n, 1)) and sh.inkernel_get
                sh.close_peer_exchange()
    # evaluation + exchange, and what fp32 models cannot do in this mode
    sh = _trainer(d, True, fused=True, split=third, f32=True)
    ref = _trainer(d, False, comm=comm, fused=True, split=third, f32=True)
    la, lb = sh.owner_step()[0].item(), ref.owner_step()[0].item()
    assert la == lb
    for _ in range(2):
        sh.owner_step_overlapped()
    assert sh.finish_overlapped().item() == la
    with pytest.raises(RuntimeError):
        sh.evaluate_local()                        # dense mode: fp64 only
    sh.close_peer_exchange()
    # evaluation + exchange without any optimiser state (the pieces one by one, as a caller with its own optimiser uses them)
    from hidenn_fem_amd.sharded import ShardedTri3Energy
    bare = ShardedTri3Energy(_model(d).float(), EnergyLoss2D(device=d, dtype=torch.float32)).setup_interfaces()
    bare.evaluate_owner()
    lc = bare.exchange_halo()[0].item()
    assert lc == la
    # the unfused steps (energy -> Adam on the owned float rows -> exchange) against value_and_grad_ + FusedAdam
    from hidenn_fem_amd.optim import FusedAdam
    m1 = _model(d).float()
    lf = EnergyLoss2D(device=d, dtype=torch.float32, arithmetic="fp64")      # the sharded steps compute in fp64 on float rows
    opt = FusedAdam([dict(params=[m1.node_coords_free], lr=LR_X), dict(params=[m1.u_free], lr=LR_U)])
    l1 = []
    for _ in range(n):
        l1.append(lf.value_and_grad_(m1).item())
        opt.step()
    for name, over in (("owner_train_step", False), ("owner_train_step_overlapped", True)):
        sh = _trainer(d, True, split=third if over else None, f32=True)
        got = _run(sh, name, n, over)
        assert _close32(got, l1, 1e-6), (name, got, l1)
        for a, b in zip(sh.model.parameters(), m1.parameters()):
            assert _close32(a.detach(), b.detach()), name
        assert sh.peer.status() == (0, n)
        sh.close_peer_exchange()
    comm.close()


def _worker_two_ranks(rank, world, port, q, per_gpu=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = torch.device("cuda", rank if per_gpu else 0)   # all ranks on the box's one GPU, or one GPU per rank (xGMI leg)
        torch.cuda.set_device(d)
        n = 6
        report = {}
        for name, fused, over in STEPS:
            ref = _trainer(d, False, fused=fused)          # torch.distributed (gloo) all_gather as the reference transport
            assert ref.world == world and ref.lo < ref.mid < ref.hi, "boundary and interior tiles exist"
            got = _trainer(d, True, fused=fused)
            ref_name = name.replace("_overlapped", "")     # same launches in another order: bit-equal (test_gpu_sharded.py)
            l_ref = _run(ref, ref_name, n, False)
            l_got = _run(got, name, n, over)
            own_x, own_u = ref.owned_rows()
            seen_x = torch.unique(torch.cat([own_x, ref._need_dst[:ref._need_n[0]].long()]))
            seen_u = torch.unique(torch.cat([own_u, ref._need_dst[ref._need_n[0]:].long()]))
            same = (_close(l_got, l_ref) and _close(got.model.node_coords_free[seen_x].detach(), ref.model.node_coords_free[seen_x].detach())
                    and _close(got.model.u_free[seen_u].detach(), ref.model.u_free[seen_u].detach()))
            same = same and got.verify_interfaces() == 0.0      # every interface row this rank reads IS the owner's current row
            st = (got.peer.status(), _puts(got, name, n, 1))
            # captured: an even number of steps per graph (the fused steps alternate between two parameter buffers)
            dist.barrier()
            g = torch.cuda.CUDAGraph()
            step = getattr(got, name)
            with torch.cuda.graph(g):
                for _ in range(4):
                    step()
                if over:
                    got.finish_overlapped()
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            l_more = _run(ref, ref_name, 4, False)
            same_g = (_close(got.loss_global.item(), l_more[-1])
                      and _close(got.model.node_coords_free[seen_x].detach(), ref.model.node_coords_free[seen_x].detach())
                      and _close(got.model.u_free[seen_u].detach(), ref.model.u_free[seen_u].detach()))
            report[name] = (bool(same), bool(same_g), st, (got.peer.status(), _puts(got, name, n + 4, 2 if over else 0)), l_got)
            dist.barrier()
            got.close_peer_exchange()
        # an fp32 model through the fused steps: peer windows against the gloo all_gather
        ok32 = True
        for name, over in (("owner_train_step_fused", False), ("owner_train_step_fused_overlapped", True)):
            ref = _trainer(d, False, fused=True, f32=True)
            got = _trainer(d, True, fused=True, f32=True)
            l_ref = _run(ref, "owner_train_step_fused", n, False)
            l_got = _run(got, name, n, over)
            own_x, own_u = ref.owned_rows()
            ok32 = ok32 and _close32(l_got, l_ref, 1e-6) and got.peer.status() == (0, _puts(got, name, n, 1)) \
                and _close32(got.model.node_coords_free[own_x].detach(), ref.model.node_coords_free[own_x].detach()) \
                and _close32(got.model.u_free[own_u].detach(), ref.model.u_free[own_u].detach())
            dist.barrier()
            got.close_peer_exchange()
        report["f32"] = bool(ok32)
        # the bounded wait: only rank 0 puts; its get gives up after 0.3 s with the sticky status bit
        lone = _trainer(d, True, timeout_s=0.3)
        if rank == 0:
            lone.owner_step()
            torch.cuda.synchronize()
            st = lone.peer.status()
            try:
                lone.peer.check()
                raised = False
            except RuntimeError:
                raised = True
            report["timeout"] = (st, raised)
        dist.barrier()
        if rank == 0:                                      # a set status bit surfaces at close (and at finish_overlapped) too
            try:
                lone.close_peer_exchange()
                report["timeout_at_close"] = False
            except RuntimeError:
                report["timeout_at_close"] = lone.peer is None      # ... after the windows were released
        else:
            lone.close_peer_exchange()
        q.put((rank, report))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_peer_windows_two_gpus():
    """The xGMI leg proper: two ranks, one GPU each -- every put is a store into a window on the OTHER GPU, every flag crosses
    the link.  Same checks as the one-GPU processes test.  Skips on boxes with fewer than two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (stores into a window on another GPU)")
    _peer_processes(2, per_gpu=True)


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_peer_windows_processes_sharing_one_gpu(world):
    _peer_processes(world, per_gpu=False)


def _peer_processes(world, per_gpu):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_two_ranks, args=(r, world, port, q, per_gpu)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for name, _, _ in STEPS:
        for r in range(world):
            same, same_g, st, st_g, losses = res[r][name]
            assert same, f"rank {r} {name}: peer-window trajectory differs from the all_gather one"
            assert same_g, f"rank {r} {name}: captured peer-window steps differ"
            assert st[0] == (0, st[1]) and st_g[0] == (0, st_g[1]), (r, name, st, st_g)
        assert all(np.array_equal(np.array(res[r][name][4]), np.array(res[0][name][4]), equal_nan=True) for r in range(world)), \
            "ranks disagree on the global energies (summed in rank order everywhere)"
    assert all(res[r]["f32"] for r in range(world)), "fp32 model: peer-window fused steps differ from the all_gather ones"
    assert res[0]["timeout"] == ((1, 1), True) and res[0]["timeout_at_close"] is True

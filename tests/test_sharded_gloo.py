"""N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.

Exercises the host logic of hidenn_fem_amd.sharded.ShardedTri3Energy -- tile-range split,
packed [gX|gU|loss] buffer, the single all-reduce, autograd plumbing -- with the oracle's closed
forms standing in for the HIP kernel (injected through the `evaluate` test seam; the product
default is the HIP kernel and has no CPU path)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK, HOME = 1023, 1 << 30


def oracle_tile_evaluator(plan, model):
    """(lo, hi, loss_v, gx_v, gu_v) -> fills the views like hfem_tri3_energy_plan does for tiles [lo, hi)."""
    from oracle import closed_form as CF
    td, ns = plan.export("tile_desc"), plan.export("node_src")
    gp = plan.export("edge_pack")
    mat, W = CF.plane_stress(), 0.25
    Tc = np.array([2e5, 0.0, 0.0, 0.0])

    def evaluate(lo, hi, loss_v, gx_v, gu_v):
        xf, uf = model.node_coords_free.detach().numpy(), model.u_free.detach().numpy()
        xfix, ufix = model.node_coords_fixed.numpy(), model.u_fixed_rows().numpy()
        total = 0.0
        for t, (eo, nel, no, nno, nown, go, ned, _) in enumerate(td[lo:hi], start=lo):
            src = ns[no:no + nno]
            X = np.where((src[:, 0] >= 0)[:, None], xf[np.maximum(src[:, 0], 0)], xfix[np.maximum(~src[:, 0], 0)] if len(xfix) else 0.0)
            U = np.where((src[:, 1] >= 0)[:, None], uf[np.maximum(src[:, 1], 0)], ufix[np.maximum(~src[:, 1], 0)] if len(ufix) else 0.0)
            _, loc, home = plan.tile_elements(t)           # whatever the record format (padding dropped, pairs expanded)
            _, gxl, gul = CF.tri3_energy(X, U, loc, mat, W)
            total += CF.tri3_energy(X, U, loc[home], mat, W, grads=False)[0]
            if ned:
                q = gp[go:go + ned]
                gl = np.stack([q & MASK, (q >> 10) & MASK], axis=1).astype(np.int64)
                CF.edge2_energy(X, U, gl, Tconst=Tc, gX=gxl, gU=gul)
                total -= CF.edge2_energy(X, U, gl[(q & HOME) != 0], Tconst=Tc)
            own = src[:nown]
            fx, fu = own[:, 0] >= 0, own[:, 1] >= 0
            gx_v[torch.from_numpy(own[fx, 0].astype(np.int64))] = torch.from_numpy(gxl[:nown][fx])
            gu_v[torch.from_numpy(own[fu, 1].astype(np.int64))] = torch.from_numpy(gul[:nown][fu])
        loss_v[0] = total

    return evaluate


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hidenn_fem_amd.mesh import structured_tri_mesh
        from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
        from hidenn_fem_amd.loss import EnergyLoss2D
        from hidenn_fem_amd.plan import TilePlan
        from hidenn_fem_amd.sharded import ShardedTri3Energy
        from oracle import closed_form as CF
        f64 = torch.float64
        coords, conn, geom, bc, mn, edges = structured_tri_mesh(41, 27, jitter=0.2, seed=5, dtype=f64)
        torch.manual_seed(3)                                   # same replicated parameters on every rank
        model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                         neumann_edges=edges)
        loss_fn = EnergyLoss2D(device=torch.device("cpu"), dtype=f64, tile_elems=96)
        plan = TilePlan(conn, coords.shape[0], coords_hint=coords, x_src=model._x_src, u_src=model._u_src,
                        edges=edges, tile_elems=96)            # host-only plan: no HIP call
        sh = ShardedTri3Energy(model, loss_fn, evaluate=oracle_tile_evaluator(plan, model), plan=plan)
        assert (sh.rank, sh.world) == (rank, world)
        lo, hi = sh.lo, sh.hi
        assert 0 <= lo < hi <= plan.n_tiles
        loss = sh()                                            # kernel stand-in + ONE all-reduce
        loss.backward()
        # local send buffer: only rows owned by this rank's tiles are non-zero
        _, gx_s, gu_s = sh._views(sh.send)
        owned_rows = int((gx_s != 0).any(dim=1).sum())
        # reference: full-mesh oracle
        X = np.zeros((coords.shape[0], 2)); U = np.zeros_like(X)
        X[~geom.numpy()] = model.node_coords_free.detach().numpy(); X[geom.numpy()] = model.node_coords_fixed.numpy()
        U[~bc.numpy()] = model.u_free.detach().numpy()
        e_ref, gX, gU = CF.tri3_energy(X, U, conn.numpy(), CF.plane_stress(), 0.25)
        e_ref -= CF.edge2_energy(X, U, edges.numpy(), Tconst=np.array([2e5, 0, 0, 0]), gX=gX, gU=gU)
        gx_ref, gu_ref = gX[~geom.numpy()], gU[~bc.numpy()]
        ok = (abs(loss.item() - e_ref) <= 1e-12 * abs(e_ref)
              and np.abs(model.node_coords_free.grad.numpy() - gx_ref).max() <= 1e-11 * np.abs(gx_ref).max()
              and np.abs(model.u_free.grad.numpy() - gu_ref).max() <= 1e-11 * np.abs(gu_ref).max())
        # every rank must hold bit-identical reduced results (identical optimiser steps follow)
        # owner-sharded mode: 8-byte exchange, local rows complete for the nodes this rank's tiles own
        sh.evaluate_local()
        loss_o, gx_o, gu_o = sh.exchange_loss_only()
        rows = (gx_o != 0).any(dim=1).numpy()
        ok = ok and abs(loss_o.item() - e_ref) <= 1e-12 * abs(e_ref) \
            and np.abs(gx_o.numpy()[rows] - gx_ref[rows]).max() <= 1e-11 * np.abs(gx_ref).max()
        mine = torch.cat([model.node_coords_free.grad.reshape(-1), model.u_free.grad.reshape(-1), loss.detach().reshape(1)])
        other = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        same = all(torch.equal(o, mine) for o in other)
        q.put((rank, bool(ok), bool(same), lo, hi, owned_rows, int(gx_s.shape[0])))
    finally:
        dist.destroy_process_group()


def torch_pack_unpack(sh):
    """torch indexing standing in for csrc/exchange.hip (hfem_iface_pack / hfem_iface_unpack) on CPU tensors."""
    def pack():
        nx, nu = sh._pub_n
        rows = sh._pub_rows.long()
        with torch.no_grad():
            sh.payload[:nx] = sh.model.node_coords_free[rows[:nx]]
            sh.payload[nx:nx + nu] = sh.model.u_free[rows[nx:]]

    def unpack():
        nx, nu = sh._need_n
        src, dst = sh._need_src.long(), sh._need_dst.long()
        with torch.no_grad():
            sh.model.node_coords_free[dst[:nx]] = sh.gathered[src[:nx]]
            sh.model.u_free[dst[nx:]] = sh.gathered[src[nx:]]
            sh.loss_global.copy_(sum(sh.gathered[r * sh.iface_stride + sh.iface_rows, 0] for r in range(sh.world)))

    return pack, unpack


def _worker_halo(rank, world, port, q):
    """Owner-sharded mode: K steps of gradient descent where each rank updates only the rows its tiles own and
    ONE all_gather per step carries interface parameter rows + partial energies; must reproduce the
    single-process trajectory (full-mesh oracle) on every rank's owned and halo rows."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hidenn_fem_amd.mesh import structured_tri_mesh
        from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
        from hidenn_fem_amd.loss import EnergyLoss2D
        from hidenn_fem_amd.plan import TilePlan
        from hidenn_fem_amd.sharded import ShardedTri3Energy
        from oracle import closed_form as CF
        f64 = torch.float64
        coords, conn, geom, bc, mn, edges = structured_tri_mesh(37, 25, jitter=0.2, seed=7, dtype=f64)
        torch.manual_seed(11)
        model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                         neumann_edges=edges)
        plan = TilePlan(conn, coords.shape[0], coords_hint=coords, x_src=model._x_src, u_src=model._u_src,
                        edges=edges, tile_elems=64)
        sh = ShardedTri3Energy(model, EnergyLoss2D(device=torch.device("cpu"), dtype=f64, tile_elems=64),
                               evaluate=oracle_tile_evaluator(plan, model), plan=plan)
        sh.setup_interfaces()
        sh._pack, sh._unpack = torch_pack_unpack(sh)
        own_x, own_u = sh.owned_rows()
        # single-process reference trajectory on the full mesh
        g_np, b_np = geom.numpy(), bc.numpy()
        xf = model.node_coords_free.detach().numpy().copy()
        uf = model.u_free.detach().numpy().copy()
        xfix = model.node_coords_fixed.numpy()
        lr_x, lr_u, K = 1e-12, 1e-13, 3
        ref_losses = []
        for _ in range(K):
            X = np.zeros((coords.shape[0], 2)); U = np.zeros_like(X)
            X[~g_np], X[g_np], U[~b_np] = xf, xfix, uf
            e, gX, gU = CF.tri3_energy(X, U, conn.numpy(), CF.plane_stress(), 0.25)
            e -= CF.edge2_energy(X, U, edges.numpy(), Tconst=np.array([2e5, 0, 0, 0]), gX=gX, gU=gU)
            ref_losses.append(e)
            xf = xf - lr_x * gX[~g_np]
            uf = uf - lr_u * gU[~b_np]
        # sharded trajectory
        losses = []
        for _ in range(K):
            sh.evaluate_owner()
            _, gx_v, gu_v = sh._views(sh.send)
            with torch.no_grad():
                model.node_coords_free[own_x] -= lr_x * gx_v[own_x]
                model.u_free[own_u] -= lr_u * gu_v[own_u]
            loss, _, _ = sh.exchange_halo()
            losses.append(loss.item())
        ok_loss = all(abs(a - b) <= 1e-12 * abs(b) for a, b in zip(losses, ref_losses))
        # rows this rank owns or reads as halo hold the reference parameters; the step really moved them
        seen_x = torch.unique(torch.cat([own_x, sh._need_dst[:sh._need_n[0]].long()]))
        seen_u = torch.unique(torch.cat([own_u, sh._need_dst[sh._need_n[0]:].long()]))
        dx = np.abs(model.node_coords_free.detach().numpy()[seen_x] - xf[seen_x]).max()
        du = np.abs(model.u_free.detach().numpy()[seen_u] - uf[seen_u]).max()
        moved = np.abs(uf - model.u_free.detach().numpy()).max() > 0 if world > 1 else True   # stale rows exist elsewhere
        st = sh.interface_stats
        q.put((rank, bool(ok_loss), float(dx), float(du), bool(moved), len(own_x), st["publish_x"], st["need_x"],
               int(model.node_coords_free.shape[0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 3])
def test_owner_sharded_halo_exchange_gloo(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_halo, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=200) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    n_rows = res[0][8]
    assert all(r[1] for r in res), "global loss trajectory differs from the single-process run"
    assert all(r[2] <= 1e-18 + 1e-13 and r[3] <= 1e-18 for r in res), f"owned/halo parameter rows diverged: {res}"
    assert sum(r[5] for r in res) == n_rows                      # ownership partitions the free rows
    assert all(0 < r[6] < r[5] and 0 < r[7] < n_rows // 2 for r in res)   # interfaces are small, non-empty


@pytest.mark.timeout(180)
def test_sharded_energy_two_ranks_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, ok0, same0, lo0, hi0, own0, n0), (r1, ok1, same1, lo1, hi1, own1, n1) = res
    assert ok0 and ok1, "sharded loss/gradients differ from the full-mesh oracle"
    assert same0 and same1, "ranks disagree after the all-reduce"
    assert lo0 == 0 and hi0 == lo1 and hi1 > lo1                # contiguous, disjoint, covering tile ranges
    assert 0 < own0 < n0 and 0 < own1 < n1 and own0 + own1 <= n0   # each rank wrote only its own rows


def test_single_process_sharded_equals_unsharded():
    """world = 1 (no process group): same code path, the exchange is a copy."""
    sys.path.insert(0, ROOT)
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.plan import TilePlan
    from hidenn_fem_amd.sharded import ShardedTri3Energy
    f64 = torch.float64
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(17, 11, jitter=0.1, seed=2, dtype=f64)
    model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges)
    plan = TilePlan(conn, coords.shape[0], coords_hint=coords, x_src=model._x_src, u_src=model._u_src, edges=edges,
                    tile_elems=64)
    sh = ShardedTri3Energy(model, EnergyLoss2D(device=torch.device("cpu"), dtype=f64, tile_elems=64),
                           evaluate=oracle_tile_evaluator(plan, model), plan=plan)
    assert (sh.lo, sh.hi) == (0, plan.n_tiles)
    loss, gx, gu = sh.value_and_grad()
    assert torch.isfinite(loss) and (gx != 0).any(dim=1).all() and (gu != 0).any(dim=1).all()


# ------------------------------------------------------------------ whole training iterations, plain and overlapped
def torch_adam(sh):
    """torch indexing standing in for hfem_adam_step_rows2_dev (csrc/optim.hip) on CPU tensors: torch.optim.Adam's update
    on the rows the rank owns, step = completed steps + 1 (the pack that follows bumps the counter)."""
    import math

    def step():
        a, m = sh._adam, sh.model
        t = int(a["step"].item()) + 1
        b1, b2 = a["betas"]
        _, gx_v, gu_v = sh._views(sh.send)
        with torch.no_grad():
            for p, g, mm, vv, rows, lr in ((m.node_coords_free, gx_v, a["mx"], a["vx"], a["rows_x"], a["lr"][0]),
                                           (m.u_free, gu_v, a["mu"], a["vu"], a["rows_u"], a["lr"][1])):
                r = rows.long()
                gi = g[r]
                mm[r] = mm[r] + (1.0 - b1) * (gi - mm[r])
                vv[r] = vv[r] * b2 + (1.0 - b2) * (gi * gi)
                p[r] = p[r] - (lr / (1.0 - b1 ** t)) * (mm[r] / (vv[r].sqrt() / math.sqrt(1.0 - b2 ** t) + a["eps"]))

    return step


def _make_trainer(world, rank_world=None):
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.plan import TilePlan
    from hidenn_fem_amd.sharded import ShardedTri3Energy
    f64 = torch.float64
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(45, 29, jitter=0.2, seed=9, dtype=f64)
    torch.manual_seed(21)
    model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges)
    plan = TilePlan(conn, coords.shape[0], coords_hint=coords, x_src=model._x_src, u_src=model._u_src, edges=edges,
                    tile_elems=48, shards=world)          # prepared for this world: boundary tiles first
    kw = {} if rank_world is None else dict(rank=rank_world[0], world=rank_world[1])
    sh = ShardedTri3Energy(model, EnergyLoss2D(device=torch.device("cpu"), dtype=f64, tile_elems=48),
                           evaluate=oracle_tile_evaluator(plan, model), plan=plan, **kw)
    sh.setup_interfaces()
    sh._pack, sh._unpack = torch_pack_unpack(sh)
    sh.init_owner_adam(lr_x=2e-6, lr_u=1e-9)
    sh._adam_step = torch_adam(sh)
    return sh, (coords, conn, geom, bc, edges)


def _worker_overlap(rank, world, port, q):
    """K Adam iterations with owner_train_step (exchange on the critical path) and with owner_train_step_overlapped
    (exchange of step k under the interior tiles of step k + 1): bit-identical parameters and energies."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        K = 4
        plain, _ = _make_trainer(world)
        over, _ = _make_trainer(world)
        assert plain.lo <= plain.mid < plain.hi and (world == 1 or plain.lo < plain.mid), "boundary and interior tiles exist"
        l_plain = []
        for _ in range(K):
            l_plain.append(plain.owner_train_step().item())
        l_over = []
        for k in range(K):
            over.owner_train_step_overlapped()
            if k > 0:
                l_over.append(over.loss_global.item())      # lags one step
        l_over.append(over.finish_overlapped().item())
        own_x, own_u = plain.owned_rows()
        seen_x = torch.unique(torch.cat([own_x, plain._need_dst[:plain._need_n[0]].long()]))
        seen_u = torch.unique(torch.cat([own_u, plain._need_dst[plain._need_n[0]:].long()]))
        # parameters bit-equal; the rank's energy is summed per tile RANGE by the CPU stand-in (two ranges -> another
        # association, last-bit differences; the HIP path sums the plan's tile energies in tile order either way)
        same = (np.allclose(l_plain, l_over, rtol=1e-14, atol=0.0)
                and torch.equal(plain.model.node_coords_free[seen_x], over.model.node_coords_free[seen_x])
                and torch.equal(plain.model.u_free[seen_u], over.model.u_free[seen_u])
                and int(over._adam["step"].item()) == K == int(plain._adam["step"].item()))
        # evaluation-only overlapped step: the energy owner_step() reports, no optimiser step counted
        plain.evaluate_owner()
        e_plain = plain.exchange_halo()[0].item()
        steps_before = int(plain._adam["step"].item())
        for _ in range(2):
            plain.owner_step_overlapped()
        same = same and abs(plain.finish_overlapped().item() - e_plain) <= 1e-14 * abs(e_plain) \
            and int(plain._adam["step"].item()) == steps_before
        moved = (plain.model.node_coords_free[own_x] - _make_trainer(world)[0].model.node_coords_free[own_x]).abs().max().item()
        q.put((rank, bool(same), l_plain, moved, plain.mid - plain.lo, plain.hi - plain.mid))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_overlapped_train_step_equals_plain_step_gloo(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_overlap, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=250) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert all(r[1] for r in res), f"overlapped trajectory differs from the plain owner-sharded step: {res}"
    assert all(r[2] == res[0][2] for r in res), "ranks disagree on the global energies"
    assert all(r[3] > 0 for r in res), "Adam moved the owned rows"
    assert all(r[4] > 0 and r[5] > 0 for r in res), "every rank has boundary and interior tiles"
    # the energies are the single-process Adam trajectory's (full-mesh oracle)
    sys.path.insert(0, ROOT)
    ref, _ = _make_trainer(1)
    l_ref = [ref.owner_train_step().item() for _ in range(len(res[0][2]))]
    assert np.allclose(res[0][2], l_ref, rtol=1e-12, atol=0.0), (res[0][2], l_ref)


def test_overlapped_train_step_single_process():
    """world = 1: everything is interior, the exchange is a copy; same numbers as the plain step."""
    sys.path.insert(0, ROOT)
    plain, _ = _make_trainer(1)
    over, _ = _make_trainer(1)
    assert (over.lo, over.mid, over.hi) == (0, 0, over.plan.n_tiles)
    a = [plain.owner_train_step().item() for _ in range(3)]
    for _ in range(3):
        over.owner_train_step_overlapped()
    last = over.finish_overlapped().item()
    assert last == a[-1]                                        # one range: the same sum
    assert torch.equal(plain.model.node_coords_free, over.model.node_coords_free)
    assert torch.equal(plain.model.u_free, over.model.u_free)

"""Owner-sharded multi-GPU mode on ONE MI355X: two 'ranks' emulated in one process (the collective replaced by a
concatenation of the two payloads), HIP kernels everywhere else -- tile-range energy kernel, hfem_iface_pack,
hfem_iface_unpack.  The real collective is covered by the world_size-2/3 gloo tests (tests/test_sharded_gloo.py)."""
import copy

import numpy as np
import pytest
import torch

F64 = torch.float64


@pytest.mark.gpu
def test_two_emulated_ranks_match_unsharded_step():
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.sharded import ShardedTri3Energy
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(161, 97, jitter=0.2, seed=3, dtype=F64)
    torch.manual_seed(5)
    base = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                    neumann_edges=edges).to(d)
    with torch.no_grad():
        base.u_free.mul_(30.0)
    lf = EnergyLoss2D(device=d, dtype=F64, tile_elems=256)
    # unsharded reference step (the parity-tested single-GPU path)
    ref = copy.deepcopy(base)
    loss_ref = lf(ref)
    loss_ref.backward()
    lr_x, lr_u = 1e-13, 1e-14
    with torch.no_grad():
        x_new = ref.node_coords_free - lr_x * ref.node_coords_free.grad
        u_new = ref.u_free - lr_u * ref.u_free.grad
    world = 2
    ranks = []
    for r in range(world):
        m = copy.deepcopy(base)
        sh = ShardedTri3Energy(m, lf, rank=r, world=world).setup_interfaces()
        ranks.append((m, sh))
    assert ranks[0][1].hi == ranks[1][1].lo and ranks[0][1].iface_stride == ranks[1][1].iface_stride
    for m, sh in ranks:                                   # evaluate own tiles, update own rows, publish
        sh.evaluate_owner()
        _, gx, gu = sh._views(sh.send)
        ox, ou = sh.owned_rows()
        with torch.no_grad():
            m.node_coords_free[ox] -= lr_x * gx[ox]
            m.u_free[ou] -= lr_u * gu[ou]
        sh._pack()
    gathered = torch.cat([sh.payload for _, sh in ranks])       # what all_gather_into_tensor delivers
    total_owned = 0
    for m, sh in ranks:
        sh.gathered.copy_(gathered)
        sh._unpack()
        torch.cuda.synchronize()
        assert abs(sh.loss_global.item() - loss_ref.item()) <= 1e-12 * abs(loss_ref.item())
        ox, ou = sh.owned_rows()
        total_owned += len(ox)
        nx = sh._need_n[0]
        seen_x = torch.unique(torch.cat([ox, sh._need_dst[:nx].long()]))
        seen_u = torch.unique(torch.cat([ou, sh._need_dst[nx:].long()]))
        assert sh._need_n[0] > 0 and sh._pub_n[0] > 0
        # owned rows and halo rows carry the unsharded step's parameters (halo rows bit-exact copies of the owner's)
        assert (m.node_coords_free[seen_x] - x_new[seen_x]).abs().max().item() <= 1e-15
        assert (m.u_free[seen_u] - u_new[seen_u]).abs().max().item() <= 1e-10 * u_new.abs().max().item() * 1e-6
        st = sh.interface_stats
        assert st["payload_bytes"] < 0.05 * m.node_coords_free.numel() * 8      # interface << field
    assert total_owned == base.node_coords_free.shape[0]
    other = ranks[1][0]
    halo_rows = ranks[0][1]._need_dst[:ranks[0][1]._need_n[0]].long()
    assert torch.equal(ranks[0][0].node_coords_free[halo_rows], other.node_coords_free[halo_rows])


@pytest.mark.gpu
def test_loss_is_bit_reproducible_and_deferred_sum_matches():
    """The energy is summed in fixed order at every level (lanes by shuffle tree, waves in wave order, tiles in tile
    order): two launches on the same inputs give the same bits.  HFEM_FLAG_NO_LOSS_SUM + hfem_plan_loss_sum is the
    same sum, deferred (also on a tile sub-range)."""
    import ctypes as C
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(401, 251, jitter=0.2, seed=4, dtype=F64)
    torch.manual_seed(1)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    with torch.no_grad():
        m.u_free.mul_(100.0)
    lf = EnergyLoss2D(device=d, dtype=F64)
    plan = m.tile_plan(lf.tile_elems)
    L = _lib.lib()
    dv = lambda v: (C.c_double * len(v))(*v)
    _, Tconst = lf._traction(m, None)
    xf, uf, xfix, ufix = m.node_coords_free.detach(), m.u_free.detach(), m.node_coords_fixed, m.u_fixed_rows()
    gx, gu = torch.empty_like(xf), torch.empty_like(uf)
    out = torch.zeros(8, dtype=F64, device=d)
    st = torch.cuda.current_stream().cuda_stream

    def run(slot, lo, hi, flags):
        _lib.check(L.hfem_tri3_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(), ufix.data_ptr(),
                                           dv(lf._mat), lf._W, dv([0.0] * 6), None, dv(Tconst), lo, hi,
                                           out[slot:slot + 1].data_ptr(), gx.data_ptr(), gu.data_ptr(), flags, st))

    nt = plan.n_tiles
    for rep in range(20):
        run(0, 0, -1, 0)
        run(1, 0, -1, 0)
        run(2, 0, -1, 8)                                   # NO_LOSS_SUM: out[2] untouched ...
        _lib.check(L.hfem_plan_loss_sum(plan.handle, 0, -1, out[3:4].data_ptr(), st))
        run(4, 5, nt - 9, 0)
        run(5, 5, nt - 9, 8)
        _lib.check(L.hfem_plan_loss_sum(plan.handle, 5, nt - 9, out[6:7].data_ptr(), st))
        o = out.tolist()
        assert o[0] == o[1] == o[3] and o[2] == 0.0 and o[4] == o[6] and o[5] == 0.0 and o[4] != o[0]


@pytest.mark.gpu
def test_rccl_backend_single_rank_collectives():
    """The N > 1 bench path uses torch.distributed's `nccl` backend (= RCCL on ROCm).  One GPU is all a test box has,
    so: a 1-rank process group on the real backend, the two collectives the exchange modes issue on the very buffers
    they issue them on (fp64 sum all-reduce of [gX|gU|loss]; all_gather_into_tensor of the interface payload), and the
    sharded evaluator end to end."""
    import socket
    import torch.distributed as dist
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.sharded import ShardedTri3Energy
    assert dist.is_nccl_available()
    d = torch.device("cuda:0")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=d)
    try:
        coords, conn, geom, bc, mn, edges = structured_tri_mesh(101, 61, jitter=0.2, seed=1, dtype=F64)
        torch.manual_seed(0)
        m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                     neumann_edges=edges).to(d)
        lf = EnergyLoss2D(device=d, dtype=F64)
        sh = ShardedTri3Energy(m, lf).setup_interfaces()
        assert (sh.rank, sh.world) == (0, 1)
        loss, gx, gu = sh.value_and_grad()
        ref = lf(m)
        ref.backward()
        assert abs(loss.item() - ref.item()) <= 1e-13 * abs(ref.item())
        assert (gx - m.node_coords_free.grad).abs().max().item() <= 1e-12 * gx.abs().max().item()
        # the collectives themselves, on the exchange buffers
        before = sh.send.clone()
        dist.all_reduce(sh.send, op=dist.ReduceOp.SUM)
        assert torch.equal(sh.send, before)
        sh.evaluate_owner()
        sh._pack()
        dist.all_gather_into_tensor(sh.gathered, sh.payload)
        sh._unpack()
        dist.barrier()
        torch.cuda.synchronize()
        assert abs(sh.loss_global.item() - ref.item()) <= 1e-13 * abs(ref.item())
        assert torch.equal(sh.gathered, sh.payload)
        # the hoisted-lookup form bench.py times is the same step
        sh.loss_global.zero_()
        loss_s, gx_s, gu_s = sh.owner_step()
        torch.cuda.synchronize()
        assert loss_s.item() == ref.item() or abs(loss_s.item() - ref.item()) <= 1e-13 * abs(ref.item())
        assert (gx_s - m.node_coords_free.grad).abs().max().item() <= 1e-12 * gx_s.abs().max().item()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_lagged_loss_sum_delivers_every_energy_bit_exactly():
    """HFEM_FLAG_SUM_PREVIOUS: the energy of evaluation k is reduced by an extra workgroup of launch k+1 (two partials
    banks), the last one by hfem_plan_loss_sum.  With the inputs changing before every launch, eagerly and back to back in
    a hipGraph, on the full plan and on a tile sub-range, every delivered energy must equal the inline sum's bits."""
    import ctypes as C
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(501, 301, jitter=0.2, seed=4, dtype=F64)
    torch.manual_seed(1)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    lf = EnergyLoss2D(device=d, dtype=F64)
    plan = m.tile_plan(lf.tile_elems)
    L = _lib.lib()
    dv = lambda v: (C.c_double * len(v))(*v)
    _, Tconst = lf._traction(m, None)
    xf, uf, xfix, ufix = m.node_coords_free.detach(), m.u_free.detach(), m.node_coords_fixed, m.u_fixed_rows()
    gx, gu = torch.empty_like(xf), torch.empty_like(uf)
    nt, K = plan.n_tiles, 40
    assert nt > 200

    def run(out, lo, hi, flags, stream):
        _lib.check(L.hfem_tri3_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(), ufix.data_ptr(),
                                           dv(lf._mat), lf._W, dv([0.0] * 6), None, dv(Tconst), lo, hi, out.data_ptr(),
                                           gx.data_ptr(), gu.data_ptr(), flags, stream), "energy")

    for lo, hi in ((0, -1), (11, nt - 7)):
        scales = [1.0 + 1e-3 * ((i % 7) - 3) for i in range(2 * K)]
        want = torch.zeros(2 * K, dtype=F64, device=d)
        uf0 = uf.clone()
        st = torch.cuda.current_stream().cuda_stream
        for i in range(2 * K):                                   # inline sums: the reference sequence
            uf.mul_(scales[i])
            run(want[i:i + 1], lo, hi, 0, st)
        torch.cuda.synchronize()
        uf.copy_(uf0)
        got = torch.zeros(2 * K, dtype=F64, device=d)
        for i in range(K):                                       # eager lagged sequence: loss k arrives with launch k+1
            uf.mul_(scales[i])
            run(got[max(i - 1, 0):max(i - 1, 0) + 1], lo, hi, 8 | (32 if i else 0), st)
        _lib.check(L.hfem_plan_loss_sum(plan.handle, lo, hi, got[K - 1:K].data_ptr(), st))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()                               # the second half back to back inside one graph
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        scratch = torch.zeros(1, dtype=F64, device=d)
        with torch.cuda.stream(s):
            run(scratch, lo, hi, 0, s.cuda_stream)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            cs = torch.cuda.current_stream().cuda_stream
            for i in range(K, 2 * K):
                uf.mul_(scales[i])
                j = max(i - 1, K)
                run(got[j:j + 1], lo, hi, 8 | (32 if i > K else 0), cs)
            _lib.check(L.hfem_plan_loss_sum(plan.handle, lo, hi, got[2 * K - 1:2 * K].data_ptr(), cs))
        g.replay()
        torch.cuda.synchronize()
        assert want.unique().numel() > K
        bad = (got != want).nonzero().flatten()
        assert bad.numel() == 0, f"range {(lo, hi)}: {bad.numel()} wrong lagged energies, first {bad[:5].tolist()}"
        uf.copy_(uf0)
    with pytest.raises(RuntimeError):                            # nothing to consume on a fresh plan
        m2 = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0).to(d)
        p2 = m2.tile_plan(0)
        _lib.check(L.hfem_tri3_energy_plan(p2.handle, m2.node_coords_free.data_ptr(), m2.node_coords_fixed.data_ptr(),
                                           m2.u_free.data_ptr(), m2.u_fixed_rows().data_ptr(), dv(lf._mat), lf._W,
                                           dv([0.0] * 6), None, dv(Tconst), 0, -1, scratch.data_ptr(), None, None, 8 | 32 | 3, st))


@pytest.mark.gpu
def test_library_comm_owner_training_step_captures_into_a_graph():
    """In-library RCCL (hfem_mg_*, csrc/mg.cpp) on one rank: a whole owner-sharded training iteration -- energy,
    Adam on the owned rows, pack, ncclAllGather on the caller's stream, unpack -- is captured into ONE hipGraph
    (4 iterations per replay) and replayed; parameters and per-iteration energies must equal the eager sequence and
    the plain single-GPU loop (EnergyLoss2D.value_and_grad_ + FusedAdam).  Also the dense mode's out-of-place
    ncclAllReduce, captured."""
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import FusedAdam
    from hidenn_fem_amd.sharded import LibraryComm, ShardedTri3Energy
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(161, 121, jitter=0.2, seed=6, dtype=F64)

    def model():
        torch.manual_seed(2)
        return PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                        neumann_edges=edges).to(d)
    lr_x, lr_u, n_iter, per = 1e-6, 1e-8, 12, 4
    # (a) the plain single-GPU loop
    m0 = model()
    lf = EnergyLoss2D(device=d, dtype=F64)
    opt = FusedAdam([dict(params=[m0.node_coords_free], lr=lr_x), dict(params=[m0.u_free], lr=lr_u)])
    ref_losses = []
    for _ in range(n_iter):
        ref_losses.append(lf.value_and_grad_(m0).item())
        opt.step()
    # (b) owner-sharded, in-library comm, eager
    comm = LibraryComm(d)
    assert comm.world == 1

    def sharded(m):
        sh = ShardedTri3Energy(m, EnergyLoss2D(device=d, dtype=F64), comm=comm)
        sh.setup_interfaces()
        sh.init_owner_adam(lr_x, lr_u)
        return sh
    m1 = model()
    sh1 = sharded(m1)
    eager = [sh1.owner_train_step().item() for _ in range(n_iter)]
    np.testing.assert_allclose(eager, ref_losses, rtol=1e-12)
    for a, b in zip(m1.parameters(), m0.parameters()):
        assert (a - b).abs().max().item() <= 1e-12 * b.abs().max().item()
    # (c) the same, `per` iterations per hipGraph replay
    m2 = model()
    sh2 = sharded(m2)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        first = sh2.owner_train_step().item()                  # warm-up iteration (counts)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    losses = torch.zeros(per, dtype=F64, device=d)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(per):
            losses[i].copy_(sh2.owner_train_step())
    got = [first]
    n_replays = (n_iter - 1) // per
    for _ in range(n_replays):
        g.replay()
        torch.cuda.synchronize()
        got += losses.tolist()
    np.testing.assert_allclose(got, ref_losses[:1 + n_replays * per], rtol=1e-12)
    assert int(sh2._adam["step"].item()) == 1 + n_replays * per
    # dense mode: out-of-place all-reduce on the caller's stream, captured
    sh2.evaluate_local()
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        sh2.exchange()
    torch.cuda.current_stream().wait_stream(s2)
    torch.cuda.synchronize()
    with torch.cuda.graph(g2):
        sh2.evaluate_local()
        sh2.exchange()
    sh2.recv.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(sh2.recv, sh2.send)
    comm.close()


@pytest.mark.gpu
def test_overlapped_training_step_hip_eager_and_captured():
    """owner_train_step_overlapped on the GPU: interior tiles -> join -> boundary tiles (HFEM_FLAG_SAME_BANK) -> Adam (one
    launch for both tensors) -> pack + energy + step counter (one launch) -> all_gather + unpack on a side stream.  One
    rank on the in-library RCCL communicator with the tile range split artificially into 'boundary' and 'interior' parts
    (the split only changes which launch evaluates a tile); eager and 4 iterations per hipGraph; against the plain
    single-GPU loop (value_and_grad_ + FusedAdam) and the non-overlapped owner_train_step."""
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.optim import FusedAdam
    from hidenn_fem_amd.sharded import LibraryComm, ShardedTri3Energy
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(201, 151, jitter=0.2, seed=8, dtype=F64)

    def model():
        torch.manual_seed(4)
        return PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                        neumann_edges=edges).to(d)
    lr_x, lr_u, n_iter, per = 1e-6, 1e-8, 13, 4
    m0 = model()
    lf = EnergyLoss2D(device=d, dtype=F64)
    opt = FusedAdam([dict(params=[m0.node_coords_free], lr=lr_x), dict(params=[m0.u_free], lr=lr_u)])
    ref_losses = []
    for _ in range(n_iter):
        ref_losses.append(lf.value_and_grad_(m0).item())
        opt.step()
    comm = LibraryComm(d)

    def sharded(m, split):
        sh = ShardedTri3Energy(m, EnergyLoss2D(device=d, dtype=F64), comm=comm)
        sh.setup_interfaces()
        sh.init_owner_adam(lr_x, lr_u)
        assert (sh.lo, sh.mid, sh.hi) == (0, 0, sh.plan.n_tiles)
        if split:
            sh.mid = sh.plan.n_tiles // 3
        return sh
    for split in (False, True):
        # eager
        m1 = model()
        sh1 = sharded(m1, split)
        got = []
        for k in range(n_iter):
            sh1.owner_train_step_overlapped()
            if k:
                got.append(sh1.loss_global.item())             # lags one step (a device-to-host read joins the streams)
        got.append(sh1.finish_overlapped().item())
        torch.cuda.synchronize()
        np.testing.assert_allclose(got, ref_losses, rtol=1e-12)
        for a, b in zip(m1.parameters(), m0.parameters()):
            assert (a - b).abs().max().item() <= 1e-12 * b.abs().max().item()
        assert int(sh1._adam["step"].item()) == n_iter
        # `per` iterations per hipGraph replay; every graph ends joined
        m2 = model()
        sh2 = sharded(m2, split)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            sh2.owner_train_step_overlapped()
            first = sh2.finish_overlapped().item()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        losses = torch.zeros(per, dtype=F64, device=d)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(per):
                sh2.owner_train_step_overlapped()
                if i:
                    losses[i - 1].copy_(sh2.loss_global)        # after the join inside the step: energy of step i - 1
            losses[per - 1].copy_(sh2.finish_overlapped())
        got = [first]
        n_replays = (n_iter - 1) // per
        for _ in range(n_replays):
            g.replay()
            torch.cuda.synchronize()
            got += losses.tolist()
        np.testing.assert_allclose(got, ref_losses[:1 + n_replays * per], rtol=1e-12)
        assert int(sh2._adam["step"].item()) == 1 + n_replays * per
    # the non-overlapped step runs the same launches in another order
    m3 = model()
    sh3 = sharded(m3, False)
    plain = [sh3.owner_train_step().item() for _ in range(n_iter)]
    np.testing.assert_allclose(plain, ref_losses, rtol=1e-12)
    # fused steps: Adam applied by the energy kernel's write-out on tile ranges (ping-pong parameter buffers, bias corrections
    # refreshed by the pack launch) -- plain and overlapped, eager and 4 iterations per hipGraph
    def sharded_fused(m, split):
        sh = ShardedTri3Energy(m, EnergyLoss2D(device=d, dtype=F64), comm=comm)
        sh.setup_interfaces()
        sh.init_owner_adam(lr_x, lr_u, fused=True)
        if split:
            sh.mid = sh.plan.n_tiles // 3
        return sh
    for name, split in (("owner_train_step_fused", False), ("owner_train_step_fused_overlapped", True)):
        m5 = model()
        sh5 = sharded_fused(m5, split)
        step = getattr(sh5, name)
        got = []
        for k in range(n_iter - 1):                            # 12: an even number, as the captured variant needs
            step()
            if not split:
                got.append(sh5.loss_global.item())
            elif k:
                got.append(sh5.loss_global.item())
        if split:
            got.append(sh5.finish_overlapped().item())
        torch.cuda.synchronize()
        np.testing.assert_allclose(got, ref_losses[:n_iter - 1], rtol=1e-12)
        assert int(sh5._adam["step"].item()) == n_iter - 1
        # one more eager step brings it level with the reference loop's parameters
        step()
        if split:
            sh5.finish_overlapped()
        for a, b in zip(m5.parameters(), m0.parameters()):
            assert (a - b).abs().max().item() <= 1e-12 * b.abs().max().item()
        # captured: 4 iterations per graph
        m6 = model()
        sh6 = sharded_fused(m6, split)
        step6 = getattr(sh6, name)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step6(); step6()
            if split:
                sh6.finish_overlapped()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g6 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g6):
            for _ in range(4):
                step6()
            if split:
                sh6.finish_overlapped()
        g6.replay(); g6.replay()
        torch.cuda.synchronize()
        assert int(sh6._adam["step"].item()) == 10
        assert abs(sh6.loss_global.item() - ref_losses[9]) <= 1e-12 * abs(ref_losses[9])
        # the model's parameters are the buffers the last step wrote
        m7 = model()
        o7 = FusedAdam([dict(params=[m7.node_coords_free], lr=lr_x), dict(params=[m7.u_free], lr=lr_u)])
        for _ in range(10):
            lf.value_and_grad_(m7)
            o7.step()
        for a, b in zip(m6.parameters(), m7.parameters()):
            assert (a - b).abs().max().item() <= 1e-12 * b.abs().max().item()
    # evaluation-only counterpart (what bench.py times as eval_exchange_overlap): same energy and gradients as owner_step
    m4 = model()
    sh4 = sharded(m4, True)
    l_ref, gx_ref, gu_ref = sh4.owner_step()
    l_ref, gx_ref, gu_ref = l_ref.item(), gx_ref.clone(), gu_ref.clone()
    sh4.send.zero_()
    for _ in range(3):
        sh4.owner_step_overlapped()
    assert sh4.finish_overlapped().item() == l_ref
    _, gx4, gu4 = sh4._views(sh4.send)
    assert (gx4 - gx_ref).abs().max().item() <= 1e-12 * gx_ref.abs().max().item()
    assert (gu4 - gu_ref).abs().max().item() <= 1e-12 * gu_ref.abs().max().item()
    assert int(sh4._adam["step"].item()) == 0
    comm.close()


@pytest.mark.gpu
def test_same_bank_ranges_sum_to_the_whole_evaluation_and_span_stamps():
    """HFEM_FLAG_SAME_BANK: two tile ranges of one evaluation leave their tile energies in one bank, hfem_plan_loss_sum
    over the union = the single-launch energy, bit for bit.  Span stamps: every tile of a launch reports start <= end
    ticks inside the launch's window, only while a buffer is set."""
    import ctypes as C
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(701, 401, jitter=0.2, seed=4, dtype=F64)
    torch.manual_seed(1)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    lf = EnergyLoss2D(device=d, dtype=F64)
    plan = m.tile_plan(0)
    assert plan.is_paired()
    L = _lib.lib()
    dv = lambda v: (C.c_double * len(v))(*v)
    _, Tconst = lf._traction(m, None)
    xf, uf, xfix, ufix = m.node_coords_free.detach(), m.u_free.detach(), m.node_coords_fixed, m.u_fixed_rows()
    gx, gu = torch.empty_like(xf), torch.empty_like(uf)
    out = torch.zeros(4, dtype=F64, device=d)
    st = torch.cuda.current_stream().cuda_stream
    nt = plan.n_tiles

    def run(slot, lo, hi, flags):
        _lib.check(L.hfem_tri3_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(), ufix.data_ptr(),
                                           dv(lf._mat), lf._W, dv([0.0] * 6), None, dv(Tconst), lo, hi,
                                           out[slot:slot + 1].data_ptr(), gx.data_ptr(), gu.data_ptr(), flags, st), "energy")
    run(0, 0, -1, 0)
    g_ref = (gx.clone(), gu.clone())
    mid = nt // 5
    gx.zero_(); gu.zero_()
    run(1, mid, nt, 8)                         # "interior" first ...
    run(1, 0, mid, 8 | 256)                    # ... then "boundary", same bank
    _lib.check(L.hfem_plan_loss_sum(plan.handle, 0, -1, out[2:3].data_ptr(), st))
    torch.cuda.synchronize()
    assert out[2].item() == out[0].item() and out[1].item() == 0.0
    assert (gx - g_ref[0]).abs().max().item() <= 1e-12 * g_ref[0].abs().max().item()
    with pytest.raises(RuntimeError):
        run(3, 0, -1, 256)                     # SAME_BANK without NO_LOSS_SUM
    # span stamps
    slots = 3
    buf = torch.zeros(slots * nt * 2, dtype=torch.int64, device=d)
    _lib.check(L.hfem_plan_set_span_stamps(plan.handle, buf.data_ptr(), slots))
    run(0, 0, -1, 8)
    run(0, 3, nt - 2, 8)
    torch.cuda.synchronize()
    sp = buf.view(slots, nt, 2).cpu().numpy()
    assert (sp[0, :, 0] > 0).all() and (sp[0, :, 1] >= sp[0, :, 0]).all()
    span0 = (sp[0, :, 1].max() - sp[0, :, 0].min()) * 0.01       # 100 MHz ticks -> us
    assert 1.0 < span0 < 200.0, span0
    assert (sp[1, :3] == 0).all() and (sp[1, nt - 2:] == 0).all() and (sp[1, 3:nt - 2, 0] >= sp[0, :, 0].min()).all()
    assert (sp[2] == 0).all()
    _lib.check(L.hfem_plan_set_span_stamps(plan.handle, None, 0))
    buf.zero_()
    run(0, 0, -1, 0)
    torch.cuda.synchronize()
    assert int(buf.abs().sum().item()) == 0


def _two_gpu_worker(rank, world, port, q):
    """One rank of the 2-GPU LibraryComm test: owner-sharded training through the in-library RCCL communicator (plain
    and overlapped) against the same steps through torch.distributed's own nccl collectives."""
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    d = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=d)
    try:
        from hidenn_fem_amd.mesh import structured_tri_mesh
        from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
        from hidenn_fem_amd.loss import EnergyLoss2D
        from hidenn_fem_amd.sharded import LibraryComm, ShardedTri3Energy
        coords, conn, geom, bc, mn, edges = structured_tri_mesh(201, 151, jitter=0.2, seed=6, dtype=F64)

        def trainer(comm):
            torch.manual_seed(2)
            m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
            sh = ShardedTri3Energy(m, EnergyLoss2D(device=d, dtype=F64), comm=comm)
            sh.setup_interfaces()
            sh.init_owner_adam(1e-6, 1e-8)
            return m, sh
        comm = LibraryComm(d)
        n_iter = 8
        m_pg, sh_pg = trainer(None)                              # torch.distributed collectives (the checked path)
        ref = [sh_pg.owner_train_step().item() for _ in range(n_iter)]
        m_lib, sh_lib = trainer(comm)                            # in-library RCCL, eager
        got = [sh_lib.owner_train_step().item() for _ in range(n_iter)]
        m_ov, sh_ov = trainer(comm)                              # in-library RCCL, overlapped, 4 iterations per hipGraph
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            sh_ov.owner_train_step_overlapped()
            first = sh_ov.finish_overlapped().item()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        per = 4
        losses = torch.zeros(per, dtype=F64, device=d)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(per):
                sh_ov.owner_train_step_overlapped()
                if i:
                    losses[i - 1].copy_(sh_ov.loss_global)
            losses[per - 1].copy_(sh_ov.finish_overlapped())
        g.replay()
        torch.cuda.synchronize()
        got_ov = [first] + losses.tolist()
        # dense mode: out-of-place all-reduce through the library vs torch.distributed
        sh_lib.evaluate_local()
        l1 = sh_lib.exchange()[0].item()
        sh_pg.evaluate_local()
        l2 = sh_pg.exchange()[0].item()
        torch.cuda.synchronize()
        own_x, _ = sh_pg.owned_rows()
        dx = (m_lib.node_coords_free[own_x] - m_pg.node_coords_free[own_x]).abs().max().item()
        q.put((rank, ref, got, got_ov, l1, l2, dx, sh_lib.mid - sh_lib.lo, sh_lib.hi - sh_lib.mid))
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node); one-GPU boxes skip")
def test_library_comm_two_gpus_matches_torch_distributed():
    """the in-library RCCL path (ncclCommInitRank with a broadcast id, ncclAllGather payload layout against
    hfem_iface_unpack's [world][stride] expectation, coexistence with torch's own communicator) with MORE than one rank:
    two processes, two GPUs -- eager, overlapped and hipGraph-captured -- against the torch.distributed collectives."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ref, got, got_ov, l1, l2, dx, nb, ni in res:
        np.testing.assert_allclose(got, ref, rtol=1e-12)
        np.testing.assert_allclose(got_ov, ref[:len(got_ov)], rtol=1e-12)
        assert abs(l1 - l2) <= 1e-12 * abs(l2) and dx <= 1e-12
        assert nb > 0 and ni > 0
    assert res[0][1] == res[1][1], "both ranks report the same global energies"

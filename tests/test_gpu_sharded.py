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

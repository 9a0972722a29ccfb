#!/usr/bin/env python3
"""Dev tool (GPU box): the paired TRI3 kernel for several tile sizes (plan_node_cap = home nodes per tile) on meshes of one
resident round (T1M) and of several (T2M, the 4 x 10^6-element cfg5 mesh through the plain API): tiles of at most 32 KB LDS
run five workgroups per CU instead of four.  Kernel only, hipGraph of K launches, HIP events, median of 5.

    python scripts/tile_cap_sweep.py [caps] [meshes]
"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hidenn_fem_amd import _lib  # noqa: E402
from hidenn_fem_amd.loss import EnergyLoss2D  # noqa: E402
from hidenn_fem_amd.mesh import structured_tri_mesh  # noqa: E402
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D  # noqa: E402

K = 60
dev, f64 = torch.device("cuda:0"), torch.float64
L = _lib.lib()
dv = lambda v: (C.c_double * len(v))(*v)
caps = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "-1,470,445,420,380").split(",")]
which = (sys.argv[2] if len(sys.argv) > 2 else "T1M,T2M,cfg5auto").split(",")
meshes = dict(
    T1M=lambda: structured_tri_mesh(1001, 501, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64),
    T2M=lambda: structured_tri_mesh(1001, 1001, length=2.0, height=2.0, jitter=0.2, seed=0, dtype=f64),
    cfg5auto=lambda: structured_tri_mesh(2001, 1001, jitter=0.3, seed=11, diagonal="random", permute=True, dtype=f64))


def timed(launch, rot):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        launch(0, s.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(K):
            launch(k % rot, torch.cuda.current_stream().cuda_stream)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / K)
    return sorted(out)[2]


for name in which:
    mesh = meshes[name]()
    for cap in caps:
        _lib.check(L.hfem_set_option(b"plan_node_cap", cap))
        c_, cn_, g_, b_, _, e_ = mesh
        torch.manual_seed(0)
        m = PiecewiseLinearShapeNN2D(c_, cn_, boundary_mask=g_, dirichlet_mask=b_, u_fixed=0.0, neumann_edges=e_).to(dev)
        pl = m.tile_plan(0)
        lf = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64)
        _, Tc = lf._traction(m, None)
        mat, Tcv, Bk = dv(lf._mat), dv(Tc), dv([0.0] * 6)
        x, u = m.node_coords_free.detach(), m.u_free.detach()
        xf, uf = m.node_coords_fixed, m.u_fixed_rows()
        ls = torch.zeros((), dtype=f64, device=dev)
        R = max(2, int(320.0 / (4 * x.numel() * 8 / 2 ** 20)) + 1)
        sets = [(x.clone(), u.clone(), torch.empty_like(x), torch.empty_like(u)) for _ in range(R)]

        def launch(i, stream):
            xs, us, gxs, gus = sets[i]
            _lib.check(L.hfem_tri3_energy_plan(pl.handle, xs.data_ptr(), xf.data_ptr(), us.data_ptr(), uf.data_ptr(), mat, lf._W, Bk,
                                               None, Tcv, 0, -1, ls.data_ptr(), gxs.data_ptr(), gus.data_ptr(), 8, stream))
        a, b = timed(launch, 1), timed(launch, R)
        st = pl.stats
        ne, nn = cn_.shape[0], c_.shape[0]
        alg = 12 * ne + 64 * nn + 8
        print(json.dumps(dict(mesh=name, cap=cap, replayed_us=round(a, 3), rotating_us=round(b, 3), replayed_frac=round(alg / a / 8e6, 3),
                              rotating_frac=round(alg / b / 8e6, 3), tiles=st["n_tiles"], lds=st["lds_bytes"], slot_rows=st["slot_rows"],
                              max_owned=st["max_tile_owned"], halo_nodes=round(st["tile_node_total"] / nn, 3),
                              stores=st["store_policy"])), flush=True)
        del sets, m, pl

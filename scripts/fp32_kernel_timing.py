#!/usr/bin/env python3
"""Dev tool (GPU box): the fp32-model energy kernel on T1M -- fp32 arithmetic (tri3_pair_f32.hip, HFEM_FLAG_FP32_MATH) against
the fp64-arithmetic float-row instance, kernel-only time (hipGraph of K launches, HIP events) and error against the fp64
kernel on the same float values.  One JSON line."""
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

K = 200
dev = torch.device("cuda:0")
f64 = torch.float64
nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1001, 501)
if len(sys.argv) > 3:                  # home nodes per tile (plan_node_cap) for this run
    _lib.check(_lib.lib().hfem_set_option(b"plan_node_cap", int(sys.argv[3])), "hfem_set_option")
coords, conn, geom, bc, mn, edges = structured_tri_mesh(nx, ny, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64)
ne, nn = conn.shape[0], coords.shape[0]
L = _lib.lib()
dv = lambda v: (C.c_double * len(v))(*v)
torch.manual_seed(0)
m32 = PiecewiseLinearShapeNN2D(coords.float(), conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(dev)
pl = m32.tile_plan(0)
x, u = m32.node_coords_free.detach(), m32.u_free.detach()
xf, uf = m32.node_coords_fixed, m32.u_fixed_rows()
gx, gu = torch.empty_like(x), torch.empty_like(u)
ls = torch.zeros((), dtype=f64, device=dev)
lf32 = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=torch.float32)
_, Tc = lf32._traction(m32, None)
mat, Tcv, Bk = dv(lf32._mat), dv(Tc), dv([0.0] * 6)


def launch(flags, stream):
    _lib.check(L.hfem_tri3_energy_plan_f32(pl.handle, x.data_ptr(), xf.data_ptr(), u.data_ptr(), uf.data_ptr(), mat, lf32._W, Bk, None,
                                           Tcv, 0, -1, ls.data_ptr(), gx.data_ptr(), gu.data_ptr(), flags, stream), "f32")


def time_it(flags):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        launch(flags, s.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K):
            launch(flags, torch.cuda.current_stream().cuda_stream)
    for _ in range(30):
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


m64 = PiecewiseLinearShapeNN2D(coords.float().double(), conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(dev)
with torch.no_grad():
    m64.u_free.copy_(u.double())
lf64 = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64)
lf64._mat, lf64._W, lf64._ci, lf64._cj = lf32._mat, lf32._W, lf32._ci, lf32._cj
l64 = lf64.value_and_grad_(m64).item()
res = dict(elements=ne, nodes=nn, tiles=pl.stats["n_tiles"], slot_rows=pl.stats["slot_rows"], max_owned=pl.stats["max_tile_owned"],
           max_nodes=pl.stats["max_tile_nodes"], alg_bytes=12 * ne + 32 * nn + 8)
for tag, fl in (("fp32_arithmetic", 1024), ("fp64_arithmetic", 0)):
    launch(fl, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    err = dict(loss=abs(ls.item() - l64) / abs(l64),
               gx=((gx.double() - m64.node_coords_free.grad).abs().max() / m64.node_coords_free.grad.abs().max()).item(),
               gu=((gu.double() - m64.u_free.grad).abs().max() / m64.u_free.grad.abs().max()).item())
    us = time_it(8 | fl)
    res[tag] = dict(kernel_us=us, frac=res["alg_bytes"] / (us * 1e-6) / 8e12, err=err)
print(json.dumps(res))

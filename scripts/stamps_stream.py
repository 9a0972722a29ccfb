#!/usr/bin/env python3
"""Lab: per-workgroup timeline of the streamed TRI3 kernel (s_memrealtime stamps, 10 ns ticks).
   python scripts/stamps_stream.py --ablate 256   (256 = stamps only; 263 = + no math / gather / stores ...)"""
import os; os.environ.setdefault("HFEM_LAB", "1")   # kernel-lab tool: needs libhidenn_hip_lab.so (build.py --lab)
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hidenn_fem_amd import _lib
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.loss import EnergyLoss2D
from hidenn_fem_amd.plan import TilePlan
ap = argparse.ArgumentParser(); ap.add_argument("--ablate", default="256"); ap.add_argument("--launches", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda:0"); f64 = torch.float64
coords, conn, geom, bc, mn, edges = structured_tri_mesh(1001, 501, jitter=0.2, seed=0, dtype=f64)
torch.manual_seed(0)
model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(dev)
lf = EnergyLoss2D(device=dev, dtype=f64)
plan = TilePlan(model.connectivity, model.Nnodes, coords_hint=model.initial_node_coords, x_src=model._x_src,
                u_src=model._u_src, edges=model.neumann_edges, device=dev, elem_order=4)
L = _lib.lib(); dv = lambda v: (C.c_double * len(v))(*v)
xf, uf = model.node_coords_free.detach(), model.u_free.detach(); xfix, ufix = model.node_coords_fixed, model.u_fixed_rows()
_, Tconst = lf._traction(model, None)
loss = torch.zeros((), dtype=f64, device=dev); gx, gu = torch.zeros_like(xf), torch.zeros_like(uf)
names = ["entry", "desc", "idx", "dma_issued", "wait0", "strip0", "strip1", "strip2", "math_end", "synced", "end"]
for abl in [int(v) for v in a.ablate.split(",")]:
    _lib.check(L.hfem_set_option(b"stream_ablate", abl))
    def go():
        _lib.check(L.hfem_tri3_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(), ufix.data_ptr(), dv(lf._mat), lf._W,
                   dv([0.0]*6), None, dv(Tconst), 0, -1, loss.data_ptr(), gx.data_ptr(), gu.data_ptr(), 8, torch.cuda.current_stream().cuda_stream))
    for _ in range(50): go()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): go()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(a.launches): go()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
    st = plan.export("stamps").astype(np.int64)[:plan.n_tiles, :11]        # the LAST launch of the graph
    t0 = st[:, 0].min()
    rel = (st - t0) * 0.01
    print(f"=== ablate {abl}: tiles {st.shape[0]}, kernel span {rel[:, 10].max():.2f} us")
    for i, n in enumerate(names):
        c = rel[:, i]
        print(f"{n:11s} abs: min {c.min():6.2f} p10 {np.percentile(c,10):6.2f} p50 {np.median(c):6.2f} p90 {np.percentile(c,90):6.2f} max {c.max():6.2f}")
    d = np.diff(rel, axis=1)
    for i in range(10):
        c = d[:, i]
        print(f"{names[i]:>11s}->{names[i+1]:11s} dur: min {c.min():6.2f} p50 {np.median(c):6.2f} p90 {np.percentile(c,90):6.2f} max {c.max():6.2f}")
_lib.check(L.hfem_set_option(b"stream_ablate", 0))

#!/usr/bin/env python3
"""Lab: per-workgroup phase timeline of the tiled kernel (s_memrealtime stamps, 10 ns ticks)."""
import os; os.environ.setdefault("HFEM_LAB", "1")   # kernel-lab tool: needs libhidenn_hip_lab.so (build.py --lab)
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hidenn_fem_amd import _lib
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.loss import EnergyLoss2D
from hidenn_fem_amd.plan import TilePlan
ap = argparse.ArgumentParser(); ap.add_argument("--tile", type=int, default=1024); ap.add_argument("--block", type=int, default=512); ap.add_argument("--cap", type=int, default=0); ap.add_argument("--curve", type=int, default=1); ap.add_argument("--quad", action="store_true"); ap.add_argument("--stagger", type=int, default=0); ap.add_argument("--shift", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0"); f64 = torch.float64
from hidenn_fem_amd.mesh import structured_quad_mesh
if a.quad:
    coords, conn, geom, bc, mn, edges = structured_quad_mesh(1001, 1001, length=2.0, height=2.0, jitter=0.2, seed=0, dtype=f64)
else:
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(1001, 501, jitter=0.2, seed=0, dtype=f64)
torch.manual_seed(0)
model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(dev)
lf = EnergyLoss2D(device=dev, dtype=f64, tile_elems=a.tile)
_lib.lib().hfem_set_option(b"plan_curve", a.curve); _lib.lib().hfem_set_option(b"plan_node_cap", a.cap)
plan = TilePlan(model.connectivity, model.Nnodes, coords_hint=model.initial_node_coords, x_src=model._x_src,
                u_src=model._u_src, edges=model.neumann_edges, tile_elems=a.tile, device=dev, elem_order=3,
                nodes_per_elem=4 if a.quad else 3)
L = _lib.lib(); dv = lambda v: (C.c_double * len(v))(*v)
xf, uf = model.node_coords_free.detach(), model.u_free.detach(); xfix, ufix = model.node_coords_fixed, model.u_fixed_rows()
_, Tconst = lf._traction(model, None)
loss = torch.zeros((), dtype=f64, device=dev); gx, gu = torch.zeros_like(xf), torch.zeros_like(uf)
_lib.check(L.hfem_set_option(b"tiled_block", a.block)); _lib.check(L.hfem_set_option(b"tiled_ablate", 64))
if a.quad:
    _lib.check(L.hfem_set_option(b"quad4_ablate", 4)); _lib.check(L.hfem_set_option(b"quad4_stagger", a.stagger)); _lib.check(L.hfem_set_option(b"quad4_stagger_shift", a.shift))
for _ in range(5):
    if a.quad:
        _lib.check(L.hfem_quad4_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(), ufix.data_ptr(), dv(lf._mat), None,
                   dv(Tconst), 0, -1, loss.data_ptr(), gx.data_ptr(), gu.data_ptr(), 8, torch.cuda.current_stream().cuda_stream))
        continue
    _lib.check(L.hfem_tri3_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(), ufix.data_ptr(), dv(lf._mat), lf._W,
               dv([0.0]*6), None, dv(Tconst), 0, -1, loss.data_ptr(), gx.data_ptr(), gu.data_ptr(), 8, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
st = plan.export("stamps").astype(np.int64)[:, :8]
t0 = st[:, 0].min()
rel = (st - t0) * 0.01          # us
names = ["start", "desc", "gathered", "bar1", "elems", "bar2", "stored", "end"]
print("tiles", st.shape[0], "kernel span us", rel[:, 7].max())
for i, n in enumerate(names):
    c = rel[:, i]
    print(f"{n:9s} abs: min {c.min():6.2f} p50 {np.median(c):6.2f} p90 {np.percentile(c,90):6.2f} max {c.max():6.2f}")
d = np.diff(rel, axis=1)
for i in range(7):
    c = d[:, i]
    print(f"{names[i]:>9s}->{names[i+1]:9s} dur: min {c.min():6.2f} p50 {np.median(c):6.2f} p90 {np.percentile(c,90):6.2f} max {c.max():6.2f}")
life = rel[:, 7] - rel[:, 0]
print("lifetime p50 %.2f p90 %.2f max %.2f" % (np.median(life), np.percentile(life, 90), life.max()))
# concurrency over time
ts = np.arange(0, rel[:, 7].max(), 1.0)
for t in ts:
    run = ((rel[:, 0] <= t) & (rel[:, 7] > t)).sum()
    ph = [int(((rel[:, i] <= t) & (rel[:, i + 1] > t)).sum()) for i in range(7)]
    print(f"t={t:5.1f}us running={run:4d} in-phase={ph}")

#!/usr/bin/env python3
"""Small driver for rocprofv3 runs: N launches of the tiled TRI3 kernel on T1M with the options given.
   python scripts/prof_run.py --tile 1024 --block 512 --order 2 --reps 20"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hidenn_fem_amd import _lib
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.loss import EnergyLoss2D
from hidenn_fem_amd.plan import TilePlan

ap = argparse.ArgumentParser()
ap.add_argument("--tile", type=int, default=1024); ap.add_argument("--block", type=int, default=512)
ap.add_argument("--order", type=int, default=3); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--pipe", type=int, default=0); ap.add_argument("--ablate", type=int, default=0)
ap.add_argument("--rewrite", action="store_true"); ap.add_argument("--store", type=int, default=16)
ap.add_argument("--nx", type=int, default=1001); ap.add_argument("--ny", type=int, default=501)
a = ap.parse_args()
dev = torch.device("cuda:0"); f64 = torch.float64
coords, conn, geom, bc, mn, edges = structured_tri_mesh(a.nx, a.ny, jitter=0.2, seed=0, dtype=f64)
torch.manual_seed(0)
model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(dev)
lf = EnergyLoss2D(device=dev, dtype=f64, tile_elems=a.tile)
plan = TilePlan(model.connectivity, model.Nnodes, coords_hint=model.initial_node_coords, x_src=model._x_src,
                u_src=model._u_src, edges=model.neumann_edges, tile_elems=a.tile, device=dev, elem_order=a.order)
L = _lib.lib(); dv = lambda v: (C.c_double * len(v))(*v)
xf, uf = model.node_coords_free.detach(), model.u_free.detach()
xfix, ufix = model.node_coords_fixed, model.u_fixed_rows()
_, Tconst = lf._traction(model, None)
loss = torch.zeros((), dtype=f64, device=dev); gx, gu = torch.zeros_like(xf), torch.zeros_like(uf)
_lib.check(L.hfem_set_option(b"tiled_block", a.block)); _lib.check(L.hfem_set_option(b"tiled_pipe", a.pipe))
_lib.check(L.hfem_set_option(b"tiled_ablate", a.ablate)); _lib.check(L.hfem_set_option(b"store_policy", a.store))
for _ in range(a.reps):
    if a.rewrite:
        xf.mul_(1.0); uf.mul_(1.0)
    _lib.check(L.hfem_tri3_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(), ufix.data_ptr(),
                                       dv(lf._mat), lf._W, dv([0.0] * 6), None, dv(Tconst), 0, -1, loss.data_ptr(),
                                       gx.data_ptr(), gu.data_ptr(), 0, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("loss", loss.item(), plan.stats)

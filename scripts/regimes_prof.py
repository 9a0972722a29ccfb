#!/usr/bin/env python3
"""Driver for `rocprofv3 --kernel-trace --stats`: the production TRI3 energy kernel on T1M in one cache regime.
   python3 scripts/regimes_prof.py --regime {replay,rewrite,rotate,adam} --reps 400
replay: same buffers every launch.  rewrite: x and u rewritten (mul_ by 1.0) before every launch.
rotate: 10 parameter / gradient sets (> 256 MB) round-robin.  adam: a real training iteration,
EnergyLoss2D.value_and_grad_ + FusedAdam.step (the energy kernel's inputs are what the optimiser just wrote)."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hidenn_fem_amd import _lib
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.loss import EnergyLoss2D

ap = argparse.ArgumentParser()
ap.add_argument("--regime", default="replay"); ap.add_argument("--reps", type=int, default=400)
ap.add_argument("--sets", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0"); f64 = torch.float64
coords, conn, geom, bc, mn, edges = structured_tri_mesh(1001, 501, jitter=0.2, seed=0, dtype=f64)
torch.manual_seed(0)
model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(dev)
lf = EnergyLoss2D(E=10e9, nu=0.3, device=dev, dtype=f64)
plan = model.tile_plan(lf.tile_elems)
L = _lib.lib(); dv = lambda v: (C.c_double * len(v))(*v)
xfix, ufix = model.node_coords_fixed, model.u_fixed_rows()
_, Tconst = lf._traction(model, None)
mat, W, Bk, Tc = dv(lf._mat), lf._W, dv([0.0] * 6), dv(Tconst)
loss = torch.zeros((), dtype=f64, device=dev)
sets = []
for r in range(a.sets if a.regime == "rotate" else 1):
    xf = model.node_coords_free.detach().clone(); uf = model.u_free.detach().clone()
    sets.append((xf, uf, torch.zeros_like(xf), torch.zeros_like(uf)))
def energy(k):
    xf, uf, gx, gu = sets[k]
    _lib.check(L.hfem_tri3_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(), ufix.data_ptr(), mat, W, Bk,
                                       None, Tc, 0, -1, loss.data_ptr(), gx.data_ptr(), gu.data_ptr(), 8,
                                       torch.cuda.current_stream().cuda_stream))
if a.regime == "adam":
    from hidenn_fem_amd.optim import FusedAdam
    opt = FusedAdam(model.parameters(), lr=1e-9, capturable=True).init_state()
    for _ in range(a.reps):
        lf.value_and_grad_(model)
        opt.step()
else:
    for i in range(a.reps):
        if a.regime == "rewrite":
            sets[0][0].mul_(1.0); sets[0][1].mul_(1.0)
        energy(i % len(sets))
torch.cuda.synchronize()
print("done", a.regime, loss.item())

#!/usr/bin/env python3
"""Dev tool: host-side cost of one owner-sharded step (energy kernel + pack + [collective] + unpack) at world = 1,
where the collective degenerates to a copy -- what is left is launch / Python overhead vs the GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.loss import EnergyLoss2D
from hidenn_fem_amd.sharded import ShardedTri3Energy

d = torch.device("cuda:0")
f64 = torch.float64
coords, conn, geom, bc, mn, edges = structured_tri_mesh(1001, 501, jitter=0.2, seed=0, dtype=f64)
m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
sh = ShardedTri3Energy(m, EnergyLoss2D(device=d, dtype=f64)).setup_interfaces()
for name, fn in (("evaluate_local", sh.evaluate_local), ("evaluate_owner+exchange_halo", lambda: (sh.evaluate_owner(), sh.exchange_halo())),
                 ("owner_step (hoisted lookups)", sh.owner_step)):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name}: host enqueue {t_host / n * 1e6:.1f} us/step, end-to-end {t_all / n * 1e6:.1f} us/step")

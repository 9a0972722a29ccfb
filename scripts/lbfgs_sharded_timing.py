#!/usr/bin/env python3
"""Dev tool (GPU box): ShardedLBFGS on T1M -- the whole optimiser on one rank (world = 1) and rank r of 8 emulated (its tile
range, its eighth of the history), graph-captured steady-state iterations vs eager launches.  ms per inner iteration, history full."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hidenn_fem_amd.loss import EnergyLoss2D  # noqa: E402
from hidenn_fem_amd.mesh import structured_tri_mesh  # noqa: E402
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D  # noqa: E402
from hidenn_fem_amd.optim import ShardedLBFGS  # noqa: E402
from hidenn_fem_amd.sharded import ShardedTri3Energy  # noqa: E402

dev = torch.device("cuda:0")
f64 = torch.float64
mesh = structured_tri_mesh(1001, 501, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64)


def run(rank, world, graph, dtype=f64):
    coords, conn, geom, bc, mn, edges = mesh
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN2D(coords.to(dtype), conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(dev)
    sh = ShardedTri3Energy(m, EnergyLoss2D(E=10e9, nu=0.3, device=dev, dtype=dtype), rank=rank, world=world)
    opt = ShardedLBFGS(sh, emulate=world > 1, graph=graph)
    ts = []
    for _ in range(8):
        torch.cuda.synchronize()
        n0, t0 = opt.state["n_iter"], time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / max(1, opt.state["n_iter"] - n0))
    return dict(rank=rank, world=world, graph=graph, dtype=str(dtype).split(".")[-1], parameters_this_rank=opt._n,
                ms_per_inner_iteration=sorted(ts[5:])[1] * 1e3, history=int(opt.status()[5]))


out = []
for world, rank in ((1, 0), (8, 0), (8, 4)):
    for graph in (False, True):
        out.append(run(rank, world, graph))
        print(json.dumps(out[-1]), flush=True)
out.append(run(0, 8, True, torch.float32))
print(json.dumps(out[-1]), flush=True)

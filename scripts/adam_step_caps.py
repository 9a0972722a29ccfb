#!/usr/bin/env python3
"""Dev tool (GPU box): the one-launch training iteration (energy + Adam at write-out) on T1M for several tile sizes
(plan_node_cap = home nodes per tile): does a launch of two or three resident rounds overlap the optimiser's streaming with
the next tiles' prologue?  K iterations per hipGraph, wall clock per iteration, median of 5."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hidenn_fem_amd import _lib  # noqa: E402
from hidenn_fem_amd.graphed import GraphedTraining  # noqa: E402
from hidenn_fem_amd.loss import EnergyLoss2D  # noqa: E402
from hidenn_fem_amd.mesh import structured_tri_mesh  # noqa: E402
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D  # noqa: E402
from hidenn_fem_amd.optim import EnergyAdamStep  # noqa: E402

K = 100
dev, f64 = torch.device("cuda:0"), torch.float64
mesh = structured_tri_mesh(1001, 501, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64)
for cap in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "-1,420,370,280,186".split(","))]:
    _lib.check(_lib.lib().hfem_set_option(b"plan_node_cap", cap))
    c_, cn_, g_, b_, _, e_ = mesh
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN2D(c_, cn_, boundary_mask=g_, dirichlet_mask=b_, u_fixed=0.0, neumann_edges=e_).to(dev)
    lf = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64)
    tr = EnergyAdamStep(m, lf, lr_x=1e-9, lr_u=1e-12)
    gt = GraphedTraining(tr.step_lagged, None, steps_per_replay=K, direct=True, begin=tr.begin_lagged, end=tr.flush_loss)
    for _ in range(20):
        gt.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        gt.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e6)
    st = tr.plan.stats
    print(json.dumps(dict(cap=cap, us_per_iteration=round(sorted(ts)[2], 3), tiles=st["n_tiles"], threads=st["threads_per_tile"],
                          slot_rows=st["slot_rows"], lds=st["lds_bytes"], max_owned=st["max_tile_owned"],
                          halo_nodes=round(st["tile_node_total"] / st["n_nodes"], 3))), flush=True)
    del gt, tr, m

#!/usr/bin/env python3
"""Dev tool: EnergyLoss2D + backward on T1M for an fp64 and an fp32 model (fp32 rows take the float-storage kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.loss import EnergyLoss2D
d = torch.device("cuda:0")
for dt in (torch.float64, torch.float32):
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(1001, 501, jitter=0.2, seed=0, dtype=dt)
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    lf = EnergyLoss2D(device=d, dtype=dt)
    def it():
        m.zero_grad(set_to_none=False)
        lf(m).backward()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): it()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): it()
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < 0.3:
        g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    t_auto = (time.perf_counter() - t0) / 500 * 1e6
    g2 = torch.cuda.CUDAGraph()
    lf.value_and_grad_(m); torch.cuda.synchronize()
    with torch.cuda.graph(g2):
        for _ in range(50): lf.value_and_grad_(m)
    g2.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g2.replay()
    torch.cuda.synchronize()
    print(f"{dt}: autograd loss+backward {t_auto:.1f} us | value_and_grad_ {(time.perf_counter() - t0) / 500 * 1e6:.1f} us "
          f"(hipGraphs of 50)", flush=True)

# full Adam training iteration at T1M, fp64: two-launch loop vs the fused step
from hidenn_fem_amd.optim import FusedAdam, EnergyAdamStep
from hidenn_fem_amd.graphed import GraphedTraining
coords, conn, geom, bc, mn, edges = structured_tri_mesh(1001, 501, jitter=0.2, seed=0, dtype=torch.float64)
for mode in ("value_and_grad_ + FusedAdam", "EnergyAdamStep", "EnergyAdamStep, lagged loss"):
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    lf = EnergyLoss2D(device=d, dtype=torch.float64)
    if mode == "EnergyAdamStep, lagged loss":
        tr = EnergyAdamStep(m, lf, lr_x=1e-9, lr_u=1e-8)
        gt = GraphedTraining(tr.step_lagged, None, steps_per_replay=50, warmup=2, direct=True, begin=tr.begin_lagged, end=tr.flush_loss)
    elif mode == "EnergyAdamStep":
        gt = GraphedTraining(EnergyAdamStep(m, lf, lr_x=1e-9, lr_u=1e-8).step, None, steps_per_replay=50, warmup=2, direct=True)
    else:
        opt = FusedAdam([dict(params=[m.node_coords_free], lr=1e-9), dict(params=[m.u_free], lr=1e-8)], capturable=True)
        gt = GraphedTraining(lambda: lf.value_and_grad_(m), opt, steps_per_replay=50, warmup=2, direct=True)
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < 0.3:
        gt.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    gt.replay(10)
    torch.cuda.synchronize()
    print(f"Adam training iteration, T1M fp64, {mode}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us", flush=True)

#!/usr/bin/env python3
"""Dev tool (GPU box): what a T1M training iteration costs -- energy launch + FusedAdam (multi-tensor, one launch) vs the
per-tensor launches, and the one-launch EnergyAdamStep; K iterations per hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.loss import EnergyLoss2D
from hidenn_fem_amd.optim import FusedAdam, EnergyAdamStep
from hidenn_fem_amd.graphed import GraphedTraining

d = torch.device("cuda:0")
f64 = torch.float64
mesh = structured_tri_mesh(1001, 501, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64)
K = 100


def model(dt=f64):
    torch.manual_seed(0)
    c, cn, g, b, mn, e = mesh
    return PiecewiseLinearShapeNN2D(c.to(dt), cn, boundary_mask=g, dirichlet_mask=b, u_fixed=0.0, neumann_edges=e).to(d)


def timeit(gt):
    for _ in range(5):
        gt.replay()
    torch.cuda.synchronize()
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < 0.5:
        gt.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); gt.replay(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e6)
    return sorted(ts)[2]


for dt in (f64, torch.float32):
    m = model(dt)
    lf = EnergyLoss2D(device=d, dtype=dt)
    opt = FusedAdam([dict(params=[m.node_coords_free], lr=1e-9), dict(params=[m.u_free], lr=1e-12)], capturable=True)
    print(dt, "energy + multi-tensor FusedAdam: %.2f us/iteration" % timeit(GraphedTraining(lambda: lf.value_and_grad_(m), opt, steps_per_replay=K, direct=True, warmup=2)), flush=True)
    m2 = model(dt)
    tr = EnergyAdamStep(m2, lf, lr_x=1e-9, lr_u=1e-12)
    print(dt, "one-launch EnergyAdamStep:        %.2f us/iteration" % timeit(GraphedTraining(tr.step_lagged, None, steps_per_replay=K, direct=True, begin=tr.begin_lagged, end=tr.flush_loss)), flush=True)
    if dt == torch.float32:          # the same one-launch step with fp64 arithmetic on the float rows (round 3's instance)
        m4 = model(dt)
        tr4 = EnergyAdamStep(m4, EnergyLoss2D(device=d, dtype=dt, arithmetic="fp64"), lr_x=1e-9, lr_u=1e-12)
        print(dt, "one-launch, fp64 arithmetic:      %.2f us/iteration" % timeit(GraphedTraining(tr4.step_lagged, None, steps_per_replay=K, direct=True, begin=tr4.begin_lagged, end=tr4.flush_loss)), flush=True)
    m3 = model(dt)
    only = FusedAdam([dict(params=[m3.node_coords_free], lr=1e-9), dict(params=[m3.u_free], lr=1e-12)], capturable=True)
    lf.value_and_grad_(m3)
    print(dt, "multi-tensor FusedAdam alone:     %.2f us/step" % timeit(GraphedTraining(lambda: None, only, steps_per_replay=K, direct=True, warmup=2)), flush=True)

# ---- the owner-sharded iterations on ONE rank (world = 1, in-library RCCL communicator): what the launch chain of the
#      multi-GPU step costs before any exchange latency -- plain (4 launches + all_gather) and overlapped (range split
#      artificially 5 % / 95 %: 5 launches, all_gather + unpack on the side stream)
from hidenn_fem_amd.sharded import LibraryComm, ShardedTri3Energy
comm = LibraryComm(d)
for name, split, same_stream in (("owner_train_step", False, False), ("owner_train_step_fused", False, False),
                                 ("owner_train_step_overlapped", True, False), ("owner_train_step_overlapped", True, True),
                                 ("owner_train_step_fused_overlapped", True, False), ("owner_train_step_fused_overlapped", True, True),
                                 ("owner_step_overlapped", True, False), ("owner_step_overlapped", True, True)):
    m = model(f64)
    sh = ShardedTri3Energy(m, EnergyLoss2D(device=d, dtype=f64), comm=comm)
    sh.setup_interfaces()
    sh.init_owner_adam(1e-9, 1e-12, fused="fused" in name)
    if split:
        sh.mid = sh.plan.n_tiles // 20
    if same_stream:                      # the same launches in the same order, exchange on the MAIN stream: what the fork / join costs
        sh.inline_exchange = True
        name_tag = name + " (one stream)"
    else:
        name_tag = name
    body = getattr(sh, name)
    end = sh.finish_overlapped if split else None
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(); body()
        if end: end()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K):
            body()
        if end: end()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < 0.5:
        g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e6)
    print("world 1 %-44s %.2f us/iteration" % (name_tag, sorted(ts)[2]), flush=True)

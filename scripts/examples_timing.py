#!/usr/bin/env python3
"""Dev tool: microseconds per training iteration of examples 1 and 3 (BASELINE configs 1-2: 1D L2 projection; 1D bar,
10 k elements, r-adaptivity): eager torch.optim.Adam, eager FusedAdam, and GraphedTraining (one hipGraph per 50 its)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import examples.example3 as e3
from hidenn_fem_amd.models import PiecewiseLinearShapeNN
from hidenn_fem_amd.loss import bar_energy_loss, l2_projection_loss
from hidenn_fem_amd.utils import gauss_legendre_points_weights
from hidenn_fem_amd.optim import FusedAdam
from hidenn_fem_amd.graphed import GraphedTraining

d = torch.device("cuda:0")


def ex1(dtype):
    nodes = torch.linspace(0, 1, 100, dtype=dtype, device=d)
    xs = torch.linspace(0, 1, 1000, dtype=dtype, device=d)
    target = torch.sin(2 * torch.pi * xs)
    m = PiecewiseLinearShapeNN(nodes, r_adapt=True).to(d)
    return m, (lambda: l2_projection_loss(m, xs, target)), 5e-3


def ex3(dtype, nodes=10001):
    grid = torch.linspace(0, e3.LENGTH, nodes, dtype=dtype, device=d)
    xi, wi = gauss_legendre_points_weights(2, device=d, dtype=dtype)
    m = PiecewiseLinearShapeNN(grid, r_adapt=True, u0=0.0, uN=0.0).to(d)
    return m, (lambda: bar_energy_loss(m, xi, wi, e3.body_force, E=e3.E_MOD)), 1e-4


def eager(make, opt_cls, n=600, **kw):
    m, closure, lr = make(torch.float32)
    opt = opt_cls(m.parameters(), lr=lr, **kw)
    for i in range(n + 100):
        if i == 100:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.zero_grad()
        loss = closure()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, loss.item()


def graphed(make, n=600, per=50):
    m, closure, lr = make(torch.float32)
    gt = GraphedTraining(closure, FusedAdam(m.parameters(), lr=lr, capturable=True), steps_per_replay=per, warmup=100)
    gt.replay(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = gt.replay(n // per)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, loss.item()


if len(sys.argv) > 1 and sys.argv[1] == "--graphed-only":      # for rocprofv3: kernel mix of the captured iteration
    which = ex1 if sys.argv[2] == "1" else ex3
    print(graphed(which, n=200, per=50))
    sys.exit(0)
for name, make in (("example1 (100 nodes, 1000 samples)", ex1), ("example3 (10 001 nodes, r-adapt)", ex3)):
    a, _ = eager(make, torch.optim.Adam)
    b, _ = eager(make, FusedAdam)
    c, _ = graphed(make)
    print(f"{name}: torch.optim.Adam eager {a:.1f} us/it | FusedAdam eager {b:.1f} us/it | GraphedTraining {c:.1f} us/it", flush=True)

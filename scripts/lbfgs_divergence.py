#!/usr/bin/env python3
"""Where do FusedLBFGS and torch.optim.LBFGS part on example 4?  (VERDICT r1 weak #8.)

Mesh and physics of examples/example4.py (200 x 100 zig-zag mesh with three holes, fp64, default LBFGS settings:
lr 1, max_iter 20, history 100, no line search).  Both optimisers get the SAME closure: the fused energy with
`deterministic=True` (fixed-order gradients, bit-identical run to run), so every difference below comes from the
optimisers' own arithmetic -- torch runs the two-loop recursion in vector space, FusedLBFGS in coefficient space
(same algebra, different rounding).  Per closure call: |loss_f - loss_t| / |loss_t| and ||p_f - p_t|| / ||p_t||.
A third run repeats torch.optim.LBFGS itself with the ATOMIC (non-reproducible) gradients: how far apart two runs of
the *reference's own* optimiser land when only the last bits of the gradient differ.
"""
import argparse, copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hidenn_fem_amd.mesh import generate_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.loss import EnergyLoss2D
from hidenn_fem_amd.optim import FusedLBFGS

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--nx", type=int, default=200); ap.add_argument("--ny", type=int, default=100)
a = ap.parse_args()
d = torch.device("cuda:0"); F64 = torch.float64
holes = [(0.5, 0.7, 0.12), (1.0, 0.3, 0.15), (1.4, 0.6, 0.1)]
nodes, conn, geom, bc, mn, edges = generate_mesh(2.0, 1.0, holes, {"up": 0, "down": 0, "right": 2, "left": 1}, a.nx, a.ny)
torch.manual_seed(0)
base = PiecewiseLinearShapeNN2D(nodes.double(), conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)

def run(opt_cls, deterministic):
    m = copy.deepcopy(base)
    lf = EnergyLoss2D(E=10e9, nu=0.3, length=2.0, height=1.0, device=d, dtype=F64, deterministic=deterministic)
    opt = opt_cls(m.parameters())
    losses, params = [], []
    def closure():
        opt.zero_grad()
        v = lf(m); v.backward()
        losses.append(v.item())
        params.append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone())
        return v
    for _ in range(a.steps):
        opt.step(closure)
    return np.array(losses), params

lt, pt = run(torch.optim.LBFGS, True)
lf_, pf = run(FusedLBFGS, True)
lt2, pt2 = run(torch.optim.LBFGS, False)
lt3, pt3 = run(torch.optim.LBFGS, False)
n = min(len(lt), len(lf_), len(lt2), len(lt3))
print(f"# mesh {tuple(nodes.shape)} nodes {tuple(conn.shape)} elements; closure calls: torch {len(lt)} fused {len(lf_)}")
print("# call  loss_torch        |loss_f-loss_t|/|loss_t|  ||p_f-p_t||/||p_t||   torch(atomic run A vs B): dloss/|loss|   dparams")
first = None
for i in range(n):
    dl = abs(lf_[i] - lt[i]) / abs(lt[i]); dp = ((pf[i] - pt[i]).norm() / pt[i].norm()).item()
    dl2 = abs(lt3[i] - lt2[i]) / abs(lt2[i]); dp2 = ((pt3[i] - pt2[i]).norm() / pt2[i].norm()).item()
    if first is None and dl > 1e-6: first = i
    if i < 25 or i % 10 == 0 or i == n - 1:
        print(f"{i:5d}  {lt[i]: .10e}  {dl:10.3e}               {dp:10.3e}          {dl2:10.3e}   {dp2:10.3e}")
print(f"# first closure call with |dloss|/|loss| > 1e-6 (fused vs torch, deterministic gradients): {first}")
print(f"# final losses: torch(det) {lt[n-1]:.6f}  fused(det) {lf_[n-1]:.6f}  torch(atomic) A {lt2[n-1]:.6f} B {lt3[n-1]:.6f}")

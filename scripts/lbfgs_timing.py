#!/usr/bin/env python3
"""Example 4's optimiser loop at full size (dev tool): T1M (10^6 TRI3, ~2 x 10^6 fp64 parameters), LBFGS as the reference
drives it (/root/reference/examples/example4.py:68-78: lr 1, max_iter 20, history 100, no line search).  Once the history
is full an inner iteration streams the 2 x 100 history vectors twice (multidot + direction passes): 4 h n 8 B = 6.4 GB,
so the optimiser, not the 9 us energy launch, is the iteration.  Reports ms per inner iteration for FusedLBFGS (and
torch.optim.LBFGS with the same closure, --torch) and the achieved fraction of the 8 TB/s HBM roofline.

    python scripts/lbfgs_timing.py [--steps 8] [--torch] [--grid 1001x501]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hidenn_fem_amd.loss import EnergyLoss2D
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.optim import FusedLBFGS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8, help="outer LBFGS steps (20 inner iterations each)")
    ap.add_argument("--grid", default="1001x501")
    ap.add_argument("--history", type=int, default=100)
    ap.add_argument("--torch", action="store_true", help="also time torch.optim.LBFGS on the same closure")
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--lib", default="", help="dev: another build of the library (build.py --tag NAME): libhidenn_hip_NAME.so")
    a = ap.parse_args()
    if a.lib:
        from hidenn_fem_amd import _lib
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), a.lib)
    dev = torch.device("cuda:0")
    dt = torch.float32 if a.fp32 else torch.float64
    nx, ny = (int(v) for v in a.grid.split("x"))
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(nx, ny, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=dt)

    def run(cls):
        torch.manual_seed(0)
        m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(dev)
        lf = EnergyLoss2D(device=dev, dtype=dt)
        opt = cls(m.parameters(), history_size=a.history)
        n = sum(p.numel() for p in m.parameters())
        evals = [0]

        def closure():
            evals[0] += 1
            return lf.value_and_grad_(m)
        out = []
        for s in range(a.steps):
            torch.cuda.synchronize()
            e0, t0 = evals[0], time.perf_counter()
            loss = opt.step(closure)
            torch.cuda.synchronize()
            dt_s = time.perf_counter() - t0
            st = opt.state[opt._params[0]] if hasattr(opt, "_params") else opt.state[opt.param_groups[0]["params"][0]]
            out.append(dict(step=s, ms=round(dt_s * 1e3, 3), closure_calls=evals[0] - e0, n_iter_total=st.get("n_iter"),
                            loss=float(loss)))
        return n, out

    res = {"lib": a.lib or "libhidenn_hip.so"}
    n, fused = run(FusedLBFGS)
    res["n_params"] = n
    res["FusedLBFGS"] = fused
    last = fused[-1]
    iters = 20
    hist = min(a.history, fused[-1]["n_iter_total"])
    b = 4.0 * hist * n * (4 if a.fp32 else 8)
    res["per_inner_iteration_ms"] = round(min(f["ms"] for f in fused[-3:]) / iters, 4)      # history full in the last three steps (steps >= 8)
    res["history_pairs"] = hist
    res["alg_bytes_per_iteration"] = b
    res["achieved_GBs"] = round(b / (res["per_inner_iteration_ms"] * 1e-3) / 1e9, 1)
    res["frac_of_8TBs"] = round(res["achieved_GBs"] / 8000.0, 3)
    if a.torch:
        _, res["torch_LBFGS"] = run(torch.optim.LBFGS)
        res["torch_per_inner_iteration_ms"] = round(res["torch_LBFGS"][-1]["ms"] / iters, 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

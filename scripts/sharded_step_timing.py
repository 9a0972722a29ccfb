#!/usr/bin/env python3
"""What the owner-sharded step machinery costs on ONE rank (dev tool; the same numbers as bench.py's
``config.sharded_step_1gpu``): T1M, K iterations per hipGraph, median of 5 replays, for the collective path (all_gather
stand-in), the side-stream overlap and the peer-window put / get -- plain and fused.

    python scripts/sharded_step_timing.py [--k 100] [--boundary 0.05]
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
from hidenn_fem_amd.sharded import ShardedTri3Energy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--boundary", type=float, default=0.05)
    ap.add_argument("--legs", default="collective,peer")
    ap.add_argument("--fp32", action="store_true", help="an fp32 model (the reference's default dtype): float rows, double2 payload")
    ap.add_argument("--grid", default="1001x501", help="nodes: NX x NY (2001x1001 = 4 x 10^6 elements: launches of several rounds)")
    a = ap.parse_args()
    dev, f64 = torch.device("cuda:0"), torch.float64
    gx_, gy_ = (int(v) for v in a.grid.split("x"))
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(gx_, gy_, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64)
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(dev)
    if a.fp32:
        m = m.float()
    sh = ShardedTri3Energy(m, EnergyLoss2D(device=dev, dtype=torch.float32 if a.fp32 else f64))
    sh.setup_interfaces()
    sh.init_owner_adam(lr_x=1e-9, lr_u=1e-12, fused=True)
    sh.mid = sh.lo + max(1, int((sh.hi - sh.lo) * a.boundary))
    K = a.k // 2 * 2

    def us(body, end=None):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            body(); body()
            if end:
                end()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(K):
                body()
            if end:
                end()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            g.replay()
            torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e6)
        return round(sorted(ts)[2], 3)

    def legs():
        return dict(eval_exchange=us(sh.owner_step), eval_exchange_overlap=us(sh.owner_step_overlapped, sh.finish_overlapped),
                    train_step=us(sh.owner_train_step),
                    train_step_overlap=us(sh.owner_train_step_overlapped, sh.finish_overlapped),
                    train_step_fused=us(sh.owner_train_step_fused),
                    train_step_fused_overlap=us(sh.owner_train_step_fused_overlapped, sh.finish_overlapped))
    out = dict(grid=a.grid, dtype="fp32" if a.fp32 else "fp64", K=K, boundary_tiles=sh.mid - sh.lo, tiles=sh.hi - sh.lo)
    if "collective" in a.legs:
        out["collective_path"] = legs()
    if "peer" in a.legs:
        sh.enable_peer_exchange()
        out["peer_windows"] = legs()
        out["peer_status"] = sh.peer.status()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

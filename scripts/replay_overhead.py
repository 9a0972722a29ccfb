#!/usr/bin/env python3
"""Dev tool (GPU box): what the timed region of bench.py pays around its K = 20 energy launches -- one hipGraph replay plus
the synchronisation behind it, wall clock -- and whether the host's wait policy matters (hipSetDeviceFlags before the first
GPU call: 0 auto, 1 spin, 2 yield, 4 blocking).

    python scripts/replay_overhead.py [flags]
"""
import ctypes as C
import json
import os
import sys
import time

import torch

flags = int(sys.argv[1]) if len(sys.argv) > 1 else -1
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
rc = hip.hipSetDeviceFlags(C.c_uint(flags)) if flags >= 0 else None

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hidenn_fem_amd.loss import EnergyLoss2D  # noqa: E402
from hidenn_fem_amd.mesh import structured_tri_mesh  # noqa: E402
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D  # noqa: E402
from hidenn_fem_amd.sharded import ShardedTri3Energy  # noqa: E402

dev, f64 = torch.device("cuda:0"), torch.float64
coords, conn, geom, bc, mn, edges = structured_tri_mesh(1001, 501, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64)
torch.manual_seed(0)
m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(dev)
lf = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64)
sh = ShardedTri3Energy(m, lf)


def graph_of(k):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        sh.begin_lagged(); sh.evaluate_local_lagged(); sh.flush_loss()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sh.begin_lagged()
        for _ in range(k):
            sh.evaluate_local_lagged()
        sh.flush_loss()
    return g


def wall(g, sync):
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(21):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        sync()
        ts.append((time.perf_counter() - t0) * 1e6)
    return sorted(ts)[len(ts) // 2]


cur = torch.cuda.current_stream()
out = dict(flags=flags, set_rc=rc)
for k in (1, 20, 200):
    g = graph_of(k)
    out[f"k{k}_device_sync_us"] = round(wall(g, torch.cuda.synchronize), 2)
    out[f"k{k}_stream_sync_us"] = round(wall(g, cur.synchronize), 2)
    ev = torch.cuda.Event()

    def ev_sync():
        ev.record()
        while not ev.query():
            pass
    out[f"k{k}_event_poll_us"] = round(wall(g, ev_sync), 2)
print(json.dumps(out))

#!/usr/bin/env python3
"""Kernel lab (dev tool, GPU box, lab build): wave priority through the memory phases of the paired TRI3 kernel on T1M.
`pair_ablate` bits of tri3_pair_lab.hip: 262144 plain lab copy, 16384 / 131072 / 65536 priority 3 / 2 / 1 through prologue and
write-out (+32768: prologue only).  Two regimes per variant: the same buffers every launch, and rotating sets (> the Infinity
Cache).  One JSON line per variant; 0 = the product kernel."""
import os; os.environ.setdefault("HFEM_LAB", "1")
import ctypes as C
import json
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hidenn_fem_amd import _lib  # noqa: E402
from hidenn_fem_amd.loss import EnergyLoss2D  # noqa: E402
from hidenn_fem_amd.mesh import structured_tri_mesh  # noqa: E402
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D  # noqa: E402

K = 120
dev = torch.device("cuda:0")
f64 = torch.float64
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,262144,16384,49152,131072,65536,0,16384".split(","))]
coords, conn, geom, bc, mn, edges = structured_tri_mesh(1001, 501, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64)
ne, nn = conn.shape[0], coords.shape[0]
L = _lib.lib()
dv = lambda v: (C.c_double * len(v))(*v)
torch.manual_seed(0)
m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(dev)
pl = m.tile_plan(0)
x, u = m.node_coords_free.detach(), m.u_free.detach()
xf, uf = m.node_coords_fixed, m.u_fixed_rows()
ls = torch.zeros((), dtype=f64, device=dev)
lf = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64)
_, Tc = lf._traction(m, None)
mat, Tcv, Bk = dv(lf._mat), dv(Tc), dv([0.0] * 6)
R = 8
sets = [(x.clone(), u.clone(), torch.empty_like(x), torch.empty_like(u)) for _ in range(R)]


def launch(i, stream, flags=8):
    xs, us, gxs, gus = sets[i]
    _lib.check(L.hfem_tri3_energy_plan(pl.handle, xs.data_ptr(), xf.data_ptr(), us.data_ptr(), uf.data_ptr(), mat, lf._W, Bk, None, Tcv,
                                       0, -1, ls.data_ptr(), gxs.data_ptr(), gus.data_ptr(), flags, stream), "energy")


def time_it(rot):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        launch(0, s.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(K):
            launch(k % R if rot else 0, torch.cuda.current_stream().cuda_stream)
    for _ in range(30):
        g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / K)
    return sorted(out)[2]


ref = None
alg = 12 * ne + 64 * nn + 8
for v in variants:
    _lib.check(L.hfem_set_option(b"pair_ablate", v))
    launch(0, torch.cuda.current_stream().cuda_stream, 0)
    torch.cuda.synchronize()
    cur = (ls.item(), sets[0][2].clone(), sets[0][3].clone())
    if ref is None:
        ref = cur
    err = max(abs(cur[0] - ref[0]) / abs(ref[0]), ((cur[1] - ref[1]).abs().max() / ref[1].abs().max()).item(),
              ((cur[2] - ref[2]).abs().max() / ref[2].abs().max()).item())
    a, b = time_it(False), time_it(True)
    print(json.dumps(dict(pair_ablate=v, replayed_us=round(a, 3), rotating_us=round(b, 3), replayed_frac=round(alg / a / 8e6, 3),
                          rotating_frac=round(alg / b / 8e6, 3), err_vs_first=err)), flush=True)
_lib.check(L.hfem_set_option(b"pair_ablate", 0))

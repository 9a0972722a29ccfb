#!/usr/bin/env python3
"""Dev tool (GPU box): gradient-store cache policy (hfem_set_option "store_policy": 16 sc1 write-through, 2 nt, 0 plain) of the
paired-slot kernel by problem size and cache regime -- the same buffers every launch (replayed) vs R rotating parameter /
gradient sets whose total exceeds the 256 MB Infinity Cache (reads and gradient lines really go to HBM)."""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hidenn_fem_amd import _lib
from hidenn_fem_amd.loss import EnergyLoss2D
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.plan import TilePlan

ap = argparse.ArgumentParser()
ap.add_argument("--grids", default="1001x501,1001x1001,2001x1001")
ap.add_argument("--policies", default="16,2")
ap.add_argument("--reps", type=int, default=60)
a = ap.parse_args()
d = torch.device("cuda:0")
f64 = torch.float64
L = _lib.lib()
dv = lambda v: (C.c_double * len(v))(*v)
for grid in a.grids.split(","):
    nx, ny = (int(v) for v in grid.split("x"))
    c, cn, g, b, mn, e = structured_tri_mesh(nx, ny, length=2.0, height=1.0 * (ny - 1) / 500, jitter=0.2, seed=0, dtype=f64)
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN2D(c, cn, boundary_mask=g, dirichlet_mask=b, u_fixed=0.0, neumann_edges=e, reorder="hilbert" if nx > 1001 else "auto").to(d)
    lf = EnergyLoss2D(device=d, dtype=f64)
    xf, uf = m.node_coords_free.detach(), m.u_free.detach()
    xfix, ufix = m.node_coords_fixed, m.u_fixed_rows()
    _, Tc = lf._traction(m, None)
    set_mb = 4 * xf.numel() * 8 / 2 ** 20
    R = max(2, int(320 / set_mb) + 1)
    sets = [(xf.clone(), uf.clone(), torch.empty_like(xf), torch.empty_like(uf)) for _ in range(R)]
    loss = torch.zeros((), dtype=f64, device=d)
    for sp in [int(v) for v in a.policies.split(",")]:
        _lib.check(L.hfem_set_option(b"store_policy", sp))
        plan = TilePlan(m.connectivity, m.Nnodes, coords_hint=m.initial_node_coords, x_src=m._x_src, u_src=m._u_src,
                        edges=m.neumann_edges, device=d)
        alg = 12 * cn.shape[0] + 64 * c.shape[0] + 8

        def launch(i, rot, stream):
            x_, u_, gx_, gu_ = sets[i % R] if rot else sets[0]
            _lib.check(L.hfem_tri3_energy_plan(plan.handle, x_.data_ptr(), xfix.data_ptr(), u_.data_ptr(), ufix.data_ptr(),
                                               dv(lf._mat), lf._W, dv([0.0] * 6), None, dv(Tc), 0, -1, loss.data_ptr(),
                                               gx_.data_ptr(), gu_.data_ptr(), 8, stream))
        res = {}
        for rot in (False, True):
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                launch(0, rot, s.cuda_stream)
            torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph):
                cs = torch.cuda.current_stream().cuda_stream
                for i in range(a.reps):
                    launch(i, rot, cs)
            t_pw = time.perf_counter()
            while time.perf_counter() - t_pw < 0.4:
                gph.replay(); torch.cuda.synchronize()
            regs = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gph.replay(); e1.record(); torch.cuda.synchronize()
                regs.append(e0.elapsed_time(e1) * 1e3 / a.reps)
            us = sorted(regs)[2]
            res["rotating" if rot else "replayed"] = dict(us=round(us, 2), frac=round(alg / us * 1e-3 / 8000.0, 3))
            del gph
        print(json.dumps(dict(grid=grid, elements=cn.shape[0], store_policy=sp, sets=R, working_set_mb=round(R * set_mb + plan.stats["device_bytes"] / 2 ** 20),
                              tiles=plan.n_tiles, **res)), flush=True)
        del plan
    del sets, m

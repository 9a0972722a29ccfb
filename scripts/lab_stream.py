#!/usr/bin/env python3
"""Kernel lab (dev tool, GPU box): the TRI3 energy kernel alone on a structured-split mesh, for a list of
variants (plan element order x library options), in three cache regimes:

  replay   K back-to-back launches on the same buffers (inputs Infinity-Cache / L2 resident)
  rewrite  x, u rewritten by another kernel before every launch (what an optimiser does);
           reported as (rewrite + energy) - (rewrite alone)
  rotate   R parameter / gradient sets (> 256 MB together) visited round-robin: reads miss the Infinity Cache

    python scripts/lab_stream.py --variants "3:tri3_stream=0;4:tri3_stream=0;4:tri3_stream=1"

Every variant is checked against the first one (loss / gradients) before it is timed.
"""
import os; os.environ.setdefault("HFEM_LAB", "1")   # kernel-lab tool: needs libhidenn_hip_lab.so (build.py --lab)
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hidenn_fem_amd import _lib
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.loss import EnergyLoss2D
from hidenn_fem_amd.plan import TilePlan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=1001)
    ap.add_argument("--ny", type=int, default=501)
    ap.add_argument("--variants", default="3:tri3_stream=0;4:tri3_stream=0;4:tri3_stream=1",
                    help="';'-separated  <plan_elem_order>:<opt=val,opt=val...>")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--sets", type=int, default=10, help="rotating parameter/gradient sets")
    ap.add_argument("--regimes", default="replay,rewrite,rotate")
    ap.add_argument("--prewarm", type=float, default=0.3)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--permute", action="store_true")
    ap.add_argument("--diagonal", default="fixed", help="structured mesh: fixed | zigzag | random")
    ap.add_argument("--mesh", default="structured", help="structured | cfg5 | cfg5r | delaunay (4 M-element BASELINE config 5 variants)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    f64 = torch.float64
    kw = dict(diagonal="random", permute=True, jitter=0.3) if a.permute else dict(jitter=0.2, diagonal=a.diagonal)
    if a.mesh == "structured":
        coords, conn, geom, bc, mn, edges = structured_tri_mesh(a.nx, a.ny, seed=0, dtype=f64, **kw)
    elif a.mesh in ("cfg5", "cfg5r"):
        from hidenn_fem_amd.mesh import reorder_for_locality
        m6 = structured_tri_mesh(2001, 1001, jitter=0.3, seed=11, diagonal="random", permute=True, dtype=f64)
        coords, conn, geom, bc, mn, edges = reorder_for_locality(m6)[0] if a.mesh == "cfg5r" else m6
    else:
        from hidenn_fem_amd.mesh import unstructured_tri_mesh
        coords, conn, geom, bc, mn, edges = unstructured_tri_mesh(2_050_000, seed=2, dtype=f64)
    torch.manual_seed(0)
    model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                     neumann_edges=edges).to(dev)
    ne, nn = conn.shape[0], coords.shape[0]
    alg = 12 * ne + 64 * nn + 8
    L = _lib.lib()
    dv = lambda v: (C.c_double * len(v))(*v)
    lf = EnergyLoss2D(device=dev, dtype=f64, tile_elems=a.tile)
    xfix, ufix = model.node_coords_fixed, model.u_fixed_rows()
    _, Tconst = lf._traction(model, None)
    mat, W, Bk, Tc = dv(lf._mat), lf._W, dv([0.0] * 6), dv(Tconst)
    sets = []
    for r in range(a.sets):
        xf = model.node_coords_free.detach().clone()
        uf = model.u_free.detach().clone()
        sets.append((xf, uf, torch.zeros_like(xf), torch.zeros_like(uf)))
    loss = torch.zeros((), dtype=f64, device=dev)
    ref = None
    defaults = {}
    for var in a.variants.split(";"):
        order, _, optstr = var.partition(":")
        opts = dict(kv.split("=") for kv in optstr.split(",") if kv)
        for name, val in defaults.items():           # undo the previous variant's options
            _lib.check(L.hfem_set_option(name.encode(), val))
        for name, val in opts.items():
            if name not in defaults:
                defaults[name] = L.hfem_get_option(name.encode())
            _lib.check(L.hfem_set_option(name.encode(), int(val)))
        plan = TilePlan(model.connectivity, model.Nnodes, coords_hint=model.initial_node_coords, x_src=model._x_src,
                        u_src=model._u_src, edges=model.neumann_edges, tile_elems=a.tile, device=dev,
                        elem_order=int(order))

        def launch(k, flags, stream):
            xf, uf, gx, gu = sets[k]
            _lib.check(L.hfem_tri3_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(),
                                               ufix.data_ptr(), mat, W, Bk, None, Tc, 0, -1, loss.data_ptr(),
                                               gx.data_ptr(), gu.data_ptr(), flags, stream))

        for g in sets[0][2:]:
            g.fill_(float("nan"))
        launch(0, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        cur = (loss.item(), sets[0][2].clone(), sets[0][3].clone())
        ok = ""
        if "ablate" in optstr:
            ok = "ablated"
        elif ref is None:
            ref = cur
        else:
            dl = abs(cur[0] - ref[0]) / abs(ref[0])
            dgx = (cur[1] - ref[1]).abs().max().item() / ref[1].abs().max().item()
            dgu = (cur[2] - ref[2]).abs().max().item() / ref[2].abs().max().item()
            ok = f"dl={dl:.1e} dgx={dgx:.1e} dgu={dgu:.1e}"
            assert dl < 1e-12 and dgx < 1e-11 and dgu < 1e-11 and not torch.isnan(cur[1]).any(), (var, ok)

        def timed(body):
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                body(s.cuda_stream, 0)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st = torch.cuda.current_stream().cuda_stream
                for i in range(a.reps):
                    body(st, i)
            t_pw = time.perf_counter()
            while time.perf_counter() - t_pw < a.prewarm:
                g.replay()
                torch.cuda.synchronize()
            samples = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                samples.append(e0.elapsed_time(e1) * 1e3 / a.reps)
            return sorted(samples)[2]

        def rw(st, i):
            sets[0][0].mul_(1.0)
            sets[0][1].mul_(1.0)

        res = {}
        for reg in a.regimes.split(","):
            if reg == "replay":
                res[reg] = timed(lambda st, i: launch(0, 8, st))
            elif reg == "rewrite":
                t_rw = timed(rw)
                res[reg] = timed(lambda st, i: (rw(st, i), launch(0, 8, st))) - t_rw
                res["rewrite_alone"] = t_rw
            elif reg == "rotate":
                res[reg] = timed(lambda st, i: launch(i % a.sets, 8, st))
        st = plan.stats
        row = dict(variant=var, tiles=st["n_tiles"], slots=st["tile_elem_total"], nodes=st["tile_node_total"],
                   lds=st["lds_bytes"], check=ok, **{k: round(v, 3) for k, v in res.items()},
                   **{"frac_" + k: round(alg / v / 1e3 / 8000, 3) for k, v in res.items() if k != "rewrite_alone"})
        print(json.dumps(row), flush=True)
        plan.close()
    for name, val in defaults.items():
        _lib.check(L.hfem_set_option(name.encode(), val))


if __name__ == "__main__":
    main()

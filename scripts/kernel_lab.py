#!/usr/bin/env python3
"""Kernel lab (dev tool): time the tiled TRI3 energy kernel alone on the T1M workload for a sweep of
tile sizes / threads per tile / ablation bits.  Usage on the GPU box:

    python scripts/kernel_lab.py --tiles 512,768,1024 --blocks 256,512 --ablate 0,1,2,4,8
"""
import os; os.environ.setdefault("HFEM_LAB", "1")   # kernel-lab tool: needs libhidenn_hip_lab.so (build.py --lab)
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hidenn_fem_amd import _lib
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.loss import EnergyLoss2D


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=1001)
    ap.add_argument("--ny", type=int, default=501)
    ap.add_argument("--tiles", default="1024")
    ap.add_argument("--blocks", default="256")
    ap.add_argument("--ablate", default="0")
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--permute", action="store_true")
    ap.add_argument("--flags", type=int, default=8)
    ap.add_argument("--orders", default="3")
    ap.add_argument("--pipes", default="0")
    ap.add_argument("--staggers", default="0", help="total start spread in 10 ns ticks")
    ap.add_argument("--smodes", default="0")
    ap.add_argument("--fstaggers", default="0", help="fast kernel: phase offset in 10 ns ticks")
    ap.add_argument("--fshifts", default="8")
    ap.add_argument("--fgroups", default="2")
    ap.add_argument("--prewarm", type=float, default=0.3, help="seconds of untimed replays before each timing")
    ap.add_argument("--fast", default="1")
    ap.add_argument("--curve", type=int, default=1)
    ap.add_argument("--caps", default="0")
    ap.add_argument("--stores", default="16")
    ap.add_argument("--rewrite", action="store_true",
                    help="rewrite x_free/u_free between launches (optimizer-like); reports the energy kernel time "
                         "as (rewrite+energy) - (rewrite only)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    f64 = torch.float64
    kw = dict(diagonal="random", permute=True, jitter=0.3) if a.permute else dict(jitter=0.2)
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(a.nx, a.ny, seed=0, dtype=f64, **kw)
    torch.manual_seed(0)
    model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                     neumann_edges=edges).to(dev)
    ne, nn = conn.shape[0], coords.shape[0]
    alg = 12 * ne + 64 * nn + 8
    L = _lib.lib()
    dv = lambda v: (C.c_double * len(v))(*v)
    ref = None
    rows = []
    from hidenn_fem_amd.plan import TilePlan
    _lib.check(L.hfem_set_option(b"plan_curve", a.curve))
    for T, order, fastv, cap in [(int(t), int(o), int(f), int(c_)) for t in a.tiles.split(",") for o in a.orders.split(",")
                                 for f in a.fast.split(",") for c_ in a.caps.split(",")]:
        _lib.check(L.hfem_set_option(b"tiled_fast", fastv))
        _lib.check(L.hfem_set_option(b"plan_node_cap", cap))
        lf = EnergyLoss2D(device=dev, dtype=f64, tile_elems=T)
        plan = TilePlan(model.connectivity, model.Nnodes, coords_hint=model.initial_node_coords,
                        x_src=model._x_src, u_src=model._u_src, edges=model.neumann_edges, tile_elems=T,
                        device=dev, elem_order=order)
        xf, uf = model.node_coords_free.detach(), model.u_free.detach()
        xfix, ufix = model.node_coords_fixed, model.u_fixed_rows()
        _, Tconst = lf._traction(model, None)
        loss = torch.zeros((), dtype=f64, device=dev)
        gx, gu = torch.zeros_like(xf), torch.zeros_like(uf)
        mat, W, Bk, Tc = dv(lf._mat), lf._W, dv([0.0] * 6), dv(Tconst)

        def launch(flags, stream):
            _lib.check(L.hfem_tri3_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(),
                                               ufix.data_ptr(), mat, W, Bk, None, Tc, 0, -1, loss.data_ptr(),
                                               gx.data_ptr(), gu.data_ptr(), flags, stream))

        for B in [int(b) for b in a.blocks.split(",")]:
            for abl, pipe, stg, smode, sp, fst, fsh, fgr in [
                    (int(x), int(q), int(g_), int(m_), int(s_), int(f1), int(f2), int(f3)) for x in a.ablate.split(",")
                    for q in a.pipes.split(",") for g_ in a.staggers.split(",") for m_ in a.smodes.split(",")
                    for s_ in a.stores.split(",") for f1 in a.fstaggers.split(",")
                    for f2 in (a.fshifts.split(",") if int(f1) else ["8"])
                    for f3 in (a.fgroups.split(",") if int(f1) else ["2"])]:
                _lib.check(L.hfem_set_option(b"store_policy", sp))
                _lib.check(L.hfem_set_option(b"fast_stagger", fst))
                _lib.check(L.hfem_set_option(b"fast_stagger_shift", fsh))
                _lib.check(L.hfem_set_option(b"fast_stagger_groups", fgr))
                if pipe and (abl or stg):
                    continue
                if stg == 0 and smode != int(a.smodes.split(",")[0]):
                    continue
                _lib.check(L.hfem_set_option(b"tiled_stagger", stg))
                _lib.check(L.hfem_set_option(b"tiled_stagger_mode", smode))
                _lib.check(L.hfem_set_option(b"tiled_pipe", pipe))
                _lib.check(L.hfem_set_option(b"tiled_block", B))
                _lib.check(L.hfem_set_option(b"tiled_ablate", abl))
                launch(0, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                ok = ""
                if abl == 0:
                    cur = (loss.item(), gx.clone(), gu.clone())
                    if ref is None:
                        ref = cur
                    else:
                        dl = abs(cur[0] - ref[0]) / abs(ref[0])
                        dgx = (cur[1] - ref[1]).abs().max().item() / ref[1].abs().max().item()
                        dgu = (cur[2] - ref[2]).abs().max().item() / ref[2].abs().max().item()
                        ok = f"dl={dl:.1e} dgx={dgx:.1e} dgu={dgu:.1e}"
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    launch(a.flags, s.cuda_stream)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                def timed(body):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        for _ in range(a.reps):
                            body()
                    import time as _t
                    t_pw = _t.perf_counter()              # warm clocks: a cold chip reads 5-7 % slow
                    while _t.perf_counter() - t_pw < a.prewarm:
                        g.replay()
                        torch.cuda.synchronize()
                    b_ = 1e9
                    for _ in range(5):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        g.replay()
                        e1.record()
                        torch.cuda.synchronize()
                        b_ = min(b_, e0.elapsed_time(e1) * 1e3 / a.reps)
                    return b_

                def energy():
                    launch(a.flags, torch.cuda.current_stream().cuda_stream)

                def rewrite():                 # what an optimiser step does to the inputs: read-modify-write
                    xf.mul_(1.0)
                    uf.mul_(1.0)

                if a.rewrite:
                    rewrite(); torch.cuda.synchronize()
                    t_rw = timed(rewrite)
                    t_both = timed(lambda: (rewrite(), energy()))
                    best = t_both - t_rw
                else:
                    best = timed(energy)
                st = plan.stats
                row = dict(T=T, cap=cap, curve=a.curve, order=order, fast=fastv, tiles=st["n_tiles"], lds=st["lds_bytes"], block=B, store=sp, pipe=pipe, stagger=stg, smode=smode, fstagger=fst, fshift=fsh, fgroups=fgr, ablate=abl, us=round(best, 2),
                           GBs=round(alg / best / 1e3, 1), frac=round(alg / best / 1e3 / 8000, 3), check=ok)
                rows.append(row)
                print(json.dumps(row), flush=True)
    _lib.check(L.hfem_set_option(b"tiled_block", 512))
    _lib.check(L.hfem_set_option(b"tiled_ablate", 0))
    _lib.check(L.hfem_set_option(b"tiled_pipe", 0))
    _lib.check(L.hfem_set_option(b"tiled_stagger", 0))
    _lib.check(L.hfem_set_option(b"store_policy", 0))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Kernel lab (dev tool): time the QUAD4 energy kernels on the cfg4-Q workload (1001 x 1001 nodes ->
10^6 QUAD4, fp64): tiled owner-computes kernel vs the planless global-atomic kernel.

    python scripts/quad4_lab.py --tiles 0,512,768
"""
import os; os.environ.setdefault("HFEM_LAB", "1")   # kernel-lab tool: needs libhidenn_hip_lab.so (build.py --lab)
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hidenn_fem_amd import _lib
from hidenn_fem_amd.mesh import structured_quad_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.loss import EnergyLoss2D
from hidenn_fem_amd.plan import TilePlan


def time_graph(launch, reps):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            launch(st.cuda_stream)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                launch(torch.cuda.current_stream().cuda_stream)
        import time as _t
        t_pw = _t.perf_counter()                  # warm clocks: a cold chip reads 5-7 % slow
        while _t.perf_counter() - t_pw < 0.3:      # prewarm
            g.replay()
            st.synchronize()
        best = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            g.replay()
            e1.record(st)
            st.synchronize()
            best.append(e0.elapsed_time(e1) * 1e3 / reps)
    best.sort()
    return best[2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1001)
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--caps", default="-1")
    ap.add_argument("--staggers", default="0", help="phase offset in 10 ns ticks")
    ap.add_argument("--shifts", default="8", help="workgroup-index bit that selects the delayed half")
    ap.add_argument("--groups", default="2")
    ap.add_argument("--ccaps", default="1", help="compile-time accumulator stride instance (1) or runtime strides (0)")
    ap.add_argument("--pipes", default="0", help="persistent pipelined kernel: workgroups per CU (0 = off)")
    ap.add_argument("--ablate", default="0", help="lab bits: 1 no element math, 2 no LDS atomics")
    ap.add_argument("--bits", default="0", help="lab bits: 1 NO raised wave priority through the memory phases, 2 rotating short slot chunk")
    ap.add_argument("--snaps", default="0", help="plan_snap values (tile cuts snap to coarse curve cells, percent of a tile)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    f64 = torch.float64
    coords, conn, geom, bc, mn, edges = structured_quad_mesh(a.n, a.n, length=2.0, height=2.0, jitter=0.2, seed=0, dtype=f64)
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                 neumann_edges=edges).to(dev)
    ne, nn = conn.shape[0], coords.shape[0]
    alg = 16 * ne + 64 * nn + 8            # conn 4 x int32 + (X, U) in + (gX, gU) out, fp64
    L = _lib.lib()
    dv = lambda v: (C.c_double * len(v))(*v)
    lf = EnergyLoss2D(device=dev, dtype=f64)
    _, Tconst = lf._traction(m, None)
    mat, Tc = dv(lf._mat), dv(Tconst)
    xf, uf = m.node_coords_free.detach(), m.u_free.detach()
    xfix, ufix = m.node_coords_fixed, m.u_fixed_rows()
    loss = torch.zeros((), dtype=f64, device=dev)
    gx, gu = torch.zeros_like(xf), torch.zeros_like(uf)
    # planless reference: zero-fill + global fp64 atomics on the assembled arrays
    X, U = m.coords.detach().contiguous(), m.u_full.detach().contiguous()
    gX, gU = torch.zeros_like(X), torch.zeros_like(U)
    acc = torch.zeros((), dtype=f64, device=dev)

    def planless(stream):
        _lib.check(L.hfem_quad4_energy_atomic(0, X.data_ptr(), U.data_ptr(), m._conn32.data_ptr(), 0, ne, nn, mat,
                                              acc.data_ptr(), gX.data_ptr(), gU.data_ptr(), stream))

    us = time_graph(planless, a.reps)
    print(json.dumps(dict(kernel="quad4_atomic(planless, no zero fill)", us=round(us, 2),
                          alg_TBps=round(alg / us * 1e-6, 3))), flush=True)
    for T, cap, snap in [(int(t), int(c), int(sn)) for t in a.tiles.split(",") for c in a.caps.split(",") for sn in a.snaps.split(",")]:
        _lib.check(L.hfem_set_option(b"plan_node_cap", cap))
        _lib.check(L.hfem_set_option(b"plan_snap", snap))
        plan = TilePlan(m.connectivity, m.Nnodes, coords_hint=m.initial_node_coords, x_src=m._x_src, u_src=m._u_src,
                        edges=m.neumann_edges, tile_elems=T, device=dev, nodes_per_elem=4)

        def tiled(stream, flags=8):
            _lib.check(L.hfem_quad4_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(), ufix.data_ptr(),
                                                mat, None, Tc, 0, -1, loss.data_ptr(), gx.data_ptr(), gu.data_ptr(), flags,
                                                stream))

        ref = None
        for abl, stg, sh, grp, pipe in [(int(x), int(y), int(z), int(w), int(q)) for x in a.ablate.split(",")
                                        for y in a.staggers.split(",") for z in (a.shifts.split(",") if int(y) else ["8"])
                                        for w in (a.groups.split(",") if int(y) else ["2"]) for q in a.pipes.split(",")]:
          for cc, bits in [(int(c_), int(b_)) for c_ in a.ccaps.split(",") for b_ in a.bits.split(",")]:
              _lib.check(L.hfem_set_option(b"quad4_const_caps", cc))
              _lib.check(L.hfem_set_option(b"quad4_bits", bits))
              _lib.check(L.hfem_set_option(b"quad4_pipe", pipe))
              if abl == 0:                                   # correctness of the variant vs the first one
                  tiled(torch.cuda.current_stream().cuda_stream, 0)
                  torch.cuda.synchronize()
                  cur = (loss.item(), gx.clone(), gu.clone())
                  if ref is None:
                      ref = cur
                  chk = "dl=%.1e dgx=%.1e dgu=%.1e" % (abs(cur[0] - ref[0]) / abs(ref[0]),
                                                       (cur[1] - ref[1]).abs().max().item() / ref[1].abs().max().item(),
                                                       (cur[2] - ref[2]).abs().max().item() / ref[2].abs().max().item())
              _lib.check(L.hfem_set_option(b"quad4_stagger_groups", grp))
              _lib.check(L.hfem_set_option(b"quad4_ablate", abl))
              _lib.check(L.hfem_set_option(b"quad4_stagger", stg))
              _lib.check(L.hfem_set_option(b"quad4_stagger_shift", sh))
              us = time_graph(tiled, a.reps)
              _lib.check(L.hfem_set_option(b"quad4_ablate", 0))
              _lib.check(L.hfem_set_option(b"quad4_stagger", 0))
              _lib.check(L.hfem_set_option(b"quad4_pipe", 0))
              st = plan.stats
              print(json.dumps(dict(kernel="quad4_tiled", ablate=abl, stagger=stg, shift=sh, groups=grp, pipe=pipe, ccaps=cc, bits=bits, check=chk if abl == 0 else "", tile_elems=T, cap=cap, snap=snap, us=round(us, 2),
                                    alg_TBps=round(alg / us * 1e-6, 3), frac=round(alg / us * 1e-6 / 8.0, 3),
                                    n_tiles=st["n_tiles"], lds=st["lds_bytes"], slots=st["tile_elem_total"],
                                    max_nodes=st["max_tile_nodes"], max_elems=st["max_tile_elems"])), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Strong-scaling rehearsal on ONE GPU (dev tool, product library): time the tiled TRI3 energy kernel over the tile
range that rank r of N would evaluate, for a sweep of tile sizes (``plan_node_cap``), so that the shard-aware tile
policy of ``hfem_plan_create`` (``plan_shards``) can be chosen from measurements.

    python scripts/shard_sweep.py --mesh t1m --caps 557,400,280,200,140,100,70 --worlds 1,2,4,8

Every (cap, N) line reports the per-rank kernel time (HIP events around a hipGraph of K back-to-back launches on the
rank's tile range) for EVERY rank of N -- max and mean -- plus the plan's tile count and halo factors.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hidenn_fem_amd import _lib
from hidenn_fem_amd.loss import EnergyLoss2D
from hidenn_fem_amd.mesh import structured_tri_mesh, unstructured_tri_mesh
from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
from hidenn_fem_amd.plan import TilePlan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="t1m", help="t1m | cfg5u | t250k | t125k")
    ap.add_argument("--caps", default="557,400,280,200,140,100,70")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--blocks", default="-1", help="plan_pair_block values (-1 auto, 256, 512)")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--prewarm", type=float, default=0.3)
    ap.add_argument("--orders", default="-1")
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--all-ranks", action="store_true", help="time every rank (default: first, middle, last)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    f64 = torch.float64
    if a.mesh == "t1m":
        mesh = structured_tri_mesh(1001, 501, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64)
    elif a.mesh == "t250k":
        mesh = structured_tri_mesh(501, 251, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64)
    elif a.mesh == "t125k":
        mesh = structured_tri_mesh(251, 251, length=1.0, height=1.0, jitter=0.2, seed=0, dtype=f64)
    elif a.mesh.startswith("grid:"):          # grid:NX:NY nodes -> 2 (NX-1)(NY-1) TRI3 on [0, 2] x [0, 1]
        _, gx_, gy_ = a.mesh.split(":")
        mesh = structured_tri_mesh(int(gx_), int(gy_), length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64)
    elif a.mesh == "cfg5u":
        mesh = unstructured_tri_mesh(2_050_000, seed=2, dtype=f64)
    else:
        raise SystemExit("unknown mesh")
    coords, conn, geom, bc, mn, edges = mesh
    torch.manual_seed(0)
    model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                     neumann_edges=edges).to(dev)
    ne, nn = conn.shape[0], coords.shape[0]
    L = _lib.lib()
    for kv in a.option:
        k_, v_ = kv.split("=")
        _lib.check(L.hfem_set_option(k_.encode(), int(v_)), "hfem_set_option")
    dv = lambda v: (C.c_double * len(v))(*v)
    lf = EnergyLoss2D(device=dev, dtype=f64)
    xf, uf = model.node_coords_free.detach(), model.u_free.detach()
    xfix, ufix = model.node_coords_fixed, model.u_fixed_rows()
    _, Tconst = lf._traction(model, None)
    loss = torch.zeros((), dtype=f64, device=dev)
    gx, gu = torch.zeros_like(xf), torch.zeros_like(uf)
    mat, W, Bk, Tc = dv(lf._mat), lf._W, dv([0.0] * 6), dv(Tconst)
    ref = None
    worlds = [int(w) for w in a.worlds.split(",")]
    for cap, order, blk in [(int(c), int(o), int(b)) for c in a.caps.split(",") for o in a.orders.split(",")
                            for b in a.blocks.split(",")]:
        # cap -1 = the library's shard-aware policy: one plan PER world size (plan_shards = world); an explicit cap is one
        # plan for all world sizes (plain contiguous ranges)
        for wi, world in enumerate(worlds):
            if cap >= 0 and wi > 0:
                pass
            else:
                _lib.check(L.hfem_set_option(b"plan_node_cap", cap), "set")
                t0 = time.perf_counter()
                plan = TilePlan(model.connectivity, model.Nnodes, coords_hint=model.initial_node_coords, x_src=model._x_src,
                                u_src=model._u_src, edges=model.neumann_edges, tile_elems=0 if cap != 0 else 1024, device=dev,
                                elem_order=None if order < 0 else order, shards=world if cap < 0 else 1,
                                pair_block=None if blk < 0 else blk)
                t_plan = time.perf_counter() - t0
                st = plan.stats

                def launch(lo, hi, flags, stream, plan=plan):
                    _lib.check(L.hfem_tri3_energy_plan(plan.handle, xf.data_ptr(), xfix.data_ptr(), uf.data_ptr(), ufix.data_ptr(),
                                                       mat, W, Bk, None, Tc, lo, hi, loss.data_ptr(), gx.data_ptr(),
                                                       gu.data_ptr(), flags, stream), "launch")

                gx.zero_(); gu.zero_()
                launch(0, -1, 0, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                cur = (loss.item(), gx.clone(), gu.clone())
                ok = ""
                if ref is None:
                    ref = cur
                else:
                    ok = "dl=%.1e dgx=%.1e dgu=%.1e" % (abs(cur[0] - ref[0]) / abs(ref[0]),
                                                        (cur[1] - ref[1]).abs().max().item() / ref[1].abs().max().item(),
                                                        (cur[2] - ref[2]).abs().max().item() / ref[2].abs().max().item())
            ranks = range(world) if a.all_ranks or world <= 3 else sorted({0, world // 2, world - 1})
            res, parts = [], []
            for r in ranks:
                lo, mid, hi = plan.shard_parts(r, world)
                parts.append((mid - lo, hi - mid))
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    launch(lo, hi, 8, s.cuda_stream)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    cs = torch.cuda.current_stream().cuda_stream
                    for _ in range(a.reps):
                        launch(lo, hi, 8, cs)
                t_pw = time.perf_counter()
                while time.perf_counter() - t_pw < a.prewarm:
                    g.replay()
                    torch.cuda.synchronize()
                regs = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    g.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    regs.append(e0.elapsed_time(e1) * 1e3 / a.reps)
                res.append(sorted(regs)[2])
                del g
            print(json.dumps(dict(mesh=a.mesh, elements=ne, nodes=nn, cap=cap, order=order, paired=plan.is_paired(),
                                  threads=st["threads_per_tile"], slot_rows=st["slot_rows"],
                                  tiles=st["n_tiles"], world=world, tiles_per_rank=st["n_tiles"] // world,
                                  boundary_interior=parts,
                                  halo_elem=round(st["tile_elem_total"] / max(ne, 1), 3),
                                  halo_node=round(st["tile_node_total"] / max(nn, 1), 3),
                                  max_nodes=st["max_tile_nodes"], max_owned=st["max_tile_owned"],
                                  max_slots=st["max_tile_elems"], lds=st["lds_bytes"],
                                  us_max=round(max(res), 3), us_mean=round(sum(res) / len(res), 3),
                                  us_ranks=[round(v, 3) for v in res], plan_s=round(t_plan, 2), check=ok)), flush=True)
        del plan


if __name__ == "__main__":
    main()

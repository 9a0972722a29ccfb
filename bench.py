#!/usr/bin/env python3
"""Headline benchmark: fused TRI3+EDGE2 elastic-energy forward+backward ("element-evals/s").

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver as ``python -m torch.distributed.run --nproc-per-node N ...``
(one rank per GPU, RCCL).  A *step* is one pass of the hot path over the synthetic mesh:
loss + d/d node_coords_free + d/d u_free, inputs resident in HBM.

Workload at N = 1: BASELINE.json configs[3] "Example 4", reading (i) of SURVEY F11 = "T1M":
plate [0,2]x[0,1], 1001x501 nodes -> 1,000,000 TRI3 (each structured quad split in two),
interior nodes jittered 0.2 h (seed 0), outer boundary fixed, left edge Dirichlet, right edge
Neumann (500 edges), u_free ~ 1e-5 N(0,1), E=10e9, nu=0.3, gauss_order=4, fp64, r-adaptivity on.
N > 1 (weak scaling): the plate grows to N x 1,000,000 elements (N*1000+1 x 501 nodes); every rank evaluates its
contiguous tile range (one process per GPU), then ONE collective.  Headline (`value`) = OWNER-SHARDED mode: owner-computes
tiles give each rank complete gradient rows for the nodes it owns, so what crosses ranks is one small all_gather of the
interface parameter rows + the partial energies -- in-library RCCL (hfem_mg_*, csrc/mg.cpp) enqueued on the kernel's
stream, the K steps captured in one hipGraph.  Reported beside it: `config.train_step` (the same loop with Adam on the
owned rows inside, so the exchanged rows change every step) and `config.alt_exchange`, the north-star's literal wording:
a dense sum all-reduce of the packed [gX|gU|loss] buffer.

One JSON line on stdout (rank 0).  ``roofline`` is for the dominant kernel
(tri3_energy_pair_kernel on the paired tile plan a split-quad mesh gets; tri3_energy_fast_kernel otherwise): algorithmic
bytes (12 Ne + 64 Nn + 8) over its average back-to-back launch time measured with HIP events on the
launch stream.
``cpu_baseline`` times the oracle's op-for-op PyTorch restatement of the reference chain on the
host cores, on the same workload (bounded number of evaluations).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nx", type=int, default=1001, help="nodes along x PER GPU (+1 shared column)")
    ap.add_argument("--ny", type=int, default=501)
    ap.add_argument("--tile-elems", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph of K steps")
    ap.add_argument("--cpu-evals", type=int, default=2)
    ap.add_argument("--inline-loss-sum", action="store_true", help="N = 1: reduce the tile energies with a separate "
                    "1-block launch after every energy kernel instead of inside the next launch")
    ap.add_argument("--prewarm", type=float, default=0.5, help="seconds of untimed replays before each timed leg")
    ap.add_argument("--option", action="append", default=[], help="name=value for hfem_set_option (lab A/B runs)")
    ap.add_argument("--no-extra", action="store_true", help="skip config.extra (Q1M, T2M, cfg5 structured / Delaunay kernel timings)")
    ap.add_argument("--no-regimes", action="store_true", help="skip the extra cache-regime legs of the roofline block")
    ap.add_argument("--only-regime", default="", help="rocprof helper: run ONLY this roofline leg (replayed | "
                    "rewritten_inputs | rotating_sets) and exit")
    ap.add_argument("--rotating-sets", type=int, default=10, help="parameter/gradient sets of the rotating leg (32 MB each)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only "
                                                      "for rehearsing the multi-rank path on fewer GPUs than ranks)")
    return ap.parse_args()


def cpu_baseline(mesh6, u_free, n_evals):
    """oracle/ref_chain.py (the reference's ATen op chain + autograd) on the host cores."""
    from oracle import ref_chain as R
    coords, conn, geom, bc, mn, edges = mesh6
    # the GPU box gives one GPU's share of the host (16 cores); more threads than that only thrash
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    mesh = dict(n_nodes=coords.shape[0], conn=conn, free_mask=~geom, boundary_mask=geom,
                coords_fixed=coords[geom], u_free_mask=~bc, dirichlet_mask=bc,
                u_fixed=torch.tensor(0.0, dtype=torch.float64), edges=edges)
    xf, uf = coords[~geom].clone(), u_free.clone()
    R.energy_and_grads(xf, uf, mesh)                       # warm-up
    best = float("inf")
    for _ in range(n_evals):
        t0 = time.perf_counter()
        loss, gx, gu = R.energy_and_grads(xf, uf, mesh)
        best = min(best, time.perf_counter() - t0)
    return dict(value=conn.shape[0] / best, unit="element-evals/s", cores=torch.get_num_threads(), kind="port",
                sample=f"same workload ({conn.shape[0]} TRI3, fp64), best of {n_evals} fwd+bwd evaluations "
                       f"after 1 warm-up, {best:.3f} s/eval"), loss.item()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a ROCm device; there is no CPU fallback"
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and world > ndev:
        sys.exit(f"bench.py: {world} ranks but {ndev} GPU(s) visible (RCCL needs one GPU per rank)")
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.sharded import LibraryComm, ShardedTri3Energy

    f64 = torch.float64
    nx = (a.nx - 1) * world + 1
    mesh6 = structured_tri_mesh(nx, a.ny, length=2.0 * world, height=1.0, jitter=0.2, seed=0, dtype=f64)
    coords, conn, geom, bc, mn, edges = mesh6
    ne, nn = conn.shape[0], coords.shape[0]
    torch.manual_seed(0)
    model = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                     neumann_edges=edges).to(dev)
    for kv in a.option:
        name, val = kv.split("=")
        _lib.check(_lib.lib().hfem_set_option(name.encode(), int(val)), "hfem_set_option")
    loss_fn = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64, tile_elems=a.tile_elems)
    comm = None
    if world > 1 and a.backend == "nccl":        # in-library RCCL communicator: collectives on the kernel's stream, capturable
        try:
            comm = LibraryComm(dev)
        except Exception as e:  # pragma: no cover
            if rank == 0:
                print(f"[bench] in-library RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
    sh = ShardedTri3Energy(model, loss_fn, comm=comm)
    plan = sh.plan
    lo, hi = sh.lo, sh.hi
    td = plan.export("tile_desc")
    ne_local_home = ne if world == 1 else None

    if world > 1:
        sh.setup_interfaces()
        sh.init_owner_adam(lr_x=1e-9, lr_u=1e-12)      # tiny steps: the mesh stays valid over any number of iterations

    def step_dense():           # north-star literal: one all-reduce of [gX|gU|loss] (every rank gets everything)
        sh.evaluate_local()
        sh.exchange()

    def step_owner():           # owner-sharded: gradient rows stay with their owner; ONE all_gather of the
        sh.owner_step()         # interface parameter rows + partial energies (what the next evaluation needs)

    # N = 1: the energy of step k is reduced by an extra workgroup of launch k+1 (HFEM_FLAG_SUM_PREVIOUS) and the last one
    # by a trailing 1-block launch, inside the timed region: every step's loss is produced, the reduction and its kernel
    # boundary just leave the critical path.  --inline-loss-sum restores the separate reduction after every launch.
    lagged = world == 1 and not a.inline_loss_sum
    step = (sh.evaluate_local_lagged if lagged else sh.evaluate_local) if world == 1 else step_owner

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- correctness guard: the benchmarked path must produce the oracle's numbers
    step()
    if lagged:
        sh.flush_loss()
    torch.cuda.synchronize()
    if world == 1:
        loss_gpu = sh._views(sh.send)[0].item()
    else:                       # the two independent exchange paths must agree on the global energy
        loss_gpu = sh.loss_global.item()
        step_dense()
        torch.cuda.synchronize()
        loss_dense = sh._views(sh.recv)[0].item()
        assert abs(loss_gpu - loss_dense) <= 1e-12 * abs(loss_dense), (loss_gpu, loss_dense)

    # ---- timed region: W warm-up steps, then exactly K steps between barrier+synchronize
    use_graph = (world == 1 or comm is not None) and not a.no_graph
    graph = None
    if use_graph:
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
                if lagged:
                    sh.flush_loss()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                if lagged:
                    sh.begin_lagged()
                for _ in range(a.steps):
                    step()
                if lagged:
                    sh.flush_loss()
        except Exception as e:  # pragma: no cover
            if rank == 0:
                print(f"[bench] hipGraph capture failed ({e}); falling back to eager launches", file=sys.stderr)
            graph = None
    # untimed pre-warm on top of the W warm-up steps: the whole default run is a few ms of GPU time, shorter than
    # the clock ramp of an idle chip (the first timed regions read 5-7 % slow without it).  Fixed count when N > 1
    # (every rank must issue the same number of collectives).
    if graph is not None:
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < a.prewarm:
            graph.replay()
            torch.cuda.synchronize()
    else:
        for _ in range(int(a.prewarm * 400)):
            step()
    for _ in range(a.warmup):
        step()
    if lagged:
        sh.flush_loss()
    if graph is not None:
        graph.replay()
    sync_all()
    t0 = time.perf_counter()
    if graph is not None:
        graph.replay()
    else:
        for _ in range(a.steps):
            step()
        if lagged:
            sh.flush_loss()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=f64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if lagged:                  # the trailing flush delivered the last step's energy: same bits as the guard evaluation
        assert sh._views(sh.send)[0].item() == loss_gpu, (sh._views(sh.send)[0].item(), loss_gpu)
    ms_per_step = elapsed / a.steps * 1e3
    value = ne / (elapsed / a.steps)          # whole-job element-evals/s (all ranks' elements)

    # ---- N = 1, reported beside the headline: the step of a caller that needs ITS OWN loss before it goes on (a line
    #      search, an L-BFGS check): the 1-block reduction launched right after every energy launch, on the critical path
    inline_step = None
    if world == 1 and lagged and not a.no_extra:
        try:
            s_ = torch.cuda.Stream()
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_):
                for _ in range(3):
                    sh.evaluate_local()
            torch.cuda.current_stream().wait_stream(s_)
            torch.cuda.synchronize()
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_):
                for _ in range(a.steps):
                    sh.evaluate_local()
            g_.replay()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            g_.replay()
            torch.cuda.synchronize()
            el_ = time.perf_counter() - t0_
            assert sh._views(sh.send)[0].item() == loss_gpu
            inline_step = dict(mode="loss of step k reduced by its own 1-block launch before step k+1 (--inline-loss-sum)",
                               ms_per_step=el_ / a.steps * 1e3, value=ne / (el_ / a.steps))
        except Exception as e:  # pragma: no cover
            print(f"[bench] inline-loss leg failed: {e}", file=sys.stderr)

    # ---- N > 1 only, reported beside the headline: the north-star's literal exchange, a dense all-reduce of the
    #      full gradient + loss (every rank ends with everything; 16 B x 2 x nodes x N on the wire)
    def timed_alt(body):
        """W warm-up + K timed iterations of `body` (one hipGraph of K when the in-library comm allows capture),
        bracketed like the headline; returns seconds (max over ranks)."""
        g = None
        if comm is not None and not a.no_graph:
            try:
                s_ = torch.cuda.Stream()
                s_.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s_):
                    for _ in range(2):
                        body()
                torch.cuda.current_stream().wait_stream(s_)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(a.steps):
                        body()
            except Exception:  # pragma: no cover
                g = None
        for _ in range(a.warmup):
            body()
        if g is not None:
            g.replay()
        sync_all()
        t0_ = time.perf_counter()
        if g is not None:
            g.replay()
        else:
            for _ in range(a.steps):
                body()
        sync_all()
        el = time.perf_counter() - t0_
        t = torch.tensor([el], dtype=f64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item(), g is not None

    alt = train = None
    if world > 1:
        el2, g2 = timed_alt(step_dense)
        alt = dict(mode=f"dense: sum all-reduce of [gX|gU|loss] fp64, {sh.send.numel() * 8} B per rank, "
                        f"{'in-library RCCL on the kernel stream' if comm is not None else 'torch.distributed ' + a.backend}",
                   value=ne / (el2 / a.steps), ms_per_step=el2 / a.steps * 1e3, launch="hipgraph" if g2 else "eager")
        el3, g3 = timed_alt(sh.owner_train_step)
        train = dict(mode="owner-sharded training iteration: energy -> Adam on the rows the rank owns -> pack -> all_gather "
                          "-> unpack (the exchanged interface rows change every step)",
                     value=ne / (el3 / a.steps), ms_per_step=el3 / a.steps * 1e3, launch="hipgraph" if g3 else "eager")

    # ---- roofline leg: the dominant kernel alone, K back-to-back launches, HIP events on its stream
    L = _lib.lib()
    dv = lambda v: (C.c_double * len(v))(*v)
    xf, uf = model.node_coords_free.detach(), model.u_free.detach()
    xfix, ufix = model.node_coords_fixed, model.u_fixed_rows()
    _, Tconst = loss_fn._traction(model, None)
    loss_s, gx_s, gu_s = sh._views(sh.send)
    mat, W, Bk, Tc = dv(loss_fn._mat), loss_fn._W, dv([0.0] * 6), dv(Tconst)
    stream = torch.cuda.current_stream()

    def kernel_only(bufs=None):
        x_, u_, gx_, gu_ = bufs if bufs is not None else (xf, uf, gx_s, gu_s)
        _lib.check(L.hfem_tri3_energy_plan(plan.handle, x_.data_ptr(), xfix.data_ptr() if xfix.numel() else None,
                                           u_.data_ptr(), ufix.data_ptr() if ufix.numel() else None, mat, W, Bk, None,
                                           Tc, lo, hi, loss_s.data_ptr(), gx_.data_ptr(), gu_.data_ptr(),
                                           8, stream.cuda_stream))           # HFEM_FLAG_NO_LOSS_SUM

    kreps = max(a.steps, 50)

    def time_launches(body):
        """Average time of one `body(i)` over `kreps` back-to-back calls (one hipGraph unless --no-graph), HIP events on
        the launch stream, median of 5 regions after the clock pre-warm.  Returns (us, regions)."""
        nonlocal stream
        g = None
        if not a.no_graph:
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    stream = s
                    body(0)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    stream = torch.cuda.current_stream()
                    for i in range(kreps):
                        body(i)
            except Exception:
                g = None
        stream = torch.cuda.current_stream()
        for i in range(5):
            body(i)
        t_pw = time.perf_counter()
        while g is not None and time.perf_counter() - t_pw < a.prewarm:
            g.replay()
            torch.cuda.synchronize()
        regions = []
        for _ in range(5):                      # 5 timed regions of `kreps` back-to-back bodies each
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(stream)
            if g is not None:
                g.replay()
            else:
                for i in range(kreps):
                    body(i)
            ev1.record(stream)
            torch.cuda.synchronize()
            regions.append(ev0.elapsed_time(ev1) * 1e3 / kreps)
        return sorted(regions)[len(regions) // 2], regions       # median region; each value is a kreps-launch average

    only = a.only_regime
    k_us, samples = time_launches(lambda i: kernel_only()) if only in ("", "replayed") else (float("nan"), [])
    # ---- the same kernel in the cache regimes a training loop sees (reported beside the headline leg, never instead):
    #  rewritten_inputs: x and u are rewritten by another kernel before every launch (what an optimiser step does);
    #                    the kernel's share = (rewrite + energy) - (rewrite alone), both as back-to-back graphs
    #  rotating_sets:    R parameter / gradient sets (R x 32 MB > the 256 MB Infinity Cache) visited round-robin, so
    #                    every read of x, u and every gradient line misses the Infinity Cache (plan arrays stay shared)
    regimes = {}
    if world == 1 and not a.no_regimes:
        def rewrite(i):
            with torch.cuda.stream(stream):
                xf.mul_(1.0)
                uf.mul_(1.0)
        if only in ("", "rewritten_inputs"):
            t_rw, _ = time_launches(rewrite)
            t_pair, _ = time_launches(lambda i: (rewrite(i), kernel_only()))
            regimes["rewritten_inputs"] = dict(pair_us=t_pair, rewrite_alone_us=t_rw, kernel_us=t_pair - t_rw)
        if only in ("", "rotating_sets"):
            R = max(2, a.rotating_sets)
            sets = [(xf.clone(), uf.clone(), torch.empty_like(gx_s), torch.empty_like(gu_s)) for _ in range(R)]
            t_rot, _ = time_launches(lambda i: kernel_only(sets[i % R]))
            regimes["rotating_sets"] = dict(kernel_us=t_rot, sets=R,
                                            working_set_mb=round(R * 4 * xf.numel() * 8 / 2 ** 20 + plan.stats["device_bytes"] / 2 ** 20, 1))
            del sets
    if only:
        if rank == 0:
            print(json.dumps(dict(only_regime=only, kernel_us=k_us, regimes=regimes)), flush=True)
        return None
    # algorithmic bytes of ONE launch on this rank: its home elements and owned nodes
    ne_launch = int(td[lo:hi, 1].sum()) if world > 1 else ne      # (halo elements are not algorithmic work)
    if world > 1:
        ne_launch = int(sum(int(plan.tile_elements(t)[2].sum()) for t in range(lo, hi)))
    nn_launch = int(td[lo:hi, 4].sum())
    alg_bytes = 12 * ne_launch + 64 * nn_launch + 8
    achieved = alg_bytes / (k_us * 1e-6) / 1e9
    # HBM traffic per launch: PMC numbers cannot be collected from inside this process; they come from the
    # committed rocprofv3 passes of the SAME kernel/workload (profiles/r02_hbm_traffic_T1M.json, scripts/prof_regimes.sh),
    # else null; likewise the profiler's own per-kernel average of every regime (profiles/r02/regimes_rocprof.json)
    traffic, rocprof_us = None, {}
    try:
        with open(os.path.join(ROOT, "profiles", "r02_hbm_traffic_T1M.json")) as f:
            tr = json.load(f)
        if world == 1 and tr["workload"] == dict(elements=ne, nodes=nn, tiles=plan.stats["n_tiles"]):
            traffic = tr["traffic_bytes_per_launch"]
        with open(os.path.join(ROOT, "profiles", "r02", "regimes_rocprof.json")) as f:
            rp = json.load(f)
        if world == 1 and rp["workload"] == dict(elements=ne, nodes=nn, tiles=plan.stats["n_tiles"]):
            rocprof_us = {k: v["avg_us"] for k, v in rp["regimes"].items()}
    except (OSError, KeyError, ValueError):
        pass
    roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic, kernel="tri3_energy_pair_kernel" if plan.is_paired() else "tri3_energy_fast_kernel", kernel_us=k_us, kernel_us_regions=[round(v, 3) for v in samples],
                    alg_bytes_per_launch=alg_bytes, elems_per_launch=ne_launch, nodes_per_launch=nn_launch)
    for name, r in regimes.items():
        r["achieved"] = alg_bytes / (r["kernel_us"] * 1e-6) / 1e9
        r["frac"] = r["achieved"] / HBM_PEAK_GBS
    if regimes:
        roofline["regimes"] = dict(replayed=dict(kernel_us=k_us, achieved=achieved, frac=achieved / HBM_PEAK_GBS), **regimes)
        if "rewritten_inputs" in regimes:
            regimes["rewritten_inputs"]["note"] = (
                "kernel_us = (rewrite + energy) - (rewrite alone): an UPPER bound -- the two rewrite kernels themselves "
                "slow down inside the pair (rocprofv3: 5.7 us each vs 3.5 us alone); the profiler's own duration of the "
                "energy kernel in this sequence is rocprof_kernel_us")
        for name, r in roofline["regimes"].items():      # the profiler's per-kernel average of the same leg (committed run)
            if name in rocprof_us:
                r["rocprof_kernel_us"] = rocprof_us[name]
                r["rocprof_frac"] = alg_bytes / (rocprof_us[name] * 1e-6) / 1e9 / HBM_PEAK_GBS

    # ---- config.extra (N = 1): the other readings of "1 M quad elements" and BASELINE config 5, kernel only, each with
    #      its own roofline figures (algorithmic bytes of ITS element type over ITS kernel's average launch time):
    #        Q1M   10^6 QUAD4-iso elements (the extension element; parity unpinned by the reference)
    #        T2M   the same 10^6 quads split in two: 2 x 10^6 TRI3 (reference-pinned element)
    #        cfg5  4 x 10^6 TRI3, structured split with random diagonals + random element / node permutation
    #        cfg5u 4.1 x 10^6 TRI3, genuinely unstructured: Delaunay of graded random points, plate with three holes
    extras = []
    if world == 1 and not a.no_extra and not only:
        from hidenn_fem_amd.mesh import structured_quad_mesh, unstructured_tri_mesh

        def extra(name, mesh, quad=False):
            c_, cn_, g_, b_, _, e_ = mesh
            torch.manual_seed(0)
            m_ = PiecewiseLinearShapeNN2D(c_, cn_, boundary_mask=g_, dirichlet_mask=b_, u_fixed=0.0, neumann_edges=e_).to(dev)
            lf_ = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64)
            pl = m_.tile_plan(0)
            x_, u_ = m_.node_coords_free.detach(), m_.u_free.detach()
            xfx, ufx = m_.node_coords_fixed, m_.u_fixed_rows()
            gx_, gu_ = torch.empty_like(x_), torch.empty_like(u_)
            ls_ = torch.zeros((), dtype=f64, device=dev)
            _, Tc_ = lf_._traction(m_, None)
            Tcv = dv(Tc_)

            def launch(i):
                if quad:
                    _lib.check(L.hfem_quad4_energy_plan(pl.handle, x_.data_ptr(), xfx.data_ptr(), u_.data_ptr(), ufx.data_ptr(),
                                                        mat, None, Tcv, 0, -1, ls_.data_ptr(), gx_.data_ptr(), gu_.data_ptr(),
                                                        8, stream.cuda_stream))
                else:
                    _lib.check(L.hfem_tri3_energy_plan(pl.handle, x_.data_ptr(), xfx.data_ptr(), u_.data_ptr(), ufx.data_ptr(),
                                                       mat, W, Bk, None, Tcv, 0, -1, ls_.data_ptr(), gx_.data_ptr(),
                                                       gu_.data_ptr(), 8, stream.cuda_stream))
            us, _ = time_launches(launch)
            ne_, nn_ = cn_.shape[0], c_.shape[0]
            ab = (16 if quad else 12) * ne_ + 64 * nn_ + 8
            st_ = pl.stats
            extras.append(dict(name=name, element="QUAD4" if quad else "TRI3", elements=ne_, nodes=nn_, tiles=st_["n_tiles"],
                               halo_elem_factor=st_["tile_elem_total"] / ne_, halo_node_factor=st_["tile_node_total"] / nn_,
                               kernel_us=us, element_evals_per_s=ne_ / (us * 1e-6), alg_bytes_per_launch=ab,
                               achieved=ab / (us * 1e-6) / 1e9, frac=ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS))
            del m_, pl

        keep = kreps
        kreps = min(kreps, 60)
        extra("Q1M: 10^6 QUAD4-iso (1001 x 1001 nodes), parity unpinned by the reference",
              structured_quad_mesh(1001, 1001, length=2.0, height=2.0, jitter=0.2, seed=0, dtype=f64), quad=True)
        extra("T2M: the same 10^6 quads split in two (2 x 10^6 TRI3)",
              structured_tri_mesh(1001, 1001, length=2.0, height=2.0, jitter=0.2, seed=0, dtype=f64))
        extra("cfg5: 4 x 10^6 TRI3, random diagonals, random element + node permutation",
              structured_tri_mesh(2001, 1001, jitter=0.3, seed=11, diagonal="random", permute=True, dtype=f64))
        from hidenn_fem_amd.mesh import reorder_for_locality
        extra("cfg5r: cfg5 after mesh.reorder_for_locality (Hilbert node renumbering, the host mesh pipeline's step)",
              reorder_for_locality(structured_tri_mesh(2001, 1001, jitter=0.3, seed=11, diagonal="random", permute=True,
                                                       dtype=f64))[0])
        extra("cfg5u: genuinely unstructured (Delaunay, plate with three holes, graded), ~4.1 x 10^6 TRI3",
              unstructured_tri_mesh(2_050_000, seed=2, dtype=f64))
        kreps = keep

    # ---- config.train_step_1gpu (N = 1): the hot path INSIDE an optimiser loop on T1M -- what a training iteration costs
    #      when the kernel's inputs are what the optimiser just wrote.  (i) energy launch + FusedAdam launch (the
    #      reference's `loss.backward(); optimizer.step()`), (ii) one launch: the tiles apply Adam to the rows they own
    #      (hfem_tri3_energy_adam_step).  K iterations in one hipGraph each; lr tiny so the mesh stays valid.
    train1 = None
    if world == 1 and not a.no_extra and not only:
        try:
            from hidenn_fem_amd.optim import FusedAdam, EnergyAdamStep
            from hidenn_fem_amd.graphed import GraphedTraining
            res = {}
            K = max(2, (a.steps // 2) * 2)
            for mode in ("two_launch", "one_launch"):
                torch.manual_seed(0)
                m_ = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                              neumann_edges=edges).to(dev)
                lf_ = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64, tile_elems=a.tile_elems)
                if mode == "two_launch":
                    opt = FusedAdam([dict(params=[m_.node_coords_free], lr=1e-9), dict(params=[m_.u_free], lr=1e-12)],
                                    capturable=True)
                    gt = GraphedTraining(lambda: lf_.value_and_grad_(m_), opt, steps_per_replay=K, direct=True, warmup=2)
                else:
                    tr = EnergyAdamStep(m_, lf_, lr_x=1e-9, lr_u=1e-12)
                    gt = GraphedTraining(tr.step_lagged, None, steps_per_replay=K, direct=True, begin=tr.begin_lagged,
                                         end=tr.flush_loss)
                for _ in range(3):
                    gt.replay()
                torch.cuda.synchronize()
                t_pw = time.perf_counter()
                while time.perf_counter() - t_pw < a.prewarm:
                    gt.replay()
                    torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    gt.replay()
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0_) / K)
                it = sorted(ts)[2]
                res[mode] = dict(us_per_iteration=it * 1e6, element_evals_per_s=ne / it)
                del gt, m_
            train1 = dict(workload="T1M, Adam on node_coords_free and u_free, K iterations per hipGraph", **res)
        except Exception as e:  # pragma: no cover
            print(f"[bench] train_step_1gpu leg failed: {e}", file=sys.stderr)

    out = None
    if rank == 0:
        cpu = None
        if not a.no_cpu_baseline:
            if world == 1:
                cpu, loss_cpu = cpu_baseline(mesh6, model.u_free.detach().cpu(), a.cpu_evals)
                rel = abs(loss_cpu - loss_gpu) / abs(loss_cpu)
                assert rel <= 1e-12, f"GPU loss {loss_gpu!r} != oracle loss {loss_cpu!r} (rel {rel:.2e})"
                cpu["loss_rel_err_vs_gpu"] = rel
        st = plan.stats
        out = dict(
            metric="element-evals/sec (fwd+bwd energy) + achieved HBM GB/s, 2D quad mesh",
            value=value, unit="element-evals/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
            ms_per_step=ms_per_step, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64",
            data="synthetic",
            config=dict(workload=f"Example 4 (T1M x {world}): 2D plate linear elasticity, {ne} TRI3 "
                                 f"(structured quads split in 2), {nn} nodes, gauss_order=4, r-adaptivity on, "
                                 f"Neumann edges {edges.shape[0]}, fwd+bwd (loss + dX + dU)",
                        elements=ne, nodes=nn, elements_per_gpu=ne // world, tiles=st["n_tiles"],
                        tile_elems=st["tile_elems"], element_order="paired slots" if plan.is_paired() else "one element per slot",
                        halo_elem_factor=sum(len(plan.tile_elements(t)[0]) for t in range(st["n_tiles"])) / max(ne, 1),
                        lds_bytes=st["lds_bytes"], launch="hipgraph" if graph is not None else "eager",
                        loss_sum=("by an extra workgroup of the next launch (HFEM_FLAG_SUM_PREVIOUS) + one trailing "
                                  "1-block launch" if lagged else "1-block launch after every energy kernel"),
                        exchange="none" if world == 1 else
                        f"owner-sharded: gradient rows stay with the rank whose tiles own the node; one all_gather per "
                        f"step of interface parameter rows + partial energy ({sh.interface_stats['payload_bytes']} B "
                        f"per rank), " + ("in-library RCCL (hfem_mg_allgather) on the kernel stream" if comm is not None
                                          else f"torch.distributed {a.backend}"),
                        loss=loss_gpu),
            roofline=roofline,
        )
        if extras:
            out["config"]["extra"] = extras
        if inline_step is not None:
            out["config"]["inline_loss_step"] = inline_step
        if train1 is not None:
            out["config"]["train_step_1gpu"] = train1
        if alt is not None:
            out["config"]["alt_exchange"] = alt
        if train is not None:
            out["config"]["train_step"] = train
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()

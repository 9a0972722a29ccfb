#!/usr/bin/env python3
"""Dev tool (GPU box): A/B timing of two builds of the library (build.py --tag NAME --define ...), alternating child
processes on the same box: the T1M paired kernel (same buffers / rotating sets), its fp32-arithmetic instance and the Q1M
QUAD4 kernel, kernel only (hipGraph of K launches between HIP events, median of 5).

    python scripts/ab_lib.py libhidenn_hip.so libhidenn_hip_noprio.so [more .so ...] [rounds]
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(lib_name):
    import torch
    from hidenn_fem_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "hidenn_fem_amd", "csrc", lib_name)
    from hidenn_fem_amd.loss import EnergyLoss2D
    from hidenn_fem_amd.mesh import structured_quad_mesh, structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    K = 100
    dev, f64 = torch.device("cuda:0"), torch.float64
    L = _lib.lib()
    dv = lambda v: (C.c_double * len(v))(*v)

    def timed(launch, rot):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            launch(0, s.cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(K):
                launch(k % rot, torch.cuda.current_stream().cuda_stream)
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

    def case(mesh, dtype, quad, R, flags):
        c_, cn_, g_, b_, _, e_ = mesh
        torch.manual_seed(0)
        m = PiecewiseLinearShapeNN2D(c_.to(dtype), cn_, boundary_mask=g_, dirichlet_mask=b_, u_fixed=0.0, neumann_edges=e_).to(dev)
        pl = m.tile_plan(0)
        lf = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=dtype)
        _, Tc = lf._traction(m, None)
        mat, Tcv, Bk = dv(lf._mat), dv(Tc), dv([0.0] * 6)
        x, u = m.node_coords_free.detach(), m.u_free.detach()
        xf, uf = m.node_coords_fixed, m.u_fixed_rows()
        ls = torch.zeros((), dtype=f64, device=dev)
        sets = [(x.clone(), u.clone(), torch.empty_like(x), torch.empty_like(u)) for _ in range(R)]

        def launch(i, stream):
            xs, us, gxs, gus = sets[i]
            if quad:
                _lib.check(L.hfem_quad4_energy_plan(pl.handle, xs.data_ptr(), xf.data_ptr(), us.data_ptr(), uf.data_ptr(), mat, None, Tcv,
                                                    0, -1, ls.data_ptr(), gxs.data_ptr(), gus.data_ptr(), flags, stream))
            elif dtype == torch.float32:
                _lib.check(L.hfem_tri3_energy_plan_f32(pl.handle, xs.data_ptr(), xf.data_ptr(), us.data_ptr(), uf.data_ptr(), mat, lf._W, Bk,
                                                       None, Tcv, 0, -1, ls.data_ptr(), gxs.data_ptr(), gus.data_ptr(), flags, stream))
            else:
                _lib.check(L.hfem_tri3_energy_plan(pl.handle, xs.data_ptr(), xf.data_ptr(), us.data_ptr(), uf.data_ptr(), mat, lf._W, Bk,
                                                   None, Tcv, 0, -1, ls.data_ptr(), gxs.data_ptr(), gus.data_ptr(), flags, stream))
        return timed(launch, 1), (timed(launch, R) if R > 1 else None)

    def fused_step(mesh):
        """the one-launch training iteration (energy + Adam at write-out), K per hipGraph, wall clock per iteration"""
        import time
        from hidenn_fem_amd.graphed import GraphedTraining
        from hidenn_fem_amd.optim import EnergyAdamStep
        c_, cn_, g_, b_, _, e_ = mesh
        torch.manual_seed(0)
        m = PiecewiseLinearShapeNN2D(c_, cn_, boundary_mask=g_, dirichlet_mask=b_, u_fixed=0.0, neumann_edges=e_).to(dev)
        lf = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=4, device=dev, dtype=f64)
        tr = EnergyAdamStep(m, lf, lr_x=1e-9, lr_u=1e-12)
        gt = GraphedTraining(tr.step_lagged, None, steps_per_replay=K, direct=True, begin=tr.begin_lagged, end=tr.flush_loss)
        for _ in range(20):
            gt.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            gt.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e6)
        return sorted(ts)[2]

    t1m = structured_tri_mesh(1001, 501, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64)
    a, b = case(t1m, f64, False, 10, 8)
    f = fused_step(t1m)
    t2, _ = case(structured_tri_mesh(1001, 1001, length=2.0, height=2.0, jitter=0.2, seed=0, dtype=f64), f64, False, 1, 8)
    c, _ = case(t1m, torch.float32, False, 1, 8 | 1024)
    q1m = structured_quad_mesh(1001, 1001, length=2.0, height=2.0, jitter=0.2, seed=0, dtype=f64)
    d, e = case(q1m, f64, True, 6, 8)
    big = {}
    if os.environ.get("AB_BIG") == "1":        # the 4 x 10^6-element extras of bench.py (several resident rounds per launch)
        cfg5 = structured_tri_mesh(2001, 1001, jitter=0.3, seed=11, diagonal="random", permute=True, dtype=f64)
        r, ro = case(cfg5, f64, False, 3, 8)
        big = dict(cfg5auto_replayed_us=r, cfg5auto_rotating_us=ro)
    if os.environ.get("AB_UNPAIRED") == "1":   # one element per slot (tri3_energy_fast_kernel): zigzag diagonals at T1M size, Delaunay at 4 M
        z, zo = case(structured_tri_mesh(1001, 501, length=2.0, height=1.0, jitter=0.2, seed=0, dtype=f64, diagonal="zigzag"), f64, False, 10, 8)
        from hidenn_fem_amd.mesh import unstructured_tri_mesh
        dl, dlo = case(unstructured_tri_mesh(2_050_000, seed=2, dtype=f64), f64, False, 3, 8)
        big.update(zigzag_replayed_us=z, zigzag_rotating_us=zo, cfg5u_replayed_us=dl, cfg5u_rotating_us=dlo)
    print(json.dumps(dict(lib=lib_name, **big, t1m_replayed_us=a, t1m_rotating_us=b, t1m_adam_step_us=f, t2m_replayed_us=t2, t1m_fp32_us=c, q1m_replayed_us=d, q1m_rotating_us=e)), flush=True)


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2])
    libs = [v for v in sys.argv[1:] if v.endswith(".so")]
    rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
    os.environ.setdefault("HFEM_PLAN_CACHE", "/tmp/hfem_plan_cache")
    os.makedirs(os.environ["HFEM_PLAN_CACHE"], exist_ok=True)
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            out = subprocess.run([sys.executable, __file__, "--child", l], capture_output=True, text=True, timeout=600)
            if out.returncode != 0:
                print(out.stderr[-2000:], file=sys.stderr)
                return 1
            line = out.stdout.strip().splitlines()[-1]
            print(line, flush=True)
            res[l].append(json.loads(line))
    med = lambda v: sorted(v)[len(v) // 2]
    for l in libs:
        print(json.dumps(dict(lib=l, median={k: round(med([r[k] for r in res[l]]), 3) for k in res[l][0] if k != "lib"})), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())

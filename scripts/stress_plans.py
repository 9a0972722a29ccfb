#!/usr/bin/env python3
"""Dev tool (GPU box): hfem_tri3_energy_plan against the closed-form oracle over 6 meshes (fixed / random / zigzag diagonals,
permuted numbering, flipped elements, Delaunay) x element orders 3 / 5 / 6 x tile sizes 0 / 48 / 333 / 1500 x slot packings
(plan_read_pack 0 / 2): 144 plans, body force and traction on."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hidenn_fem_amd import _lib
from hidenn_fem_amd.plan import TilePlan
from hidenn_fem_amd.mesh import structured_tri_mesh, unstructured_tri_mesh
from oracle import closed_form as CF
d = torch.device("cuda:0"); L = _lib.lib(); s = _lib.stream_ptr(d)
dv = lambda a: (C.c_double * len(a))(*a)
mat, W = CF.plane_stress(), 0.25
rng = np.random.default_rng(0)
n = 0
for mi, mk in enumerate([dict(nx=37, ny=29), dict(nx=130, ny=77, diagonal="random"), dict(nx=90, ny=140, diagonal="zigzag"),
                         dict(nx=101, ny=64, diagonal="random", permute=True), dict(nx=150, ny=151, flip_fraction=0.3), "delaunay"]):
    if mk == "delaunay":
        coords, conn, geom, bc, mnn, edges = unstructured_tri_mesh(9000, seed=3, dtype=torch.float64)
    else:
        coords, conn, geom, bc, mnn, edges = structured_tri_mesh(jitter=0.25, seed=mi, dtype=torch.float64, **mk)
    X, cn, ed = coords.numpy(), conn.numpy(), edges.numpy()
    U = 1e-4 * rng.standard_normal(X.shape)
    Bk = rng.standard_normal(6) * 1e4; Tc = np.array([2e5, 0.0, 0.0, 1e4])
    e_ref, gX_ref, gU_ref = CF.tri3_energy(X, U, cn, mat, W, Bk)
    e_ref -= CF.edge2_energy(X, U, ed, Tconst=Tc, gX=gX_ref, gU=gU_ref)
    Xd, Ud = torch.from_numpy(X).to(d), torch.from_numpy(U).to(d)
    for order in (3, 5, 6):
        for T in (0, 48, 333, 1500):
            for rp in (0, 2):
                _lib.check(L.hfem_set_option(b"plan_read_pack", rp))
                plan = TilePlan(cn, X.shape[0], coords_hint=X, edges=ed, tile_elems=T, device=d, elem_order=order)
                loss = torch.full((), 7.0, dtype=torch.float64, device=d)
                gX, gU = torch.full_like(Xd, float("nan")), torch.full_like(Ud, float("nan"))
                _lib.check(L.hfem_tri3_energy_plan(plan.handle, Xd.data_ptr(), None, Ud.data_ptr(), None, dv(mat), W, dv(Bk), None, dv(Tc), 0, -1,
                                                   loss.data_ptr(), gX.data_ptr(), gU.data_ptr(), 0, s))
                torch.cuda.synchronize()
                assert abs(loss.item() - e_ref) <= 1e-12 * abs(e_ref), (mi, order, T, rp, loss.item(), e_ref)
                assert np.abs(gX.cpu().numpy() - gX_ref).max() <= 1e-10 * np.abs(gX_ref).max(), (mi, order, T, rp)
                assert np.abs(gU.cpu().numpy() - gU_ref).max() <= 1e-10 * np.abs(gU_ref).max(), (mi, order, T, rp)
                plan.close(); n += 1
_lib.check(L.hfem_set_option(b"plan_read_pack", 2))
print("stress ok", n, "plans")

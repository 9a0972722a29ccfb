"""Pin the plain-C closed forms (oracle/hfem_oracle.c -- the arithmetic the HIP
kernels implement) against the golden vectors generated from the imported
reference.  Closed forms re-associate the reference's per-Gauss-point sums
(W = sum w_q once per element), so the bar is the fp64 tolerance of BASELINE.md:
loss rel <= 1e-12, gradients max-abs <= 1e-10 * max|g| (observed ~1e-15)."""
import numpy as np
import torch

from oracle import closed_form as CF
from oracle import ref_chain as R
from conftest import tri_mesh_dict, tri_case_forces

LOSS_RTOL = 1e-12
GRAD_RTOL = 1e-10


def _assembled(g, case):
    mesh, xf, uf = tri_mesh_dict(g, case)
    X = R.assemble_coords(mesh["n_nodes"], mesh["free_mask"], xf, mesh["boundary_mask"], mesh["coords_fixed"]).numpy()
    U = R.assemble_u(mesh["n_nodes"], mesh["u_free_mask"], uf, mesh["dirichlet_mask"], mesh["u_fixed"]).numpy()
    return mesh, X, U


def body_table(b_force, order):
    """B_k = sum_q w_q N_k(xi_q) b(xi_q): what the fused kernel gets instead of a callable."""
    xg, wg = R.triangle_gauss(order)
    if b_force is None:
        return np.zeros((3, 2))
    b = b_force(xg)
    N = torch.stack([xg[:, 0], xg[:, 1], 1.0 - xg[:, 0] - xg[:, 1]], dim=1)      # [ng,3]
    return torch.einsum("q,qk,qi->ki", wg, N, b).numpy()


def traction_table(t_force, X, edges, order1):
    """Per-edge {Ti, Tj} = sum_q w_q {(1-xi_q), xi_q} t(x_q) with raw Legendre xi (F3)."""
    xg1, wg1 = R.interval_gauss(order1)
    if t_force is None:
        t = np.array([1.0e5, 0.0])                                            # loss.py:47-51
        ci, cj = float((wg1 * (1 - xg1)).sum()), float((wg1 * xg1).sum())
        return None, np.concatenate([ci * t, cj * t])
    Xt = torch.from_numpy(X)
    x_i, x_j = Xt[edges[:, 0]], Xt[edges[:, 1]]
    xq = (1.0 - xg1[None, :, None]) * x_i[:, None, :] + xg1[None, :, None] * x_j[:, None, :]
    tq = t_force(xq.reshape(-1, 2)).reshape(edges.shape[0], -1, 2)
    Ti = torch.einsum("q,eqi->ei", wg1 * (1 - xg1), tq)
    Tj = torch.einsum("q,eqi->ei", wg1 * xg1, tq)
    return torch.cat([Ti, Tj], dim=1).numpy(), None


def closed_form_energy(g, case):
    mesh, X, U = _assembled(g, case)
    go, go1 = (int(v) for v in g[case + "/gauss_order"])
    b, t = tri_case_forces(case)
    W = float(R.triangle_gauss(go)[1].sum())
    dom, gX, gU = CF.tri3_energy(X, U, mesh["conn"].numpy(), CF.plane_stress(), W, body_table(b, go))
    T, Tc = traction_table(t, X, mesh["edges"], go1)
    edg = CF.edge2_energy(X, U, mesh["edges"].numpy(), T=T, Tconst=Tc, gX=gX, gU=gU)
    return mesh, dom, edg, gX, gU


def test_closed_form_matches_reference_golden(g_tri):
    for case in g_tri.cases():
        mesh, dom, edg, gX, gU = closed_form_energy(g_tri, case)
        want = g_tri[case + "/loss"].item()
        assert abs(dom - g_tri[case + "/domain"].item()) <= LOSS_RTOL * abs(g_tri[case + "/domain"].item()), case
        assert abs(edg - g_tri[case + "/edge"].item()) <= LOSS_RTOL * max(abs(g_tri[case + "/edge"].item()), 1e-300), case
        assert abs((dom - edg) - want) <= LOSS_RTOL * abs(want), case
        gu_ref, gx_ref = g_tri[case + "/g_u_free"], g_tri[case + "/g_coords_free"]
        gu = gU[mesh["u_free_mask"].numpy()]
        gx = gX[mesh["free_mask"].numpy()]
        assert np.abs(gu - gu_ref).max() <= GRAD_RTOL * np.abs(gu_ref).max(), case
        assert np.abs(gx - gx_ref).max() <= GRAD_RTOL * np.abs(gx_ref).max(), case


def test_closed_form_per_point_eval(g_tri):
    case = "order4"
    mesh, X, U = _assembled(g_tri, case)
    p = case + "/pp_"
    u_h, detJ, grad_u = CF.tri3_eval(X, U, mesh["conn"].numpy(), g_tri[p + "x_eval"], g_tri[p + "elem_id"])
    np.testing.assert_allclose(u_h, g_tri[p + "u_h"], rtol=1e-13, atol=1e-20)
    np.testing.assert_allclose(detJ, g_tri[p + "detJ"], rtol=1e-13)
    np.testing.assert_allclose(grad_u, g_tri[p + "grad_u"], rtol=1e-11, atol=1e-16)


def test_closed_form_line2_and_grid(g_line):
    for name in ("free", "dir0", "dirN", "dir"):
        p = f"line_{name}/"
        xg, inc = g_line[p + "x_grid"], g_line[p + "incr"]
        grid = CF.grid_param_fwd(inc, xg[0], xg[-1])
        np.testing.assert_allclose(grid, g_line[p + "grid"], rtol=1e-14, atol=1e-15)
        bc = g_line[p + "bc"]
        u_full = np.concatenate(([np.float32(bc[0])] if not np.isnan(bc[0]) else [], g_line[p + "u"],
                                 [np.float32(bc[1])] if not np.isnan(bc[1]) else []))
        pred, gg, gu, gx = CF.line2(g_line[p + "grid"], u_full, g_line[p + "x_eval"], g_line[p + "cot"])
        np.testing.assert_allclose(pred, g_line[p + "pred"], rtol=1e-12, atol=1e-16)
        lo = 0 if np.isnan(bc[0]) else 1
        hi = len(u_full) if np.isnan(bc[1]) else len(u_full) - 1
        np.testing.assert_allclose(gu[lo:hi], g_line[p + "g_u"], rtol=1e-11, atol=1e-15)
        np.testing.assert_allclose(gx, g_line[p + "g_x_eval"], rtol=1e-11, atol=1e-15)
        gp = CF.grid_param_bwd(inc, xg[0], xg[-1], gg)
        ref = g_line[p + "g_incr"]
        assert np.abs(gp - ref).max() <= 1e-10 * np.abs(ref).max()


def test_closed_form_bar_energy(g_line):
    xi, wi = (t.numpy() for t in R.interval_gauss(2))
    for tag in ("n89", "n1001"):
        p = f"ex3_{tag}/"
        xg, inc = g_line[p + "x_grid"], g_line[p + "incr"]
        grid = CF.grid_param_fwd(inc, xg[0], xg[-1])
        u_full = np.concatenate(([0.0], g_line[p + "u"], [0.0]))
        x_i, x_j = grid[:-1, None], grid[1:, None]
        xq = 0.5 * (x_j - x_i) * xi + 0.5 * (x_j + x_i)
        wq = 0.5 * (x_j - x_i) * wi
        bq = R.example3_body_force(torch.from_numpy(xq)).numpy()
        e, gg, gu = CF.bar_energy(grid, u_full, xq, wq, bq, 175.0)
        want = g_line[p + "loss"].item()
        assert abs(e - want) <= 1e-12 * abs(want)
        ref = g_line[p + "g_u"]
        assert np.abs(gu[1:-1] - ref).max() <= 1e-10 * np.abs(ref).max()
        gp = CF.grid_param_bwd(inc, xg[0], xg[-1], gg)
        ref = g_line[p + "g_incr"]
        assert np.abs(gp - ref).max() <= 1e-10 * np.abs(ref).max()


def test_closed_form_rectq4(g_rect):
    for case in g_rect.cases():
        p = case + "/"
        gx, gy = g_rect[p + "gx_full"], g_rect[p + "gy_full"]
        u = g_rect[p + "u"].copy()
        uf = g_rect[p + "u_fixed"][0]
        mask = g_rect[p + "mask_x"][:, None] | g_rect[p + "mask_y"][None, :]
        if not np.isnan(uf):
            u[mask] = np.float32(uf)
        pred, ggx, ggy, gu, gxe = CF.rectq4(gx, gy, u, g_rect[p + "x_eval"], g_rect[p + "cot"])
        np.testing.assert_allclose(pred, g_rect[p + "pred"], rtol=1e-12, atol=1e-15)
        if not np.isnan(uf):
            gu[mask] = 0.0                                   # where(node_mask, u_fixed, u): no grad
        np.testing.assert_allclose(gu, g_rect[p + "g_u"], rtol=1e-11, atol=1e-15)
        np.testing.assert_allclose(gxe, g_rect[p + "g_x_eval"], rtol=1e-10, atol=1e-13)
        if g_rect[p + "r_adapt"][0]:
            x0 = g_rect[p + "grid_x"]
            y0 = g_rect[p + "grid_y"]
            ggx[g_rect[p + "mask_x"]] = 0.0                  # torch.where(mask, initial, grid)
            ggy[g_rect[p + "mask_y"]] = 0.0
            gpx = CF.grid_param_bwd(g_rect[p + "incr_x"], x0[0], x0[-1], ggx)
            gpy = CF.grid_param_bwd(g_rect[p + "incr_y"], y0[0], y0[-1], ggy)
            rx, ry = g_rect[p + "g_incr_x"], g_rect[p + "g_incr_y"]
            assert np.abs(gpx - rx).max() <= 1e-10 * max(np.abs(rx).max(), 1e-300)
            assert np.abs(gpy - ry).max() <= 1e-10 * max(np.abs(ry).max(), 1e-300)

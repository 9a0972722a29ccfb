"""Pin the oracle (oracle/ref_chain.py: CPU/PyTorch restatement of the reference op
chain) against golden vectors produced by the imported reference itself
(tests/golden/make_golden.py).  Same ATen ops in the same order -> the bar here is
bit-exact (atol=rtol=0) for the TRI3/EDGE2 chain and 1D/structured forward; a few
ulp where torch's autograd accumulation order is not contractually fixed."""
import numpy as np
import pytest
import torch

from oracle import ref_chain as R
from conftest import tri_mesh_dict, tri_case_forces, tri_case_forces_f32

F64 = torch.float64


def test_quadrature_tables_bit_exact(g_quad):
    for o in (1, 3, 4, 6, 7):
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            rs, w = R.triangle_gauss(o, dt)
            assert np.array_equal(rs.numpy(), g_quad[f"tri{o}_{tag}_rs"])
            assert np.array_equal(w.numpy(), g_quad[f"tri{o}_{tag}_w"])
    for o in (1, 2, 3, 4, 5):
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            x, w = R.interval_gauss(o, dt)
            assert np.array_equal(x.numpy(), g_quad[f"gl{o}_{tag}_x"])
            assert np.array_equal(w.numpy(), g_quad[f"gl{o}_{tag}_w"])
    # the reference's quirks are part of the contract
    assert abs(R.triangle_gauss(4)[1].sum().item() - 0.25) < 1e-15
    assert abs(R.triangle_gauss(6)[1].sum().item() - 0.25) < 1e-12
    assert abs(R.triangle_gauss(3)[1].sum().item() - 0.5) < 1e-15
    assert abs(R.interval_gauss(2)[1].sum().item() - 2.0) < 1e-15
    with pytest.raises(NotImplementedError):
        R.triangle_gauss(2)


def test_tri3_energy_and_grads_match_reference(g_tri):
    cases = g_tri.cases()
    assert len(cases) >= 16
    for case in cases:
        mesh, xf, uf = tri_mesh_dict(g_tri, case)
        go, go1 = (int(v) for v in g_tri[case + "/gauss_order"])
        b, t = tri_case_forces(case)
        loss, gx, gu = R.energy_and_grads(xf, uf, mesh, gauss_order=go, gauss_order_1d=go1,
                                          b_force=b, t_force=t)
        assert loss.item() == g_tri[case + "/loss"].item(), case
        np.testing.assert_allclose(gu.numpy(), g_tri[case + "/g_u_free"], rtol=0, atol=0, err_msg=case)
        np.testing.assert_allclose(gx.numpy(), g_tri[case + "/g_coords_free"], rtol=0, atol=0, err_msg=case)


def test_tri3_fp32_as_shipped_matches_reference(g_tri_f32):
    """The reference's DEFAULT dtype (src/loss.py:16, src/models.py:274): the oracle's op chain in fp32 against the reference
    run in fp32, and in fp64 on the same float inputs against the reference's fp64 run -- same ATen ops in the same order."""
    cases = g_tri_f32.cases()
    assert len(cases) >= 9
    worst = 0.0
    for case in cases:
        mesh, xf, uf = tri_mesh_dict(g_tri_f32, case)
        assert xf.dtype == torch.float32 and uf.dtype == torch.float32
        go, go1 = (int(v) for v in g_tri_f32[case + "/gauss_order"])
        b, t = tri_case_forces_f32(case)
        mesh32 = dict(mesh, u_fixed=torch.tensor(0.0, dtype=torch.float32))
        loss, gx, gu = R.energy_and_grads(xf, uf, mesh32, gauss_order=go, gauss_order_1d=go1, b_force=b, t_force=t)
        assert loss.dtype == torch.float32
        assert loss.item() == g_tri_f32[case + "/loss"].item(), case
        np.testing.assert_allclose(gu.numpy(), g_tri_f32[case + "/g_u_free"], rtol=0, atol=0, err_msg=case)
        np.testing.assert_allclose(gx.numpy(), g_tri_f32[case + "/g_coords_free"], rtol=0, atol=0, err_msg=case)
        mesh64 = dict(mesh, coords_fixed=mesh["coords_fixed"].double())
        l64, gx64, gu64 = R.energy_and_grads(xf.double(), uf.double(), mesh64, gauss_order=go, gauss_order_1d=go1, b_force=b, t_force=t)
        assert l64.item() == g_tri_f32[case + "/loss64"].item(), case
        np.testing.assert_allclose(gu64.numpy(), g_tri_f32[case + "/g_u_free64"], rtol=0, atol=0, err_msg=case)
        np.testing.assert_allclose(gx64.numpy(), g_tri_f32[case + "/g_coords_free64"], rtol=0, atol=0, err_msg=case)
        worst = max(worst, float(np.abs(gx.numpy() - gx64.numpy()).max() / np.abs(gx64.numpy()).max()))
    # what fp32 arithmetic costs the reference itself on these meshes: the tolerance the fp32 GPU tests are held to
    assert 1e-8 < worst < 2e-6, worst


def test_tri3_domain_edge_split(g_tri):
    for case in ("order4", "order4_body", "traction_fn", "flipped"):
        mesh, xf, uf = tri_mesh_dict(g_tri, case)
        go, go1 = (int(v) for v in g_tri[case + "/gauss_order"])
        b, t = tri_case_forces(case)
        coords = R.assemble_coords(mesh["n_nodes"], mesh["free_mask"], xf, mesh["boundary_mask"], mesh["coords_fixed"])
        u = R.assemble_u(mesh["n_nodes"], mesh["u_free_mask"], uf, mesh["dirichlet_mask"], mesh["u_fixed"])
        C = R.plane_stress_C()
        dom = R.domain_energy(coords, u, mesh["conn"], C, *R.triangle_gauss(go), b_force=b)
        edg = R.edge_energy(coords, u, mesh["edges"], *R.interval_gauss(go1), t_force=t)
        assert dom.item() == g_tri[case + "/domain"].item()
        assert edg.item() == g_tri[case + "/edge"].item()


def test_tri3_unfused_forward_backward(g_tri):
    case = "order4"
    mesh, xf, uf = tri_mesh_dict(g_tri, case)
    xf.requires_grad_(True)
    uf.requires_grad_(True)
    coords = R.assemble_coords(mesh["n_nodes"], mesh["free_mask"], xf, mesh["boundary_mask"], mesh["coords_fixed"])
    u = R.assemble_u(mesh["n_nodes"], mesh["u_free_mask"], uf, mesh["dirichlet_mask"], mesh["u_fixed"])
    p = case + "/pp_"
    u_h, detJ, grad_u = R.tri3_forward(coords, u, mesh["conn"], g_tri.t(p + "x_eval"), g_tri.t(p + "elem_id"))
    assert np.array_equal(u_h.detach().numpy(), g_tri[p + "u_h"])
    assert np.array_equal(detJ.detach().numpy(), g_tri[p + "detJ"])
    assert np.array_equal(grad_u.detach().numpy(), g_tri[p + "grad_u"])
    ((u_h * g_tri.t(p + "cu")).sum() + (detJ * g_tri.t(p + "cd")).sum()
     + (grad_u * g_tri.t(p + "cg")).sum()).backward()
    np.testing.assert_allclose(uf.grad.numpy(), g_tri[p + "g_u_free"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(xf.grad.numpy(), g_tri[p + "g_coords_free"], rtol=1e-14, atol=1e-18)
    q = case + "/pe_"
    ue, ds = R.edge2_forward(coords.detach(), u.detach(), mesh["edges"], g_tri.t(q + "x_eval"), g_tri.t(q + "edge_id"))
    assert np.array_equal(ue.numpy(), g_tri[q + "u_h"])
    assert np.array_equal(ds.numpy(), g_tri[q + "ds"])


def _line_u_full(u, bc):
    # the reference stores the Dirichlet end values as float32 buffers (models.py:25,30);
    # ``.double()`` then widens the rounded value -- 0.1 becomes 0.10000000149...
    parts = []
    if not np.isnan(bc[0]):
        parts.append(torch.tensor([bc[0]], dtype=torch.float32).to(u.dtype))
    parts.append(u)
    if not np.isnan(bc[1]):
        parts.append(torch.tensor([bc[1]], dtype=torch.float32).to(u.dtype))
    return torch.cat(parts)                       # models.py:58-67


def test_line2_forward_backward(g_line):
    for name in ("free", "dir0", "dirN", "dir"):
        p = f"line_{name}/"
        xg = g_line.t(p + "x_grid")
        u = g_line.t(p + "u").requires_grad_(True)
        inc = g_line.t(p + "incr").requires_grad_(True)
        xe = g_line.t(p + "x_eval").requires_grad_(True)
        grid = R.grid_param(inc, xg[0:1], xg[-1:])
        assert np.array_equal(grid.detach().numpy(), g_line[p + "grid"])
        pred = R.line2_forward(grid, _line_u_full(u, g_line[p + "bc"]), xe)
        assert np.array_equal(pred.detach().numpy(), g_line[p + "pred"])
        (pred * g_line.t(p + "cot")).sum().backward()
        np.testing.assert_allclose(u.grad.numpy(), g_line[p + "g_u"], rtol=1e-14, atol=1e-18)
        np.testing.assert_allclose(inc.grad.numpy(), g_line[p + "g_incr"], rtol=1e-13, atol=1e-18)
        np.testing.assert_allclose(xe.grad.numpy(), g_line[p + "g_x_eval"], rtol=1e-14, atol=1e-18)
    p = "line_fixed/"
    xg = g_line.t(p + "x_grid")
    u = g_line.t(p + "u").requires_grad_(True)
    xe = g_line.t(p + "x_eval").requires_grad_(True)
    pred = R.line2_forward(xg, _line_u_full(u, np.array([0.1, np.nan])), xe)
    assert np.array_equal(pred.detach().numpy(), g_line[p + "pred"])
    (pred * g_line.t(p + "cot")).sum().backward()
    np.testing.assert_allclose(u.grad.numpy(), g_line[p + "g_u"], rtol=1e-14, atol=1e-18)
    np.testing.assert_allclose(xe.grad.numpy(), g_line[p + "g_x_eval"], rtol=1e-14, atol=1e-18)


def test_example1_adam_trajectory(g_line):
    """examples/example1.py:25-42 driven through the oracle functions + torch Adam."""
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        for r_adapt in (True, False):
            xg = torch.linspace(0, 1, 100, dtype=dt)
            xt = torch.linspace(0, 1, 1000, dtype=dt)
            ut = torch.sin(2 * torch.pi * xt)
            u = torch.zeros(100, dtype=dt, requires_grad=True)
            params = [u]
            if r_adapt:
                inc = (xg[1:] - xg[:-1]).clone().requires_grad_(True)
                params = [inc, u]
            opt = torch.optim.Adam(params, lr=0.005)
            got = []
            for _ in range(20):
                opt.zero_grad()
                grid = R.grid_param(inc, xg[0:1], xg[-1:]) if r_adapt else xg
                loss = R.mse_loss(R.line2_forward(grid, u, xt), ut)
                loss.backward()
                opt.step()
                got.append(loss.item())
            want = g_line[f"ex1_{tag}_r{int(r_adapt)}/adam_losses"]
            np.testing.assert_allclose(got, want, rtol=1e-12 if dt == F64 else 1e-5)


def test_example3_bar_energy(g_line):
    xi, wi = R.interval_gauss(2)
    for tag in ("n89", "n1001"):
        p = f"ex3_{tag}/"
        xg = g_line.t(p + "x_grid")
        u = g_line.t(p + "u").requires_grad_(True)
        inc = g_line.t(p + "incr").requires_grad_(True)
        grid = R.grid_param(inc, xg[0:1], xg[-1:])
        zero = torch.zeros(1, dtype=F64)
        loss = R.bar_energy(grid, torch.cat([zero, u, zero]), xi, wi, R.example3_body_force, 175.0)
        loss.backward()
        assert loss.item() == g_line[p + "loss"].item()
        np.testing.assert_allclose(u.grad.numpy(), g_line[p + "g_u"], rtol=1e-13, atol=1e-18)
        np.testing.assert_allclose(inc.grad.numpy(), g_line[p + "g_incr"], rtol=1e-12, atol=1e-16)


def test_rectq4_forward_backward(g_rect):
    for case in g_rect.cases():
        p = case + "/"
        gx0, gy0 = g_rect.t(p + "grid_x"), g_rect.t(p + "grid_y")
        u = g_rect.t(p + "u").requires_grad_(True)
        r_adapt = bool(g_rect[p + "r_adapt"][0])
        mx, my = g_rect.t(p + "mask_x"), g_rect.t(p + "mask_y")
        if r_adapt:
            ix = g_rect.t(p + "incr_x").requires_grad_(True)
            iy = g_rect.t(p + "incr_y").requires_grad_(True)
            gx = R.masked_grid(R.grid_param(ix, gx0[0:1], gx0[-1:]), mx, gx0)
            gy = R.masked_grid(R.grid_param(iy, gy0[0:1], gy0[-1:]), my, gy0)
        else:
            gx, gy = R.masked_grid(gx0, mx, gx0), R.masked_grid(gy0, my, gy0)
        assert np.array_equal(gx.detach().numpy(), g_rect[p + "gx_full"])
        assert np.array_equal(gy.detach().numpy(), g_rect[p + "gy_full"])
        uf = g_rect[p + "u_fixed"][0]
        node_mask = mx[:, None] | my[None, :]                     # models.py:134
        # u_fixed is a float32 buffer widened by .double() (models.py:137): 0.3 -> 0.30000001192...
        u_full = u if np.isnan(uf) else torch.where(node_mask, torch.tensor([uf], dtype=torch.float32).to(F64), u)
        xe = g_rect.t(p + "x_eval").requires_grad_(True)
        pred = R.rectq4_forward(gx, gy, u_full, xe)
        assert np.array_equal(pred.detach().numpy(), g_rect[p + "pred"]), case
        (pred * g_rect.t(p + "cot")).sum().backward()
        np.testing.assert_allclose(u.grad.numpy(), g_rect[p + "g_u"], rtol=1e-14, atol=1e-18)
        np.testing.assert_allclose(xe.grad.numpy(), g_rect[p + "g_x_eval"], rtol=1e-13, atol=1e-16)
        if r_adapt:
            np.testing.assert_allclose(ix.grad.numpy(), g_rect[p + "g_incr_x"], rtol=1e-12, atol=1e-16)
            np.testing.assert_allclose(iy.grad.numpy(), g_rect[p + "g_incr_y"], rtol=1e-12, atol=1e-16)

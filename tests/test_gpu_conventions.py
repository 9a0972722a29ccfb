"""The two opt-in switches of the TRI3 energy path (flags of hfem_tri3_energy_plan):
HFEM_FLAG_DETERMINISTIC (fixed-order accumulation: bit-identical run to run) and HFEM_FLAG_PHYSICAL_GRAD
(grad_u = G Jinv instead of the reference's G Jinv^T, /root/reference/src/models.py:351).  The default everywhere
stays the reference convention with LDS atomics; these tests pin that the switches do what they say."""
import numpy as np
import pytest
import torch

from conftest import tri_mesh_dict, tri_case_forces

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _model(d, nx=61, ny=47, jitter=0.3, seed=5, flip=0.0, u_scale=1.0):
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(nx, ny, jitter=jitter, seed=seed, flip_fraction=flip, dtype=F64)
    torch.manual_seed(seed)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges).to(d)
    with torch.no_grad():
        m.u_free.mul_(u_scale)
    return m, (coords, conn, geom, bc, mn, edges)


def _grads(lf, m):
    m.zero_grad()
    loss = lf(m)
    loss.backward()
    return loss.item(), m.node_coords_free.grad.clone(), m.u_free.grad.clone()


def test_deterministic_flag_is_bit_reproducible_and_matches_the_atomic_kernel():
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = torch.device("cuda:0")
    m, _ = _model(d, nx=301, ny=201, flip=0.2, u_scale=30.0)
    det = EnergyLoss2D(device=d, dtype=F64, deterministic=True)
    fast = EnergyLoss2D(device=d, dtype=F64)
    runs = [_grads(det, m) for _ in range(4)]
    for r in runs[1:]:
        assert r[0] == runs[0][0]
        assert torch.equal(r[1], runs[0][1]) and torch.equal(r[2], runs[0][2])          # bit-identical, run after run
    lf, gx, gu = _grads(fast, m)
    assert abs(lf - runs[0][0]) <= 1e-12 * abs(lf)
    assert (gx - runs[0][1]).abs().max().item() <= 1e-11 * gx.abs().max().item()
    assert (gu - runs[0][2]).abs().max().item() <= 1e-11 * gu.abs().max().item()
    # the atomic kernel itself is NOT bit-reproducible in general (ds_add_f64 order): nothing to assert, but the
    # deterministic one must also be usable through value_and_grad_
    v = det.value_and_grad_(m)
    assert v.item() == runs[0][0] and torch.equal(m.u_free.grad, runs[0][2])


def test_deterministic_flag_on_the_reference_golden_cases(g_tri):
    """Loss + gradients of the fixed-order kernel against what the reference itself produced (body force, traction
    function, flipped elements, permuted mesh, all gauss orders)."""
    from hidenn_fem_amd.loss import EnergyLoss2D
    from test_gpu_parity import tri_model_from_golden, assert_grad_close
    d = torch.device("cuda:0")
    for case in g_tri.cases():
        go, go1 = (int(v) for v in g_tri[case + "/gauss_order"])
        b, t = tri_case_forces(case)
        m = tri_model_from_golden(g_tri, case, d)
        lf = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=go, gauss_order_1d=go1, device=d, dtype=F64, deterministic=True)
        bd = (lambda x: b(x.cpu()).to(d)) if b else None
        td = (lambda x: t(x.cpu()).to(d)) if t else None
        loss = lf(m, b_force=bd, t_force=td)
        loss.backward()
        want = g_tri[case + "/loss"].item()
        assert abs(loss.item() - want) <= 1e-12 * abs(want), case
        assert_grad_close(m.u_free.grad, g_tri[case + "/g_u_free"], case + " gu")
        if m.node_coords_free.numel():
            assert_grad_close(m.node_coords_free.grad, g_tri[case + "/g_coords_free"], case + " gx")


def test_physical_convention_is_exact_for_linear_fields_and_order_invariant():
    """u = A x + b on a skewed mesh: the physical gradient is A in every element (models.py:351's contraction is not,
    SURVEY F4), the energy is psi(sym A) x area whatever the mesh, and rotating every element's local node order
    changes nothing -- under the reference convention all three fail."""
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.loss import EnergyLoss2D
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(41, 33, jitter=0.35, seed=9, dtype=F64)
    A = torch.tensor([[2.0, 3.0], [-1.0, 5.0]], dtype=F64) * 1e-4
    bvec = torch.tensor([0.3, -0.2], dtype=F64) * 1e-4

    def build(cn):
        m = PiecewiseLinearShapeNN2D(coords, cn, boundary_mask=geom, dirichlet_mask=None, u_fixed=None, neumann_edges=None).to(d)
        with torch.no_grad():
            m.u_free.copy_((coords @ A.T + bvec).to(d))
        return m
    m = build(conn)
    ne = conn.shape[0]
    xe = torch.full((ne, 2), 1.0 / 3.0, dtype=F64, device=d)
    eid = torch.arange(ne, device=d)
    m.grad_convention = "physical"
    _, detJ, gu_phys = m(xe, eid)
    assert (gu_phys - A.to(d)).abs().max().item() <= 1e-13 * A.abs().max().item() * 50
    m.grad_convention = "reference"
    _, _, gu_ref = m(xe, eid)
    assert (gu_ref - A.to(d)).abs().max().item() > 1e-2 * A.abs().max().item()          # the reference contraction is not A
    # energy: gauss_order 3 has W = 1/2, so E = psi(sym A) * sum |detJ| / 2 = psi * area
    lf_p = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=3, device=d, dtype=F64, grad_convention="physical")
    lf_r = EnergyLoss2D(E=10e9, nu=0.3, gauss_order=3, device=d, dtype=F64)
    eps = torch.tensor([A[0, 0], A[1, 1], A[0, 1] + A[1, 0]], dtype=F64)
    psi = 0.5 * float(eps @ (lf_p.C.cpu() @ eps))
    e_p = lf_p(m).item()
    X, cn = coords.numpy(), conn.numpy()
    a_, b_, c_ = X[cn[:, 0]], X[cn[:, 1]], X[cn[:, 2]]
    sum_abs_det = float(np.abs((a_[:, 0] - c_[:, 0]) * (b_[:, 1] - c_[:, 1]) - (b_[:, 0] - c_[:, 0]) * (a_[:, 1] - c_[:, 1])).sum())
    assert abs(sum_abs_det - 4.0) < 1e-3                # = 2 x the plate area (2 x 1), up to a sliver the 0.35 jitter folds over
    assert abs(e_p - psi * 0.5 * sum_abs_det) <= 1e-12 * psi * 2.0
    assert abs(lf_r(m).item() - psi * 2.0) > 1e-3 * psi                                   # mesh-dependent under the reference rule
    m_rot = build(conn[:, [1, 2, 0]])
    assert abs(lf_p(m_rot).item() - e_p) <= 1e-12 * abs(e_p)                              # local node order does not matter
    assert abs(lf_r(m_rot).item() - lf_r(m).item()) > 1e-6 * abs(e_p)                     # ... it does for the reference (F4)


def test_physical_convention_gradients_tiled_vs_fixed_order_vs_autograd_oracle():
    from hidenn_fem_amd.loss import EnergyLoss2D
    from oracle import ref_chain as R
    d = torch.device("cuda:0")
    m, (coords, conn, geom, bc, mn, edges) = _model(d, nx=37, ny=29, flip=0.25, u_scale=40.0)
    tiled = EnergyLoss2D(device=d, dtype=F64, grad_convention="physical")
    fixed = EnergyLoss2D(device=d, dtype=F64, grad_convention="physical", deterministic=True)
    lt, gxt, gut = _grads(tiled, m)
    lf_, gxf, guf = _grads(fixed, m)
    assert abs(lt - lf_) <= 1e-12 * abs(lt)
    assert (gxt - gxf).abs().max().item() <= 1e-11 * gxf.abs().max().item()
    assert (gut - guf).abs().max().item() <= 1e-11 * guf.abs().max().item()
    mesh = dict(n_nodes=coords.shape[0], conn=conn, free_mask=~geom, boundary_mask=geom, coords_fixed=coords[geom],
                u_free_mask=~bc, dirichlet_mask=bc, u_fixed=torch.tensor(0.0, dtype=F64), edges=edges)
    lo, gxo, guo = R.energy_and_grads(coords[~geom], m.u_free.detach().cpu(), mesh, convention="physical")
    assert abs(lt - lo.item()) <= 1e-12 * abs(lo.item())
    assert (gxt.cpu() - gxo).abs().max().item() <= 1e-10 * gxo.abs().max().item()
    assert (gut.cpu() - guo).abs().max().item() <= 1e-10 * guo.abs().max().item()
    # per-point forward + backward in the physical convention against the same oracle
    m.grad_convention = "physical"
    ne = conn.shape[0]
    xe = torch.rand(ne, 2, dtype=F64, generator=torch.Generator().manual_seed(1)) * 0.5
    eid = torch.arange(ne)
    m.zero_grad()
    uh, dj, gu = m(xe.to(d), eid.to(d))
    w = torch.randn(ne, 2, 2, dtype=F64, generator=torch.Generator().manual_seed(2))
    (gu * w.to(d)).sum().backward()
    cf = coords[~geom].clone().requires_grad_(True)
    uf = m.u_free.detach().cpu().clone().requires_grad_(True)
    co = R.assemble_coords(mesh["n_nodes"], mesh["free_mask"], cf, mesh["boundary_mask"], mesh["coords_fixed"])
    uo = R.assemble_u(mesh["n_nodes"], mesh["u_free_mask"], uf, mesh["dirichlet_mask"], mesh["u_fixed"])
    _, _, gu_o = R.tri3_forward(co, uo, conn, xe, eid, "physical")
    (gu_o * w).sum().backward()
    assert (gu.cpu() - gu_o.detach()).abs().max().item() <= 1e-12 * gu_o.abs().max().item()
    assert (m.u_free.grad.cpu() - uf.grad).abs().max().item() <= 1e-10 * uf.grad.abs().max().item()
    assert (m.node_coords_free.grad.cpu() - cf.grad).abs().max().item() <= 1e-10 * cf.grad.abs().max().item()

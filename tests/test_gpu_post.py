"""Post-processing kernels against the reference's own chains, restated with the oracle:
plot_von_mises = per-point forward at the centroid (oracle/ref_chain.tri3_forward, pinned by the G1 golden
vectors) + the numpy formulas of /root/reference/src/plots.py:189-198; compute_du_dx_per_element = autograd of
the 1D forward at every element midpoint (plots.py:15-25)."""
import numpy as np
import pytest
import torch

F64 = torch.float64


@pytest.mark.gpu
def test_von_mises_matches_the_reference_chain():
    from oracle import ref_chain as R
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN2D
    from hidenn_fem_amd.post import von_mises
    d = torch.device("cuda:0")
    coords, conn, geom, bc, mn, edges = structured_tri_mesh(57, 43, jitter=0.3, seed=6, flip_fraction=0.4, dtype=F64)
    torch.manual_seed(3)
    m = PiecewiseLinearShapeNN2D(coords, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                 neumann_edges=edges).to(d)
    with torch.no_grad():
        m.u_free.mul_(200.0)
    vm, gu = von_mises(m, return_grad_u=True)
    ne = conn.shape[0]
    X, U = m.coords.detach().cpu(), m.u_full.detach().cpu()
    x_eval = torch.tensor([[1 / 3, 1 / 3]], dtype=F64).expand(ne, 2)            # plots.py:183-184
    _, _, g = R.tri3_forward(X, U, conn, x_eval, torch.arange(ne))
    g = g.numpy()
    E, nu = 10e9, 0.3                                                           # plots.py:189-198
    exx, eyy, exy = g[:, 0, 0], g[:, 1, 1], 0.5 * (g[:, 0, 1] + g[:, 1, 0])
    sxx, syy, sxy = E / (1 - nu ** 2) * (exx + nu * eyy), E / (1 - nu ** 2) * (eyy + nu * exx), E / (1 + nu) * exy
    ref = np.sqrt(sxx ** 2 - sxx * syy + syy ** 2 + 3 * sxy ** 2)
    np.testing.assert_allclose(gu.cpu().numpy(), g, rtol=1e-12, atol=1e-16)
    np.testing.assert_allclose(vm.cpu().numpy(), ref, rtol=1e-12)
    # fp32 models get fp32 results (computed in fp64 inside)
    m32 = PiecewiseLinearShapeNN2D(coords.float(), conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0).to(d)
    assert von_mises(m32).dtype == torch.float32


@pytest.mark.gpu
def test_du_dx_per_element_matches_autograd_at_midpoints():
    from oracle import ref_chain as R
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN
    from hidenn_fem_amd.post import compute_du_dx_per_element
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN(torch.linspace(0, 10, 301, dtype=F64), r_adapt=True, u0=0.0, uN=0.0).to(d)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    got = compute_du_dx_per_element(m)
    grid, u = m.grid.detach().cpu(), m.u_full.detach().cpu()
    xm = (0.5 * (grid[:-1] + grid[1:])).clone().requires_grad_(True)            # plots.py:19-22, all elements at once
    uh = R.line2_forward(grid, u, xm)
    ref = torch.autograd.grad(uh.sum(), xm)[0]
    assert got.device.type == "cpu" and got.shape == ref.shape
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-11, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2049, 10000, 300001])
def test_long_grid_parametrisation_matches_the_reference_chain(n):
    """Grids longer than one workgroup's worth take the three-launch workspace kernels: same chain as
    src/models.py:45-56 (softplus -> clamp -> cumsum -> renormalise, optional mask), forward and backward."""
    from oracle import ref_chain as R
    from hidenn_fem_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    p = (torch.rand(n, generator=g, dtype=F64) * 3e-3 - 1e-3)            # some negative increments (softplus regime)
    p[::97] = -30.0                                                       # clamp(1e-6) rows: zero gradient
    p[5::211] = 25.0                                                      # softplus threshold rows
    mask = torch.zeros(n + 1, dtype=torch.bool)
    mask[0] = mask[-1] = True
    mask[torch.randint(1, n, (n // 50,), generator=g)] = True
    initial = torch.linspace(0.0, 10.0, n + 1, dtype=F64)
    cot = torch.randn(n + 1, generator=g, dtype=F64)
    assert n > ops.GRID_PARAM_ONE_BLOCK
    for use_mask in (False, True):
        pr = p.clone().requires_grad_(True)
        ref = R.grid_param(pr, torch.tensor([0.0], dtype=F64), torch.tensor([10.0], dtype=F64))
        if use_mask:
            ref = R.masked_grid(ref, mask, initial)
        (ref * cot).sum().backward()
        pg = p.clone().to(d).requires_grad_(True)
        m8 = mask.to(torch.uint8).to(d) if use_mask else None
        got = ops.GridParamFn.apply(pg, 0.0, 10.0, m8, initial.to(d) if use_mask else None)
        (got * cot.to(d)).sum().backward()
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-12, atol=1e-13)
        gr = pr.grad.numpy()
        np.testing.assert_allclose(pg.grad.cpu().numpy(), gr, rtol=1e-9, atol=1e-12 * np.abs(gr).max())

"""TEST INFRASTRUCTURE ONLY -- op-for-op CPU/PyTorch restatement of the reference path.

This is the *checker* and the reported CPU baseline ("port"), never the product.
It deliberately keeps the reference's ATen op chain (bool-mask ``index_put``
assembly, ``[M,3,2]`` element gathers with ``M = Ne*ng``, batched
``linalg.det`` / ``linalg.inv``, the two ``einsum`` contractions, autograd for
the backward) so that timing it is timing the reference's algorithm, and so that
its results are the reference's results bit for bit (pinned by
``tests/golden/*.npz``, which were generated from the imported reference).

All citations are ``file:line`` into ``/root/reference``.

The functions are stateless: they take plain tensors, not ``nn.Module``s.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# quadrature tables                                   src/utils.py:4-81
# --------------------------------------------------------------------------

def interval_gauss(order: int = 1, dtype=torch.float64):
    """Gauss-Legendre nodes/weights exactly as ``src/utils.py:8`` returns them:
    raw ``leggauss`` on [-1, 1] (weights sum to 2) although the docstring there
    says [0, 1]."""
    x, w = np.polynomial.legendre.leggauss(order)
    return torch.tensor(x, dtype=dtype), torch.tensor(w, dtype=dtype)


_THIRD = 1 / 3
_TRI_RULES = {
    # order: (points, weights-before-scale, scale)        src/utils.py:20-76
    1: ([[_THIRD, _THIRD]], [0.5], 1.0),
    3: ([[1 / 6, 1 / 6], [4 * (1 / 6), 1 / 6], [1 / 6, 4 * (1 / 6)]],
        [1 / 6, 1 / 6, 1 / 6], 1.0),
    # orders 4 and 6 carry an extra 0.5 on already area-scaled weights -> they
    # sum to 0.25.  Parity = keep it.
    4: ([[_THIRD, _THIRD], [0.6, 0.2], [0.2, 0.6], [0.2, 0.2]],
        [-27 / 96, 25 / 96, 25 / 96, 25 / 96], 0.5),
    6: (None, None, 0.5),   # filled below (needs a, b)
    7: ([[_THIRD, _THIRD],
         [0.0597158717, 0.4701420641], [0.4701420641, 0.0597158717],
         [0.4701420641, 0.4701420641],
         [0.7974269853, 0.1012865073], [0.1012865073, 0.7974269853],
         [0.1012865073, 0.1012865073]],
        [0.225, 0.1323941527, 0.1323941527, 0.1323941527,
         0.1259391805, 0.1259391805, 0.1259391805], 0.5),
}
_a6, _b6 = 0.445948490915965, 0.091576213509771
_TRI_RULES[6] = ([[_a6, _a6], [1 - 2 * _a6, _a6], [_a6, 1 - 2 * _a6],
                  [_b6, _b6], [1 - 2 * _b6, _b6], [_b6, 1 - 2 * _b6]],
                 [0.111690794839005] * 3 + [0.054975871827661] * 3, 0.5)


def triangle_gauss(order: int = 1, dtype=torch.float64):
    """Reference-triangle rules of ``src/utils.py:13-81`` (orders 1,3,4,6,7).
    The scale is applied as a tensor multiply *after* the cast to ``dtype``,
    as the reference does (``0.5 * torch.tensor(...)``, utils.py:39,55,68)."""
    if order not in _TRI_RULES:
        raise NotImplementedError("Supported orders: 1, 3, 4, 6, 7")   # utils.py:78-79
    pts, w, scale = _TRI_RULES[order]
    rs = torch.tensor(pts, dtype=dtype)
    wt = torch.tensor(w, dtype=dtype)
    if scale != 1.0:
        wt = scale * wt
    return rs, wt


# --------------------------------------------------------------------------
# triangular model: assembly + forward                src/models.py:292-376
# --------------------------------------------------------------------------

def assemble_coords(n_nodes, free_mask, coords_free, boundary_mask, coords_fixed):
    """``coords`` property, models.py:292-297: zeros + two bool-mask index_puts."""
    out = torch.zeros(n_nodes, 2, dtype=coords_free.dtype)
    out[free_mask] = coords_free
    out[boundary_mask] = coords_fixed
    return out


def assemble_u(n_nodes, u_free_mask, u_free, dirichlet_mask, u_fixed):
    """``u_full`` property, models.py:299-305 (``u_fixed`` broadcast into the rows)."""
    out = torch.zeros(n_nodes, 2, dtype=u_free.dtype)
    out[u_free_mask] = u_free
    if u_fixed is not None:
        out[dirichlet_mask] = u_fixed
    return out


def tri3_forward(coords, u_full, conn, x_eval, elem_id, convention="reference"):
    """Domain branch of ``forward``, models.py:317-357.

    Returns ``u_h [M,2]``, ``detJ [M]``, ``grad_u [M,2,2]``.  ``dN_dx`` is
    ``Jinv @ dN_dxi`` exactly as models.py:351 (not the transposed, textbook
    contraction -- SURVEY F4).  ``convention="physical"`` is NOT the reference: it is
    the build's opt-in switch (``Jinv^T @ dN_dxi``), restated here so that its kernels
    have an autograd checker too."""
    tri = coords[conn[elem_id]]                              # models.py:320,235
    xi, eta = x_eval[:, 0:1], x_eval[:, 1:2]
    N = torch.cat([xi, eta, 1.0 - xi - eta], dim=1)          # models.py:323-328
    u_nodes = u_full[conn[elem_id]]                          # models.py:331
    u_h = torch.sum(N.unsqueeze(2) * u_nodes, dim=1)         # models.py:333
    v0, v1, v2 = tri[:, 0, :], tri[:, 1, :], tri[:, 2, :]
    Jmat = torch.stack([v0 - v2, v1 - v2], dim=2)            # models.py:339
    detJ = torch.linalg.det(Jmat)                            # models.py:340
    Jinv = torch.linalg.inv(Jmat)                            # models.py:343
    dN_dxi = torch.tensor([[1., 0., -1.], [0., 1., -1.]], dtype=coords.dtype)
    if convention == "physical":
        dN_dx = torch.einsum("mji,jk->mik", Jinv, dN_dxi)    # opt-in: Jinv^T (no reference counterpart)
    else:
        dN_dx = torch.einsum("mij,jk->mik", Jinv, dN_dxi)    # models.py:351
    grad_u = torch.einsum("mai,mja->mij", u_nodes, dN_dx)    # models.py:355
    return u_h, detJ, grad_u


def edge2_forward(coords, u_full, edges, x_eval, edge_id):
    """Edge branch of ``forward``, models.py:359-376 -> ``u_h [M,2]``, ``ds [M]``."""
    x_i = coords[edges[edge_id, 0]]                          # models.py:221-222
    x_j = coords[edges[edge_id, 1]]
    xi = x_eval[:, 0:1]
    N = torch.cat([1.0 - xi, xi], dim=1)                     # models.py:366
    u_nodes = u_full[edges[edge_id]]                         # models.py:369
    u_h = torch.sum(N.unsqueeze(2) * u_nodes, dim=1)
    ds = torch.norm(x_j - x_i, dim=1)                        # models.py:375
    return u_h, ds


# --------------------------------------------------------------------------
# EnergyLoss2D                                        src/loss.py:6-116
# --------------------------------------------------------------------------

def plane_stress_C(E=10e9, nu=0.3, dtype=torch.float64):
    """loss.py:29-32."""
    return torch.tensor([[1.0, nu, 0.0], [nu, 1.0, 0.0], [0.0, 0.0, (1.0 - nu) / 2.0]],
                        dtype=dtype) * (E / (1 - nu ** 2))


def default_traction(x, L=1.0, F_total=100e3):
    """``uniform_edge_force``, loss.py:47-51: (F_total/L, 0) whatever the plate size (F9)."""
    tx = torch.full((x.shape[0],), F_total / L, dtype=x.dtype)
    return torch.stack([tx, torch.zeros_like(tx)], dim=1)


def domain_energy(coords, u_full, conn, C, xg, wg, b_force=None, convention="reference"):
    """loss.py:55-88.  ``b_force`` receives the *reference* points (F6)."""
    ne, ng = conn.shape[0], xg.shape[0]
    x_eval = xg.unsqueeze(0).expand(ne, ng, 2).reshape(-1, 2)             # loss.py:60
    elem_id = torch.arange(ne).unsqueeze(1).repeat(1, ng).reshape(-1)     # loss.py:61
    w_flat = wg.unsqueeze(0).repeat(ne, 1).reshape(-1)                    # loss.py:62
    u_eval, detJ, grad_u = tri3_forward(coords, u_full, conn, x_eval, elem_id, convention)
    gx, gy = grad_u[:, 0, :], grad_u[:, 1, :]
    eps = torch.stack([gx[:, 0], gy[:, 1], 2 * (0.5 * (gx[:, 1] + gy[:, 0]))], dim=1)  # loss.py:70-73
    sig = eps @ C.T                                                       # loss.py:76
    psi = 0.5 * torch.sum(eps * sig, dim=1)                               # loss.py:77
    b_vec = b_force(x_eval) if b_force is not None else torch.zeros_like(x_eval)
    body = torch.sum(b_vec * u_eval, dim=1)                               # loss.py:81
    qw = w_flat * detJ.abs()                                              # loss.py:84
    return torch.sum(qw * psi) - torch.sum(qw * body)                     # loss.py:85-88


def edge_energy(coords, u_full, edges, xg1, wg1, t_force=None):
    """loss.py:91-110.  ``xg1`` are the raw Legendre nodes used as xi in [0,1] (F3)."""
    ned, n1 = edges.shape[0], xg1.shape[0]
    x_i, x_j = coords[edges[:, 0]], coords[edges[:, 1]]                   # loss.py:92
    xq = (1.0 - xg1[None, :, None]) * x_i[:, None, :] + xg1[None, :, None] * x_j[:, None, :]
    xq_flat = xq.reshape(-1, 2)                                           # loss.py:96-97
    wq_flat = wg1[None, :].expand(ned, n1).reshape(-1)                    # loss.py:98
    x_eval = xg1[None, :].expand(ned, n1).reshape(-1, 1)                  # loss.py:101
    edge_id = torch.repeat_interleave(torch.arange(ned), repeats=n1)      # loss.py:102
    u_edge, ds = edge2_forward(coords, u_full, edges, x_eval, edge_id)
    t = t_force(xq_flat) if t_force is not None else default_traction(xq_flat)
    return torch.sum((u_edge * t).sum(dim=1) * (wq_flat * ds))            # loss.py:109-110


def total_energy(coords_free, u_free, mesh, E=10e9, nu=0.3, gauss_order=4,
                 gauss_order_1d=2, b_force=None, t_force=None, convention="reference"):
    """``EnergyLoss2D.__call__`` (loss.py:113-116) on top of the model's
    ``coords`` / ``u_full`` assembly (models.py:292-305).

    ``mesh`` is a dict with ``n_nodes, conn, free_mask, boundary_mask,
    coords_fixed, u_free_mask, dirichlet_mask, u_fixed, edges``.  ``coords`` and
    ``u_full`` are assembled once per use exactly as often as the reference
    does (coords: domain + edge; u_full: domain + edge)."""
    dt = coords_free.dtype
    C = plane_stress_C(E, nu, dt)
    xg, wg = triangle_gauss(gauss_order, dt)
    xg1, wg1 = interval_gauss(gauss_order_1d, dt)

    def coords():
        return assemble_coords(mesh["n_nodes"], mesh["free_mask"], coords_free,
                               mesh["boundary_mask"], mesh["coords_fixed"])

    def ufull():
        return assemble_u(mesh["n_nodes"], mesh["u_free_mask"], u_free,
                          mesh["dirichlet_mask"], mesh["u_fixed"])

    dom = domain_energy(coords(), ufull(), mesh["conn"], C, xg, wg, b_force, convention)
    if mesh.get("edges") is not None and mesh["edges"].shape[0] > 0:
        edg = edge_energy(coords(), ufull(), mesh["edges"], xg1, wg1, t_force)
    else:
        edg = torch.zeros((), dtype=dt)
    return dom - edg


def energy_and_grads(coords_free, u_free, mesh, **kw):
    """One fwd+bwd "element-eval" pass of the reference chain: loss + autograd
    gradients w.r.t. ``node_coords_free`` and ``u_free``."""
    xf = coords_free.detach().clone().requires_grad_(True)
    uf = u_free.detach().clone().requires_grad_(True)
    loss = total_energy(xf, uf, mesh, **kw)
    loss.backward()
    return loss.detach(), xf.grad, uf.grad


# --------------------------------------------------------------------------
# 1D model                                            src/models.py:6-90
# --------------------------------------------------------------------------

def grid_param(increments, x0, xN):
    """r-adaptive grid, models.py:45-56 / 146-155: softplus -> clamp(1e-6) ->
    cumsum -> renormalise; returns ``cat([x0, inner])`` (the last inner value is xN)."""
    inc = torch.clamp(F.softplus(increments), min=1e-6)
    cum = torch.cumsum(inc, dim=0)
    inner = x0 + (xN - x0) * cum / cum[-1]
    return torch.cat([x0, inner], dim=0)


def line2_forward(grid, u_full, x_eval, eps=1e-10):
    """models.py:70-90.  ``searchsorted(right=False) - 1`` then clamp: a point on
    node k>0 belongs to the *left* element."""
    n = grid.shape[0]
    e = (torch.searchsorted(grid, x_eval) - 1).clamp(0, n - 2)   # models.py:73-74
    x_i, x_j = grid[e], grid[e + 1]
    u_i, u_j = u_full[e], u_full[e + 1]
    N1 = (x_j - x_eval) / (x_j - x_i).clamp(eps)                 # models.py:84
    N2 = (x_eval - x_i) / (x_j - x_i).clamp(eps)                 # models.py:85
    return u_i * N1 + u_j * N2                                   # models.py:88


def mse_loss(pred, target):
    """examples/example1.py:38, example2.py:46."""
    return ((pred - target) ** 2).mean()


def bar_energy(grid, u_full, xi, wi, b_force, E):
    """examples/example3.py:27-70.  Quadrature points and weights are built
    under ``no_grad`` (F8), ``du/dx`` comes from ``autograd.grad(create_graph=True)``."""
    with torch.no_grad():                                        # example3.py:41-50
        x_i = grid[:-1].unsqueeze(1)
        x_j = grid[1:].unsqueeze(1)
        xq = 0.5 * (x_j - x_i) * xi + 0.5 * (x_j + x_i)
        wq = 0.5 * (x_j - x_i) * wi
    xq.requires_grad_(True)
    u = line2_forward(grid, u_full, xq)
    du = torch.autograd.grad(u, xq, grad_outputs=torch.ones_like(u), create_graph=True)[0]
    return torch.sum(wq * (0.5 * E * du ** 2 - b_force(xq) * u))  # example3.py:59-68


def example3_body_force(x):
    """examples/example3.py:16-24."""
    pi = torch.pi
    n1 = 4 * pi ** 2 * (x - 2.5) ** 2 - 2 * pi
    d1 = torch.exp(pi * (x - 2.5) ** 2)
    n2 = 8 * pi ** 2 * (x - 7.5) ** 2 - 4 * pi
    d2 = torch.exp(pi * (x - 7.5) ** 2)
    return -n1 / d1 - n2 / d2


# --------------------------------------------------------------------------
# structured 2D model (the shadowed class)            src/models.py:93-212
# --------------------------------------------------------------------------

def masked_grid(grid_full, boundary_mask, initial_grid):
    """models.py:165-166: boundary-masked coordinates stay at their initial value."""
    return torch.where(boundary_mask, initial_grid, grid_full)


def rectq4_forward(gx, gy, u_full, x_eval, eps=1e-10):
    """models.py:180-212: two searchsorted, 4-node gather, bilinear interpolation."""
    nx, ny = gx.shape[0], gy.shape[0]
    px, py = x_eval[:, 0].contiguous(), x_eval[:, 1].contiguous()
    ix = (torch.searchsorted(gx, px) - 1).clamp(0, nx - 2)
    iy = (torch.searchsorted(gy, py) - 1).clamp(0, ny - 2)
    x_i, x_j, y_i, y_j = gx[ix], gx[ix + 1], gy[iy], gy[iy + 1]
    u00, u10 = u_full[ix, iy], u_full[ix + 1, iy]
    u01, u11 = u_full[ix, iy + 1], u_full[ix + 1, iy + 1]
    N1x = (x_j - px) / (x_j - x_i).clamp(eps)
    N2x = (px - x_i) / (x_j - x_i).clamp(eps)
    N1y = (y_j - py) / (y_j - y_i).clamp(eps)
    N2y = (py - y_i) / (y_j - y_i).clamp(eps)
    return N1x * N1y * u00 + N2x * N1y * u10 + N1x * N2y * u01 + N2x * N2y * u11

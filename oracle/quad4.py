"""TEST INFRASTRUCTURE ONLY -- PyTorch/autograd restatement of the QUAD4-iso EXTENSION.

PARITY UNPINNED BY THE REFERENCE: achraf-15/HiDeNN-FEM has no isoparametric quadrilateral
(SURVEY F11).  This file states the element exactly as SURVEY section 8a specifies it, with the
reference's triangle conventions carried over op for op (``Jmat = stack(...)`` with
``J[i][j] = d x_i / d xi_j`` as /root/reference/src/models.py:339, ``linalg.det`` / ``linalg.inv``,
``dN_dx = einsum("mij,mjk->mik", Jinv, dN_dxi)`` as models.py:351, ``grad_u`` as models.py:355,
Voigt strain / stress / psi and ``abs(detJ)`` as /root/reference/src/loss.py:66-88), and autograd
supplies the backward.  It is the only oracle the QUAD4 kernels are checked against.
"""
import torch

XI = torch.tensor([-1.0, 1.0, 1.0, -1.0])      # local nodes CCW from (-1,-1)
ETA = torch.tensor([-1.0, -1.0, 1.0, 1.0])


def quad4_forward(coords, u_full, conn4, x_eval, elem_id, convention="reference"):
    """-> u_h [M,2], detJ [M], grad_u [M,2,2] at (xi, eta) in [-1,1]^2.  ``convention="physical"`` is the OPT-IN
    switch of the build (dN_dx = Jinv^T dN_dxi) -- not the reference's formula -- so that the
    physical-convention QUAD4 kernels have an autograd checker too."""
    dt = coords.dtype
    xi, eta = x_eval[:, 0:1], x_eval[:, 1:2]
    xk, ek = XI.to(dt)[None, :], ETA.to(dt)[None, :]
    N = 0.25 * (1 + xk * xi) * (1 + ek * eta)                         # [M,4]
    dN_dxi = torch.stack([0.25 * xk * (1 + ek * eta), 0.25 * ek * (1 + xk * xi)], dim=1)   # [M,2,4]
    nodes = coords[conn4[elem_id]]                                    # [M,4,2]
    u_nodes = u_full[conn4[elem_id]]                                  # [M,4,2]
    u_h = torch.sum(N.unsqueeze(2) * u_nodes, dim=1)
    Jmat = torch.einsum("mki,mjk->mij", nodes, dN_dxi)                # J[i][j] = sum_k x_k[i] D_N[j][k]
    detJ = torch.linalg.det(Jmat)
    Jinv = torch.linalg.inv(Jmat)
    if convention == "physical":
        dN_dx = torch.einsum("mji,mjk->mik", Jinv, dN_dxi)            # Jinv^T: the physical gradient (opt-in)
    else:
        dN_dx = torch.einsum("mij,mjk->mik", Jinv, dN_dxi)            # reference convention (F4)
    grad_u = torch.einsum("mai,mja->mij", u_nodes, dN_dx)
    return u_h, detJ, grad_u


def gauss_2x2(dtype=torch.float64):
    """The 2x2 Gauss points in the order (-,-) (+,-) (-,+) (+,+) (weights 1)."""
    g = 1.0 / torch.sqrt(torch.tensor(3.0, dtype=dtype))
    return torch.stack([torch.stack([-g, -g]), torch.stack([g, -g]), torch.stack([-g, g]), torch.stack([g, g])])


def quad4_domain_energy(coords, u_full, conn4, C, b_force=None, convention="reference"):
    """sum_e sum_{2x2 Gauss} w |detJ| (psi - b(xi_q).u_h)  (weights 1, points +-1/sqrt(3)); ``b_force`` receives
    the REFERENCE points, as the reference's triangle path does (loss.py:60,80)."""
    dt = coords.dtype
    g = 1.0 / torch.sqrt(torch.tensor(3.0, dtype=dt))
    pts = torch.stack([torch.stack([-g, -g]), torch.stack([g, -g]), torch.stack([-g, g]), torch.stack([g, g])])
    ne = conn4.shape[0]
    x_eval = pts.unsqueeze(0).expand(ne, 4, 2).reshape(-1, 2)
    elem_id = torch.arange(ne).unsqueeze(1).repeat(1, 4).reshape(-1)
    u_eval, detJ, grad_u = quad4_forward(coords, u_full, conn4, x_eval, elem_id, convention)
    gx, gy = grad_u[:, 0, :], grad_u[:, 1, :]
    eps = torch.stack([gx[:, 0], gy[:, 1], 2 * (0.5 * (gx[:, 1] + gy[:, 0]))], dim=1)
    sig = eps @ C.T
    psi = 0.5 * torch.sum(eps * sig, dim=1)
    if b_force is not None:
        return torch.sum(detJ.abs() * psi) - torch.sum(detJ.abs() * torch.sum(b_force(x_eval) * u_eval, dim=1))
    return torch.sum(detJ.abs() * psi)

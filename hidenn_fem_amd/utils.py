"""Quadrature tables and the gradient smoke check.

Mirrors the public names of ``/root/reference/src/utils.py`` (plus
``gauss_legendre_points_weights``, which ``examples/example3.py:5`` imports but the
reference never defined -- SURVEY F2).  The tables are the reference's numbers,
quirks included: parity with the reference energy needs them
verbatim.
"""
from __future__ import annotations

import numpy as np
import torch


def interval_gauss_points(order=1, device=None, dtype=torch.float32):
    """Gauss-Legendre nodes and weights as ``src/utils.py:4-11`` returns them: raw
    ``leggauss`` on **[-1, 1]** (weights sum to 2), whatever the upstream docstring says."""
    xi, wi = np.polynomial.legendre.leggauss(order)
    return (torch.tensor(xi, dtype=dtype, device=device),
            torch.tensor(wi, dtype=dtype, device=device))


def gauss_legendre_points_weights(n, device=None, dtype=torch.float32):
    """Name used by ``examples/example3.py:5,85``; same rule on [-1, 1], which is what
    that example maps with ``0.5*(b-a)*xi + 0.5*(b+a)`` (example3.py:49-50)."""
    return interval_gauss_points(n, device=device, dtype=dtype)


def _tri_rule(order):
    """(points [ng][2], weights [ng], post-scale).  ``src/utils.py:20-76``.  Orders 4
    and 6 apply an extra 0.5 to area-scaled weights (they sum to 0.25, F5); order 7 uses
    10-digit truncated Dunavant constants."""
    t = 1 / 3
    if order == 1:
        return [[t, t]], [0.5], None
    if order == 3:
        a = 1 / 6
        return [[a, a], [4 * a, a], [a, 4 * a]], [1 / 6] * 3, None
    if order == 4:
        return [[t, t], [0.6, 0.2], [0.2, 0.6], [0.2, 0.2]], [-27 / 96] + [25 / 96] * 3, 0.5
    if order == 6:
        a, b = 0.445948490915965, 0.091576213509771
        w1, w2 = 0.111690794839005, 0.054975871827661
        pts = [[a, a], [1 - 2 * a, a], [a, 1 - 2 * a], [b, b], [1 - 2 * b, b], [b, 1 - 2 * b]]
        return pts, [w1] * 3 + [w2] * 3, 0.5
    if order == 7:
        p, q = 0.0597158717, 0.4701420641
        r, s = 0.7974269853, 0.1012865073
        pts = [[t, t], [p, q], [q, p], [q, q], [r, s], [s, r], [s, s]]
        return pts, [0.225] + [0.1323941527] * 3 + [0.1259391805] * 3, 0.5
    raise NotImplementedError("Supported orders: 1, 3, 4, 6, 7")


def triangle_gauss_points(order=1, device=None, dtype=torch.float32):
    """Points (r,s) and weights on the triangle (0,0),(1,0),(0,1); ``src/utils.py:13-81``."""
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    pts, w, scale = _tri_rule(order)
    rs = torch.tensor(pts, dtype=dtype, device=device)
    wt = torch.tensor(w, dtype=dtype, device=device)
    if scale is not None:
        wt = scale * wt          # applied after the cast, as upstream (utils.py:39,55,68)
    return rs, wt


def test_gradients(model, loss_fn):
    """Smoke check of ``src/utils.py:83-97``: gradients exist and are finite."""
    loss = loss_fn(model)
    loss.backward()
    assert model.u_free.grad is not None
    assert not torch.isnan(model.u_free.grad).any()
    assert model.node_coords_free.grad is not None
    assert not torch.isnan(model.node_coords_free.grad).any()
    print("Gradient magnitudes:")
    print(f"u_free: {model.u_free.grad.norm()}")
    print(f"node_coords: {model.node_coords_free.grad.norm()}")


test_gradients.__test__ = False   # not a pytest test

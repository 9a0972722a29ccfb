"""The PRODUCT's quadrature tables (hidenn_fem_amd/utils.py) against the tables the reference itself returned
(/root/reference/src/utils.py:4-81, frozen in tests/golden/g5_quadrature.npz by make_golden.py): bit-exact in fp32
and fp64, quirks included (raw Legendre nodes on [-1,1]; F5: order-4/6 weights sum to 0.25)."""
import numpy as np
import pytest
import torch

from hidenn_fem_amd import utils as U

CPU = torch.device("cpu")


def test_product_triangle_tables_bit_exact(g_quad):
    for o in (1, 3, 4, 6, 7):
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            rs, w = U.triangle_gauss_points(o, device=CPU, dtype=dt)
            assert rs.dtype == dt and w.dtype == dt
            assert np.array_equal(rs.numpy(), g_quad[f"tri{o}_{tag}_rs"]), (o, tag)
            assert np.array_equal(w.numpy(), g_quad[f"tri{o}_{tag}_w"]), (o, tag)
    with pytest.raises(NotImplementedError):
        U.triangle_gauss_points(2, device=CPU)
    with pytest.raises(NotImplementedError):
        U.triangle_gauss_points(5, device=CPU)


def test_product_interval_tables_bit_exact(g_quad):
    for o in (1, 2, 3, 4, 5):
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            for fn in (U.interval_gauss_points, U.gauss_legendre_points_weights):     # F2: the alias example 3 imports
                x, w = fn(o, device=CPU, dtype=dt)
                assert np.array_equal(x.numpy(), g_quad[f"gl{o}_{tag}_x"]), (o, tag)
                assert np.array_equal(w.numpy(), g_quad[f"gl{o}_{tag}_w"]), (o, tag)


def test_energy_loss_constants_follow_the_tables(g_quad):
    """EnergyLoss2D hands the kernel W = sum(w) and c_i, c_j of the raw Legendre rule: fixed by the tables above."""
    from hidenn_fem_amd.loss import EnergyLoss2D
    for go in (1, 3, 4, 6, 7):
        lf = EnergyLoss2D(gauss_order=go, gauss_order_1d=2, device=CPU, dtype=torch.float64)
        assert lf._W == float(torch.from_numpy(g_quad[f"tri{go}_f64_w"]).sum())        # the same torch.sum the product runs
    lf = EnergyLoss2D(device=CPU, dtype=torch.float64)
    x, w = g_quad["gl2_f64_x"], g_quad["gl2_f64_w"]
    assert abs(lf._ci - float((w * (1.0 - x)).sum())) < 1e-15 and abs(lf._cj - float((w * x).sum())) < 1e-15
    assert abs(lf._ci - 2.0) < 1e-15 and abs(lf._cj) < 1e-15            # F3: c_i = 2, c_j = 0 for order 2

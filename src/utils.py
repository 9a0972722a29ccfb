from hidenn_fem_amd.utils import (interval_gauss_points, gauss_legendre_points_weights,  # noqa: F401
                                  triangle_gauss_points, test_gradients)

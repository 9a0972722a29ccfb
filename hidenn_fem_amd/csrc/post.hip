// Post-processing callers of the unfused forward: the reference evaluates the P1
// gradient at every element centroid through the per-point forward and finishes in numpy
// (/root/reference/src/plots.py:177-201, plot_von_mises), and gets the 1D per-element slope with one
// autograd call per element in a Python loop (plots.py:5-27, compute_du_dx_per_element).  grad_u is
// constant per TRI3 element, so one thread per element does the whole chain:
//   grad_u (reference convention dN_dx = Jinv * D_N, models.py:351-355) -> eps -> plane-stress sigma
//   (sigma_xy = E/(1+nu) eps_xy, as plots.py:196) -> von Mises; 28 B/elem in (conn + gathered nodes via L2),
//   8 B/elem out -- HBM-bound, launch-bound at example sizes.
#include <hip/hip_runtime.h>

#include "hfem_device.h"

namespace hfem {

__global__ __launch_bounds__(256) void tri3_von_mises_kernel(const double2 *__restrict__ X, const double2 *__restrict__ U,
                                                             const int32_t *__restrict__ conn, int64_t ne, double E,
                                                             double nu, double *__restrict__ vm,
                                                             double4 *__restrict__ grad_u) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const double k1 = E / (1.0 - nu * nu), k2 = E / (1.0 + nu);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < ne; e += stride) {
        const int32_t n0 = conn[3 * e], n1 = conn[3 * e + 1], n2 = conn[3 * e + 2];
        const double2 X0 = X[n0], X1 = X[n1], X2 = X[n2], U0 = U[n0], U1 = U[n1], U2 = U[n2];
        const double a = X0.x - X2.x, b = X1.x - X2.x, c = X0.y - X2.y, d = X1.y - X2.y;
        const double inv = 1.0 / (a * d - b * c);
        const double g0x = U0.x - U2.x, g0y = U0.y - U2.y, g1x = U1.x - U2.x, g1y = U1.y - U2.y;
        const double h00 = (g0x * d - g1x * b) * inv, h01 = (g1x * a - g0x * c) * inv;
        const double h10 = (g0y * d - g1y * b) * inv, h11 = (g1y * a - g0y * c) * inv;
        if (grad_u) grad_u[e] = make_double4(h00, h01, h10, h11);
        const double exy = 0.5 * (h01 + h10);
        const double sxx = k1 * (h00 + nu * h11), syy = k1 * (h11 + nu * h00), sxy = k2 * exy;
        vm[e] = sqrt(sxx * sxx - sxx * syy + syy * syy + 3.0 * sxy * sxy);
    }
}

// slope of the piecewise-linear 1D field on element i: (u[i+1] - u[i]) / (grid[i+1] - grid[i]), dim_u components
__global__ __launch_bounds__(256) void line2_slopes_kernel(const double *__restrict__ grid, const double *__restrict__ u,
                                                           int64_t n_elem, int32_t dim_u, double *__restrict__ out) {
    const int64_t tot = n_elem * dim_u, stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < tot; t += stride) {
        const int64_t i = t / dim_u, c = t - i * dim_u;
        out[t] = (u[(i + 1) * dim_u + c] - u[i * dim_u + c]) / (grid[i + 1] - grid[i]);
    }
}

}  // namespace hfem

using namespace hfem;

static int post_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

extern "C" int hfem_tri3_von_mises(int device, const double *X, const double *U, const int32_t *conn, int64_t ne,
                                   double E, double nu, double *von_mises, double *grad_u, void *stream) {
    HFEM_ARG_CHECK(ne >= 0, "negative element count");
    if (ne == 0) return 0;
    HFEM_ARG_CHECK(X && U && conn && von_mises, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(tri3_von_mises_kernel, dim3(post_grid(ne)), dim3(256), 0, (hipStream_t)stream, (const double2 *)X,
                       (const double2 *)U, conn, ne, E, nu, von_mises, (double4 *)grad_u);
    return launch_status("hfem_tri3_von_mises");
}

extern "C" int hfem_line2_slopes(int device, const double *grid, const double *u, int64_t n_nodes, int32_t dim_u,
                                 double *out, void *stream) {
    HFEM_ARG_CHECK(n_nodes >= 0 && dim_u >= 1, "bad sizes");
    if (n_nodes < 2) return 0;
    HFEM_ARG_CHECK(grid && u && out, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(line2_slopes_kernel, dim3(post_grid((n_nodes - 1) * dim_u)), dim3(256), 0, (hipStream_t)stream, grid,
                       u, n_nodes - 1, dim_u, out);
    return launch_status("hfem_line2_slopes");
}

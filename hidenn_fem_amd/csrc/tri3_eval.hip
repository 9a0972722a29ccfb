// Unfused per-point TRI3 / EDGE2 evaluation (forward + backward) and the
// free/fixed row assembly, gfx950.
//
// These keep the reference's `(x_ref, element_id)` forward contract alive for
// callers that need per-point outputs -- plots.plot_von_mises
// (/root/reference/src/plots.py:183-187) and any user loss built on
// model(x_eval, elem_id) -- while the fused energy kernel (tri3_energy.hip) is
// what EnergyLoss2D uses.
#include <hip/hip_runtime.h>

#include "hfem_device.h"

namespace hfem {

constexpr int kBlockE = 256;

// forward of /root/reference/src/models.py:317-357
__global__ __launch_bounds__(kBlockE) void tri3_eval_fwd_kernel(
    const double2 *__restrict__ X, const double2 *__restrict__ U, const int32_t *__restrict__ conn,
    const double2 *__restrict__ x_eval, const int64_t *__restrict__ elem_id, int64_t m,
    double2 *__restrict__ u_h, double *__restrict__ detJ, double4 *__restrict__ grad_u, int phys) {
    const int64_t stride = (int64_t)gridDim.x * kBlockE;
    for (int64_t p = (int64_t)blockIdx.x * kBlockE + threadIdx.x; p < m; p += stride) {
        const int64_t e = elem_id[p];
        const int32_t n0 = conn[3 * e], n1 = conn[3 * e + 1], n2 = conn[3 * e + 2];
        const double2 X0 = X[n0], X1 = X[n1], X2 = X[n2], U0 = U[n0], U1 = U[n1], U2 = U[n2];
        const double2 r = x_eval[p];
        const double zeta = 1.0 - r.x - r.y;
        u_h[p] = make_double2(r.x * U0.x + r.y * U1.x + zeta * U2.x, r.x * U0.y + r.y * U1.y + zeta * U2.y);
        // phys: grad_u = G Jinv instead of the reference's G Jinv^T -- b and c trade places (hfem_device.h)
        const double a = X0.x - X2.x, d = X1.y - X2.y;
        const double b = phys ? X0.y - X2.y : X1.x - X2.x, c = phys ? X1.x - X2.x : X0.y - X2.y;
        const double det = a * d - b * c, inv = 1.0 / det;
        detJ[p] = det;
        const double g0x = U0.x - U2.x, g0y = U0.y - U2.y, g1x = U1.x - U2.x, g1y = U1.y - U2.y;
        grad_u[p] = make_double4((g0x * d - g1x * b) * inv, (g1x * a - g0x * c) * inv,
                                 (g0y * d - g1y * b) * inv, (g1y * a - g0y * c) * inv);
    }
}

// backward: cotangents cu[M][2], cd[M], cg[M][2][2] -> gX,gU (accumulated)
__global__ __launch_bounds__(kBlockE) void tri3_eval_bwd_kernel(
    const double2 *__restrict__ X, const double2 *__restrict__ U, const int32_t *__restrict__ conn,
    const double2 *__restrict__ x_eval, const int64_t *__restrict__ elem_id, int64_t m,
    const double2 *__restrict__ cu, const double *__restrict__ cd, const double4 *__restrict__ cg,
    double *__restrict__ gX, double *__restrict__ gU, int phys) {
    const int64_t stride = (int64_t)gridDim.x * kBlockE;
    for (int64_t p = (int64_t)blockIdx.x * kBlockE + threadIdx.x; p < m; p += stride) {
        const int64_t e = elem_id[p];
        const int32_t n[3] = {conn[3 * e], conn[3 * e + 1], conn[3 * e + 2]};
        const double2 X0 = X[n[0]], X1 = X[n[1]], X2 = X[n[2]], U0 = U[n[0]], U1 = U[n[1]], U2 = U[n[2]];
        const double a = X0.x - X2.x, d = X1.y - X2.y;
        const double b = phys ? X0.y - X2.y : X1.x - X2.x, c = phys ? X1.x - X2.x : X0.y - X2.y;
        const double det = a * d - b * c, inv = 1.0 / det;
        const double g0x = U0.x - U2.x, g0y = U0.y - U2.y, g1x = U1.x - U2.x, g1y = U1.y - U2.y;
        const double h00 = (g0x * d - g1x * b) * inv, h01 = (g1x * a - g0x * c) * inv;
        const double h10 = (g0y * d - g1y * b) * inv, h11 = (g1y * a - g0y * c) * inv;
        const double4 P = cg ? cg[p] : make_double4(0, 0, 0, 0);   // {P00,P01,P10,P11}
        const double2 q = cu ? cu[p] : make_double2(0, 0);
        const double cdet = cd ? cd[p] : 0.0;
        const double2 r = x_eval[p];
        const double zeta = 1.0 - r.x - r.y;
        const double dg0x = (P.x * d - P.y * c) * inv, dg0y = (P.z * d - P.w * c) * inv;
        const double dg1x = (P.y * a - P.x * b) * inv, dg1y = (P.w * a - P.z * b) * inv;
        const double2 gu[3] = {make_double2(dg0x + r.x * q.x, dg0y + r.x * q.y),
                               make_double2(dg1x + r.y * q.x, dg1y + r.y * q.y),
                               make_double2(zeta * q.x - (dg0x + dg1x), zeta * q.y - (dg0y + dg1y))};
        const double ddet = cdet - (P.x * h00 + P.y * h01 + P.z * h10 + P.w * h11) * inv;
        const double da = (P.y * g1x + P.w * g1y) * inv + ddet * d;
        const double db = -(P.x * g1x + P.z * g1y) * inv - ddet * c;
        const double dc = -(P.y * g0x + P.w * g0y) * inv - ddet * b;
        const double dd = (P.x * g0x + P.z * g0y) * inv + ddet * a;
        const double2 gx[3] = {phys ? make_double2(da, db) : make_double2(da, dc),
                               phys ? make_double2(dc, dd) : make_double2(db, dd),
                               phys ? make_double2(-(da + dc), -(db + dd)) : make_double2(-(da + db), -(dc + dd))};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            unsafeAtomicAdd(&gX[2 * (int64_t)n[j]], gx[j].x);
            unsafeAtomicAdd(&gX[2 * (int64_t)n[j] + 1], gx[j].y);
            unsafeAtomicAdd(&gU[2 * (int64_t)n[j]], gu[j].x);
            unsafeAtomicAdd(&gU[2 * (int64_t)n[j] + 1], gu[j].y);
        }
    }
}

// forward of /root/reference/src/models.py:359-376
__global__ __launch_bounds__(kBlockE) void edge2_eval_fwd_kernel(
    const double2 *__restrict__ X, const double2 *__restrict__ U, const int32_t *__restrict__ edges,
    const double *__restrict__ xi, const int64_t *__restrict__ edge_id, int64_t m,
    double2 *__restrict__ u_h, double *__restrict__ ds) {
    const int64_t stride = (int64_t)gridDim.x * kBlockE;
    for (int64_t p = (int64_t)blockIdx.x * kBlockE + threadIdx.x; p < m; p += stride) {
        const int64_t g = edge_id[p];
        const int32_t i = edges[2 * g], j = edges[2 * g + 1];
        const double s = xi[p], w = 1.0 - s;
        const double2 Ui = U[i], Uj = U[j], Xi = X[i], Xj = X[j];
        u_h[p] = make_double2(w * Ui.x + s * Uj.x, w * Ui.y + s * Uj.y);
        const double rx = Xj.x - Xi.x, ry = Xj.y - Xi.y;
        ds[p] = sqrt(rx * rx + ry * ry);
    }
}

__global__ __launch_bounds__(kBlockE) void edge2_eval_bwd_kernel(
    const double2 *__restrict__ X, const int32_t *__restrict__ edges, const double *__restrict__ xi,
    const int64_t *__restrict__ edge_id, int64_t m, const double2 *__restrict__ cu,
    const double *__restrict__ cds, double *__restrict__ gX, double *__restrict__ gU) {
    const int64_t stride = (int64_t)gridDim.x * kBlockE;
    for (int64_t p = (int64_t)blockIdx.x * kBlockE + threadIdx.x; p < m; p += stride) {
        const int64_t g = edge_id[p];
        const int32_t i = edges[2 * g], j = edges[2 * g + 1];
        const double s = xi[p], w = 1.0 - s;
        if (cu) {
            const double2 q = cu[p];
            unsafeAtomicAdd(&gU[2 * (int64_t)i], w * q.x);
            unsafeAtomicAdd(&gU[2 * (int64_t)i + 1], w * q.y);
            unsafeAtomicAdd(&gU[2 * (int64_t)j], s * q.x);
            unsafeAtomicAdd(&gU[2 * (int64_t)j + 1], s * q.y);
        }
        if (cds) {
            const double2 Xi = X[i], Xj = X[j];
            const double rx = Xj.x - Xi.x, ry = Xj.y - Xi.y;
            const double f = cds[p] / sqrt(rx * rx + ry * ry);
            unsafeAtomicAdd(&gX[2 * (int64_t)j], f * rx);
            unsafeAtomicAdd(&gX[2 * (int64_t)j + 1], f * ry);
            unsafeAtomicAdd(&gX[2 * (int64_t)i], -f * rx);
            unsafeAtomicAdd(&gX[2 * (int64_t)i + 1], -f * ry);
        }
    }
}

// dst[idx[r]][:] = src[r][:]   /   dst[r][:] = src[idx[r]][:]
__global__ __launch_bounds__(kBlockE) void scatter_rows_kernel(const double *__restrict__ src,
                                                              const int32_t *__restrict__ idx, int64_t rows,
                                                              int width, double *__restrict__ dst) {
    const int64_t total = rows * width, stride = (int64_t)gridDim.x * kBlockE;
    for (int64_t t = (int64_t)blockIdx.x * kBlockE + threadIdx.x; t < total; t += stride) {
        const int64_t r = t / width;
        const int c = (int)(t - r * width);
        dst[(int64_t)idx[r] * width + c] = src[t];
    }
}

__global__ __launch_bounds__(kBlockE) void gather_rows_kernel(const double *__restrict__ src,
                                                             const int32_t *__restrict__ idx, int64_t rows,
                                                             int width, double *__restrict__ dst) {
    const int64_t total = rows * width, stride = (int64_t)gridDim.x * kBlockE;
    for (int64_t t = (int64_t)blockIdx.x * kBlockE + threadIdx.x; t < total; t += stride) {
        const int64_t r = t / width;
        const int c = (int)(t - r * width);
        dst[t] = src[(int64_t)idx[r] * width + c];
    }
}

static int grid_e(int64_t n) {
    int64_t g = (n + kBlockE - 1) / kBlockE;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace hfem

using namespace hfem;

extern "C" int hfem_tri3_eval_fwd(int device, const double *X, const double *U, const int32_t *conn,
                                  const double *x_eval, const int64_t *elem_id, int64_t m, double *u_h,
                                  double *detJ, double *grad_u, void *stream) {
    return hfem_tri3_eval_fwd_conv(device, X, U, conn, x_eval, elem_id, m, u_h, detJ, grad_u, HFEM_GRAD_REFERENCE, stream);
}

extern "C" int hfem_tri3_eval_fwd_conv(int device, const double *X, const double *U, const int32_t *conn,
                                       const double *x_eval, const int64_t *elem_id, int64_t m, double *u_h,
                                       double *detJ, double *grad_u, int32_t convention, void *stream) {
    HFEM_ARG_CHECK(convention == HFEM_GRAD_REFERENCE || convention == HFEM_GRAD_PHYSICAL, "unknown gradient convention");
    HFEM_ARG_CHECK(m >= 0, "negative point count");
    if (m == 0) return 0;
    HFEM_ARG_CHECK(X && U && conn && x_eval && elem_id && u_h && detJ && grad_u, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(tri3_eval_fwd_kernel, dim3(grid_e(m)), dim3(kBlockE), 0, (hipStream_t)stream,
                       (const double2 *)X, (const double2 *)U, conn, (const double2 *)x_eval, elem_id, m,
                       (double2 *)u_h, detJ, (double4 *)grad_u, (int)convention);
    return launch_status("hfem_tri3_eval_fwd");
}

extern "C" int hfem_tri3_eval_bwd(int device, const double *X, const double *U, const int32_t *conn,
                                  const double *x_eval, const int64_t *elem_id, int64_t m, const double *cu,
                                  const double *cd, const double *cg, double *gX, double *gU, void *stream) {
    return hfem_tri3_eval_bwd_conv(device, X, U, conn, x_eval, elem_id, m, cu, cd, cg, gX, gU, HFEM_GRAD_REFERENCE, stream);
}

extern "C" int hfem_tri3_eval_bwd_conv(int device, const double *X, const double *U, const int32_t *conn,
                                       const double *x_eval, const int64_t *elem_id, int64_t m, const double *cu,
                                       const double *cd, const double *cg, double *gX, double *gU,
                                       int32_t convention, void *stream) {
    HFEM_ARG_CHECK(convention == HFEM_GRAD_REFERENCE || convention == HFEM_GRAD_PHYSICAL, "unknown gradient convention");
    HFEM_ARG_CHECK(m >= 0, "negative point count");
    if (m == 0) return 0;
    HFEM_ARG_CHECK(X && U && conn && x_eval && elem_id && gX && gU, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(tri3_eval_bwd_kernel, dim3(grid_e(m)), dim3(kBlockE), 0, (hipStream_t)stream,
                       (const double2 *)X, (const double2 *)U, conn, (const double2 *)x_eval, elem_id, m,
                       (const double2 *)cu, cd, (const double4 *)cg, gX, gU, (int)convention);
    return launch_status("hfem_tri3_eval_bwd");
}

extern "C" int hfem_edge2_eval_fwd(int device, const double *X, const double *U, const int32_t *edges,
                                   const double *xi, const int64_t *edge_id, int64_t m, double *u_h,
                                   double *ds, void *stream) {
    HFEM_ARG_CHECK(m >= 0, "negative point count");
    if (m == 0) return 0;
    HFEM_ARG_CHECK(X && U && edges && xi && edge_id && u_h && ds, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(edge2_eval_fwd_kernel, dim3(grid_e(m)), dim3(kBlockE), 0, (hipStream_t)stream,
                       (const double2 *)X, (const double2 *)U, edges, xi, edge_id, m, (double2 *)u_h, ds);
    return launch_status("hfem_edge2_eval_fwd");
}

extern "C" int hfem_edge2_eval_bwd(int device, const double *X, const double *U, const int32_t *edges,
                                   const double *xi, const int64_t *edge_id, int64_t m, const double *cu,
                                   const double *cds, double *gX, double *gU, void *stream) {
    (void)U;
    HFEM_ARG_CHECK(m >= 0, "negative point count");
    if (m == 0) return 0;
    HFEM_ARG_CHECK(X && edges && xi && edge_id && gX && gU, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(edge2_eval_bwd_kernel, dim3(grid_e(m)), dim3(kBlockE), 0, (hipStream_t)stream,
                       (const double2 *)X, edges, xi, edge_id, m, (const double2 *)cu, cds, gX, gU);
    return launch_status("hfem_edge2_eval_bwd");
}

extern "C" int hfem_scatter_rows(int device, const double *src, const int32_t *idx, int64_t rows,
                                 int32_t width, double *dst, void *stream) {
    HFEM_ARG_CHECK(rows >= 0 && width > 0, "bad shape");
    if (rows == 0) return 0;
    HFEM_ARG_CHECK(src && idx && dst, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_e(rows * width)), dim3(kBlockE), 0, (hipStream_t)stream,
                       src, idx, rows, (int)width, dst);
    return launch_status("hfem_scatter_rows");
}

extern "C" int hfem_gather_rows(int device, const double *src, const int32_t *idx, int64_t rows,
                                int32_t width, double *dst, void *stream) {
    HFEM_ARG_CHECK(rows >= 0 && width > 0, "bad shape");
    if (rows == 0) return 0;
    HFEM_ARG_CHECK(src && idx && dst, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_e(rows * width)), dim3(kBlockE), 0, (hipStream_t)stream,
                       src, idx, rows, (int)width, dst);
    return launch_status("hfem_gather_rows");
}

// QUAD4-iso extension (SURVEY F11 reading (ii), section 8a spec): bilinear isoparametric quadrilateral,
// 2x2 Gauss, same conventions as the reference's triangle (J[i][j] = d x_i / d xi_j, dN_dx = Jinv * D_N,
// abs(detJ); /root/reference/src/models.py:336-355, src/loss.py:66-88).  The reference has no Q4
// element: parity is pinned by the test-side autograd restatement oracle/quad4.py only.
//
// At a point (xi, eta) with D_N[j][k] = d N_k / d xi_j:
//   a = sum_k x_k D0k, b = sum_k x_k D1k, c = sum_k y_k D0k, d = sum_k y_k D1k      (J = [[a,b],[c,d]])
//   G0 = sum_k U_k D0k, G1 = sum_k U_k D1k
// and from there H = G Jinv^T, eps, sigma, psi and the whole backward are the TRI3 closed forms
// (hfem_device.h) with (a,b,c,d,G0,G1); the chain rule back to the nodes multiplies by D0k / D1k.
// Planless this round (one thread per element, fp64 global atomics); the tiled plan is TRI3-only.
#include <hip/hip_runtime.h>

#include "hfem_device.h"

namespace hfem {

constexpr int kBlockQ = 256;

struct JacGrad {          // dL/d(a,b,c,d), dL/dG0, dL/dG1
    double da, db, dc, dd;
    double2 dg0, dg1;
};

// energy density * |det| at one point with weight w, and its gradient w.r.t. (a..d, G0, G1)
template <bool GRAD>
__device__ __forceinline__ double jac_point(double a, double b, double c, double d, double2 g0, double2 g1,
                                            double w, const Tri3Consts &k, JacGrad &o) {
    const double det = a * d - b * c;
    const double inv = fast_rcp(det);
    const double A = fabs(det);
    const double ai = a * inv, bi = b * inv, ci = c * inv, di = d * inv;
    const double h00 = g0.x * di - g1.x * bi, h01 = g1.x * ai - g0.x * ci;
    const double h10 = g0.y * di - g1.y * bi, h11 = g1.y * ai - g0.y * ci;
    const double gam = h01 + h10;
    const double sxx = k.c11 * h00 + k.c12 * h11, syy = k.c12 * h00 + k.c22 * h11, sxy = k.c33 * gam;
    const double wpsi = (0.5 * w) * (h00 * sxx + h11 * syy + gam * sxy);
    if (GRAD) {
        const double aw = A * w;
        const double p00 = aw * sxx, p01 = aw * sxy, p11 = aw * syy;
        o.dg0 = make_double2(p00 * di - p01 * ci, p01 * di - p11 * ci);
        o.dg1 = make_double2(p01 * ai - p00 * bi, p11 * ai - p01 * bi);
        const double q = det < 0.0 ? -w : w;
        const double ddet = det < 0.0 ? wpsi : -wpsi;
        o.da = q * (sxy * g1.x + syy * g1.y) + ddet * d;
        o.db = -q * (sxx * g1.x + sxy * g1.y) - ddet * c;
        o.dc = -q * (sxy * g0.x + syy * g0.y) - ddet * b;
        o.dd = q * (sxx * g0.x + sxy * g0.y) + ddet * a;
    }
    return A * wpsi;
}

// reference-square corner signs, CCW from (-1,-1): xi_k = {-1,1,1,-1}, eta_k = {-1,-1,1,1}
__device__ __forceinline__ constexpr double corner_xi(int k) { return (k == 1 || k == 2) ? 1.0 : -1.0; }
__device__ __forceinline__ constexpr double corner_eta(int k) { return k >= 2 ? 1.0 : -1.0; }

__device__ __forceinline__ void shape_derivs(double xi, double eta, double (&D0)[4], double (&D1)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        D0[k] = 0.25 * corner_xi(k) * (1.0 + corner_eta(k) * eta);
        D1[k] = 0.25 * corner_eta(k) * (1.0 + corner_xi(k) * xi);
    }
}

// fused QUAD4 energy: sum over 2x2 Gauss points (+-1/sqrt(3), weights 1) of |detJ| psi, fwd + bwd
__global__ __launch_bounds__(kBlockQ) void quad4_energy_atomic_kernel(
    const double2 *__restrict__ X, const double2 *__restrict__ U, const int32_t *__restrict__ conn,
    int64_t e_begin, int64_t e_end, Tri3Consts k, double *__restrict__ loss_acc, double *__restrict__ gX,
    double *__restrict__ gU) {
    __shared__ double red[kBlockQ / 64];
    double e_loc = 0.0;
    const double gp = 0.57735026918962576451;   // 1/sqrt(3)
    const int64_t stride = (int64_t)gridDim.x * kBlockQ;
    for (int64_t e = e_begin + (int64_t)blockIdx.x * kBlockQ + threadIdx.x; e < e_end; e += stride) {
        int32_t n[4];
        double2 Xn[4], Un[4], gx[4], gu[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            n[j] = conn[4 * e + j];
            Xn[j] = X[n[j]];
            Un[j] = U[n[j]];
            gx[j] = gu[j] = make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double xi = (q & 1) ? gp : -gp, eta = (q & 2) ? gp : -gp;
            double D0[4], D1[4];
            shape_derivs(xi, eta, D0, D1);
            double a = 0, b = 0, c = 0, d = 0;
            double2 g0 = make_double2(0, 0), g1 = make_double2(0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a += Xn[j].x * D0[j]; b += Xn[j].x * D1[j]; c += Xn[j].y * D0[j]; d += Xn[j].y * D1[j];
                g0.x += Un[j].x * D0[j]; g0.y += Un[j].y * D0[j]; g1.x += Un[j].x * D1[j]; g1.y += Un[j].y * D1[j];
            }
            JacGrad o;
            if (gX) {
                e_loc += jac_point<true>(a, b, c, d, g0, g1, 1.0, k, o);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gx[j].x += o.da * D0[j] + o.db * D1[j];
                    gx[j].y += o.dc * D0[j] + o.dd * D1[j];
                    gu[j].x += o.dg0.x * D0[j] + o.dg1.x * D1[j];
                    gu[j].y += o.dg0.y * D0[j] + o.dg1.y * D1[j];
                }
            } else {
                e_loc += jac_point<false>(a, b, c, d, g0, g1, 1.0, k, o);
            }
        }
        if (gX) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsafeAtomicAdd(&gX[2 * (int64_t)n[j]], gx[j].x);
                unsafeAtomicAdd(&gX[2 * (int64_t)n[j] + 1], gx[j].y);
                unsafeAtomicAdd(&gU[2 * (int64_t)n[j]], gu[j].x);
                unsafeAtomicAdd(&gU[2 * (int64_t)n[j] + 1], gu[j].y);
            }
        }
    }
    const double tot = block_sum(e_loc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_acc, tot);
}

// per-point forward: u_h, detJ, grad_u at (xi, eta) in [-1,1]^2 of element elem_id
__global__ __launch_bounds__(kBlockQ) void quad4_eval_fwd_kernel(
    const double2 *__restrict__ X, const double2 *__restrict__ U, const int32_t *__restrict__ conn,
    const double2 *__restrict__ x_eval, const int64_t *__restrict__ elem_id, int64_t m,
    double2 *__restrict__ u_h, double *__restrict__ detJ, double4 *__restrict__ grad_u) {
    const int64_t stride = (int64_t)gridDim.x * kBlockQ;
    for (int64_t p = (int64_t)blockIdx.x * kBlockQ + threadIdx.x; p < m; p += stride) {
        const int64_t e = elem_id[p];
        const double2 r = x_eval[p];
        double D0[4], D1[4];
        shape_derivs(r.x, r.y, D0, D1);
        double a = 0, b = 0, c = 0, d = 0;
        double2 g0 = make_double2(0, 0), g1 = make_double2(0, 0), uh = make_double2(0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t nj = conn[4 * e + j];
            const double2 Xj = X[nj], Uj = U[nj];
            const double Nj = 0.25 * (1.0 + corner_xi(j) * r.x) * (1.0 + corner_eta(j) * r.y);
            uh.x += Nj * Uj.x; uh.y += Nj * Uj.y;
            a += Xj.x * D0[j]; b += Xj.x * D1[j]; c += Xj.y * D0[j]; d += Xj.y * D1[j];
            g0.x += Uj.x * D0[j]; g0.y += Uj.y * D0[j]; g1.x += Uj.x * D1[j]; g1.y += Uj.y * D1[j];
        }
        const double det = a * d - b * c, inv = 1.0 / det;
        u_h[p] = uh;
        detJ[p] = det;
        grad_u[p] = make_double4((g0.x * d - g1.x * b) * inv, (g1.x * a - g0.x * c) * inv,
                                 (g0.y * d - g1.y * b) * inv, (g1.y * a - g0.y * c) * inv);
    }
}

// per-point backward: cotangents cu [M][2], cd [M], cg [M][2][2] -> gX, gU (accumulated)
__global__ __launch_bounds__(kBlockQ) void quad4_eval_bwd_kernel(
    const double2 *__restrict__ X, const double2 *__restrict__ U, const int32_t *__restrict__ conn,
    const double2 *__restrict__ x_eval, const int64_t *__restrict__ elem_id, int64_t m,
    const double2 *__restrict__ cu, const double *__restrict__ cd, const double4 *__restrict__ cg,
    double *__restrict__ gX, double *__restrict__ gU) {
    const int64_t stride = (int64_t)gridDim.x * kBlockQ;
    for (int64_t p = (int64_t)blockIdx.x * kBlockQ + threadIdx.x; p < m; p += stride) {
        const int64_t e = elem_id[p];
        const double2 r = x_eval[p];
        double D0[4], D1[4];
        shape_derivs(r.x, r.y, D0, D1);
        int32_t n[4];
        double a = 0, b = 0, c = 0, d = 0;
        double2 g0 = make_double2(0, 0), g1 = make_double2(0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            n[j] = conn[4 * e + j];
            const double2 Xj = X[n[j]], Uj = U[n[j]];
            a += Xj.x * D0[j]; b += Xj.x * D1[j]; c += Xj.y * D0[j]; d += Xj.y * D1[j];
            g0.x += Uj.x * D0[j]; g0.y += Uj.y * D0[j]; g1.x += Uj.x * D1[j]; g1.y += Uj.y * D1[j];
        }
        const double det = a * d - b * c, inv = 1.0 / det;
        const double h00 = (g0.x * d - g1.x * b) * inv, h01 = (g1.x * a - g0.x * c) * inv;
        const double h10 = (g0.y * d - g1.y * b) * inv, h11 = (g1.y * a - g0.y * c) * inv;
        const double4 P = cg ? cg[p] : make_double4(0, 0, 0, 0);
        const double2 q = cu ? cu[p] : make_double2(0, 0);
        const double cdet = cd ? cd[p] : 0.0;
        const double dg0x = (P.x * d - P.y * c) * inv, dg0y = (P.z * d - P.w * c) * inv;
        const double dg1x = (P.y * a - P.x * b) * inv, dg1y = (P.w * a - P.z * b) * inv;
        const double ddet = cdet - (P.x * h00 + P.y * h01 + P.z * h10 + P.w * h11) * inv;
        const double da = (P.y * g1.x + P.w * g1.y) * inv + ddet * d;
        const double db = -(P.x * g1.x + P.z * g1.y) * inv - ddet * c;
        const double dc = -(P.y * g0.x + P.w * g0.y) * inv - ddet * b;
        const double dd = (P.x * g0.x + P.z * g0.y) * inv + ddet * a;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double Nj = 0.25 * (1.0 + corner_xi(j) * r.x) * (1.0 + corner_eta(j) * r.y);
            unsafeAtomicAdd(&gX[2 * (int64_t)n[j]], da * D0[j] + db * D1[j]);
            unsafeAtomicAdd(&gX[2 * (int64_t)n[j] + 1], dc * D0[j] + dd * D1[j]);
            unsafeAtomicAdd(&gU[2 * (int64_t)n[j]], dg0x * D0[j] + dg1x * D1[j] + Nj * q.x);
            unsafeAtomicAdd(&gU[2 * (int64_t)n[j] + 1], dg0y * D0[j] + dg1y * D1[j] + Nj * q.y);
        }
    }
}

static int grid_q(int64_t n) {
    int64_t g = (n + kBlockQ - 1) / kBlockQ;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace hfem

using namespace hfem;

extern "C" int hfem_quad4_energy_atomic(int device, const double *X, const double *U, const int32_t *conn,
                                        int64_t e_begin, int64_t e_end, int64_t nn, const double mat[4],
                                        double *loss_acc, double *gX, double *gU, void *stream) {
    HFEM_ARG_CHECK(X && U && conn && mat && loss_acc, "null pointer");
    HFEM_ARG_CHECK(e_begin >= 0 && e_end >= e_begin && nn >= 0, "bad element range");
    HFEM_ARG_CHECK((gX == nullptr) == (gU == nullptr), "gX and gU must both be given or both NULL");
    if (int rc = use_device(device)) return rc;
    if (e_end == e_begin) return 0;
    Tri3Consts k;
    k.c11 = mat[0]; k.c12 = mat[1]; k.c22 = mat[2]; k.c33 = mat[3];
    k.W = 1.0;
    for (int i = 0; i < 6; ++i) k.Bk[i] = 0.0;
    hipLaunchKernelGGL(quad4_energy_atomic_kernel, dim3(grid_q(e_end - e_begin)), dim3(kBlockQ), 0,
                       (hipStream_t)stream, (const double2 *)X, (const double2 *)U, conn, e_begin, e_end, k,
                       loss_acc, gX, gU);
    return launch_status("hfem_quad4_energy_atomic");
}

extern "C" int hfem_quad4_eval_fwd(int device, const double *X, const double *U, const int32_t *conn,
                                   const double *x_eval, const int64_t *elem_id, int64_t m, double *u_h,
                                   double *detJ, double *grad_u, void *stream) {
    HFEM_ARG_CHECK(m >= 0, "negative point count");
    if (m == 0) return 0;
    HFEM_ARG_CHECK(X && U && conn && x_eval && elem_id && u_h && detJ && grad_u, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(quad4_eval_fwd_kernel, dim3(grid_q(m)), dim3(kBlockQ), 0, (hipStream_t)stream,
                       (const double2 *)X, (const double2 *)U, conn, (const double2 *)x_eval, elem_id, m,
                       (double2 *)u_h, detJ, (double4 *)grad_u);
    return launch_status("hfem_quad4_eval_fwd");
}

extern "C" int hfem_quad4_eval_bwd(int device, const double *X, const double *U, const int32_t *conn,
                                   const double *x_eval, const int64_t *elem_id, int64_t m, const double *cu,
                                   const double *cd, const double *cg, double *gX, double *gU, void *stream) {
    HFEM_ARG_CHECK(m >= 0, "negative point count");
    if (m == 0) return 0;
    HFEM_ARG_CHECK(X && U && conn && x_eval && elem_id && gX && gU, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(quad4_eval_bwd_kernel, dim3(grid_q(m)), dim3(kBlockQ), 0, (hipStream_t)stream,
                       (const double2 *)X, (const double2 *)U, conn, (const double2 *)x_eval, elem_id, m,
                       (const double2 *)cu, cd, (const double4 *)cg, gX, gU);
    return launch_status("hfem_quad4_eval_bwd");
}

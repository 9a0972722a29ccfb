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
// Two energy kernels: quad4_energy_fast_kernel on the owner-computes tile plan (production) and the planless
// quad4_energy_atomic_kernel (one thread per element, fp64 global atomics; independent algebra, cross-check).
#include <hip/hip_runtime.h>

#include "hfem_device.h"
#include "hfem_plan_dev.h"

namespace hfem {

constexpr int kBlockQ = 256;

int g_quad4_const_caps = 1;   // default tile shape: instance with a compile-time accumulator stride
// Lab build only (-DHFEM_LAB): phase offset of groups of resident workgroups (bit `shift` of the launch index
// upward), in 10 ns ticks; -1 = 2 us when the launch has at least 1.5 rounds of tiles.  On a warm chip it changes
// nothing on Q1M (27.8 us with and without; the gain first read on a cold chip was clock ramp).
int g_quad4_stagger_groups = 2;
int g_quad4_stagger = 0, g_quad4_stagger_shift = 8;
int g_quad4_pipe = 0;     // 0: one workgroup per tile; k > 0: persistent pipelined kernel, k workgroups per CU
int g_quad4_bits = 0;     // lab: bit 0 = NO raised wave priority through the memory phases, bit 1 = leftover slot chunk rotates over the waves
int g_quad4_ablate = 0;   // bit 0 = no element math, bit 1 = no LDS atomics (tiled kernel)

struct JacGrad {          // dL/d(a,b,c,d), dL/dG0, dL/dG1
    double da, db, dc, dd;
    double2 dg0, dg1;
};

// energy density * |det| at one point with weight w, and its gradient w.r.t. (a..d, G0, G1)
// beta_w (body-force instances): w * b(xi_q).u_h(xi_q) of this point -- the density is w psi - beta_w; with the
// cotangents below it adds only the sign(det) (-beta_w) cof(J) term (u_h's own gradient is added by the caller).
// Cofactor form (round 3): with adj = [[d, -b], [-c, a]] the UNSCALED h^ = G adj^T = det H, eps^ = (h^00, h^11, h^01 + h^10),
// sigma^ = C eps^ and  w |det| psi = t2/2 (eps^ . sigma^)  with  t2 = w / |det| = copysign(w, det) / det.  Nothing is
// pre-scaled by 1/det (no a/det .. d/det), dE/dh^ = t2 sigma^, and the only trace of the 1/|det| factor in the backward
// is  dE/d det = -E / det.  51 fp64 operations per point with gradients (57 in the scaled form); `inv` = 1 / det comes
// from the caller, which inverts the four determinants of an element with ONE reciprocal.
template <bool GRAD>
__device__ __forceinline__ double jac_point(double a, double b, double c, double d, double2 g0, double2 g1, double det, double inv,
                                            double w, const Tri3Consts &k, JacGrad &o, double beta_w = 0.0, double *A_out = nullptr) {
    const double t2 = __builtin_copysign(w, det) * inv;                    // w / |det|  (sign through one v_bfi)
    const double h00 = g0.x * d - g1.x * b, h01 = g1.x * a - g0.x * c;
    const double h10 = g0.y * d - g1.y * b, h11 = g1.y * a - g0.y * c;
    const double gam = h01 + h10;
    const double sxx = k.c11 * h00 + k.c12 * h11, syy = k.c12 * h00 + k.c22 * h11, sxy = k.c33 * gam;
    const double hs = h00 * sxx + h11 * syy + gam * sxy;
    double e = (0.5 * t2) * hs;                                            // w |det| psi
    if (A_out) *A_out = fabs(det);
    if (GRAD) {
        const double p00 = t2 * sxx, p01 = t2 * sxy, p11 = t2 * syy;       // dE / d h^  (symmetric)
        o.dg0 = make_double2(p00 * d - p01 * c, p01 * d - p11 * c);
        o.dg1 = make_double2(p01 * a - p00 * b, p11 * a - p01 * b);
        double ddet = -e * inv;                                            // through t2 = w sign(det) / det
        if (A_out) ddet -= __builtin_copysign(1.0, det) * beta_w;          // body-force instances only: -sign(det) beta_w
        o.da = (p01 * g1.x + p11 * g1.y) + ddet * d;
        o.db = -(p00 * g1.x + p01 * g1.y) - ddet * c;
        o.dc = -(p01 * g0.x + p11 * g0.y) - ddet * b;
        o.dd = (p00 * g0.x + p01 * g0.y) + ddet * a;
    }
    if (A_out) e -= fabs(det) * beta_w;
    return e;
}

// the four determinants of an element inverted with ONE v_rcp_f64 (quarter-rate) + Newton: 1/d_i from 1/(d0 d1 d2 d3).
// A vanishing determinant makes all four results Inf / NaN -- the element's energy is NaN either way.
__device__ __forceinline__ void rcp4(const double (&dt)[4], double (&iv)[4]) {
    const double p01 = dt[0] * dt[1], p23 = dt[2] * dt[3];
    const double r = fast_rcp(p01 * p23);
    const double r01 = r * p23, r23 = r * p01;                             // 1/(d0 d1), 1/(d2 d3)
    iv[0] = r01 * dt[1]; iv[1] = r01 * dt[0]; iv[2] = r23 * dt[3]; iv[3] = r23 * dt[2];
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

// Whole-element energy (2x2 Gauss, weights 1) and its gradient w.r.t. the four nodes, in the bilinear
// coefficient form.  For a nodal field v_0..v_3:  4 dv/dxi = a1 + eta a3,  4 dv/deta = a2 + xi a3  with
//   a1 = (v1-v0)+(v2-v3),  a2 = (v3-v0)+(v2-v1),  a3 = (v2-v3)-(v1-v0).
// H = G J^-T does not see the common factor 4 and |det| sees 16, so the points are evaluated on the unscaled
// coefficients with weight 1/16; cotangents are accumulated per coefficient (the a3 part as +-da +-db, scaled by
// 1/sqrt(3) once) and spread to the nodes with +-1 at the end.  ~370 fp64 instructions per element with
// gradients (the node-by-node D_N form above costs ~490).
// HASB: body force b(xi_q) at the four Gauss points (k.Bk... is the TRI3 table; QUAD4 takes Bq[q] = b at point q):
// e -= sum_q |det_q| u_h(q).b_q, dU_k -= sum_q |det_q| N_k(q) b_q, and the |det_q| dependence through jac_point.
// PHYS: the opt-in physical gradient convention, grad_u = G Jinv instead of the reference's G Jinv^T.  As for
// TRI3 (hfem_device.h), E_phys(a, b, c, d) = E_ref(a, c, b, d): b and c swap on the way into jac_point and their cotangents
// swap on the way out.
template <bool GRAD, bool HASB = false, bool PHYS = false>
__device__ __forceinline__ double quad4_element(const double2 (&Xn)[4], const double2 (&Un)[4], const Tri3Consts &k,
                                                double2 (&gx)[4], double2 (&gu)[4], const double2 *Bq = nullptr) {
    const double gp = 0.57735026918962576451;   // 1/sqrt(3)
    double a1[4], a2[4], a3[4];                  // components: x, y, ux, uy
    {
        const double v[4][4] = {{Xn[0].x, Xn[1].x, Xn[2].x, Xn[3].x}, {Xn[0].y, Xn[1].y, Xn[2].y, Xn[3].y},
                                {Un[0].x, Un[1].x, Un[2].x, Un[3].x}, {Un[0].y, Un[1].y, Un[2].y, Un[3].y}};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double s = v[i][1] - v[i][0], t = v[i][2] - v[i][3];
            a1[i] = s + t;
            a3[i] = t - s;
            a2[i] = (v[i][3] - v[i][0]) + (v[i][2] - v[i][1]);
        }
    }
    // Points in the order (-,-) (+,-) (-,+) (+,+).  With u_q = dE/d(4 dv/dxi) and v_q = dE/d(4 dv/deta) at point q:
    //   d a1 = sum u_q, d a2 = sum v_q, d a3 = gp ((u2+u3) - (u0+u1) + (v1+v3) - (v0+v2)); the pair sums are shared.
    double U[4][4], V[4][4];
    double2 gub[4] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
    double e = 0.0;
    // J(xi, eta) is affine: a, c depend on eta only, b, d on xi only -- two values each; the four determinants first, one
    // reciprocal for all of them
    const double am = a1[0] - gp * a3[0], ap = a1[0] + gp * a3[0], cm = a1[1] - gp * a3[1], cp = a1[1] + gp * a3[1];
    const double bm = a2[0] - gp * a3[0], bp = a2[0] + gp * a3[0], dm = a2[1] - gp * a3[1], dp = a2[1] + gp * a3[1];
    const double dets[4] = {am * dm - bm * cm, am * dp - bp * cm, ap * dm - bm * cp, ap * dp - bp * cp};
    double invs[4];
    rcp4(dets, invs);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double xi = (q & 1) ? gp : -gp, eta = (q & 2) ? gp : -gp;
        const double a = (q & 2) ? ap : am, b = (q & 1) ? bp : bm;
        const double c = (q & 2) ? cp : cm, d = (q & 1) ? dp : dm;
        const double2 g0 = make_double2(a1[2] + eta * a3[2], a1[3] + eta * a3[3]);
        const double2 g1 = make_double2(a2[2] + xi * a3[2], a2[3] + xi * a3[3]);
        JacGrad o;
        if (HASB) {
            // N_k(q) = (1 + xi_k xi)(1 + eta_k eta) / 4; the unscaled coefficients carry |det| x 16: weight 1/16
            double nk[4], uhx = 0.0, uhy = 0.0, A16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                nk[j] = 0.25 * (1.0 + corner_xi(j) * xi) * (1.0 + corner_eta(j) * eta);
                uhx += nk[j] * Un[j].x;
                uhy += nk[j] * Un[j].y;
            }
            const double2 bq = Bq[q];
            e += jac_point<GRAD>(a, PHYS ? c : b, PHYS ? b : c, d, g0, g1, dets[q], invs[q], 0.0625, k, o, 0.0625 * (uhx * bq.x + uhy * bq.y), &A16);
            if (GRAD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gub[j].x -= (0.0625 * A16) * nk[j] * bq.x;
                    gub[j].y -= (0.0625 * A16) * nk[j] * bq.y;
                }
            }
        } else {
            e += jac_point<GRAD>(a, PHYS ? c : b, PHYS ? b : c, d, g0, g1, dets[q], invs[q], 0.0625, k, o);
        }
        if (GRAD) {
            U[q][0] = o.da; U[q][1] = PHYS ? o.db : o.dc; U[q][2] = o.dg0.x; U[q][3] = o.dg0.y;
            V[q][0] = PHYS ? o.dc : o.db; V[q][1] = o.dd; V[q][2] = o.dg1.x; V[q][3] = o.dg1.y;
        }
    }
    if (GRAD) {
        double g[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double u01 = U[0][i] + U[1][i], u23 = U[2][i] + U[3][i];
            const double v02 = V[0][i] + V[2][i], v13 = V[1][i] + V[3][i];
            const double d1 = u01 + u23, d2 = v02 + v13;
            const double d3 = gp * ((u23 - u01) + (v13 - v02)), p = d1 + d2, m = d1 - d2;
            g[0][i] = d3 - p;
            g[1][i] = m - d3;
            g[2][i] = p + d3;
            g[3][i] = -m - d3;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gx[j] = make_double2(g[j][0], g[j][1]);
            gu[j] = HASB ? make_double2(g[j][2] + gub[j].x, g[j][3] + gub[j].y) : make_double2(g[j][2], g[j][3]);
        }
    }
    return e;
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
                const double det = a * d - b * c;
                e_loc += jac_point<true>(a, b, c, d, g0, g1, det, fast_rcp(det), 1.0, k, o);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gx[j].x += o.da * D0[j] + o.db * D1[j];
                    gx[j].y += o.dc * D0[j] + o.dd * D1[j];
                    gu[j].x += o.dg0.x * D0[j] + o.dg1.x * D1[j];
                    gu[j].y += o.dg0.y * D0[j] + o.dg1.y * D1[j];
                }
            } else {
                const double det = a * d - b * c;
                e_loc += jac_point<false>(a, b, c, d, g0, g1, det, fast_rcp(det), 1.0, k, o);
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

// ------------------------------------------------------------------ tiled (owner-computes plan)
// Same structure as tri3_energy_fast_kernel (tri3_energy.hip): row maps + element records requested
// up front, nodes gathered into LDS through the free/fixed maps, one thread per element slot
// (8 x ds_read_b128, 4 Gauss points in registers, 16 conflict-free ds_add_f64 on owned corners),
// every owned gradient row written once with an sc1 (write-through) store.  Element records are two
// words per slot: {l0 | l1<<10 | l2<<20 | home<<30 | skip<<31} and {l3}.
// CAPO > 0: compile-time stride of the four accumulator arrays (their LDS addresses become one scaled id + immediate).
struct Quad4Body { double2 b[4]; };      // body force at the 2x2 Gauss points (reference coordinates)

// V2: row storage type (double2, or float2 for fp32 models -- the reference's default dtype: widened on load, gradient rows
// rounded once on store, fp64 arithmetic); PHYS: the opt-in physical gradient convention; SP: cache policy of the gradient
// stores (16 sc1 write-through, 2 nt: the plan's choice for meshes of 750 k nodes and more, tri3_energy.hip).
template <int BLOCK, int NPT, int EPT, int ABL, int CAPO = 0, bool HASB = false, int CAPN = 0, typename V2 = double2, bool PHYS = false, int SP = 16>
__global__ __launch_bounds__(BLOCK) void quad4_energy_fast_kernel(
    PlanDev pd, int tile_begin, const V2 *__restrict__ x_free, const V2 *__restrict__ x_fixed,
    const V2 *__restrict__ u_free, const V2 *__restrict__ u_fixed, Tri3Consts k,
    const double4 *__restrict__ T_edge, double4 Tconst, double *__restrict__ partials,
    V2 *__restrict__ gx_free, V2 *__restrict__ gu_free, int cap_nodes, int cap_owned_rt, int skip_edges,
    int stagger_ticks, int stagger_shift, unsigned long long *__restrict__ stamps, Quad4Body body) {
#define HFEM_QSTAMP(I)                                                                              \
    if ((ABL & 4) && threadIdx.x == 0) stamps[16 * (size_t)blockIdx.x + (I)] = __builtin_amdgcn_s_memrealtime();
    HFEM_QSTAMP(0)
    const int cap_owned = CAPO > 0 ? CAPO : cap_owned_rt;
    const int cap_n = CAPN > 0 ? CAPN : cap_nodes;       // CAPN > 0: the uv array's offset folds into the ds_read immediates
    extern __shared__ double2 lds[];
    double2 *nd_xy = lds;
    double2 *nd_uv = lds + cap_n;
    double *acc0 = reinterpret_cast<double *>(lds + 2 * cap_n);
    double *acc1 = acc0 + cap_owned, *acc2 = acc1 + cap_owned, *acc3 = acc2 + cap_owned;
    double *red = acc3 + cap_owned;

    const int tid = threadIdx.x;
    const int slot = xcd_tile(blockIdx.x, gridDim.x);
#ifdef HFEM_LAB
    const int lab_bits = skip_edges >> 8;
    skip_edges &= 1;
    if (!(lab_bits & 1)) mem_phase_begin();             // lab bit 0: NO raised priority through the memory phases
    const int rtid = (lab_bits & 2) ? ((tid + 64 * (slot & 3)) & (BLOCK - 1)) : tid;      // record lane: which wave takes the short last chunk
#else
    const int rtid = tid;
    mem_phase_begin();                                  // prologue at raised wave priority (hfem_plan_dev.h)
#endif
#ifdef HFEM_LAB
    // Phase offset between co-resident workgroups: without it every resident tile gathers at the same time and
    // then computes at the same time (HBM idle while the fp64 VALU works and vice versa).
    if (stagger_ticks > 0) {
        const int grp = (blockIdx.x >> (stagger_shift & 31)) & (stagger_shift >> 8);   // shift | (groups-1) << 8
        const long long wait = (long long)stagger_ticks * grp / (stagger_shift >> 8);
        const long long t_start = __builtin_amdgcn_s_memrealtime();       // 100 MHz: 10 ns ticks
        while ((long long)__builtin_amdgcn_s_memrealtime() - t_start < wait) __builtin_amdgcn_s_sleep(2);
    }
#endif
    // ---- row maps and element records from the tile index alone (uniform node / slot strides, plan.cpp), in flight together
    //      with the descriptor; unguarded loads (padding = valid rows / skip records), then the gather, issued back to back
    int2 s[NPT];
    uint32_t pk[EPT], pk3[EPT];
    const int2 *src = pd.node_src + (size_t)(tile_begin + slot) * pd.node_stride;
    const size_t rec0 = (size_t)(tile_begin + slot) * pd.elem_stride;
#pragma unroll
    for (int j = 0; j < NPT; ++j) s[j] = src[min(tid + j * BLOCK, pd.node_stride - 1)];      // lanes past the stride repeat its last record
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const size_t i = rec0 + min(rtid + j * BLOCK, pd.elem_stride - 1);
        pk[j] = pd.elem_pack[i];
        pk3[j] = pd.elem_pack_hi[i];
    }
    const TileDesc d = pd.tiles[tile_begin + slot];
    const int n_owned = d.n_owned;
    if ((ABL & 4) && threadIdx.x == 0 && n_owned >= 0) stamps[16 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    double vxx[NPT], vxy[NPT], vux[NPT], vuy[NPT];      // plain doubles (double2 arrays end up in scratch)
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const V2 *px = s[j].x >= 0 ? x_free + s[j].x : x_fixed + ~s[j].x;
        const V2 *pu = s[j].y >= 0 ? u_free + s[j].y : u_fixed + ~s[j].y;
        const V2 tx = *px, tu = *pu;
        vxx[j] = tx.x; vxy[j] = tx.y; vux[j] = tu.x; vuy[j] = tu.y;
    }
    __builtin_amdgcn_sched_barrier(0);                  // all gather loads are issued before the first is waited for
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (rtid + j * BLOCK >= d.n_elem) { pk[j] = kSkipBit; pk3[j] = 0u; }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int l = tid + j * BLOCK;
        if (l < d.n_node) { nd_xy[l] = make_double2(vxx[j], vxy[j]); nd_uv[l] = make_double2(vux[j], vuy[j]); }
        if (l < n_owned) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
    }
    HFEM_QSTAMP(2)
    __syncthreads();
    HFEM_QSTAMP(3)
#ifdef HFEM_LAB
    if (!(lab_bits & 1))
#endif
    mem_phase_end();

    double e_loc = 0.0;
#pragma unroll
    for (int jj = 0; jj < EPT; ++jj) {
        const uint32_t p = pk[jj];
        if (!(p & kSkipBit)) {
            const int l[4] = {(int)(p & kLocalMask), (int)((p >> kLocalBits) & kLocalMask),
                              (int)((p >> (2 * kLocalBits)) & kLocalMask), (int)(pk3[jj] & kLocalMask)};
            double2 Xn[4], Un[4], gx[4], gu[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Xn[j] = nd_xy[l[j]];
                Un[j] = nd_uv[l[j]];
            }
            double e;
            if (ABL & 1) {          // lab: no element math
                e = Xn[0].x + Un[3].y;
#pragma unroll
                for (int j = 0; j < 4; ++j) { gx[j] = Xn[j]; gu[j] = Un[j]; }
            } else {
                e = HASB ? quad4_element<true, true, PHYS>(Xn, Un, k, gx, gu, body.b) : quad4_element<true, false, PHYS>(Xn, Un, k, gx, gu);
            }
            if (ABL & 2) {          // lab: no LDS atomics (keep the math live)
#pragma unroll
                for (int j = 0; j < 4; ++j) e += gx[j].x + gx[j].y + gu[j].x + gu[j].y;
            }
            if (p & kHomeBit) e_loc += e;
            if (!(ABL & 2)) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (l[j] < n_owned) {
                        unsafeAtomicAdd(&acc0[l[j]], gx[j].x); unsafeAtomicAdd(&acc1[l[j]], gx[j].y);
                        unsafeAtomicAdd(&acc2[l[j]], gu[j].x); unsafeAtomicAdd(&acc3[l[j]], gu[j].y);
                    }
            }
        }
    }
    const int n_edge = skip_edges ? 0 : d.n_edge;
    for (int i = tid; i < n_edge; i += BLOCK) {          // boundary tiles only
        const uint32_t p = pd.edge_pack[d.edge_off + i];
        const int l0 = (int)(p & kLocalMask), l1 = (int)((p >> kLocalBits) & kLocalMask);
        const double4 tt = T_edge ? T_edge[pd.edge_gid[d.edge_off + i]] : Tconst;
        double2 gx[2], gu[2];
        const double wk = edge2_element<true>(nd_xy[l0], nd_xy[l1], nd_uv[l0], nd_uv[l1], tt, gx, gu);
        if (p & kHomeBit) e_loc -= wk;
        if (l0 < n_owned) {
            unsafeAtomicAdd(&acc0[l0], gx[0].x); unsafeAtomicAdd(&acc1[l0], gx[0].y);
            unsafeAtomicAdd(&acc2[l0], gu[0].x); unsafeAtomicAdd(&acc3[l0], gu[0].y);
        }
        if (l1 < n_owned) {
            unsafeAtomicAdd(&acc0[l1], gx[1].x); unsafeAtomicAdd(&acc1[l1], gx[1].y);
            unsafeAtomicAdd(&acc2[l1], gu[1].x); unsafeAtomicAdd(&acc3[l1], gu[1].y);
        }
    }
    {
        const double w = wave_sum(e_loc);
        if ((tid & 63) == 0) red[tid >> 6] = w;          // one slot per wave: summed in wave order below
    }
    HFEM_QSTAMP(4)
#ifdef HFEM_LAB
    if (!(lab_bits & 1))
#endif
    mem_phase_begin();                                  // write-out
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): all loads returned long ago; keeps per-store vmcnt waits out of the write-out
    __syncthreads();
    HFEM_QSTAMP(5)

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    constexpr bool kWide = sizeof(V2) == 16;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)gx_free, 0, 0x7FFFFFF0, 0x00020000);
    __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void *)gu_free, 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int l = tid + j * BLOCK;
        if (l < n_owned) {
            if (gx_free && s[j].x >= 0) {
                V2 v;
                v.x = acc0[l]; v.y = acc1[l];           // rounds once for float2
                if (kWide) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), rx, s[j].x * 16, 0, SP);
                else __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), rx, s[j].x * 8, 0, SP);
            }
            if (gu_free && s[j].y >= 0) {
                V2 v;
                v.x = acc2[l]; v.y = acc3[l];
                if (kWide) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), ru, s[j].y * 16, 0, SP);
                else __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), ru, s[j].y * 8, 0, SP);
            }
        }
    }
    if (tid == 0) {                                     // fixed order: the tile energy is bit-reproducible
        double tile_e = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) tile_e += red[w];
        partials[slot] = tile_e;
    }
    HFEM_QSTAMP(6)
    if (ABL & 4) {
        __builtin_amdgcn_s_waitcnt(0);
        HFEM_QSTAMP(7)
    }
#undef HFEM_QSTAMP
}

#ifdef HFEM_LAB
// ------------------------------------------------------------------ persistent, pipelined (QUAD4)
// The one-workgroup-per-tile kernel above spends its memory phases (gather, write-out: ~16 us on Q1M) with the
// fp64 VALU idle and its element stage (~11 us) with the memory system idle -- the two add up.  QUAD4's element
// stage is as long as a tile's gather latency, so a register-staged software pipeline pays here (it does not for
// TRI3, whose element stage is a third of it: tri3_energy_pipe_kernel): a persistent workgroup walks a contiguous
// run of tiles; the gather of tile t+1 is issued right after the barrier that starts tile t's element loop and
// lands in VGPRs while the loop runs; tile t+2's row maps are requested at the same time; tile t's gradient stores
// are issued last and drain under tile t+1's loop.  The element loop touches no VMEM-loaded register (its records
// and the tile descriptors are staged through LDS), and every prefetch load is straight-line code (idle lanes clamp
// to row 0, "no next tile" is a zero-sized tile), so the compiler emits counted vmcnt(N) waits, none inside the loop.
// LDS: xy[cap_nodes] double2 | uv[cap_nodes] double2 | acc[4][cap_owned] | red[16] | desc[16][8] | pk[cap_elems] | pk3[cap_elems]
template <int BLOCK, int NPT, int EPT>
__global__ __launch_bounds__(BLOCK) void quad4_energy_pipe_kernel(
    PlanDev pd, int tile_begin, int n_tiles, const double2 *__restrict__ x_free, const double2 *__restrict__ x_fixed,
    const double2 *__restrict__ u_free, const double2 *__restrict__ u_fixed, Tri3Consts k,
    const double4 *__restrict__ T_edge, double4 Tconst, double *__restrict__ partials, double2 *__restrict__ gx_free,
    double2 *__restrict__ gu_free, int cap_nodes, int cap_owned, int cap_elems, int skip_edges) {
    extern __shared__ double2 lds[];
    double2 *nd_xy = lds;
    double2 *nd_uv = lds + cap_nodes;
    double *acc0 = reinterpret_cast<double *>(lds + 2 * cap_nodes);
    double *acc1 = acc0 + cap_owned, *acc2 = acc1 + cap_owned, *acc3 = acc2 + cap_owned;
    double *red = acc3 + cap_owned;
    int *ldesc = reinterpret_cast<int *>(red + 16);                      // [kPipeMaxTiles][8]
    uint32_t *lpk = reinterpret_cast<uint32_t *>(ldesc + 8 * kPipeMaxTiles);
    uint32_t *lpk3 = lpk + cap_elems;

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)gx_free, 0, 0x7FFFFFF0, 0x00020000);
    __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc((void *)gu_free, 0, 0x7FFFFFF0, 0x00020000);
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const int w = xcd_tile(blockIdx.x, G);           // consecutive w share an XCD (L2 reuse of halos)
    const int t0 = (int)(((long long)w * n_tiles) / G), t1 = (int)(((long long)(w + 1) * n_tiles) / G);
    double e_loc = 0.0;

    if (tid < 8 * (t1 - t0)) ldesc[tid] = reinterpret_cast<const int *>(pd.tiles + tile_begin + t0)[tid];
    __syncthreads();
#define HFEM_DESC(T, F) __builtin_amdgcn_readfirstlane(ldesc[8 * ((T) - t0) + (F)])
    // TileDesc fields: 0 elem_off 1 n_elem 2 node_off 3 n_node 4 n_owned 5 edge_off 6 n_edge
    if (t0 < t1) {
        int d_n_node, d_n_owned, d_n_elem, d_edge_off, d_n_edge;
        int q_n_node, q_n_owned, q_elem_off, q_n_elem, q_edge_off, q_n_edge;
        int2 s[NPT], s1[NPT], s2[NPT];
        static_assert(NPT == 3, "the prefetch registers are named scalars (arrays of double2 end up in scratch)");
        double2 xy0, xy1, xy2, uv0, uv1, uv2;
        uint32_t pkr[EPT], pkr3[EPT];
        const int2 *nsrc = pd.node_src;
        const uint32_t *epk = pd.elem_pack, *epk3 = pd.elem_pack_hi;
#define HFEM_GATHER1(J, XY, UV, SRC, NN)                                                           \
    {                                                                                              \
        const bool ok = tid + J * BLOCK < (NN);                                                    \
        const int ix = ok ? SRC[J].x : 0, iu = ok ? SRC[J].y : 0;                                  \
        XY = *(ix >= 0 ? x_free + ix : x_fixed + ~ix);                                             \
        UV = *(iu >= 0 ? u_free + iu : u_fixed + ~iu);                                             \
    }
#define HFEM_GATHER(SRC, NN)                                                                       \
    HFEM_GATHER1(0, xy0, uv0, SRC, NN) HFEM_GATHER1(1, xy1, uv1, SRC, NN) HFEM_GATHER1(2, xy2, uv2, SRC, NN)
#define HFEM_LOAD_SRC(DST, OFF, NN)                                                                \
    _Pragma("unroll") for (int j = 0; j < NPT; ++j) {                                              \
        const int l = tid + j * BLOCK;                                                             \
        DST[j] = nsrc[(OFF) + (l < (NN) ? l : 0)];                                                 \
    }
#define HFEM_LOAD_PK(OFF, NN)                                                                      \
    _Pragma("unroll") for (int j = 0; j < EPT; ++j) {                                              \
        const int i = tid + j * BLOCK;                                                             \
        pkr[j] = epk[(OFF) + (i < (NN) ? i : 0)];                                                  \
        pkr3[j] = epk3[(OFF) + (i < (NN) ? i : 0)];                                                \
    }
#define HFEM_STAGE1(J, XY, UV)                                                                     \
    {                                                                                              \
        const int l = tid + J * BLOCK;                                                             \
        if (l < d_n_node) { nd_xy[l] = XY; nd_uv[l] = UV; }                                        \
        if (l < d_n_owned) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }         \
    }
#define HFEM_STAGE()                                                                               \
    HFEM_STAGE1(0, xy0, uv0) HFEM_STAGE1(1, xy1, uv1) HFEM_STAGE1(2, xy2, uv2)                     \
    _Pragma("unroll") for (int j = 0; j < EPT; ++j) {                                              \
        const int i = tid + j * BLOCK;                                                             \
        if (i < d_n_elem) { lpk[i] = pkr[j]; lpk3[i] = pkr3[j]; }                                  \
    }
        {
            const int c_elem_off = HFEM_DESC(t0, 0), c_node_off = HFEM_DESC(t0, 2);
            d_n_elem = HFEM_DESC(t0, 1); d_n_node = HFEM_DESC(t0, 3); d_n_owned = HFEM_DESC(t0, 4);
            d_edge_off = HFEM_DESC(t0, 5); d_n_edge = HFEM_DESC(t0, 6);
            const bool hn = t0 + 1 < t1;
            const int tq = hn ? t0 + 1 : t0;
            const int q_node_off = hn ? HFEM_DESC(tq, 2) : 0;
            q_elem_off = hn ? HFEM_DESC(tq, 0) : 0; q_n_elem = hn ? HFEM_DESC(tq, 1) : 0;
            q_n_node = hn ? HFEM_DESC(tq, 3) : 0; q_n_owned = hn ? HFEM_DESC(tq, 4) : 0;
            q_edge_off = hn ? HFEM_DESC(tq, 5) : 0; q_n_edge = hn ? HFEM_DESC(tq, 6) : 0;
            HFEM_LOAD_SRC(s, c_node_off, d_n_node)
            HFEM_LOAD_SRC(s1, q_node_off, q_n_node)
#pragma unroll
            for (int j = 0; j < NPT; ++j) s2[j] = make_int2(0, 0);
            HFEM_LOAD_PK(c_elem_off, d_n_elem)
            HFEM_GATHER(s, d_n_node)
            HFEM_STAGE()
        }
        for (int t = t0; t < t1; ++t) {
            __syncthreads();                       // tile t is staged
            {
                const bool hn2 = t + 2 < t1;
                const int tr = hn2 ? t + 2 : t;
                const int r_node_off = hn2 ? HFEM_DESC(tr, 2) : 0, r_n_node = hn2 ? HFEM_DESC(tr, 3) : 0;
                HFEM_GATHER(s1, q_n_node)
                HFEM_LOAD_PK(q_elem_off, q_n_elem)
                HFEM_LOAD_SRC(s2, r_node_off, r_n_node)
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- elements of tile t: LDS in, LDS out -- no VMEM dependence
            const int n_owned = d_n_owned;
#pragma unroll
            for (int jj = 0; jj < EPT; ++jj) {
                const int i = tid + jj * BLOCK;
                if (i < d_n_elem && !(lpk[i] & kSkipBit)) {
                    const uint32_t p = lpk[i];
                    const int l[4] = {(int)(p & kLocalMask), (int)((p >> kLocalBits) & kLocalMask),
                                      (int)((p >> (2 * kLocalBits)) & kLocalMask), (int)(lpk3[i] & kLocalMask)};
                    double2 Xn[4], Un[4], gx[4], gu[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { Xn[j] = nd_xy[l[j]]; Un[j] = nd_uv[l[j]]; }
                    const double e = quad4_element<true>(Xn, Un, k, gx, gu);
                    if (p & kHomeBit) e_loc += e;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (l[j] < n_owned) {
                            unsafeAtomicAdd(&acc0[l[j]], gx[j].x); unsafeAtomicAdd(&acc1[l[j]], gx[j].y);
                            unsafeAtomicAdd(&acc2[l[j]], gu[j].x); unsafeAtomicAdd(&acc3[l[j]], gu[j].y);
                        }
                }
            }
            const int n_edge = skip_edges ? 0 : d_n_edge;
            for (int i = tid; i < n_edge; i += BLOCK) {      // boundary tiles only
                const uint32_t p = pd.edge_pack[d_edge_off + i];
                const int l0 = (int)(p & kLocalMask), l1 = (int)((p >> kLocalBits) & kLocalMask);
                const double4 tt = T_edge ? T_edge[pd.edge_gid[d_edge_off + i]] : Tconst;
                double2 gx[2], gu[2];
                const double wk = edge2_element<true>(nd_xy[l0], nd_xy[l1], nd_uv[l0], nd_uv[l1], tt, gx, gu);
                if (p & kHomeBit) e_loc -= wk;
                if (l0 < n_owned) {
                    unsafeAtomicAdd(&acc0[l0], gx[0].x); unsafeAtomicAdd(&acc1[l0], gx[0].y);
                    unsafeAtomicAdd(&acc2[l0], gu[0].x); unsafeAtomicAdd(&acc3[l0], gu[0].y);
                }
                if (l1 < n_owned) {
                    unsafeAtomicAdd(&acc0[l1], gx[1].x); unsafeAtomicAdd(&acc1[l1], gx[1].y);
                    unsafeAtomicAdd(&acc2[l1], gu[1].x); unsafeAtomicAdd(&acc3[l1], gu[1].y);
                }
            }
            __syncthreads();                       // all accumulation of tile t done
            // ---- tail: A store addresses  B accumulators -> registers  C stage tile t+1 (waits for the prefetch)
            //      D rotate row maps  E issue the stores LAST, so they drain under the next tile's element loop
            double2 ogx[NPT], ogu[NPT];
            int rx[NPT], ru[NPT];                  // destination rows (-1: nothing to store); no pointer arrays (scratch)
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                const int l = tid + j * BLOCK;
                rx[j] = (l < n_owned && gx_free && s[j].x >= 0) ? s[j].x : -1;                        // A
                ru[j] = (l < n_owned && gu_free && s[j].y >= 0) ? s[j].y : -1;
                ogx[j] = ogu[j] = make_double2(0.0, 0.0);
                if (l < n_owned) {                                                                   // B
                    ogx[j] = make_double2(acc0[l], acc1[l]);
                    ogu[j] = make_double2(acc2[l], acc3[l]);
                }
            }
            d_n_node = q_n_node; d_n_owned = q_n_owned; d_n_elem = q_n_elem; d_edge_off = q_edge_off;    // C
            d_n_edge = q_n_edge;
            HFEM_STAGE()
#pragma unroll
            for (int j = 0; j < NPT; ++j) { s[j] = s1[j]; s1[j] = s2[j]; }                          // D
            {
                const bool hn2 = t + 2 < t1;
                const int tr = hn2 ? t + 2 : t;
                q_elem_off = hn2 ? HFEM_DESC(tr, 0) : 0; q_n_elem = hn2 ? HFEM_DESC(tr, 1) : 0;
                q_n_node = hn2 ? HFEM_DESC(tr, 3) : 0; q_n_owned = hn2 ? HFEM_DESC(tr, 4) : 0;
                q_edge_off = hn2 ? HFEM_DESC(tr, 5) : 0; q_n_edge = hn2 ? HFEM_DESC(tr, 6) : 0;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NPT; ++j) {                                                          // E (sc1: write-through)
                // branch-free: a lane with nothing to store passes an offset beyond the descriptor's range and the
                // hardware drops it -- the compiler can then count the stores (vmcnt(N), not vmcnt(0), at the loop top)
                const double2 vx = ogx[j], vu = ogu[j];           // scalars: taking an array element's address pins it in scratch
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&vx), rsx,
                                                       rx[j] >= 0 ? rx[j] * 16 : (int)0x80000000, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&vu), rsu,
                                                       ru[j] >= 0 ? ru[j] * 16 : (int)0x80000000, 0, 16);
            }
        }
#undef HFEM_GATHER
#undef HFEM_GATHER1
#undef HFEM_STAGE1
#undef HFEM_LOAD_SRC
#undef HFEM_LOAD_PK
#undef HFEM_STAGE
    }
#undef HFEM_DESC
    const double tot = block_sum(e_loc, red);
    if (tid == 0) partials[blockIdx.x] = tot;
}
#endif  // HFEM_LAB


// ------------------------------------------------------------------ deterministic (fixed-order) QUAD4
// HFEM_FLAG_DETERMINISTIC on a QUAD4 plan: the node-centric cross-check of tri3_det.hip for the extension element.  One
// thread per NODE walks the node -> (element, corner) adjacency in ascending element id, re-evaluates every adjacent element
// (quad4_element, all four corners -- 4x the flops, no atomics) and keeps its own corner's rows, then its Neumann edges,
// then ONE store per row; the loss is one thread per element / edge, shuffle-tree + wave-order block sums and a one-block
// sum.  Bit-identical run to run.  A checker, several times slower than the tiled kernel.
__device__ __forceinline__ double2 q4_row(const double2 *__restrict__ free_rows, const double2 *__restrict__ fixed_rows, int32_t src) {
    return src >= 0 ? free_rows[src] : fixed_rows[~src];
}

template <bool PHYS>
__global__ __launch_bounds__(256) void quad4_det_grad_kernel(
    int32_t nn, const int32_t *__restrict__ conn, const int32_t *__restrict__ x_src, const int32_t *__restrict__ u_src,
    const int32_t *__restrict__ adj_ptr, const int32_t *__restrict__ adj, const int32_t *__restrict__ edges,
    const int32_t *__restrict__ eadj_ptr, const int32_t *__restrict__ eadj, const double2 *__restrict__ x_free,
    const double2 *__restrict__ x_fixed, const double2 *__restrict__ u_free, const double2 *__restrict__ u_fixed,
    Tri3Consts k, Quad4Body body, const double4 *__restrict__ T_edge, double4 Tconst, int skip_edges,
    double2 *__restrict__ gx_free, double2 *__restrict__ gu_free) {
    const int32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= nn) return;
    double2 sx = make_double2(0.0, 0.0), su = make_double2(0.0, 0.0);
    for (int32_t i = adj_ptr[n]; i < adj_ptr[n + 1]; ++i) {
        const int32_t ec = adj[i], e = ec >> 2, c = ec & 3;
        double2 Xn[4], Un[4], gx[4], gu[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t nj = conn[4 * (int64_t)e + j];
            Xn[j] = q4_row(x_free, x_fixed, x_src[nj]);
            Un[j] = q4_row(u_free, u_fixed, u_src[nj]);
        }
        quad4_element<true, true, PHYS>(Xn, Un, k, gx, gu, body.b);
        // this node's corner by 0 / 1 weights, not by a select chain (which the compiler turns into an indexed load of a scratch
        // copy of the arrays); exact for finite rows (tri3_det.hip)
        const double m0 = c == 0 ? 1.0 : 0.0, m1 = c == 1 ? 1.0 : 0.0, m2 = c == 2 ? 1.0 : 0.0, m3 = c == 3 ? 1.0 : 0.0;
        sx.x += (m0 * gx[0].x + m1 * gx[1].x) + (m2 * gx[2].x + m3 * gx[3].x); sx.y += (m0 * gx[0].y + m1 * gx[1].y) + (m2 * gx[2].y + m3 * gx[3].y);
        su.x += (m0 * gu[0].x + m1 * gu[1].x) + (m2 * gu[2].x + m3 * gu[3].x); su.y += (m0 * gu[0].y + m1 * gu[1].y) + (m2 * gu[2].y + m3 * gu[3].y);
    }
    if (!skip_edges)
        for (int32_t i = eadj_ptr[n]; i < eadj_ptr[n + 1]; ++i) {
            const int32_t ge = eadj[i], g = ge >> 1, end = ge & 1;
            const int32_t ni = edges[2 * (int64_t)g], nj = edges[2 * (int64_t)g + 1];
            const double4 tt = T_edge ? T_edge[g] : Tconst;
            double2 gx[2], gu[2];
            edge2_element<true>(q4_row(x_free, x_fixed, x_src[ni]), q4_row(x_free, x_fixed, x_src[nj]),
                                q4_row(u_free, u_fixed, u_src[ni]), q4_row(u_free, u_fixed, u_src[nj]), tt, gx, gu);
            const double2 px = end ? gx[1] : gx[0], pu = end ? gu[1] : gu[0];      // selects, not a dynamic index (that puts the arrays in scratch)
            sx.x += px.x; sx.y += px.y;
            su.x += pu.x; su.y += pu.y;
        }
    const int32_t rx = x_src[n], ru = u_src[n];
    if (gx_free && rx >= 0) gx_free[rx] = sx;
    if (gu_free && ru >= 0) gu_free[ru] = su;
}

template <bool PHYS>
__global__ __launch_bounds__(256) void quad4_det_loss_kernel(
    int32_t ne, int32_t ned, const int32_t *__restrict__ conn, const int32_t *__restrict__ x_src,
    const int32_t *__restrict__ u_src, const int32_t *__restrict__ edges, const double2 *__restrict__ x_free,
    const double2 *__restrict__ x_fixed, const double2 *__restrict__ u_free, const double2 *__restrict__ u_fixed,
    Tri3Consts k, Quad4Body body, const double4 *__restrict__ T_edge, double4 Tconst, double *__restrict__ partials) {
    __shared__ double red[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (i < ne) {
        double2 Xn[4], Un[4], gx[4], gu[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t nj = conn[4 * i + j];
            Xn[j] = q4_row(x_free, x_fixed, x_src[nj]);
            Un[j] = q4_row(u_free, u_fixed, u_src[nj]);
        }
        v = quad4_element<false, true, PHYS>(Xn, Un, k, gx, gu, body.b);
    } else if (i < (int64_t)ne + ned) {
        const int64_t g = i - ne;
        const int32_t ni = edges[2 * g], nj = edges[2 * g + 1];
        double2 ex[2], eu[2];
        v = -edge2_element<false>(q4_row(x_free, x_fixed, x_src[ni]), q4_row(x_free, x_fixed, x_src[nj]),
                                  q4_row(u_free, u_fixed, u_src[ni]), q4_row(u_free, u_fixed, u_src[nj]),
                                  T_edge ? T_edge[g] : Tconst, ex, eu);
    }
    const double tot = block_sum(v, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void quad4_sum_partials_kernel(const double *__restrict__ partials, int n,
                                                                 double *__restrict__ out) {
    __shared__ double red[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partials[i];
    const double tot = block_sum(v, red);
    if (threadIdx.x == 0) out[0] = tot;
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

extern "C" int hfem_quad4_energy_plan(hfem_plan *plan, const double *x_free, const double *x_fixed,
                                      const double *u_free, const double *u_fixed, const double mat[4],
                                      const double *T_edge, const double Tconst[4], int32_t tile_begin,
                                      int32_t tile_end, double *loss_out, double *gx_free, double *gu_free,
                                      int32_t flags, void *stream) {
    return hfem_quad4_energy_plan_ex(plan, 0, x_free, x_fixed, u_free, u_fixed, mat, nullptr, T_edge, Tconst, tile_begin,
                                     tile_end, loss_out, gx_free, gu_free, flags, stream);
}

extern "C" int hfem_quad4_energy_plan_body(hfem_plan *plan, const double *x_free, const double *x_fixed,
                                           const double *u_free, const double *u_fixed, const double mat[4],
                                           const double Bq[8], const double *T_edge, const double Tconst[4],
                                           int32_t tile_begin, int32_t tile_end, double *loss_out, double *gx_free,
                                           double *gu_free, int32_t flags, void *stream) {
    return hfem_quad4_energy_plan_ex(plan, 0, x_free, x_fixed, u_free, u_fixed, mat, Bq, T_edge, Tconst, tile_begin, tile_end,
                                     loss_out, gx_free, gu_free, flags, stream);
}

// General form: dtype 0 = fp64 rows, 1 = fp32 rows (an fp32 model -- the reference's default dtype -- without widening copies:
// rows widened on load, gradient rows rounded once on store, arithmetic and loss_out fp64); flags may carry
// HFEM_FLAG_PHYSICAL_GRAD (grad_u = G Jinv) and, for fp64 rows on the whole plan, HFEM_FLAG_DETERMINISTIC.
extern "C" int hfem_quad4_energy_plan_ex(hfem_plan *plan, int32_t dtype, const void *x_free, const void *x_fixed,
                                         const void *u_free, const void *u_fixed, const double mat[4], const double Bq[8],
                                         const double *T_edge, const double Tconst[4], int32_t tile_begin,
                                         int32_t tile_end, double *loss_out, void *gx_free, void *gu_free, int32_t flags,
                                         void *stream) {
    HFEM_ARG_CHECK(plan && mat && loss_out, "null pointer");
    HFEM_ARG_CHECK(dtype == 0 || dtype == 1, "dtype: 0 = fp64 rows, 1 = fp32 rows");
    HFEM_ARG_CHECK(!(flags & (HFEM_FLAG_SUM_PREVIOUS | HFEM_FLAG_SAME_BANK | HFEM_FLAG_PEER_GET)), "QUAD4 extension: inline or deferred loss sum only, no in-launch get");
    Quad4Body body;
    bool hasb = false;
    for (int q = 0; q < 4; ++q) {
        body.b[q] = Bq ? make_double2(Bq[2 * q], Bq[2 * q + 1]) : make_double2(0.0, 0.0);
        hasb = hasb || body.b[q].x != 0.0 || body.b[q].y != 0.0;
    }
    HFEM_ARG_CHECK(plan->device >= 0, "host-only plan (created with device < 0) cannot launch");
    HFEM_ARG_CHECK(plan->host.npe == 4, "this plan was built for TRI3: use hfem_tri3_energy_plan");
    HFEM_ARG_CHECK(x_free && u_free, "x_free / u_free must be given");
    const HostPlan &h = plan->host;
    const int32_t nt = (int32_t)h.tiles.size();
    if (tile_end < 0) tile_end = nt;
    HFEM_ARG_CHECK(tile_begin >= 0 && tile_begin <= tile_end && tile_end <= nt, "bad tile range");
    HFEM_ARG_CHECK(h.ned == 0 || T_edge || Tconst, "plan has Neumann edges: need a traction table");
    HFEM_ARG_CHECK(h.max_nodes <= 4 * 256 && h.max_elems <= 4 * 256, "QUAD4 tile exceeds the kernel's register tiling");
    if (int rc = use_device(plan->device)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int n = tile_end - tile_begin;
    const double4 tc = Tconst ? make_double4(Tconst[0], Tconst[1], Tconst[2], Tconst[3]) : make_double4(0, 0, 0, 0);
    Tri3Consts k = make_consts(mat, 1.0, nullptr);
    const bool phys = (flags & HFEM_FLAG_PHYSICAL_GRAD) != 0;
    const int skip_edges = (flags & HFEM_FLAG_NO_EDGES) ? 1 : 0;
    void *gx = (flags & HFEM_FLAG_NO_GX) ? nullptr : gx_free, *gu = (flags & HFEM_FLAG_NO_GU) ? nullptr : gu_free;
    if (flags & HFEM_FLAG_DETERMINISTIC) {
        HFEM_ARG_CHECK(dtype == 0, "HFEM_FLAG_DETERMINISTIC: fp64 rows");
        HFEM_ARG_CHECK(tile_begin == 0 && tile_end == nt, "HFEM_FLAG_DETERMINISTIC: whole plan only");
        HFEM_ARG_CHECK(!(flags & HFEM_FLAG_NO_LOSS_SUM), "HFEM_FLAG_DETERMINISTIC always delivers the loss");
        if (int rc = det_prepare(plan, s)) return rc;
        const hfem_plan::Det &D = plan->det;
        const int32_t nn = (int32_t)h.nn, ne = (int32_t)h.ne, ned = skip_edges ? 0 : (int32_t)h.ned;
        if (nn > 0 && (gx || gu)) {
#define HFEM_Q4_DET_GRAD(P)                                                                                                 \
    hipLaunchKernelGGL((quad4_det_grad_kernel<P>), dim3((nn + 255) / 256), dim3(256), 0, s, nn, D.conn, D.x_src, D.u_src,     \
                       D.adj_ptr, D.adj, D.edges, D.eadj_ptr, D.eadj, (const double2 *)x_free, (const double2 *)x_fixed,     \
                       (const double2 *)u_free, (const double2 *)u_fixed, k, body, (const double4 *)T_edge, tc, skip_edges, \
                       (double2 *)gx, (double2 *)gu)
            if (phys) HFEM_Q4_DET_GRAD(true); else HFEM_Q4_DET_GRAD(false);
#undef HFEM_Q4_DET_GRAD
            if (int rc = launch_status("hfem_quad4_energy_plan(deterministic gradients)")) return rc;
        }
        const int nb = (int)(((int64_t)ne + ned + 255) / 256);
        if (nb > 0) {
#define HFEM_Q4_DET_LOSS(P)                                                                                                 \
    hipLaunchKernelGGL((quad4_det_loss_kernel<P>), dim3(nb), dim3(256), 0, s, ne, ned, D.conn, D.x_src, D.u_src, D.edges,    \
                       (const double2 *)x_free, (const double2 *)x_fixed, (const double2 *)u_free, (const double2 *)u_fixed, \
                       k, body, (const double4 *)T_edge, tc, D.partials)
            if (phys) HFEM_Q4_DET_LOSS(true); else HFEM_Q4_DET_LOSS(false);
#undef HFEM_Q4_DET_LOSS
        }
        hipLaunchKernelGGL(quad4_sum_partials_kernel, dim3(1), dim3(256), 0, s, D.partials, nb, loss_out);
        return launch_status("hfem_quad4_energy_plan(deterministic loss)");
    }
    if (n > 0) {
        const int sshift = g_quad4_stagger_shift | ((g_quad4_stagger_groups - 1) << 8);
        // generic launch of one instance (NPT, EPT, ABL, CAPO, HASB, CAPN, V2, PHYS) with the plan's runtime LDS strides
#define HFEM_LAUNCH_Q4V(V, ...)                                                                                              \
    hipLaunchKernelGGL((quad4_energy_fast_kernel<256, __VA_ARGS__>), dim3(n), dim3(256), (size_t)plan->lds_bytes, s,          \
                       plan_dev(plan), (int)tile_begin, (const V *)x_free, (const V *)x_fixed, (const V *)u_free,            \
                       (const V *)u_fixed, k, (const double4 *)T_edge, tc, plan->d_partials + tile_begin, (V *)gx, (V *)gu,   \
                       h.max_nodes, h.max_owned, skip_k, stagger, sshift, plan->d_stamps, body)
#define HFEM_LAUNCH_Q4(NPT, EPT, ...) HFEM_LAUNCH_Q4V(double2, NPT, EPT, __VA_ARGS__)
#ifdef HFEM_LAB
        const int abl = g_quad4_ablate;
        const int stagger = g_quad4_stagger >= 0 ? g_quad4_stagger : (n >= 1536 ? 200 : 0);
        const int skip_k = skip_edges | (g_quad4_bits << 8);      // the tiled kernel's lab bits ride above the flag
        if (dtype == 0 && !phys && g_quad4_pipe > 0 && abl == 0 && h.max_nodes <= 3 * 256 && h.max_elems <= 4 * 256 &&
            plan->lds_bytes_pipe <= 64 * 1024) {
            int G = g_quad4_pipe * 256;
            if (G > n) G = n;
            if ((n + G - 1) / G > kPipeMaxTiles) G = (n + kPipeMaxTiles - 1) / kPipeMaxTiles;
            const int cap_elems = (h.max_elems + 3) & ~3;
            hipLaunchKernelGGL((quad4_energy_pipe_kernel<256, 3, 4>), dim3(G), dim3(256), (size_t)plan->lds_bytes_pipe, s,
                               plan_dev(plan), (int)tile_begin, n, (const double2 *)x_free, (const double2 *)x_fixed,
                               (const double2 *)u_free, (const double2 *)u_fixed, k, (const double4 *)T_edge, tc,
                               plan->d_partials + tile_begin, (double2 *)gx, (double2 *)gu, h.max_nodes, h.max_owned, cap_elems,
                               skip_edges);
            if (int rc = launch_status("hfem_quad4_energy_plan(pipe)")) return rc;
            if (flags & HFEM_FLAG_NO_LOSS_SUM) return 0;
            hipLaunchKernelGGL(quad4_sum_partials_kernel, dim3(1), dim3(256), 0, s, plan->d_partials + tile_begin, G, loss_out);
            return launch_status("hfem_quad4_energy_plan(sum)");
        }
        if (dtype == 0 && !phys && !hasb && (g_quad4_bits & 4) && h.max_nodes <= 3 * 128 && h.max_elems <= 3 * 128) {
            // lab: 128-thread workgroups on half-size tiles (8 per CU by LDS and registers alike)
            hipLaunchKernelGGL((quad4_energy_fast_kernel<128, 3, 3, 0, 0, false, 0, double2, false, 2>), dim3(n), dim3(128),
                               (size_t)plan->lds_bytes, s, plan_dev(plan), (int)tile_begin, (const double2 *)x_free,
                               (const double2 *)x_fixed, (const double2 *)u_free, (const double2 *)u_fixed, k, (const double4 *)T_edge, tc,
                               plan->d_partials + tile_begin, (double2 *)gx, (double2 *)gu, h.max_nodes, h.max_owned, skip_k, stagger,
                               sshift, plan->d_stamps, body);
        } else
        if (dtype == 0 && !phys && abl == 1) HFEM_LAUNCH_Q4(4, 4, 1);        // lab instances (hfem_set_option("quad4_ablate"))
        else if (dtype == 0 && !phys && abl == 2) HFEM_LAUNCH_Q4(4, 4, 2);
        else if (dtype == 0 && !phys && abl == 3) HFEM_LAUNCH_Q4(4, 4, 3);
        else if (dtype == 0 && !phys && abl == 4) HFEM_LAUNCH_Q4(4, 4, 4);   // s_memrealtime phase stamps (scripts/stamps.py)
        else
#else
        const int stagger = 0;
        const int skip_k = skip_edges;
#endif
        if (dtype == 1) {                                                // fp32 rows: general instances (runtime strides)
            if (phys && hasb) HFEM_LAUNCH_Q4V(float2, 4, 4, 0, 0, true, 0, float2, true);
            else if (phys) HFEM_LAUNCH_Q4V(float2, 4, 4, 0, 0, false, 0, float2, true);
            else if (hasb) HFEM_LAUNCH_Q4V(float2, 4, 4, 0, 0, true, 0, float2, false);
            else HFEM_LAUNCH_Q4V(float2, 4, 4, 0, 0, false, 0, float2, false);
        } else if (phys) {                                               // opt-in physical convention: general instances
            if (hasb) HFEM_LAUNCH_Q4V(double2, 4, 4, 0, 0, true, 0, double2, true);
            else HFEM_LAUNCH_Q4V(double2, 4, 4, 0, 0, false, 0, double2, true);
        } else if (hasb) HFEM_LAUNCH_Q4(4, 4, 0, 0, true);               // body force: general instance (runtime strides)
        else if (plan->tune.store_policy == 2 && g_quad4_const_caps && h.max_nodes <= 672 && h.max_owned <= 560 && h.max_elems <= 3 * 256) {
            // big meshes: nt gradient stores (the plan's store policy), default tile shape
            hipLaunchKernelGGL((quad4_energy_fast_kernel<256, 3, 3, 0, 560, false, 672, double2, false, 2>), dim3(n), dim3(256), (size_t)39552, s,
                               plan_dev(plan), (int)tile_begin, (const double2 *)x_free, (const double2 *)x_fixed,
                               (const double2 *)u_free, (const double2 *)u_fixed, k, (const double4 *)T_edge, tc,
                               plan->d_partials + tile_begin, (double2 *)gx, (double2 *)gu, 672, 560, skip_k, stagger, sshift,
                               plan->d_stamps, body);
        } else if (plan->tune.store_policy == 2) HFEM_LAUNCH_Q4(4, 4, 0, 0, false, 0, double2, false, 2);
        else if (g_quad4_const_caps && h.max_nodes <= 672 && h.max_owned <= 560 && h.max_elems <= 3 * 256) {
            // default tile shape (557 owned nodes): compile-time LDS strides, (672 + 560) * 32 + 128 = 39552 B: 4 workgroups per CU
            hipLaunchKernelGGL((quad4_energy_fast_kernel<256, 3, 3, 0, 560, false, 672>), dim3(n), dim3(256), (size_t)39552, s,
                               plan_dev(plan), (int)tile_begin, (const double2 *)x_free, (const double2 *)x_fixed,
                               (const double2 *)u_free, (const double2 *)u_fixed, k, (const double4 *)T_edge, tc,
                               plan->d_partials + tile_begin, (double2 *)gx, (double2 *)gu, 672, 560, skip_k, stagger, sshift,
                               plan->d_stamps, body);
        } else if (h.max_nodes <= 3 * 256 && h.max_elems <= 3 * 256) HFEM_LAUNCH_Q4(3, 3, 0);
        else if (g_quad4_const_caps && h.max_owned <= 560 && h.max_nodes * 32 + 560 * 32 + 128 <= 40960) {
            // default tile shape: compile-time accumulator stride (launched with the matching LDS size)
            hipLaunchKernelGGL((quad4_energy_fast_kernel<256, 4, 4, 0, 560>), dim3(n), dim3(256),
                               (size_t)(h.max_nodes * 32 + 560 * 32 + 128), s, plan_dev(plan), (int)tile_begin,
                               (const double2 *)x_free, (const double2 *)x_fixed, (const double2 *)u_free,
                               (const double2 *)u_fixed, k, (const double4 *)T_edge, tc, plan->d_partials + tile_begin,
                               (double2 *)gx, (double2 *)gu, h.max_nodes, 560, skip_k, stagger, sshift, plan->d_stamps, body);
        } else HFEM_LAUNCH_Q4(4, 4, 0);
#undef HFEM_LAUNCH_Q4
#undef HFEM_LAUNCH_Q4V
        if (int rc = launch_status("hfem_quad4_energy_plan")) return rc;
    }
    if (flags & HFEM_FLAG_NO_LOSS_SUM) return 0;
    hipLaunchKernelGGL(quad4_sum_partials_kernel, dim3(1), dim3(256), 0, s, plan->d_partials + tile_begin, n, loss_out);
    return launch_status("hfem_quad4_energy_plan(sum)");
}

/* TEST INFRASTRUCTURE ONLY -- plain-C closed forms of the HiDeNN-FEM hot path.
 *
 * Scalar, single-threaded restatement (forward + hand-derived backward) of what
 * the reference computes with ATen op chains + autograd:
 *   TRI3 element energy   /root/reference/src/models.py:316-357 + src/loss.py:55-88
 *   EDGE2 Neumann work    /root/reference/src/models.py:359-376 + src/loss.py:91-110
 *   LINE2 interpolation   /root/reference/src/models.py:70-90
 *   grid parametrisation  /root/reference/src/models.py:45-56,146-168
 *   bar energy            /root/reference/examples/example3.py:27-70
 *   RECT-Q4 interpolation /root/reference/src/models.py:180-212
 * Formulas: SURVEY.md section 8a.  Pinned against tests/golden/ (generated from
 * the imported reference) by tests/test_oracle_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load
 * this library; the product never does.
 *
 * Build: python oracle/build.py   (gcc -O2 -fPIC -shared -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* ---------------------------------------------------------------- TRI3 ------
 * X,U: full node arrays [Nn][2].  conn: [Ne][3] int64 (reference layout,
 * models.py:252).  mat = {c11, c12, c22, c33} of the plane-stress matrix
 * (loss.py:29-32).  W = sum_q w_q; Bk[k][i] = sum_q w_q N_k(xi_q) b_i(xi_q)
 * (loss.py:80-81 with the reference-coordinate argument).
 * Accumulates into gX,gU [Nn][2] (caller zeroes) and returns the domain energy.
 * grad_u = G * Jinv^T is the reference's Jinv*dN_dxi convention (models.py:351,
 * SURVEY F4); abs(detJ) as loss.py:84.                                         */
double oracle_tri3_energy(const double *X, const double *U, const int64_t *conn,
                          int64_t ne, const double *mat, double W, const double *Bk,
                          double *gX, double *gU)
{
    const double c11 = mat[0], c12 = mat[1], c22 = mat[2], c33 = mat[3];
    double total = 0.0;
    for (int64_t e = 0; e < ne; ++e) {
        const int64_t n0 = conn[3 * e], n1 = conn[3 * e + 1], n2 = conn[3 * e + 2];
        const double x0 = X[2 * n0], y0 = X[2 * n0 + 1];
        const double x1 = X[2 * n1], y1 = X[2 * n1 + 1];
        const double x2 = X[2 * n2], y2 = X[2 * n2 + 1];
        const double a = x0 - x2, b = x1 - x2, c = y0 - y2, d = y1 - y2; /* J=[[a,b],[c,d]] models.py:339 */
        const double det = a * d - b * c;
        const double s = det < 0.0 ? -1.0 : 1.0, A = fabs(det);
        const double ux0 = U[2 * n0], uy0 = U[2 * n0 + 1];
        const double ux1 = U[2 * n1], uy1 = U[2 * n1 + 1];
        const double ux2 = U[2 * n2], uy2 = U[2 * n2 + 1];
        const double g0x = ux0 - ux2, g0y = uy0 - uy2;   /* G0 = U0-U2 */
        const double g1x = ux1 - ux2, g1y = uy1 - uy2;   /* G1 = U1-U2 */
        /* H[i][0] = (G0[i] d - G1[i] b)/det ; H[i][1] = (-G0[i] c + G1[i] a)/det */
        const double h00 = (g0x * d - g1x * b) / det, h01 = (-g0x * c + g1x * a) / det;
        const double h10 = (g0y * d - g1y * b) / det, h11 = (-g0y * c + g1y * a) / det;
        const double exx = h00, eyy = h11, gam = h01 + h10;               /* loss.py:70-73 */
        const double sxx = c11 * exx + c12 * eyy, syy = c12 * exx + c22 * eyy, sxy = c33 * gam;
        const double psi = 0.5 * (exx * sxx + eyy * syy + gam * sxy);     /* loss.py:77 */
        const double beta = ux0 * Bk[0] + uy0 * Bk[1] + ux1 * Bk[2] + uy1 * Bk[3]
                          + ux2 * Bk[4] + uy2 * Bk[5];
        const double dens = W * psi - beta;
        total += A * dens;                                                /* loss.py:84-88 */
        if (!gX) continue;
        /* P[i][j] = dL/dH[i][j] */
        const double aw = A * W;
        const double p00 = aw * sxx, p01 = aw * sxy, p10 = aw * sxy, p11 = aw * syy;
        /* dG0[i] = (P[i][0] d - P[i][1] c)/det ; dG1[i] = (-P[i][0] b + P[i][1] a)/det */
        const double dg0x = (p00 * d - p01 * c) / det, dg0y = (p10 * d - p11 * c) / det;
        const double dg1x = (-p00 * b + p01 * a) / det, dg1y = (-p10 * b + p11 * a) / det;
        gU[2 * n0]     += dg0x - A * Bk[0];  gU[2 * n0 + 1] += dg0y - A * Bk[1];
        gU[2 * n1]     += dg1x - A * Bk[2];  gU[2 * n1 + 1] += dg1y - A * Bk[3];
        gU[2 * n2]     += -(dg0x + dg1x) - A * Bk[4];
        gU[2 * n2 + 1] += -(dg0y + dg1y) - A * Bk[5];
        /* dJ entries with H's explicit dependence, then through det */
        double da = (p01 * g1x + p11 * g1y) / det;
        double db = -(p00 * g1x + p10 * g1y) / det;
        double dc = -(p01 * g0x + p11 * g0y) / det;
        double dd = (p00 * g0x + p10 * g0y) / det;
        const double ddet = -(p00 * h00 + p01 * h01 + p10 * h10 + p11 * h11) / det + s * dens;
        da += ddet * d;  dd += ddet * a;  db -= ddet * c;  dc -= ddet * b;
        gX[2 * n0] += da;  gX[2 * n0 + 1] += dc;        /* g0 = (da, dc) */
        gX[2 * n1] += db;  gX[2 * n1 + 1] += dd;        /* g1 = (db, dd) */
        gX[2 * n2] += -(da + db);  gX[2 * n2 + 1] += -(dc + dd);
    }
    return total;
}

/* ---------------------------------------------------------------- QUAD4 -----
 * PARITY UNPINNED BY THE REFERENCE: achraf-15/HiDeNN-FEM has no isoparametric quadrilateral.  This is
 * the extension element as SURVEY section 8a specifies it, in the plain node-by-node D_N form (the kernels use a
 * bilinear-coefficient form: an independent derivation), with the reference's triangle conventions:
 * J[i][j] = sum_k x_k[i] D_N[j][k] (models.py:339), dN_dx = Jinv * D_N (models.py:351, F4), abs(detJ) (loss.py:84).
 * conn4 [Ne][4] int64, local nodes CCW from (-1,-1); 2x2 Gauss (+-1/sqrt(3), weights 1) in the order
 * (-,-) (+,-) (-,+) (+,+); Bq [4][2] = body force at those REFERENCE points (loss.py:80 passes reference
 * coordinates, F6) or NULL.  Accumulates gX,gU [Nn][2] (caller zeroes; may be NULL) and returns
 * sum_e sum_q |detJ_q| (psi_q - u_h(q).b_q).                                                                  */
double oracle_quad4_energy(const double *X, const double *U, const int64_t *conn4, int64_t ne, const double *mat,
                           const double *Bq, double *gX, double *gU)
{
    const double c11 = mat[0], c12 = mat[1], c22 = mat[2], c33 = mat[3];
    const double gp = 0.57735026918962576451;
    static const double XI[4] = {-1.0, 1.0, 1.0, -1.0}, ETA[4] = {-1.0, -1.0, 1.0, 1.0};
    double total = 0.0;
    for (int64_t e = 0; e < ne; ++e) {
        int64_t n[4];
        double x[4], y[4], ux[4], uy[4], gxx[4] = {0, 0, 0, 0}, gxy[4] = {0, 0, 0, 0}, gux[4] = {0, 0, 0, 0}, guy[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; ++k) {
            n[k] = conn4[4 * e + k];
            x[k] = X[2 * n[k]]; y[k] = X[2 * n[k] + 1]; ux[k] = U[2 * n[k]]; uy[k] = U[2 * n[k] + 1];
        }
        for (int q = 0; q < 4; ++q) {
            const double xi = (q & 1) ? gp : -gp, eta = (q & 2) ? gp : -gp;
            double N[4], D0[4], D1[4];
            double a = 0, b = 0, c = 0, d = 0, g0x = 0, g0y = 0, g1x = 0, g1y = 0, uhx = 0, uhy = 0;
            for (int k = 0; k < 4; ++k) {
                N[k] = 0.25 * (1.0 + XI[k] * xi) * (1.0 + ETA[k] * eta);
                D0[k] = 0.25 * XI[k] * (1.0 + ETA[k] * eta);
                D1[k] = 0.25 * ETA[k] * (1.0 + XI[k] * xi);
                a += x[k] * D0[k]; b += x[k] * D1[k]; c += y[k] * D0[k]; d += y[k] * D1[k];
                g0x += ux[k] * D0[k]; g0y += uy[k] * D0[k]; g1x += ux[k] * D1[k]; g1y += uy[k] * D1[k];
                uhx += N[k] * ux[k]; uhy += N[k] * uy[k];
            }
            const double det = a * d - b * c, sg = det < 0.0 ? -1.0 : 1.0, A = fabs(det);
            const double h00 = (g0x * d - g1x * b) / det, h01 = (-g0x * c + g1x * a) / det;
            const double h10 = (g0y * d - g1y * b) / det, h11 = (-g0y * c + g1y * a) / det;
            const double gam = h01 + h10;
            const double sxx = c11 * h00 + c12 * h11, syy = c12 * h00 + c22 * h11, sxy = c33 * gam;
            const double psi = 0.5 * (h00 * sxx + h11 * syy + gam * sxy);
            const double bx = Bq ? Bq[2 * q] : 0.0, by = Bq ? Bq[2 * q + 1] : 0.0;
            const double dens = psi - (uhx * bx + uhy * by);
            total += A * dens;
            if (!gX) continue;
            const double p00 = A * sxx, p01 = A * sxy, p10 = A * sxy, p11 = A * syy;
            const double dg0x = (p00 * d - p01 * c) / det, dg0y = (p10 * d - p11 * c) / det;
            const double dg1x = (-p00 * b + p01 * a) / det, dg1y = (-p10 * b + p11 * a) / det;
            double da = (p01 * g1x + p11 * g1y) / det, db = -(p00 * g1x + p10 * g1y) / det;
            double dc = -(p01 * g0x + p11 * g0y) / det, dd = (p00 * g0x + p10 * g0y) / det;
            const double ddet = -(p00 * h00 + p01 * h01 + p10 * h10 + p11 * h11) / det + sg * dens;
            da += ddet * d; dd += ddet * a; db -= ddet * c; dc -= ddet * b;
            for (int k = 0; k < 4; ++k) {
                gxx[k] += da * D0[k] + db * D1[k];
                gxy[k] += dc * D0[k] + dd * D1[k];
                gux[k] += dg0x * D0[k] + dg1x * D1[k] - A * N[k] * bx;
                guy[k] += dg0y * D0[k] + dg1y * D1[k] - A * N[k] * by;
            }
        }
        if (gX)
            for (int k = 0; k < 4; ++k) {
                gX[2 * n[k]] += gxx[k]; gX[2 * n[k] + 1] += gxy[k];
                gU[2 * n[k]] += gux[k]; gU[2 * n[k] + 1] += guy[k];
            }
    }
    return total;
}

/* ---------------------------------------------------------------- EDGE2 -----
 * edges [Ned][2] int64, index-sorted (i<j, mesh.py:130,255).  T [Ned][4] =
 * {Ti.x, Ti.y, Tj.x, Tj.y} with Ti = sum_q w_q (1-xi_q) t(x_q), Tj = sum_q w_q
 * xi_q t(x_q) (raw Legendre xi); if T is NULL the constant table
 * Tconst[4] is used for every edge.  Returns the edge work (to be SUBTRACTED,
 * loss.py:116) and accumulates the gradient of (-work) into gX,gU.            */
double oracle_edge2_energy(const double *X, const double *U, const int64_t *edges,
                           int64_t ned, const double *T, const double *Tconst,
                           double *gX, double *gU)
{
    double total = 0.0;
    for (int64_t e = 0; e < ned; ++e) {
        const int64_t i = edges[2 * e], j = edges[2 * e + 1];
        const double *t = T ? T + 4 * e : Tconst;
        const double rx = X[2 * j] - X[2 * i], ry = X[2 * j + 1] - X[2 * i + 1];
        const double ds = sqrt(rx * rx + ry * ry);                         /* models.py:375 */
        const double m = U[2 * i] * t[0] + U[2 * i + 1] * t[1] + U[2 * j] * t[2] + U[2 * j + 1] * t[3];
        total += ds * m;
        if (!gX) continue;
        gU[2 * i] -= ds * t[0];  gU[2 * i + 1] -= ds * t[1];
        gU[2 * j] -= ds * t[2];  gU[2 * j + 1] -= ds * t[3];
        const double fx = m * rx / ds, fy = m * ry / ds;
        gX[2 * j] -= fx;  gX[2 * j + 1] -= fy;
        gX[2 * i] += fx;  gX[2 * i + 1] += fy;
    }
    return total;
}

/* ------------------------------------------------- TRI3 per-point forward ---
 * models.py:316-357: u_h[M][2], detJ[M], grad_u[M][2][2] at (xi,eta)[M][2].    */
void oracle_tri3_eval(const double *X, const double *U, const int64_t *conn,
                      const double *x_eval, const int64_t *elem_id, int64_t m,
                      double *u_h, double *detJ, double *grad_u)
{
    for (int64_t p = 0; p < m; ++p) {
        const int64_t e = elem_id[p];
        const int64_t n0 = conn[3 * e], n1 = conn[3 * e + 1], n2 = conn[3 * e + 2];
        const double xi = x_eval[2 * p], eta = x_eval[2 * p + 1], zeta = 1.0 - xi - eta;
        u_h[2 * p]     = xi * U[2 * n0] + eta * U[2 * n1] + zeta * U[2 * n2];
        u_h[2 * p + 1] = xi * U[2 * n0 + 1] + eta * U[2 * n1 + 1] + zeta * U[2 * n2 + 1];
        const double a = X[2 * n0] - X[2 * n2], b = X[2 * n1] - X[2 * n2];
        const double c = X[2 * n0 + 1] - X[2 * n2 + 1], d = X[2 * n1 + 1] - X[2 * n2 + 1];
        const double det = a * d - b * c;
        detJ[p] = det;
        for (int i = 0; i < 2; ++i) {
            const double g0 = U[2 * n0 + i] - U[2 * n2 + i], g1 = U[2 * n1 + i] - U[2 * n2 + i];
            grad_u[4 * p + 2 * i]     = (g0 * d - g1 * b) / det;
            grad_u[4 * p + 2 * i + 1] = (-g0 * c + g1 * a) / det;
        }
    }
}

/* ---------------------------------------------------------------- 1D --------
 * grid parametrisation (models.py:45-56): p[n] -> grid[n+1] = {x0, x0+(xN-x0) cum_k/S}. */
static double softplus(double v) { return v > 20.0 ? v : log1p(exp(v)); }   /* torch threshold=20 */

void oracle_grid_param_fwd(const double *p, int64_t n, double x0, double xN, double *grid)
{
    double cum = 0.0, S = 0.0;
    for (int64_t k = 0; k < n; ++k) { double s = softplus(p[k]); S += s < 1e-6 ? 1e-6 : s; }
    grid[0] = x0;
    for (int64_t k = 0; k < n; ++k) {
        double s = softplus(p[k]);
        cum += s < 1e-6 ? 1e-6 : s;
        grid[k + 1] = x0 + (xN - x0) * cum / S;
    }
}

/* backward of the above: ggrid[n+1] -> gp[n] (ggrid[0] hits the constant x0). */
void oracle_grid_param_bwd(const double *p, int64_t n, double x0, double xN,
                           const double *ggrid, double *gp)
{
    double *cum = (double *)malloc(sizeof(double) * (size_t)n);
    double acc = 0.0;
    for (int64_t k = 0; k < n; ++k) { double s = softplus(p[k]); acc += s < 1e-6 ? 1e-6 : s; cum[k] = acc; }
    const double S = acc, L = xN - x0;
    double dot = 0.0;
    for (int64_t k = 0; k < n; ++k) dot += ggrid[k + 1] * cum[k];
    double run = 0.0;                                   /* reverse cumsum of gcum */
    for (int64_t k = n - 1; k >= 0; --k) {
        double gcum = L * ggrid[k + 1] / S;
        if (k == n - 1) gcum -= L * dot / (S * S);
        run += gcum;
        const double s = softplus(p[k]);
        const double sig = 1.0 / (1.0 + exp(-p[k]));
        gp[k] = s > 1e-6 ? run * (p[k] > 20.0 ? 1.0 : sig) : 0.0;
    }
    free(cum);
}

/* searchsorted(right=False)-1, clamp(0, n-2): models.py:73-74 */
static int64_t find_elem(const double *grid, int64_t n, double x)
{
    int64_t lo = 0, hi = n;                 /* first index with grid[idx] >= x */
    while (lo < hi) { int64_t mid = (lo + hi) / 2; if (grid[mid] < x) lo = mid + 1; else hi = mid; }
    int64_t e = lo - 1;
    if (e < 0) e = 0;
    if (e > n - 2) e = n - 2;
    return e;
}

/* LINE2 forward (models.py:70-90) + backward with upstream cot[m].
 * Outputs pred[m]; accumulates ggrid[n], gu[n]; writes gx_eval[m].  Any of the
 * gradient pointers may be NULL.                                               */
void oracle_line2(const double *grid, const double *u, int64_t n, const double *x_eval,
                  int64_t m, double *pred, const double *cot, double *ggrid, double *gu,
                  double *gx_eval)
{
    for (int64_t q = 0; q < m; ++q) {
        const double x = x_eval[q];
        const int64_t e = find_elem(grid, n, x);
        const double xi = grid[e], xj = grid[e + 1], ui = u[e], uj = u[e + 1];
        const double raw = xj - xi, h = raw < 1e-10 ? 1e-10 : raw;         /* clamp(eps) */
        const double N1 = (xj - x) / h, N2 = (x - xi) / h;
        if (pred) pred[q] = ui * N1 + uj * N2;
        if (!cot) continue;
        const double g = cot[q], gN1 = g * ui, gN2 = g * uj;
        if (gu) { gu[e] += g * N1; gu[e + 1] += g * N2; }
        if (ggrid) {
            const double gh = raw < 1e-10 ? 0.0 : -(gN1 * N1 + gN2 * N2) / h;
            ggrid[e + 1] += gN1 / h + gh;
            ggrid[e]     += -gN2 / h - gh;
        }
        if (gx_eval) gx_eval[q] = (gN2 - gN1) / h;
    }
}

/* Bar energy of examples/example3.py:27-70 with detached quadrature:
 * xq[ne][ng], wq[ne][ng], bq[ne][ng] = b(xq) are CONSTANTS; grid[n], u[n] full
 * arrays.  The point->element lookup is still the searchsorted of models.py:73.
 * Returns the loss; accumulates ggrid[n], gu[n].                               */
double oracle_bar_energy(const double *grid, const double *u, int64_t n, const double *xq,
                         const double *wq, const double *bq, int64_t npts, double E,
                         double *ggrid, double *gu)
{
    double total = 0.0;
    for (int64_t q = 0; q < npts; ++q) {
        const double x = xq[q];
        const int64_t e = find_elem(grid, n, x);
        const double xi = grid[e], xj = grid[e + 1], ui = u[e], uj = u[e + 1];
        const double raw = xj - xi, h = raw < 1e-10 ? 1e-10 : raw;
        const double N1 = (xj - x) / h, N2 = (x - xi) / h;
        const double uu = ui * N1 + uj * N2, du = (uj - ui) / h;           /* example3.py:52-56 */
        total += wq[q] * (0.5 * E * du * du - bq[q] * uu);                 /* example3.py:59-68 */
        if (!gu) continue;
        const double gdu = wq[q] * E * du, guq = -wq[q] * bq[q];
        gu[e]     += -gdu / h + guq * N1;
        gu[e + 1] +=  gdu / h + guq * N2;
        if (ggrid) {
            const double gN1 = guq * ui, gN2 = guq * uj;
            const double gh = raw < 1e-10 ? 0.0 : (-gdu * du - (gN1 * N1 + gN2 * N2)) / h;
            ggrid[e + 1] += gN1 / h + gh;
            ggrid[e]     += -gN2 / h - gh;
        }
    }
    return total;
}

/* RECT-Q4 forward/backward (models.py:180-212): gx[nx], gy[ny], u[nx][ny]. */
void oracle_rectq4(const double *gx, int64_t nx, const double *gy, int64_t ny, const double *u,
                   const double *x_eval, int64_t m, double *pred, const double *cot,
                   double *ggx, double *ggy, double *gu, double *gx_eval)
{
    for (int64_t q = 0; q < m; ++q) {
        const double px = x_eval[2 * q], py = x_eval[2 * q + 1];
        const int64_t ix = find_elem(gx, nx, px), iy = find_elem(gy, ny, py);
        const double xi = gx[ix], xj = gx[ix + 1], yi = gy[iy], yj = gy[iy + 1];
        const double rawx = xj - xi, hx = rawx < 1e-10 ? 1e-10 : rawx;
        const double rawy = yj - yi, hy = rawy < 1e-10 ? 1e-10 : rawy;
        const double N1x = (xj - px) / hx, N2x = (px - xi) / hx;
        const double N1y = (yj - py) / hy, N2y = (py - yi) / hy;
        const double u00 = u[ix * ny + iy], u10 = u[(ix + 1) * ny + iy];
        const double u01 = u[ix * ny + iy + 1], u11 = u[(ix + 1) * ny + iy + 1];
        if (pred) pred[q] = N1x * N1y * u00 + N2x * N1y * u10 + N1x * N2y * u01 + N2x * N2y * u11;
        if (!cot) continue;
        const double g = cot[q];
        if (gu) {
            gu[ix * ny + iy] += g * N1x * N1y;        gu[(ix + 1) * ny + iy] += g * N2x * N1y;
            gu[ix * ny + iy + 1] += g * N1x * N2y;    gu[(ix + 1) * ny + iy + 1] += g * N2x * N2y;
        }
        const double gN1x = g * (N1y * u00 + N2y * u01), gN2x = g * (N1y * u10 + N2y * u11);
        const double gN1y = g * (N1x * u00 + N2x * u10), gN2y = g * (N1x * u01 + N2x * u11);
        if (ggx) {
            const double gh = rawx < 1e-10 ? 0.0 : -(gN1x * N1x + gN2x * N2x) / hx;
            ggx[ix + 1] += gN1x / hx + gh;  ggx[ix] += -gN2x / hx - gh;
        }
        if (ggy) {
            const double gh = rawy < 1e-10 ? 0.0 : -(gN1y * N1y + gN2y * N2y) / hy;
            ggy[iy + 1] += gN1y / hy + gh;  ggy[iy] += -gN2y / hy - gh;
        }
        if (gx_eval) {
            gx_eval[2 * q]     = (gN2x - gN1x) / hx;
            gx_eval[2 * q + 1] = (gN2y - gN1y) / hy;
        }
    }
}

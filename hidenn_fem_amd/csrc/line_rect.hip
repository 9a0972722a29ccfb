// 1D (LINE2) and structured-2D (RECT-Q4) paths of the reference, gfx950:
//   grid parametrisation   /root/reference/src/models.py:45-56, 146-168
//   LINE2 interpolation    /root/reference/src/models.py:70-90
//   RECT-Q4 interpolation  /root/reference/src/models.py:180-212
//   inline losses          /root/reference/examples/example1.py:38, example2.py:46,
//                          example3.py:27-70
// Forward + hand-derived backward.  These problems are tiny
// (10^2..10^5 points): the kernels are launch-bound, so each loss is ONE launch
// that produces the scalar and all parameter gradients.
#include <hip/hip_runtime.h>

#include "hfem_device.h"

namespace hfem {

constexpr int kBlockL = 256;
constexpr int kScan = 1024;   // single-block scan width for the grid parametrisation

__device__ __forceinline__ double softplus_clamped(double v) {
    const double s = v > 20.0 ? v : log1p(exp(v));   // torch softplus, beta=1, threshold=20
    return s < 1e-6 ? 1e-6 : s;                      // clamp(min=1e-6)
}
__device__ __forceinline__ double softplus_clamped_grad(double v) {
    const double s = v > 20.0 ? v : log1p(exp(v));
    if (s < 1e-6) return 0.0;
    return v > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-v));
}

// inclusive Hillis-Steele scan over kScan LDS entries; forward (dir=+1) or suffix (dir=-1)
__device__ __forceinline__ double block_scan(double v, double *buf, bool suffix) {
    const int tid = threadIdx.x;
    buf[tid] = v;
    __syncthreads();
    for (int off = 1; off < kScan; off <<= 1) {
        const int o = suffix ? tid + off : tid - off;
        const double add = (o >= 0 && o < kScan) ? buf[o] : 0.0;
        __syncthreads();
        buf[tid] += add;
        __syncthreads();
    }
    return buf[tid];
}

// p[n] -> grid[n+1]; one workgroup of kScan threads, each owning a contiguous chunk.
template <typename T>
__global__ __launch_bounds__(kScan) void grid_param_fwd_kernel(const T *__restrict__ p, int64_t n,
                                                              double x0, double xN,
                                                              const uint8_t *__restrict__ mask,
                                                              const T *__restrict__ initial,
                                                              T *__restrict__ grid) {
    __shared__ double buf[kScan];
    const int tid = threadIdx.x;
    const int64_t chunk = (n + kScan - 1) / kScan;
    const int64_t k0 = tid * chunk, k1 = (k0 + chunk < n) ? k0 + chunk : n;
    double s = 0.0;
    for (int64_t k = k0; k < k1; ++k) s += softplus_clamped((double)p[k]);
    const double incl = block_scan(s, buf, false);
    const double S = buf[kScan - 1];
    double run = incl - s;
    for (int64_t k = k0; k < k1; ++k) {
        run += softplus_clamped((double)p[k]);
        const double cum = (k == n - 1) ? S : run;              // cum[-1]/cum[-1] == 1 exactly
        double g = x0 + (xN - x0) * cum / S;                    // models.py:52
        if (mask && mask[k + 1]) g = (double)initial[k + 1];    // models.py:165-166
        grid[k + 1] = (T)g;
    }
    if (tid == 0) grid[0] = (mask && mask[0]) ? initial[0] : (T)x0;
}

// ggrid[n+1] -> gp[n]
template <typename T>
__global__ __launch_bounds__(kScan) void grid_param_bwd_kernel(const T *__restrict__ p, int64_t n,
                                                              double x0, double xN,
                                                              const uint8_t *__restrict__ mask,
                                                              const T *__restrict__ ggrid,
                                                              T *__restrict__ gp) {
    __shared__ double buf[kScan];
    __shared__ double red[kScan / 64];
    __shared__ double dot_s;
    const int tid = threadIdx.x;
    const int64_t chunk = (n + kScan - 1) / kScan;
    const int64_t k0 = tid * chunk, k1 = (k0 + chunk < n) ? k0 + chunk : n;
    const double L = xN - x0;
    double s = 0.0;
    for (int64_t k = k0; k < k1; ++k) s += softplus_clamped((double)p[k]);
    const double incl = block_scan(s, buf, false);
    const double S = buf[kScan - 1];
    __syncthreads();
    // dot = sum_k gx_k cum_k (masked rows carry no gradient)
    double run = incl - s, dot = 0.0;
    for (int64_t k = k0; k < k1; ++k) {
        run += softplus_clamped((double)p[k]);
        const double g = (mask && mask[k + 1]) ? 0.0 : (double)ggrid[k + 1];
        dot += g * ((k == n - 1) ? S : run);
    }
    const double dtot = block_sum(dot, red);
    if (tid == 0) dot_s = dtot;
    __syncthreads();
    const double corr = L * dot_s / (S * S);
    // gcum_k = L g_k / S  (last: minus corr); ginc = suffix sum of gcum
    double loc = 0.0;
    for (int64_t k = k0; k < k1; ++k) {
        const double g = (mask && mask[k + 1]) ? 0.0 : (double)ggrid[k + 1];
        loc += L * g / S - (k == n - 1 ? corr : 0.0);
    }
    const double suf = block_scan(loc, buf, true);   // sum over chunks >= tid
    double acc = suf - loc;                           // chunks strictly after this one
    for (int64_t k = k1 - 1; k >= k0; --k) {
        const double g = (mask && mask[k + 1]) ? 0.0 : (double)ggrid[k + 1];
        acc += L * g / S - (k == n - 1 ? corr : 0.0);
        gp[k] = (T)(acc * softplus_clamped_grad((double)p[k]));
    }
}

// ---- multi-workgroup form for long grids (the single-workgroup kernels above put every softplus -- an fp64
// exp + log1p -- on ONE CU: 31 us forward / 53 us backward at n = 10^4, most of an example-3 iteration).
// Three small launches each way: per-block local scan (+ block sums), one-block scan of the block sums,
// finalize.  Blocks own kGpChunk = 1024 consecutive increments (256 threads x 4, coalesced).
constexpr int kGpThreads = 256, kGpPer = 4, kGpChunk = kGpThreads * kGpPer;

// inclusive scan (prefix or suffix) of one value per thread over the block's kGpThreads threads
__device__ __forceinline__ double gp_block_scan(double v, double *buf, bool suffix) {
    const int tid = threadIdx.x;
    buf[tid] = v;
    __syncthreads();
    for (int off = 1; off < kGpThreads; off <<= 1) {
        const int o = suffix ? tid + off : tid - off;
        const double add = (o >= 0 && o < kGpThreads) ? buf[o] : 0.0;
        __syncthreads();
        buf[tid] += add;
        __syncthreads();
    }
    return buf[tid];
}

// F1: cumloc[k] = inclusive prefix of softplus within the block; bsum[b] = block total
template <typename T>
__global__ __launch_bounds__(kGpThreads) void gp_fwd_local_kernel(const T *__restrict__ p, int64_t n,
                                                                  double *__restrict__ cum, double *__restrict__ bsum) {
    __shared__ double buf[kGpThreads];
    const int64_t k0 = (int64_t)blockIdx.x * kGpChunk + (int64_t)threadIdx.x * kGpPer;
    double v[kGpPer], run = 0.0;
#pragma unroll
    for (int j = 0; j < kGpPer; ++j) {
        v[j] = (k0 + j < n) ? softplus_clamped((double)p[k0 + j]) : 0.0;
        run += v[j];
        v[j] = run;
    }
    const double incl = gp_block_scan(run, buf, false);
    const double before = incl - run;
#pragma unroll
    for (int j = 0; j < kGpPer; ++j)
        if (k0 + j < n) cum[k0 + j] = before + v[j];
    if (threadIdx.x == kGpThreads - 1) bsum[blockIdx.x] = incl;
}

// F2 / B2: one block; exclusive scan (prefix or suffix) of the nb block sums in place; totals to scal
__global__ __launch_bounds__(kScan) void gp_block_offsets_kernel(double *__restrict__ bsum, int64_t nb, bool suffix,
                                                                 const double *__restrict__ dsum, double *__restrict__ scal) {
    __shared__ double buf[kScan];
    __shared__ double red[kScan / 64];
    const int tid = threadIdx.x;
    const int64_t chunk = (nb + kScan - 1) / kScan;
    const int64_t b0 = tid * chunk, b1 = (b0 + chunk < nb) ? b0 + chunk : nb;
    double s = 0.0, d = 0.0;
    for (int64_t b = b0; b < b1; ++b) { s += bsum[b]; if (dsum) d += dsum[b]; }
    const double incl = block_scan(s, buf, suffix);
    const double total = suffix ? buf[0] : buf[kScan - 1];
    __syncthreads();
    double run = incl - s;                                  // blocks strictly before (after) this thread's chunk
    if (!suffix) for (int64_t b = b0; b < b1; ++b) { const double t = bsum[b]; bsum[b] = run; run += t; }
    else for (int64_t b = b1 - 1; b >= b0; --b) { const double t = bsum[b]; bsum[b] = run; run += t; }
    const double dtot = block_sum(d, red);
    if (tid == 0) { scal[0] = total; scal[1] = dtot; }
}

// F3: cum += block offset (last entry := S exactly), grid
template <typename T>
__global__ __launch_bounds__(kGpThreads) void gp_fwd_final_kernel(int64_t n, double x0, double xN,
                                                                  const uint8_t *__restrict__ mask,
                                                                  const T *__restrict__ initial,
                                                                  const double *__restrict__ boff,
                                                                  const double *__restrict__ scal, double *__restrict__ cum,
                                                                  T *__restrict__ grid) {
    const double S = scal[0], off = boff[blockIdx.x];
    const int64_t k0 = (int64_t)blockIdx.x * kGpChunk + (int64_t)threadIdx.x * kGpPer;
#pragma unroll
    for (int j = 0; j < kGpPer; ++j) {
        const int64_t k = k0 + j;
        if (k < n) {
            const double c = (k == n - 1) ? S : off + cum[k];       // cum[-1]/cum[-1] == 1 exactly
            cum[k] = c;
            double g = x0 + (xN - x0) * c / S;                      // models.py:52
            if (mask && mask[k + 1]) g = (double)initial[k + 1];    // models.py:165-166
            grid[k + 1] = (T)g;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) grid[0] = (mask && mask[0]) ? initial[0] : (T)x0;
}

// B1: a_k = L g_k / S; sufloc[k] = inclusive suffix of a within the block; asum[b]; dsum[b] = sum g_k cum_k
template <typename T>
__global__ __launch_bounds__(kGpThreads) void gp_bwd_local_kernel(int64_t n, double L, const uint8_t *__restrict__ mask,
                                                                  const T *__restrict__ ggrid,
                                                                  const double *__restrict__ cum, double *__restrict__ suf,
                                                                  double *__restrict__ asum, double *__restrict__ dsum) {
    __shared__ double buf[kGpThreads];
    __shared__ double red[kGpThreads / 64];
    const double S = cum[n - 1];
    const int64_t k0 = (int64_t)blockIdx.x * kGpChunk + (int64_t)threadIdx.x * kGpPer;
    double a[kGpPer], run = 0.0, dot = 0.0;
#pragma unroll
    for (int j = kGpPer - 1; j >= 0; --j) {
        const int64_t k = k0 + j;
        double g = 0.0;
        if (k < n) {
            g = (mask && mask[k + 1]) ? 0.0 : (double)ggrid[k + 1];
            dot += g * cum[k];
        }
        run += L * g / S;
        a[j] = run;                                          // suffix within the thread
    }
    const double incl = gp_block_scan(run, buf, true);
    const double after = incl - run;
#pragma unroll
    for (int j = 0; j < kGpPer; ++j)
        if (k0 + j < n) suf[k0 + j] = after + a[j];
    if (threadIdx.x == 0) asum[blockIdx.x] = incl;
    __syncthreads();
    const double dt = block_sum(dot, red);
    if (threadIdx.x == 0) dsum[blockIdx.x] = dt;
}

// B3: gp[k] = (suffix over later blocks + local suffix - corr) * softplus'(p[k])
template <typename T>
__global__ __launch_bounds__(kGpThreads) void gp_bwd_final_kernel(const T *__restrict__ p, int64_t n, double L,
                                                                  const double *__restrict__ cum,
                                                                  const double *__restrict__ suf,
                                                                  const double *__restrict__ aoff,
                                                                  const double *__restrict__ scal, T *__restrict__ gp) {
    const double S = cum[n - 1], corr = L * scal[1] / (S * S), off = aoff[blockIdx.x];
    const int64_t k0 = (int64_t)blockIdx.x * kGpChunk + (int64_t)threadIdx.x * kGpPer;
#pragma unroll
    for (int j = 0; j < kGpPer; ++j) {
        const int64_t k = k0 + j;
        if (k < n) gp[k] = (T)((off + suf[k] - corr) * softplus_clamped_grad((double)p[k]));
    }
}

// searchsorted(grid, x, right=False) - 1, clamp(0, n-2)      models.py:73-74
template <typename T>
__device__ __forceinline__ int find_elem(const T *__restrict__ grid, int n, double x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((double)grid[mid] < x) lo = mid + 1; else hi = mid;
    }
    int e = lo - 1;
    e = e < 0 ? 0 : e;
    return e > n - 2 ? n - 2 : e;
}

struct Hat {   // one axis of the hat-function pair, models.py:84-85
    int e;
    double N1, N2, h, raw, ui, uj;
};
// T = row type of the caller's arrays (double, or float for the reference's default dtype: widened on load, fp64 arithmetic
// in between, per-point outputs rounded once on store; ACCUMULATED outputs -- grid / nodal-value gradients, the loss --
// are atomics of type T, so a float caller gets one rounding per contribution, in varying order)
template <typename T>
__device__ __forceinline__ Hat hat_eval(const T *__restrict__ grid, int n, double x) {
    Hat t;
    t.e = find_elem(grid, n, x);
    const double xi = (double)grid[t.e], xj = (double)grid[t.e + 1];
    t.raw = xj - xi;
    t.h = t.raw < 1e-10 ? 1e-10 : t.raw;        // .clamp(self.epsilon)
    t.N1 = (xj - x) / t.h;
    t.N2 = (x - xi) / t.h;
    return t;
}
// accumulate d/dgrid of one axis given dL/dN1, dL/dN2
template <typename T>
__device__ __forceinline__ void hat_grid_grad(const Hat &t, double gN1, double gN2, double extra_gh,
                                              T *__restrict__ ggrid) {
    const double gh = t.raw < 1e-10 ? 0.0 : (extra_gh - (gN1 * t.N1 + gN2 * t.N2) / t.h);
    unsafeAtomicAdd(&ggrid[t.e + 1], (T)(gN1 / t.h + gh));
    unsafeAtomicAdd(&ggrid[t.e], (T)(-gN2 / t.h - gh));
}

template <typename T>
__global__ __launch_bounds__(kBlockL) void line2_eval_fwd_kernel(const T *__restrict__ grid, const T *__restrict__ u, int n,
                                                                const T *__restrict__ x_eval, int64_t m,
                                                                T *__restrict__ pred, T *__restrict__ dudx) {
    const int64_t stride = (int64_t)gridDim.x * kBlockL;
    for (int64_t q = (int64_t)blockIdx.x * kBlockL + threadIdx.x; q < m; q += stride) {
        const Hat t = hat_eval(grid, n, (double)x_eval[q]);
        const double ui = (double)u[t.e], uj = (double)u[t.e + 1];
        if (pred) pred[q] = (T)(ui * t.N1 + uj * t.N2);          // models.py:88
        if (dudx) dudx[q] = (T)((uj - ui) / t.h);
    }
}

template <typename T>
__global__ __launch_bounds__(kBlockL) void line2_eval_bwd_kernel(
    const T *__restrict__ grid, const T *__restrict__ u, int n, const T *__restrict__ x_eval,
    int64_t m, const T *__restrict__ cot, const T *__restrict__ cot_d, T *__restrict__ ggrid,
    T *__restrict__ gu, T *__restrict__ gx_eval) {
    const int64_t stride = (int64_t)gridDim.x * kBlockL;
    for (int64_t q = (int64_t)blockIdx.x * kBlockL + threadIdx.x; q < m; q += stride) {
        const Hat t = hat_eval(grid, n, (double)x_eval[q]);
        const double ui = (double)u[t.e], uj = (double)u[t.e + 1];
        const double g = cot ? (double)cot[q] : 0.0, gd = cot_d ? (double)cot_d[q] : 0.0;
        const double du = (uj - ui) / t.h;
        if (gu) {
            unsafeAtomicAdd(&gu[t.e], (T)(g * t.N1 - gd / t.h));
            unsafeAtomicAdd(&gu[t.e + 1], (T)(g * t.N2 + gd / t.h));
        }
        if (ggrid) hat_grid_grad(t, g * ui, g * uj, -gd * du / t.h, ggrid);
        if (gx_eval) gx_eval[q] = (T)(g * du);
    }
}

// examples/example3.py:27-70 with detached xq,wq (F8): loss += sum wq (E/2 du^2 - b u)
template <typename T>
__global__ __launch_bounds__(kBlockL) void bar_energy_kernel(const T *__restrict__ grid, const T *__restrict__ u, int n,
                                                            const T *__restrict__ xq, const T *__restrict__ wq,
                                                            const T *__restrict__ bq, int64_t npts, double E,
                                                            T *__restrict__ loss, T *__restrict__ ggrid,
                                                            T *__restrict__ gu) {
    __shared__ double red[kBlockL / 64];
    double loc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlockL;
    for (int64_t q = (int64_t)blockIdx.x * kBlockL + threadIdx.x; q < npts; q += stride) {
        const Hat t = hat_eval(grid, n, (double)xq[q]);
        const double ui = (double)u[t.e], uj = (double)u[t.e + 1], w = (double)wq[q], b = (double)bq[q];
        const double uu = ui * t.N1 + uj * t.N2, du = (uj - ui) / t.h;
        loc += w * (0.5 * E * du * du - b * uu);
        const double gdu = w * E * du, guq = -w * b;
        if (gu) {
            unsafeAtomicAdd(&gu[t.e], (T)(guq * t.N1 - gdu / t.h));
            unsafeAtomicAdd(&gu[t.e + 1], (T)(guq * t.N2 + gdu / t.h));
        }
        if (ggrid) hat_grid_grad(t, guq * ui, guq * uj, -gdu * du / t.h, ggrid);
    }
    const double tot = block_sum(loc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, (T)tot);
}

// examples/example1.py:38: loss += mean((pred - target)^2) and its backward
template <typename T>
__global__ __launch_bounds__(kBlockL) void line2_mse_kernel(const T *__restrict__ grid, const T *__restrict__ u, int n,
                                                           const T *__restrict__ x_eval, const T *__restrict__ target,
                                                           int64_t m, T *__restrict__ loss, T *__restrict__ ggrid,
                                                           T *__restrict__ gu) {
    __shared__ double red[kBlockL / 64];
    double loc = 0.0;
    const double invm = 1.0 / (double)m;
    const int64_t stride = (int64_t)gridDim.x * kBlockL;
    for (int64_t q = (int64_t)blockIdx.x * kBlockL + threadIdx.x; q < m; q += stride) {
        const Hat t = hat_eval(grid, n, (double)x_eval[q]);
        const double ui = (double)u[t.e], uj = (double)u[t.e + 1];
        const double diff = ui * t.N1 + uj * t.N2 - (double)target[q];
        loc += diff * diff;
        const double g = 2.0 * diff * invm;
        if (gu) {
            unsafeAtomicAdd(&gu[t.e], (T)(g * t.N1));
            unsafeAtomicAdd(&gu[t.e + 1], (T)(g * t.N2));
        }
        if (ggrid) hat_grid_grad(t, g * ui, g * uj, 0.0, ggrid);
    }
    const double tot = block_sum(loc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, (T)(tot * invm));
}

// ---- RECT-Q4 ---------------------------------------------------------------
template <typename T> struct Row2;                 // [.][2] rows of the caller's dtype
template <> struct Row2<double> { typedef double2 type; };
template <> struct Row2<float> { typedef float2 type; };

struct Q4 {
    Hat x, y;
    double u00, u10, u01, u11;
};
template <typename T>
__device__ __forceinline__ Q4 q4_eval(const T *__restrict__ gx, int nx, const T *__restrict__ gy, int ny,
                                      const T *__restrict__ u, double px, double py) {
    Q4 c;
    c.x = hat_eval(gx, nx, px);
    c.y = hat_eval(gy, ny, py);
    const int64_t b = (int64_t)c.x.e * ny + c.y.e;
    c.u00 = (double)u[b]; c.u01 = (double)u[b + 1]; c.u10 = (double)u[b + ny]; c.u11 = (double)u[b + ny + 1];   // models.py:197-200
    return c;
}
__device__ __forceinline__ double q4_value(const Q4 &c) {   // models.py:210
    return c.x.N1 * c.y.N1 * c.u00 + c.x.N2 * c.y.N1 * c.u10 + c.x.N1 * c.y.N2 * c.u01 + c.x.N2 * c.y.N2 * c.u11;
}
template <typename T>
__device__ __forceinline__ void q4_backward(const Q4 &c, double g, int ny, T *__restrict__ ggx, T *__restrict__ ggy,
                                            T *__restrict__ gu, double2 *gpt) {
    if (gu) {
        const int64_t b = (int64_t)c.x.e * ny + c.y.e;
        unsafeAtomicAdd(&gu[b], (T)(g * c.x.N1 * c.y.N1));
        unsafeAtomicAdd(&gu[b + ny], (T)(g * c.x.N2 * c.y.N1));
        unsafeAtomicAdd(&gu[b + 1], (T)(g * c.x.N1 * c.y.N2));
        unsafeAtomicAdd(&gu[b + ny + 1], (T)(g * c.x.N2 * c.y.N2));
    }
    const double gN1x = g * (c.y.N1 * c.u00 + c.y.N2 * c.u01), gN2x = g * (c.y.N1 * c.u10 + c.y.N2 * c.u11);
    const double gN1y = g * (c.x.N1 * c.u00 + c.x.N2 * c.u10), gN2y = g * (c.x.N1 * c.u01 + c.x.N2 * c.u11);
    if (ggx) hat_grid_grad(c.x, gN1x, gN2x, 0.0, ggx);
    if (ggy) hat_grid_grad(c.y, gN1y, gN2y, 0.0, ggy);
    if (gpt) *gpt = make_double2((gN2x - gN1x) / c.x.h, (gN2y - gN1y) / c.y.h);
}

template <typename T>
__global__ __launch_bounds__(kBlockL) void rectq4_eval_fwd_kernel(const T *__restrict__ gx, int nx, const T *__restrict__ gy,
                                                                 int ny, const T *__restrict__ u,
                                                                 const typename Row2<T>::type *__restrict__ x_eval,
                                                                 int64_t m, T *__restrict__ pred) {
    const int64_t stride = (int64_t)gridDim.x * kBlockL;
    for (int64_t q = (int64_t)blockIdx.x * kBlockL + threadIdx.x; q < m; q += stride) {
        const typename Row2<T>::type pt = x_eval[q];
        pred[q] = (T)q4_value(q4_eval(gx, nx, gy, ny, u, (double)pt.x, (double)pt.y));
    }
}

template <typename T>
__global__ __launch_bounds__(kBlockL) void rectq4_eval_bwd_kernel(
    const T *__restrict__ gx, int nx, const T *__restrict__ gy, int ny, const T *__restrict__ u,
    const typename Row2<T>::type *__restrict__ x_eval, int64_t m, const T *__restrict__ cot, T *__restrict__ ggx,
    T *__restrict__ ggy, T *__restrict__ gu, typename Row2<T>::type *__restrict__ gx_eval) {
    const int64_t stride = (int64_t)gridDim.x * kBlockL;
    for (int64_t q = (int64_t)blockIdx.x * kBlockL + threadIdx.x; q < m; q += stride) {
        const typename Row2<T>::type pt = x_eval[q];
        const Q4 c = q4_eval(gx, nx, gy, ny, u, (double)pt.x, (double)pt.y);
        double2 gpt;                                        // always taken (a conditional address of a local lands in scratch)
        q4_backward(c, (double)cot[q], ny, ggx, ggy, gu, &gpt);
        if (gx_eval) {
            typename Row2<T>::type o;
            o.x = (T)gpt.x; o.y = (T)gpt.y;
            gx_eval[q] = o;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlockL) void rectq4_mse_kernel(
    const T *__restrict__ gx, int nx, const T *__restrict__ gy, int ny, const T *__restrict__ u,
    const typename Row2<T>::type *__restrict__ x_eval, const T *__restrict__ target, int64_t m, T *__restrict__ loss,
    T *__restrict__ ggx, T *__restrict__ ggy, T *__restrict__ gu) {
    __shared__ double red[kBlockL / 64];
    double loc = 0.0;
    const double invm = 1.0 / (double)m;
    const int64_t stride = (int64_t)gridDim.x * kBlockL;
    for (int64_t q = (int64_t)blockIdx.x * kBlockL + threadIdx.x; q < m; q += stride) {
        const typename Row2<T>::type pt = x_eval[q];
        const Q4 c = q4_eval(gx, nx, gy, ny, u, (double)pt.x, (double)pt.y);
        const double diff = q4_value(c) - (double)target[q];
        loc += diff * diff;
        q4_backward(c, 2.0 * diff * invm, ny, ggx, ggy, gu, (double2 *)nullptr);
    }
    const double tot = block_sum(loc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, (T)(tot * invm));
}

static int grid_l(int64_t n) {
    int64_t g = (n + kBlockL - 1) / kBlockL;
    return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}

}  // namespace hfem

using namespace hfem;

#define HFEM_N_CHECK(n) HFEM_ARG_CHECK((n) >= 2 && (n) < (1ll << 31), "need 2 <= n < 2^31 grid nodes")

// ---- host side: one implementation per entry point, instantiated for double rows (the fp64 ABI) and float rows (the
//      *_f32 ABI: the reference's default dtype without widening copies; scratch of the workspace forms stays fp64)
namespace {

template <typename T>
int grid_param_fwd_impl(int device, const T *p, int64_t n, double x0, double xN, const uint8_t *mask, const T *initial,
                        T *grid, void *stream) {
    HFEM_ARG_CHECK(p && grid && n >= 1, "null pointer / empty increments");
    HFEM_ARG_CHECK(!mask || initial, "mask given without initial grid");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(grid_param_fwd_kernel<T>, dim3(1), dim3(kScan), 0, (hipStream_t)stream, p, n, x0, xN, mask, initial, grid);
    return launch_status("hfem_grid_param_fwd");
}

template <typename T>
int grid_param_bwd_impl(int device, const T *p, int64_t n, double x0, double xN, const uint8_t *mask, const T *ggrid, T *gp,
                        void *stream) {
    HFEM_ARG_CHECK(p && ggrid && gp && n >= 1, "null pointer / empty increments");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(grid_param_bwd_kernel<T>, dim3(1), dim3(kScan), 0, (hipStream_t)stream, p, n, x0, xN, mask, ggrid, gp);
    return launch_status("hfem_grid_param_bwd");
}

template <typename T>
int grid_param_fwd_ws_impl(int device, const T *p, int64_t n, double x0, double xN, const uint8_t *mask, const T *initial,
                           T *grid, double *cum, double *ws, void *stream) {
    HFEM_ARG_CHECK(p && grid && cum && ws && n >= 1, "null pointer / empty increments");
    HFEM_ARG_CHECK(!mask || initial, "mask given without initial grid");
    if (int rc = use_device(device)) return rc;
    const int64_t nb = (n + kGpChunk - 1) / kGpChunk;
    HFEM_ARG_CHECK(nb <= 2147483647, "grid too long");
    double *bsum = ws, *scal = ws + nb;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gp_fwd_local_kernel<T>, dim3((int)nb), dim3(kGpThreads), 0, s, p, n, cum, bsum);
    hipLaunchKernelGGL(gp_block_offsets_kernel, dim3(1), dim3(kScan), 0, s, bsum, nb, false, (const double *)nullptr, scal);
    hipLaunchKernelGGL(gp_fwd_final_kernel<T>, dim3((int)nb), dim3(kGpThreads), 0, s, n, x0, xN, mask, initial,
                       (const double *)bsum, (const double *)scal, cum, grid);
    return launch_status("hfem_grid_param_fwd_ws");
}

template <typename T>
int grid_param_bwd_ws_impl(int device, const T *p, int64_t n, double x0, double xN, const uint8_t *mask, const T *ggrid,
                           const double *cum, T *gp, double *ws, void *stream) {
    HFEM_ARG_CHECK(p && ggrid && cum && gp && ws && n >= 1, "null pointer / empty increments");
    if (int rc = use_device(device)) return rc;
    const int64_t nb = (n + kGpChunk - 1) / kGpChunk;
    HFEM_ARG_CHECK(nb <= 2147483647, "grid too long");
    double *suf = ws, *asum = ws + n, *dsum = asum + nb, *scal = dsum + nb;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gp_bwd_local_kernel<T>, dim3((int)nb), dim3(kGpThreads), 0, s, n, xN - x0, mask, ggrid, cum, suf, asum,
                       dsum);
    hipLaunchKernelGGL(gp_block_offsets_kernel, dim3(1), dim3(kScan), 0, s, asum, nb, true, (const double *)dsum, scal);
    hipLaunchKernelGGL(gp_bwd_final_kernel<T>, dim3((int)nb), dim3(kGpThreads), 0, s, p, n, xN - x0, cum, (const double *)suf,
                       (const double *)asum, (const double *)scal, gp);
    return launch_status("hfem_grid_param_bwd_ws");
}

template <typename T>
int line2_eval_fwd_impl(int device, const T *grid, const T *u, int64_t n, const T *x_eval, int64_t m, T *pred, T *dudx,
                        void *stream) {
    HFEM_N_CHECK(n);
    HFEM_ARG_CHECK(m >= 0, "negative point count");
    if (m == 0) return 0;
    HFEM_ARG_CHECK(grid && u && x_eval && (pred || dudx), "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(line2_eval_fwd_kernel<T>, dim3(grid_l(m)), dim3(kBlockL), 0, (hipStream_t)stream, grid, u, (int)n,
                       x_eval, m, pred, dudx);
    return launch_status("hfem_line2_eval_fwd");
}

template <typename T>
int line2_eval_bwd_impl(int device, const T *grid, const T *u, int64_t n, const T *x_eval, int64_t m, const T *cot,
                        const T *cot_dudx, T *ggrid, T *gu, T *gx_eval, void *stream) {
    HFEM_N_CHECK(n);
    HFEM_ARG_CHECK(m >= 0, "negative point count");
    if (m == 0) return 0;
    HFEM_ARG_CHECK(grid && u && x_eval && (cot || cot_dudx), "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(line2_eval_bwd_kernel<T>, dim3(grid_l(m)), dim3(kBlockL), 0, (hipStream_t)stream, grid, u, (int)n,
                       x_eval, m, cot, cot_dudx, ggrid, gu, gx_eval);
    return launch_status("hfem_line2_eval_bwd");
}

template <typename T>
int bar_energy_impl(int device, const T *grid, const T *u, int64_t n, const T *xq, const T *wq, const T *bq, int64_t npts,
                    double E, T *loss_acc, T *ggrid, T *gu, void *stream) {
    HFEM_N_CHECK(n);
    HFEM_ARG_CHECK(npts >= 0, "negative point count");
    if (npts == 0) return 0;
    HFEM_ARG_CHECK(grid && u && xq && wq && bq && loss_acc, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(bar_energy_kernel<T>, dim3(grid_l(npts)), dim3(kBlockL), 0, (hipStream_t)stream, grid, u, (int)n, xq,
                       wq, bq, npts, E, loss_acc, ggrid, gu);
    return launch_status("hfem_bar_energy");
}

template <typename T>
int line2_mse_impl(int device, const T *grid, const T *u, int64_t n, const T *x_eval, const T *target, int64_t m, T *loss_acc,
                   T *ggrid, T *gu, void *stream) {
    HFEM_N_CHECK(n);
    HFEM_ARG_CHECK(m >= 1, "need at least one point (mean of an empty set)");
    HFEM_ARG_CHECK(grid && u && x_eval && target && loss_acc, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(line2_mse_kernel<T>, dim3(grid_l(m)), dim3(kBlockL), 0, (hipStream_t)stream, grid, u, (int)n, x_eval,
                       target, m, loss_acc, ggrid, gu);
    return launch_status("hfem_line2_mse");
}

template <typename T>
int rectq4_eval_fwd_impl(int device, const T *gx, int64_t nx, const T *gy, int64_t ny, const T *u, const T *x_eval, int64_t m,
                         T *pred, void *stream) {
    HFEM_N_CHECK(nx);
    HFEM_N_CHECK(ny);
    HFEM_ARG_CHECK(m >= 0, "negative point count");
    if (m == 0) return 0;
    HFEM_ARG_CHECK(gx && gy && u && x_eval && pred, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(rectq4_eval_fwd_kernel<T>, dim3(grid_l(m)), dim3(kBlockL), 0, (hipStream_t)stream, gx, (int)nx, gy,
                       (int)ny, u, (const typename Row2<T>::type *)x_eval, m, pred);
    return launch_status("hfem_rectq4_eval_fwd");
}

template <typename T>
int rectq4_eval_bwd_impl(int device, const T *gx, int64_t nx, const T *gy, int64_t ny, const T *u, const T *x_eval, int64_t m,
                         const T *cot, T *ggx, T *ggy, T *gu, T *gx_eval, void *stream) {
    HFEM_N_CHECK(nx);
    HFEM_N_CHECK(ny);
    HFEM_ARG_CHECK(m >= 0, "negative point count");
    if (m == 0) return 0;
    HFEM_ARG_CHECK(gx && gy && u && x_eval && cot, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(rectq4_eval_bwd_kernel<T>, dim3(grid_l(m)), dim3(kBlockL), 0, (hipStream_t)stream, gx, (int)nx, gy,
                       (int)ny, u, (const typename Row2<T>::type *)x_eval, m, cot, ggx, ggy, gu,
                       (typename Row2<T>::type *)gx_eval);
    return launch_status("hfem_rectq4_eval_bwd");
}

template <typename T>
int rectq4_mse_impl(int device, const T *gx, int64_t nx, const T *gy, int64_t ny, const T *u, const T *x_eval, const T *target,
                    int64_t m, T *loss_acc, T *ggx, T *ggy, T *gu, void *stream) {
    HFEM_N_CHECK(nx);
    HFEM_N_CHECK(ny);
    HFEM_ARG_CHECK(m >= 1, "need at least one point (mean of an empty set)");
    HFEM_ARG_CHECK(gx && gy && u && x_eval && target && loss_acc, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(rectq4_mse_kernel<T>, dim3(grid_l(m)), dim3(kBlockL), 0, (hipStream_t)stream, gx, (int)nx, gy, (int)ny,
                       u, (const typename Row2<T>::type *)x_eval, target, m, loss_acc, ggx, ggy, gu);
    return launch_status("hfem_rectq4_mse");
}

}  // namespace

// Workspace forms for long grids (three launches each way over all CUs; the forms without _ws are one workgroup):
// ws holds hfem_grid_param_ws_elems(n) doubles of scratch; cum[n] is an OUTPUT of the forward (the clamped-softplus
// running sums, cum[n-1] = their total) that the backward reads back.  Both stay fp64 in the _f32 forms.
extern "C" int64_t hfem_grid_param_ws_elems(int64_t n) {
    if (n < 1) return 8;
    const int64_t nb = (n + kGpChunk - 1) / kGpChunk;
    return n + 2 * nb + 8;
}

#define HFEM_ROW_ABI(SUFFIX, T)                                                                                                  \
    extern "C" int hfem_grid_param_fwd##SUFFIX(int device, const T *p, int64_t n, double x0, double xN, const uint8_t *mask,       \
                                               const T *initial, T *grid, void *stream) {                                         \
        return grid_param_fwd_impl<T>(device, p, n, x0, xN, mask, initial, grid, stream);                                        \
    }                                                                                                                            \
    extern "C" int hfem_grid_param_bwd##SUFFIX(int device, const T *p, int64_t n, double x0, double xN, const uint8_t *mask,       \
                                               const T *ggrid, T *gp, void *stream) {                                             \
        return grid_param_bwd_impl<T>(device, p, n, x0, xN, mask, ggrid, gp, stream);                                            \
    }                                                                                                                            \
    extern "C" int hfem_grid_param_fwd_ws##SUFFIX(int device, const T *p, int64_t n, double x0, double xN, const uint8_t *mask,    \
                                                  const T *initial, T *grid, double *cum, double *ws, void *stream) {             \
        return grid_param_fwd_ws_impl<T>(device, p, n, x0, xN, mask, initial, grid, cum, ws, stream);                            \
    }                                                                                                                            \
    extern "C" int hfem_grid_param_bwd_ws##SUFFIX(int device, const T *p, int64_t n, double x0, double xN, const uint8_t *mask,    \
                                                  const T *ggrid, const double *cum, T *gp, double *ws, void *stream) {           \
        return grid_param_bwd_ws_impl<T>(device, p, n, x0, xN, mask, ggrid, cum, gp, ws, stream);                                \
    }                                                                                                                            \
    extern "C" int hfem_line2_eval_fwd##SUFFIX(int device, const T *grid, const T *u, int64_t n, const T *x_eval, int64_t m,       \
                                               T *pred, T *dudx, void *stream) {                                                  \
        return line2_eval_fwd_impl<T>(device, grid, u, n, x_eval, m, pred, dudx, stream);                                        \
    }                                                                                                                            \
    extern "C" int hfem_line2_eval_bwd##SUFFIX(int device, const T *grid, const T *u, int64_t n, const T *x_eval, int64_t m,       \
                                               const T *cot, const T *cot_dudx, T *ggrid, T *gu, T *gx_eval, void *stream) {      \
        return line2_eval_bwd_impl<T>(device, grid, u, n, x_eval, m, cot, cot_dudx, ggrid, gu, gx_eval, stream);                 \
    }                                                                                                                            \
    extern "C" int hfem_bar_energy##SUFFIX(int device, const T *grid, const T *u, int64_t n, const T *xq, const T *wq,             \
                                           const T *bq, int64_t npts, double E, T *loss_acc, T *ggrid, T *gu, void *stream) {     \
        return bar_energy_impl<T>(device, grid, u, n, xq, wq, bq, npts, E, loss_acc, ggrid, gu, stream);                         \
    }                                                                                                                            \
    extern "C" int hfem_line2_mse##SUFFIX(int device, const T *grid, const T *u, int64_t n, const T *x_eval, const T *target,      \
                                          int64_t m, T *loss_acc, T *ggrid, T *gu, void *stream) {                                \
        return line2_mse_impl<T>(device, grid, u, n, x_eval, target, m, loss_acc, ggrid, gu, stream);                            \
    }                                                                                                                            \
    extern "C" int hfem_rectq4_eval_fwd##SUFFIX(int device, const T *gx, int64_t nx, const T *gy, int64_t ny, const T *u,          \
                                                const T *x_eval, int64_t m, T *pred, void *stream) {                              \
        return rectq4_eval_fwd_impl<T>(device, gx, nx, gy, ny, u, x_eval, m, pred, stream);                                      \
    }                                                                                                                            \
    extern "C" int hfem_rectq4_eval_bwd##SUFFIX(int device, const T *gx, int64_t nx, const T *gy, int64_t ny, const T *u,          \
                                                const T *x_eval, int64_t m, const T *cot, T *ggx, T *ggy, T *gu, T *gx_eval,      \
                                                void *stream) {                                                                  \
        return rectq4_eval_bwd_impl<T>(device, gx, nx, gy, ny, u, x_eval, m, cot, ggx, ggy, gu, gx_eval, stream);                \
    }                                                                                                                            \
    extern "C" int hfem_rectq4_mse##SUFFIX(int device, const T *gx, int64_t nx, const T *gy, int64_t ny, const T *u,               \
                                           const T *x_eval, const T *target, int64_t m, T *loss_acc, T *ggx, T *ggy, T *gu,       \
                                           void *stream) {                                                                       \
        return rectq4_mse_impl<T>(device, gx, nx, gy, ny, u, x_eval, target, m, loss_acc, ggx, ggy, gu, stream);                 \
    }

HFEM_ROW_ABI(, double)
HFEM_ROW_ABI(_f32, float)
#undef HFEM_ROW_ABI

extern "C" int hfem_version(void) { return HFEM_VERSION; }
extern "C" const char *hfem_last_error(void) { return hfem::get_error(); }
extern "C" int hfem_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : -1;
}

// Fused Adam step on a flat parameter tensor (the first "next" row after the
// energy path): the reference drives its models with torch.optim.Adam
// (/root/reference/examples/example1.py:31, example2.py:37, example3.py:89), which costs ~10 small
// kernels per parameter tensor per step; once the energy is one launch the optimiser dominates.
// One launch here reads p,g,m,v and writes p,m,v (56 B/param fp64) -- bandwidth-bound, no reuse.
// Arithmetic follows torch.optim.Adam (betas, eps, bias corrections, no weight decay / amsgrad)
// operation for operation:  m += (g-m)(1-b1);  v = v b2 + (1-b2) g g;
//                           p -= (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps)   (same association as torch's foreach kernels).
#include <hip/hip_runtime.h>

#include <cmath>

#include "hfem_device.h"

namespace hfem {

// One element of torch.optim.Adam's update (betas, eps, bias correction folded into step_size / sqrt_bc2; no weight decay,
// no amsgrad), operation for operation as torch's single-tensor path.
template <typename T>
__device__ __forceinline__ void adam_one(T &p, const T g, T &m, T &v, double w1, double b2, double w2, double step_size,
                                         double sqrt_bc2, double eps) {
    const T mi = m + (T)w1 * (g - m);                            // lerp_(grad, 1 - beta1)
    const T vi = v * (T)b2 + (T)w2 * (g * g);                    // mul_(beta2).addcmul_(g, g, 1 - beta2): a + alpha (b c)
    const T denom = (T)sqrt((double)vi) / (T)sqrt_bc2 + (T)eps;  // (sqrt(v) / sqrt(bc2)).add_(eps): a true division
    m = mi;
    v = vi;
    p = p - (T)step_size * (mi / denom);
}
template <typename T> struct AdamVec;                           // 16-byte vectors of the parameter dtype
template <> struct AdamVec<double> { typedef double2 type; static constexpr int N = 2; };
template <> struct AdamVec<float> { typedef float4 type; static constexpr int N = 4; };

// Body shared by the two kernels below: 16-byte loads / stores (four arrays in flight per thread), grid-stride over the
// vector part, the < N-element tail by the first threads.  The update is memory-bound: 7 array passes per parameter.
template <typename T>
__device__ __forceinline__ void adam_body(T *__restrict__ p, const T *__restrict__ g, T *__restrict__ m, T *__restrict__ v,
                                          int64_t n, double w1, double b2, double w2, double step_size, double sqrt_bc2,
                                          double eps) {
    typedef typename AdamVec<T>::type V;
    constexpr int N = AdamVec<T>::N;
    // views that do not start on a 16-byte boundary take the scalar loop for everything
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const int64_t nv = aligned ? n / N : 0, stride = (int64_t)gridDim.x * 256;
    V *pv = reinterpret_cast<V *>(p), *mv = reinterpret_cast<V *>(m), *vv = reinterpret_cast<V *>(v);
    const V *gv = reinterpret_cast<const V *>(g);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
        V P = pv[i], M = mv[i], Vv = vv[i];
        const V G = gv[i];
        T *pp = reinterpret_cast<T *>(&P), *mm = reinterpret_cast<T *>(&M), *vx = reinterpret_cast<T *>(&Vv);
        const T *gg = reinterpret_cast<const T *>(&G);
#pragma unroll
        for (int k = 0; k < N; ++k) adam_one<T>(pp[k], gg[k], mm[k], vx[k], w1, b2, w2, step_size, sqrt_bc2, eps);
        mv[i] = M;
        vv[i] = Vv;
        pv[i] = P;
    }
    for (int64_t t = nv * N + (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += stride)   // tail / unaligned views
        adam_one<T>(p[t], g[t], m[t], v[t], w1, b2, w2, step_size, sqrt_bc2, eps);
}

template <typename T>
__global__ __launch_bounds__(256) void adam_step_kernel(T *__restrict__ p, const T *__restrict__ g,
                                                        T *__restrict__ m, T *__restrict__ v, int64_t n,
                                                        double w1, double b2, double w2, double step_size,
                                                        double sqrt_bc2, double eps) {
    adam_body<T>(p, g, m, v, n, w1, b2, w2, step_size, sqrt_bc2, eps);
}

// hipGraph-capturable variant: the step count lives on the device (kernel arguments are frozen inside a graph),
// the bias corrections are computed per thread from it (two pow() on uniform values: noise next to 56 B/param).
template <typename T>
__global__ __launch_bounds__(256) void adam_step_dev_kernel(T *__restrict__ p, const T *__restrict__ g,
                                                            T *__restrict__ m, T *__restrict__ v, int64_t n,
                                                            double b1, double b2, double lr, double eps,
                                                            const int64_t *__restrict__ step_dev) {
    const double step = (double)step_dev[0];
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    adam_body<T>(p, g, m, v, n, 1.0 - b1, b2, 1.0 - b2, lr / bc1, sqrt(bc2), eps);
}

// the same update on the [.][2] fp64 rows listed in rows[] only (owner-sharded multi-GPU mode: a rank updates exactly the
// parameter rows its tiles own); one thread per row
__global__ __launch_bounds__(256) void adam_step_rows_dev_kernel(double2 *__restrict__ p, const double2 *__restrict__ g,
                                                                 double2 *__restrict__ m, double2 *__restrict__ v,
                                                                 const int32_t *__restrict__ rows, int64_t n_rows,
                                                                 double b1, double b2, double lr, double eps,
                                                                 const int64_t *__restrict__ step_dev) {
    const double step = (double)step_dev[0];
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    const double step_size = lr / bc1, sqrt_bc2 = sqrt(bc2), w1 = 1.0 - b1, w2 = 1.0 - b2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows) return;
    const int32_t r = rows[i];
    const double2 gi = g[r], m0 = m[r], v0 = v[r], p0 = p[r];
    double2 mi, vi, pi;
    mi.x = m0.x + w1 * (gi.x - m0.x); mi.y = m0.y + w1 * (gi.y - m0.y);
    vi.x = v0.x * b2 + w2 * (gi.x * gi.x); vi.y = v0.y * b2 + w2 * (gi.y * gi.y);
    pi.x = p0.x - step_size * (mi.x / (sqrt(vi.x) / sqrt_bc2 + eps));
    pi.y = p0.y - step_size * (mi.y / (sqrt(vi.y) / sqrt_bc2 + eps));
    m[r] = mi; v[r] = vi; p[r] = pi;
}

// both parameter tensors of the TRI3 model in one launch: rows_x[n_x] of (px, gx, mx, vx) with lr_x, then rows_u[n_u] of
// (pu, gu, mu, vu) with lr_u; the step the bias corrections use is step_dev[0] + step_offset
__global__ __launch_bounds__(256) void adam_step_rows2_dev_kernel(double2 *__restrict__ px, const double2 *__restrict__ gx,
                                                                  double2 *__restrict__ mx, double2 *__restrict__ vx,
                                                                  const int32_t *__restrict__ rows_x, int64_t n_x, double lr_x,
                                                                  double2 *__restrict__ pu, const double2 *__restrict__ gu,
                                                                  double2 *__restrict__ mu, double2 *__restrict__ vu,
                                                                  const int32_t *__restrict__ rows_u, int64_t n_u, double lr_u,
                                                                  double b1, double b2, double eps,
                                                                  const int64_t *__restrict__ step_dev, int64_t step_offset) {
    const double step = (double)(step_dev[0] + step_offset);
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    const double sqrt_bc2 = sqrt(bc2), w1 = 1.0 - b1, w2 = 1.0 - b2;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_x + n_u) return;
    const bool isx = i < n_x;
    if (!isx) i -= n_x;
    double2 *p = isx ? px : pu, *m = isx ? mx : mu, *v = isx ? vx : vu;
    const double2 *g = isx ? gx : gu;
    const double step_size = (isx ? lr_x : lr_u) / bc1;
    const int32_t r = (isx ? rows_x : rows_u)[i];
    const double2 gi = g[r], m0 = m[r], v0 = v[r], p0 = p[r];
    double2 mi, vi, pi;
    mi.x = m0.x + w1 * (gi.x - m0.x); mi.y = m0.y + w1 * (gi.y - m0.y);
    vi.x = v0.x * b2 + w2 * (gi.x * gi.x); vi.y = v0.y * b2 + w2 * (gi.y * gi.y);
    pi.x = p0.x - step_size * (mi.x / (sqrt(vi.x) / sqrt_bc2 + eps));
    pi.y = p0.y - step_size * (mi.y / (sqrt(vi.y) / sqrt_bc2 + eps));
    m[r] = mi; v[r] = vi; p[r] = pi;
}

// float rows (an fp32 model): torch's fp32 arithmetic, component by component (adam_one<float>)
__global__ __launch_bounds__(256) void adam_step_rows2_dev_f32_kernel(float2 *__restrict__ px, const float2 *__restrict__ gx,
                                                                      float2 *__restrict__ mx, float2 *__restrict__ vx,
                                                                      const int32_t *__restrict__ rows_x, int64_t n_x, double lr_x,
                                                                      float2 *__restrict__ pu, const float2 *__restrict__ gu,
                                                                      float2 *__restrict__ mu, float2 *__restrict__ vu,
                                                                      const int32_t *__restrict__ rows_u, int64_t n_u, double lr_u,
                                                                      double b1, double b2, double eps,
                                                                      const int64_t *__restrict__ step_dev, int64_t step_offset) {
    const double step = (double)(step_dev[0] + step_offset);
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    const double sqrt_bc2 = sqrt(bc2), w1 = 1.0 - b1, w2 = 1.0 - b2;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_x + n_u) return;
    const bool isx = i < n_x;
    if (!isx) i -= n_x;
    float2 *p = isx ? px : pu, *m = isx ? mx : mu, *v = isx ? vx : vu;
    const float2 *g = isx ? gx : gu;
    const double step_size = (isx ? lr_x : lr_u) / bc1;
    const int32_t r = (isx ? rows_x : rows_u)[i];
    const float2 gi = g[r];
    float2 mi = m[r], vi = v[r], pi = p[r];
    adam_one<float>(pi.x, gi.x, mi.x, vi.x, w1, b2, w2, step_size, sqrt_bc2, eps);
    adam_one<float>(pi.y, gi.y, mi.y, vi.y, w1, b2, w2, step_size, sqrt_bc2, eps);
    m[r] = mi; v[r] = vi; p[r] = pi;
}

// ---- multi-tensor form: ONE launch for every parameter tensor of the optimiser.  table[t] (device memory) describes tensor
// t and the first block that works on it; block b finds its tensor by a scan of the (few) table entries and updates one
// contiguous chunk of 16-byte vectors.  The step count: step_dev[0] holds the number of COMPLETED steps, every block reads
// it when it starts and uses step_dev[0] + step_offset; when step_dev is given the LAST block to finish (ticket counter)
// bumps it -- strictly after every block's read, so no separate counter launch and no race.
constexpr int kAdamTicketClasses = 32;                      // ticket buffer: (1 + 32) counters, 128 bytes apart

template <typename T>
__device__ __forceinline__ void adam_chunk(const hfem_adam_tensor &e, int64_t chunk_vecs, int64_t b, double step_size,
                                           double sqrt_bc2) {
    typedef typename AdamVec<T>::type V;
    constexpr int N = AdamVec<T>::N;
    T *p = (T *)e.p, *m = (T *)e.m, *v = (T *)e.v;
    const T *g = (const T *)e.g;
    const double w1 = 1.0 - e.beta1, w2 = 1.0 - e.beta2;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const int64_t nv = aligned ? e.n / N : 0;
    V *pv = reinterpret_cast<V *>(p), *mv = reinterpret_cast<V *>(m), *vv = reinterpret_cast<V *>(v);
    const V *gv = reinterpret_cast<const V *>(g);
    const int64_t hi = min(nv, (b + 1) * chunk_vecs);
    for (int64_t i = b * chunk_vecs + threadIdx.x; i < hi; i += 256) {
        V P = pv[i], M = mv[i], Vv = vv[i];
        const V G = gv[i];
        T *pp = reinterpret_cast<T *>(&P), *mm = reinterpret_cast<T *>(&M), *vx = reinterpret_cast<T *>(&Vv);
        const T *gg = reinterpret_cast<const T *>(&G);
#pragma unroll
        for (int k = 0; k < N; ++k) adam_one<T>(pp[k], gg[k], mm[k], vx[k], w1, e.beta2, w2, step_size, sqrt_bc2, e.eps);
        mv[i] = M;
        vv[i] = Vv;
        pv[i] = P;
    }
    if (b == 0)                                             // tail / unaligned views: the tensor's first block
        for (int64_t t = nv * N + threadIdx.x; t < e.n; t += 256)
            adam_one<T>(p[t], g[t], m[t], v[t], w1, e.beta2, w2, step_size, sqrt_bc2, e.eps);
}

__global__ __launch_bounds__(256) void adam_multi_dev_kernel(const hfem_adam_tensor *__restrict__ table, int n_tensors,
                                                             int64_t chunk_vecs, int64_t *__restrict__ step_dev,
                                                             int64_t step_offset, int32_t *__restrict__ ticket) {
    int t = 0;
    while (t + 1 < n_tensors && (int)blockIdx.x >= table[t + 1].block_begin) ++t;
    const hfem_adam_tensor e = table[t];
    const double step = (double)((step_dev ? step_dev[0] : 0) + step_offset);
    const double bc1 = 1.0 - pow(e.beta1, step), bc2 = 1.0 - pow(e.beta2, step);
    const int64_t b = (int64_t)blockIdx.x - e.block_begin;
    if (e.dtype == 0) adam_chunk<double>(e, chunk_vecs, b, e.lr / bc1, sqrt(bc2));
    else adam_chunk<float>(e, chunk_vecs, b, e.lr / bc1, sqrt(bc2));
    if (step_dev) {
        // Last-arriver bump of the step counter, two levels: same-address device atomics serialise at ~30 ns each (4096 blocks
        // on ONE ticket cost 130 us, measured), so block b takes a ticket of class b % 32 (its own 128-byte line) and only
        // the last block of a class takes one of the top-level counter: <= 2 x 64 serialised atomics, off the data path.
        // No fence: a release fence at agent scope writes back the XCD's whole L2 (buffer_wbl2) -- 2048 of them cost 60 us,
        // measured -- and none is needed: a thread that reaches the barrier has USED the step it loaded, so every read of
        // step_dev precedes its block's ticket; the counter's new value reaches the next launch through the kernel boundary.
        __syncthreads();                                    // every thread of this block has read (and used) the step
        if (threadIdx.x == 0) {
            const int n_cls = (int)gridDim.x < kAdamTicketClasses ? (int)gridDim.x : kAdamTicketClasses;
            const int cls = (int)blockIdx.x % n_cls;
            const int in_cls = ((int)gridDim.x - cls + n_cls - 1) / n_cls;
            int32_t *sub = ticket + 32 * (1 + cls);
            if (atomicAdd(sub, 1) == in_cls - 1) {           // last block of its class
                atomicExch(sub, 0);
                if (atomicAdd(ticket, 1) == n_cls - 1) {     // last class: every block has read the step
                    atomicExch(ticket, 0);
                    step_dev[0] += 1;
                }
            }
        }
    }
}

__global__ void counter_add_kernel(int64_t *c, int64_t inc) { c[0] += inc; }

// step += 1; bc = {1 - b1^step, sqrt(1 - b2^step)}: the scalars of one Adam step, for kernels that fuse the update
__global__ void adam_prep_kernel(int64_t *step, double b1, double b2, double *bc) {
    const int64_t st = step[0] + 1;
    step[0] = st;
    bc[0] = 1.0 - pow(b1, (double)st);
    bc[1] = sqrt(1.0 - pow(b2, (double)st));
}

}  // namespace hfem

using namespace hfem;

extern "C" int hfem_adam_step(int device, void *p, const void *g, void *m, void *v, int64_t n, int32_t dtype,
                              double lr, double beta1, double beta2, double eps, int64_t step, void *stream) {
    HFEM_ARG_CHECK(n >= 0 && step >= 1, "need n >= 0 and step >= 1");
    if (n == 0) return 0;
    HFEM_ARG_CHECK(p && g && m && v, "null pointer");
    HFEM_ARG_CHECK(dtype == 0 || dtype == 1, "dtype: 0 = fp64, 1 = fp32");
    if (int rc = use_device(device)) return rc;
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    const double step_size = lr / bc1, sqrt_bc2 = std::sqrt(bc2);
    int64_t grid = (n / (dtype == 0 ? 2 : 4) + 255) / 256;     // one 16-byte vector per thread, grid-stride beyond 8 blocks / CU
    if (grid < 1) grid = 1;
    if (grid > 2048) grid = 2048;
    if (dtype == 0)
        hipLaunchKernelGGL(adam_step_kernel<double>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (double *)p,
                           (const double *)g, (double *)m, (double *)v, n, 1.0 - beta1, beta2, 1.0 - beta2, step_size,
                           sqrt_bc2, eps);
    else
        hipLaunchKernelGGL(adam_step_kernel<float>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (float *)p,
                           (const float *)g, (float *)m, (float *)v, n, 1.0 - beta1, beta2, 1.0 - beta2, step_size,
                           sqrt_bc2, eps);
    return launch_status("hfem_adam_step");
}

// Same update with the step count read from device memory (1-based: the value the counter holds when the kernel
// runs), for optimiser steps captured in a hipGraph.  hfem_counter_add bumps such a counter in stream order.
extern "C" int hfem_adam_step_dev(int device, void *p, const void *g, void *m, void *v, int64_t n, int32_t dtype,
                                  double lr, double beta1, double beta2, double eps, const int64_t *step_dev,
                                  void *stream) {
    HFEM_ARG_CHECK(n >= 0, "need n >= 0");
    if (n == 0) return 0;
    HFEM_ARG_CHECK(p && g && m && v && step_dev, "null pointer");
    HFEM_ARG_CHECK(dtype == 0 || dtype == 1, "dtype: 0 = fp64, 1 = fp32");
    if (int rc = use_device(device)) return rc;
    int64_t grid = (n / (dtype == 0 ? 2 : 4) + 255) / 256;     // one 16-byte vector per thread, grid-stride beyond 8 blocks / CU
    if (grid < 1) grid = 1;
    if (grid > 2048) grid = 2048;
    if (dtype == 0)
        hipLaunchKernelGGL(adam_step_dev_kernel<double>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (double *)p,
                           (const double *)g, (double *)m, (double *)v, n, beta1, beta2, lr, eps, step_dev);
    else
        hipLaunchKernelGGL(adam_step_dev_kernel<float>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (float *)p,
                           (const float *)g, (float *)m, (float *)v, n, beta1, beta2, lr, eps, step_dev);
    return launch_status("hfem_adam_step_dev");
}

extern "C" int hfem_adam_step_rows_dev(int device, double *p, const double *g, double *m, double *v, const int32_t *rows,
                                       int64_t n_rows, double lr, double beta1, double beta2, double eps,
                                       const int64_t *step_dev, void *stream) {
    HFEM_ARG_CHECK(n_rows >= 0, "negative row count");
    if (n_rows == 0) return 0;
    HFEM_ARG_CHECK(p && g && m && v && rows && step_dev, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(adam_step_rows_dev_kernel, dim3((int)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (double2 *)p, (const double2 *)g, (double2 *)m, (double2 *)v, rows, n_rows, beta1, beta2, lr, eps,
                       step_dev);
    return launch_status("hfem_adam_step_rows_dev");
}

extern "C" int hfem_adam_step_rows2_dev(int device, double *px, const double *gx, double *mx, double *vx,
                                        const int32_t *rows_x, int64_t n_x, double lr_x, double *pu, const double *gu,
                                        double *mu, double *vu, const int32_t *rows_u, int64_t n_u, double lr_u,
                                        double beta1, double beta2, double eps, const int64_t *step_dev,
                                        int64_t step_offset, void *stream) {
    HFEM_ARG_CHECK(n_x >= 0 && n_u >= 0, "negative row count");
    if (n_x + n_u == 0) return 0;
    HFEM_ARG_CHECK(step_dev && (n_x == 0 || (px && gx && mx && vx && rows_x)) && (n_u == 0 || (pu && gu && mu && vu && rows_u)),
                   "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(adam_step_rows2_dev_kernel, dim3((int)((n_x + n_u + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (double2 *)px, (const double2 *)gx, (double2 *)mx, (double2 *)vx, rows_x, n_x, lr_x, (double2 *)pu,
                       (const double2 *)gu, (double2 *)mu, (double2 *)vu, rows_u, n_u, lr_u, beta1, beta2, eps, step_dev,
                       step_offset);
    return launch_status("hfem_adam_step_rows2_dev");
}

extern "C" int hfem_adam_step_rows2_dev_f32(int device, float *px, const float *gx, float *mx, float *vx,
                                            const int32_t *rows_x, int64_t n_x, double lr_x, float *pu, const float *gu,
                                            float *mu, float *vu, const int32_t *rows_u, int64_t n_u, double lr_u,
                                            double beta1, double beta2, double eps, const int64_t *step_dev,
                                            int64_t step_offset, void *stream) {
    HFEM_ARG_CHECK(n_x >= 0 && n_u >= 0, "negative row count");
    if (n_x + n_u == 0) return 0;
    HFEM_ARG_CHECK(step_dev && (n_x == 0 || (px && gx && mx && vx && rows_x)) && (n_u == 0 || (pu && gu && mu && vu && rows_u)),
                   "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(adam_step_rows2_dev_f32_kernel, dim3((int)((n_x + n_u + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float2 *)px, (const float2 *)gx, (float2 *)mx, (float2 *)vx, rows_x, n_x, lr_x, (float2 *)pu,
                       (const float2 *)gu, (float2 *)mu, (float2 *)vu, rows_u, n_u, lr_u, beta1, beta2, eps, step_dev,
                       step_offset);
    return launch_status("hfem_adam_step_rows2_dev_f32");
}

extern "C" int hfem_adam_multi_dev(int device, const hfem_adam_tensor *table_dev, int32_t n_tensors, int32_t n_blocks,
                                   int64_t chunk_vecs, int64_t *step_dev, int64_t step_offset, int32_t *ticket_dev,
                                   void *stream) {
    HFEM_ARG_CHECK(n_tensors >= 0 && n_blocks >= 0 && chunk_vecs >= 1, "bad sizes");
    if (n_tensors == 0 || n_blocks == 0) return 0;
    HFEM_ARG_CHECK(table_dev, "null table");
    HFEM_ARG_CHECK(!step_dev || ticket_dev, "a device step counter needs a ticket counter");
    HFEM_ARG_CHECK(step_dev || step_offset >= 1, "need a device step counter or a host step >= 1");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(adam_multi_dev_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n_tensors,
                       chunk_vecs, step_dev, step_offset, ticket_dev);
    return launch_status("hfem_adam_multi_dev");
}

extern "C" int hfem_counter_add(int device, int64_t *counter, int64_t inc, void *stream) {
    HFEM_ARG_CHECK(counter, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, inc);
    return launch_status("hfem_counter_add");
}

extern "C" int hfem_adam_prep(int device, int64_t *step_dev, double beta1, double beta2, double *bc_dev, void *stream) {
    HFEM_ARG_CHECK(step_dev && bc_dev, "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, beta1, beta2, bc_dev);
    return launch_status("hfem_adam_prep");
}

// L-BFGS on the flat parameter vector, device-resident (the reference drives example 4
// with torch.optim.LBFGS(model.parameters()) -- /root/reference/examples/example4.py:68-78 -- i.e. lr 1,
// max_iter 20, history 100, no line search).  torch's step is ~4*history small vector ops plus ~6 host syncs
// per inner iteration; once the energy is one 14 us launch that is the whole run time.
//
// Same algorithm (torch/optim/lbfgs.py: memory update if y.s > 1e-10, H_diag = y.s / y.y, two-loop recursion,
// first step t = min(1, 1/|g|_1) lr, break tests), different evaluation order: the two-loop recursion only
// ever needs inner products between g and the history vectors, so it is run in COEFFICIENT space.  With
//   q = -g - sum_j al_j y_j,   r = H q + sum_j c_j s_j,   c_j = al_j - be_j
//   al_i = ro_i ( -s_i.g - sum_{j newer} al_j s_i.y_j )
//   be_i = ro_i ( H ( -y_i.g - sum_j al_j y_i.y_j ) + sum_{j older} c_j s_j.y_i )
// only S^T g, Y^T g and the Gram blocks S^T Y, Y^T Y are needed.  Per iteration:
//   pair_kernel        y = g - g_prev, s = t d into the spare ring slot (+ y.s, y.y partials)      [1 pass over N]
//   multidot_kernel    ONE pass over the history: per slot j  Y_j.g, S_j.g and, for a new pair,
//                      Y_j.s_new, Y_j.y_new, S_j.y_new (row/column of the Gram blocks)              [2m vectors read once]
//   recursion_kernel   one workgroup: the two loops on (m+1)^2 Gram entries, gtd = g.d, t, break flag
//   direction_kernel   ONE pass: d = -H g + sum_j (-H al_j) y_j + sum_j c_j s_j (+ max|d| partials) [2m vectors read once]
// i.e. the history is streamed twice (the two-loop recursion also reads every vector twice) in 2 launches instead
// of 4m, all scalars stay on the device, and the host reads one status record per iteration.  Dots accumulate in
// fp64 for fp32 and fp64 vectors alike; every reduction has a fixed order (bit-reproducible).
#include <hip/hip_runtime.h>

#include <mutex>

#include <cmath>
#include <new>

#include "hfem_device.h"

namespace hfem {

constexpr int kLb = 256;           // threads per block everywhere
constexpr int kLbPer = 8;          // elements per thread per chunk
constexpr int kLbChunk = kLb * kLbPer;

struct LbfgsState {                // one record in device memory
    int n_iter;                    // torch state["n_iter"]
    int count, head;               // ring: logical i (0 = oldest) lives in slot (head + i) % M1, i < count
    int new_slot;                  // slot of the pair accepted in this iteration, -1 if none
    int stop_gtd;                  // g.d > -tolerance_change: no update this iteration
    int skip;                      // sharded flow: the break tests fired (or the host wants no direction): recursion / direction passes return at once
    double H_diag, t, gtd, cg;     // cg = -H_diag: coefficient of g in d
    double loss, prev_loss;
    double g_absmax, g_abssum, gg, d_absmax, ys, yy;
    double flags;                  // bit 0 opt_cond, 1 small step, 2 small loss change, 3 gtd break  (check_kernel)
    int halt;                      // sharded flow, several iterations per captured graph: the PREVIOUS iteration ended the step (a
                                   // break test fired, or g.d > -tolerance_change): this iteration's finish, recursion, direction and
                                   // status write do nothing -- the state and the status record stay those of the iteration that ended it
};

struct LbfgsArrays {               // device pointers, by value to kernels
    LbfgsState *st;
    double *ro, *al, *cy, *cs;     // [M1] per slot
    double *dots;                  // [M1][5]: Y.g, S.g, Y.s_new, Y.y_new, S.y_new
    double *SY, *YY;               // [M1][M1] by slot: SY[i][j] = s_i.y_j, YY[i][j] = y_i.y_j
    double *part;                  // partial sums scratch
};

__device__ __forceinline__ double block_sum_all(double v, double *scratch) {   // result in every thread
    const double r = block_sum(v, scratch);
    __shared__ double bc;
    if (threadIdx.x == 0) bc = r;
    __syncthreads();
    const double out = bc;
    __syncthreads();
    return out;
}

// ---- statistics of a gradient: max|g|, sum|g|, g.g  (partials per block: [nb][3])
template <typename T>
__global__ __launch_bounds__(kLb) void gstats_kernel(const T *__restrict__ g, int64_t n, double *__restrict__ part) {
    __shared__ double red[kLb / 64];
    double mx = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kLb + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLb) {
        const double v = (double)g[i], a = fabs(v);
        mx = a > mx || a != a ? a : mx;          // NaN propagates
        s1 += a;
        s2 += v * v;
    }
    // max via sum trick is not possible: dedicated shuffle max
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(mx, off);
        mx = (o > mx || o != o) ? o : mx;
    }
    __shared__ double mred[kLb / 64];
    if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = mx;
    const double t1 = block_sum(s1, red);
    __syncthreads();
    const double t2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        double m = mred[0];
        for (int w = 1; w < kLb / 64; ++w) m = (mred[w] > m || mred[w] != mred[w]) ? mred[w] : m;
        part[3 * blockIdx.x] = m;
        part[3 * blockIdx.x + 1] = t1;
        part[3 * blockIdx.x + 2] = t2;
    }
}

// finalize the statistics and the break tests that follow a closure evaluation (torch lbfgs.py: opt_cond,
// "lack of progress" tests); `loss` is a device scalar (may be null at the first evaluation of a step)
__global__ __launch_bounds__(kLb) void check_kernel(LbfgsArrays A, int nb, const double *__restrict__ loss, int have_prev,
                                                    double tol_grad, double tol_change, double *__restrict__ status) {
    __shared__ double red[kLb / 64];
    double mx = 0.0, s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nb; b += kLb) {
        const double m = A.part[3 * b];
        mx = (m > mx || m != m) ? m : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(mx, off);
        mx = (o > mx || o != o) ? o : mx;
    }
    __shared__ double mred[kLb / 64];
    if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = mx;
    for (int b = threadIdx.x; b < nb; b += kLb) { s1 += A.part[3 * b + 1]; s2 += A.part[3 * b + 2]; }
    const double t1 = block_sum(s1, red);
    __syncthreads();
    const double t2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        double m = mred[0];
        for (int w = 1; w < kLb / 64; ++w) m = (mred[w] > m || mred[w] != mred[w]) ? mred[w] : m;
        LbfgsState &S = *A.st;
        S.g_absmax = m; S.g_abssum = t1; S.gg = t2;
        if (loss) S.loss = loss[0];
        int f = 0;
        if (m <= tol_grad) f |= 1;
        if (have_prev) {
            if (S.d_absmax * fabs(S.t) <= tol_change) f |= 2;
            if (fabs(S.loss - S.prev_loss) < tol_change) f |= 4;
            if (S.stop_gtd) f |= 8;
        }
        S.flags = (double)f;
        status[0] = S.loss; status[1] = (double)f; status[2] = m; status[3] = S.gtd; status[4] = S.t;
        status[5] = (double)S.count; status[6] = (double)S.n_iter; status[7] = S.H_diag;
    }
}

// ---- y = g - g_prev, s = t d into the spare slot; g_prev = g; partials of y.s and y.y: [nb][2]
template <typename T>
__global__ __launch_bounds__(kLb) void pair_kernel(LbfgsArrays A, const T *__restrict__ g, T *__restrict__ g_prev,
                                                   const T *__restrict__ d, T *__restrict__ Sring, T *__restrict__ Yring,
                                                   int64_t n, int M1, int first) {
    __shared__ double red[kLb / 64];
    const LbfgsState &S = *A.st;
    const int spare = (S.head + S.count) % M1;
    const double t = S.t;
    T *ys_ = Yring + (int64_t)spare * n, *ss_ = Sring + (int64_t)spare * n;
    double a = 0.0, b = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kLb + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLb) {
        const T gi = g[i];
        if (!first) {
            const T y = gi - g_prev[i];                     // flat_grad.sub(prev_flat_grad)
            const T s = (T)((double)d[i] * t);              // d.mul(t)
            ys_[i] = y; ss_[i] = s;
            a += (double)y * (double)s;
            b += (double)y * (double)y;
        }
        g_prev[i] = gi;
    }
    const double ta = block_sum(a, red);
    __syncthreads();
    const double tb = block_sum(b, red);
    if (threadIdx.x == 0) { A.part[2 * blockIdx.x] = ta; A.part[2 * blockIdx.x + 1] = tb; }
}

// accept / reject the pair, advance the ring, n_iter, prev_loss (one block)
__global__ __launch_bounds__(kLb) void pair_reduce_kernel(LbfgsArrays A, int nb, int M1, int first) {
    __shared__ double red[kLb / 64];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nb; k += kLb) { a += A.part[2 * k]; b += A.part[2 * k + 1]; }
    const double ys = block_sum(a, red);
    __syncthreads();
    const double yy = block_sum(b, red);
    if (threadIdx.x == 0) {
        LbfgsState &S = *A.st;
        S.n_iter += 1;
        S.new_slot = -1;
        if (first) { S.count = 0; S.head = 0; S.H_diag = 1.0; }
        else {
            S.ys = ys; S.yy = yy;
            if (ys > 1e-10) {
                const int m = M1 - 1, slot = (S.head + S.count) % M1;
                if (S.count == m) S.head = (S.head + 1) % M1;    // drop the oldest: its slot becomes the spare
                else S.count += 1;
                A.ro[slot] = 1.0 / ys;
                S.H_diag = ys / yy;
                S.new_slot = slot;
            }
        }
        S.prev_loss = S.loss;
    }
}

// ---- one pass over the history: block b owns chunk b (kLbChunk elements); per active slot five dots.
// The pass is latency-bound unless loads stay in flight across the per-slot reduction (five wave sums, a barrier, a store):
// the next slot's 2 x kLbPer values are requested BEFORE the current slot is reduced (two register sets, the slot loop unrolled
// by two; the partials buffer in LDS alternates so that one barrier per slot is enough).  10^6-element mesh (2 x 10^6 fp64
// parameters), 100 pairs, 3.2 GB per pass: 936 -> 598 us (5.35 TB/s); 4 or 2 elements per thread are slower (1.39 / 1.86 ms per
// L-BFGS iteration against 1.35).
// VEC consecutive elements per load (fp32 histories: 2 -> 8-byte loads like the fp64 ones, needs an even n); a thread owns
// PER loads = PER * VEC elements of the chunk.
// HFEM_LBFGS_VARIANT (build.py --tag/--define, A/B timing): 2 = non-temporal loads of the history in the multidot pass, 8 = 16-byte
// loads in the fp64 multidot pass.  Measured on T1M (2 x 10^6 fp64 parameters, 100 pairs, three alternating runs each,
// profiles/r04/lbfgs_load_variants.jsonl): 0 -> 1.335 ms per inner iteration, 2 -> 1.255 (the history is read once per pass and
// is 12 x the Infinity Cache: keeping it out of the caches leaves them to g, the new pair and the partials), 8 -> 1.32.  The same
// two changes in the direction pass (bits 1 and 4 of that experiment) gained nothing in fp64 -- 1.265 / 1.35 -- and its
// unguarded-load form ran the fp32 pass at HALF speed (760 us against ~300: profiles/r04/lbfgs_fp32_direction_regression.csv), so
// the direction pass keeps its round-3 form.  Default: 2.  scripts/micro/history_stream_bench.hip: the access pattern itself
// allows 5.5-6.0 TB/s in every shape tried (block 64...1024, 8 / 16-byte loads, 1...8 loads per thread, nt or not).
#ifndef HFEM_LBFGS_VARIANT
#define HFEM_LBFGS_VARIANT 2
#endif
template <typename T, int VEC, bool NT = false>
__device__ __forceinline__ void lb_load(const T *__restrict__ p, int64_t i, int64_t n, T *out) {
    if constexpr (VEC == 2) {
        if (i < n) {                                           // n even, i even: the pair is inside and 2 sizeof(T)-aligned
            typedef T pair_t __attribute__((ext_vector_type(2)));
            const pair_t *q = reinterpret_cast<const pair_t *>(p + i);
            const pair_t v = NT ? __builtin_nontemporal_load(q) : *q;
            out[0] = v.x; out[1] = v.y;
        } else { out[0] = (T)0; out[1] = (T)0; }
    } else {
        out[0] = i < n ? (NT ? __builtin_nontemporal_load(p + i) : p[i]) : (T)0;
    }
}

template <typename T, int PER, int VEC = 1>
__global__ __launch_bounds__(kLb) void multidot_kernel(LbfgsArrays A, const T *__restrict__ g, const T *__restrict__ Sring,
                                                       const T *__restrict__ Yring, int64_t n, int M1, int spec = 0) {
    constexpr int E = PER * VEC;                            // elements per thread
    __shared__ double red[2][5][kLb / 64];
    const LbfgsState &S = *A.st;
    // spec (sharded flow): the pair sits in the ring's SPARE slot, not yet accepted: dots with it, and its own row of dots
    const int head = S.head, ns = spec ? (spec > 1 ? -1 : (S.head + S.count) % M1) : S.new_slot;
    const int count = S.count + (spec == 1 ? 1 : 0);
    const int64_t base = (int64_t)blockIdx.x * (kLb * E) + (int64_t)threadIdx.x * VEC;
    T gv[E], sv[E], yv[E];                                  // in the vectors' own type (fp32 histories: half the registers)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int64_t i = base + (int64_t)k * kLb * VEC;
        lb_load<T, VEC>(g, i, n, gv + k * VEC);
        lb_load<T, VEC>(Sring + (int64_t)(ns >= 0 ? ns : 0) * n, ns >= 0 ? i : n, n, sv + k * VEC);
        lb_load<T, VEC>(Yring + (int64_t)(ns >= 0 ? ns : 0) * n, ns >= 0 ? i : n, n, yv + k * VEC);
    }
    auto load = [&](int l, T (&y)[E], T (&s)[E]) {
        const int slot = (head + l) % M1;
        const T *Yj = Yring + (int64_t)slot * n, *Sj = Sring + (int64_t)slot * n;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t i = base + (int64_t)k * kLb * VEC;
            lb_load<T, VEC, (HFEM_LBFGS_VARIANT & 2) != 0>(Yj, i, n, y + k * VEC);
            lb_load<T, VEC, (HFEM_LBFGS_VARIANT & 2) != 0>(Sj, i, n, s + k * VEC);
        }
    };
    auto reduce_store = [&](int l, const T (&yr)[E], const T (&sr)[E], int buf) {
        double acc[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < E; ++k) {                        // out-of-range lanes hold zeros: same sums as a guarded loop
            const double y = (double)yr[k], s = (double)sr[k];
            const double gk = (double)gv[k], sk = (double)sv[k], yk = (double)yv[k];
            acc[0] += y * gk; acc[1] += s * gk;
            acc[2] += y * sk; acc[3] += y * yk; acc[4] += s * yk;
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const double w = wave_sum(acc[q]);
            if ((threadIdx.x & 63) == 0) red[buf][q][threadIdx.x >> 6] = w;
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            double r = 0.0;
            for (int w = 0; w < kLb / 64; ++w) r += red[buf][threadIdx.x][w];
            A.part[((int64_t)blockIdx.x * M1 + (head + l) % M1) * 5 + threadIdx.x] = r;
        }
    };
    // gridDim.y workgroups share a chunk and take the slots l = blockIdx.y, + gridDim.y, ...: short vectors (few
    // chunks) still fill the chip; each (chunk, slot) partial is produced by exactly one workgroup, in the same order
    const int gy = gridDim.y;
    T ya[E], sa[E], yb[E], sb[E];
    int l = blockIdx.y;
    if (l < count) load(l, ya, sa);
    for (; l < count; l += 2 * gy) {
        const int l2 = l + gy;
        if (l2 < count) load(l2, yb, sb);
        reduce_store(l, ya, sa, 0);
        if (l2 < count) {
            if (l2 + gy < count) load(l2 + gy, ya, sa);
            reduce_store(l2, yb, sb, 1);
        }
    }
}

// sum the per-chunk partials: grid = count blocks (logical slot l each)
__global__ __launch_bounds__(kLb) void multidot_reduce_kernel(LbfgsArrays A, int nb, int M1, double *__restrict__ out = nullptr) {
    __shared__ double red[kLb / 64];
    const LbfgsState &S = *A.st;
    if ((int)blockIdx.x >= S.count + (out ? 1 : 0)) return;       // out != NULL: the sharded flow's payload (spare slot included)
    const int slot = (S.head + blockIdx.x) % M1;
    double *dst = out ? out : A.dots;
    for (int q = 0; q < 5; ++q) {
        double v = 0.0;
        for (int b = threadIdx.x; b < nb; b += kLb) v += A.part[((int64_t)b * M1 + slot) * 5 + q];
        const double r = block_sum(v, red);
        if (threadIdx.x == 0) dst[slot * 5 + q] = r;
        __syncthreads();
    }
}

// ---- the two loops in coefficient space (one block); writes cy, cs, cg, t, gtd, stop
__global__ __launch_bounds__(kLb) void recursion_kernel(LbfgsArrays A, int M1, double lr, double tol_change) {
    __shared__ double red[kLb / 64];
    LbfgsState &S = *A.st;
    if (S.skip || S.halt) return;
    const int count = S.count, head = S.head, ns = S.new_slot, tid = threadIdx.x;
    const int nthr = blockDim.x;         // launched with ONE wave: the 2 x count sequential steps then sync at wave cost
    // Gram row / column of the new pair
    if (ns >= 0) {
        for (int l = tid; l < count; l += nthr) {
            const int j = (head + l) % M1;
            A.SY[ns * M1 + j] = A.dots[j * 5 + 2];        // s_new . y_j
            A.SY[j * M1 + ns] = A.dots[j * 5 + 4];        // s_j . y_new
            A.YY[ns * M1 + j] = A.dots[j * 5 + 3];        // y_new . y_j
            A.YY[j * M1 + ns] = A.dots[j * 5 + 3];
        }
    }
    __syncthreads();
    const double H = S.H_diag;
    // first loop, newest to oldest: al_i = ro_i ( -s_i.g - sum_{j newer} al_j s_i.y_j )
    for (int l = count - 1; l >= 0; --l) {
        const int i = (head + l) % M1;
        double v = 0.0;
        for (int l2 = l + 1 + tid; l2 < count; l2 += nthr) {
            const int j = (head + l2) % M1;
            v += A.al[j] * A.SY[i * M1 + j];
        }
        const double sum = block_sum_all(v, red);
        if (tid == 0) A.al[i] = A.ro[i] * (-A.dots[i * 5 + 1] - sum);
        __syncthreads();
    }
    // second loop, oldest to newest: be_i = ro_i ( H(-y_i.g - sum_j al_j y_i.y_j) + sum_{j older} c_j s_j.y_i )
    for (int l = 0; l < count; ++l) {
        const int i = (head + l) % M1;
        double v = 0.0, w = 0.0;
        for (int l2 = tid; l2 < count; l2 += nthr) {
            const int j = (head + l2) % M1;
            v += A.al[j] * A.YY[i * M1 + j];
            if (l2 < l) w += A.cs[j] * A.SY[j * M1 + i];
        }
        const double sv = block_sum_all(v, red);
        const double sw = block_sum_all(w, red);
        if (tid == 0) {
            const double be = A.ro[i] * (H * (-A.dots[i * 5] - sv) + sw);
            A.cs[i] = A.al[i] - be;
        }
        __syncthreads();
    }
    // g.d = -H g.g + sum_j (-H al_j) y_j.g + sum_j c_j s_j.g
    double v = 0.0;
    for (int l = tid; l < count; l += nthr) {
        const int j = (head + l) % M1;
        const double cyj = -H * A.al[j];
        A.cy[j] = cyj;
        v += cyj * A.dots[j * 5] + A.cs[j] * A.dots[j * 5 + 1];
    }
    const double sum = block_sum_all(v, red);
    if (tid == 0) {
        S.cg = -H;
        S.gtd = -H * S.gg + sum;
        S.t = S.n_iter == 1 ? fmin(1.0, 1.0 / S.g_abssum) * lr : lr;
        S.stop_gtd = S.gtd > -tol_change ? 1 : 0;
    }
}

// ---- the same two loops for histories up to 256 pairs, without a reduction on the critical path.
// The reduction form above pays a global-load round trip plus two block reductions per sequential step (178 us at
// 100 pairs -- more than the two passes over the history on a 35 k-element mesh).  Here every pending sum is kept as a
// running vector in LDS: after al_i is known, every older row k takes t_k += al_i s_k.y_i; after c_i is known, every
// newer row k takes w_k += c_i s_i.y_k (rank-1 updates); a sequential step is then one LDS read, a few flops and a
// broadcast, and the Gram entries it needs are fetched four steps ahead (their addresses do not depend on the values).
// Logical index l = 0 (oldest) .. count-1; thread k owns rows k and k + 128.
// Thread k owns row k and keeps its running sums t_k, w_k in REGISTERS; the owner of row i computes al_i (c_i) itself and
// publishes it through LDS: one barrier per sequential step (the first version kept the sums in LDS for thread 0 to read
// and paid two: 105 us at 100 pairs; this one 52 us; a single wavefront with four rows per lane 170 us -- the step is
// bound by the LDS round trips per row, not by the barrier).
constexpr int kRecT = 256, kRecMax = 256;
__global__ __launch_bounds__(kRecT) void recursion_rank1_kernel(LbfgsArrays A, int M1, double lr, double tol_change) {
    __shared__ double al_[kRecMax], c_[kRecMax];
    __shared__ int slot_[kRecMax];
    __shared__ double red[kRecT / 64];
    LbfgsState &S = *A.st;
    if (S.skip || S.halt) return;
    const int count = S.count, head = S.head, ns = S.new_slot, k = threadIdx.x;
    if (ns >= 0 && k < count) {                           // Gram row / column of the new pair
        const int j = (head + k) % M1;
        A.SY[ns * M1 + j] = A.dots[j * 5 + 2];
        A.SY[j * M1 + ns] = A.dots[j * 5 + 4];
        A.YY[ns * M1 + j] = A.dots[j * 5 + 3];
        A.YY[j * M1 + ns] = A.dots[j * 5 + 3];
    }
    const bool live = k < count;
    const int sk = live ? (head + k) % M1 : 0;
    const double ro = live ? A.ro[sk] : 0.0, sg = live ? A.dots[sk * 5 + 1] : 0.0, yg = live ? A.dots[sk * 5] : 0.0;
    slot_[k] = sk;
    __syncthreads();                                      // also orders the Gram writes above before the reads below (one block)
    __threadfence_block();
    const double H = S.H_diag;
    double t = 0.0, w = 0.0, al_k = 0.0, c_k = 0.0;
    // ---- first loop, i = count-1 .. 0, four steps per prefetch group: column i of S^T Y
    {
        double a[4], b[4];
        auto fetch = [&](int i_hi, double (&x)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i_hi - q;
                x[q] = (i >= 0 && k < i) ? A.SY[sk * M1 + slot_[i]] : 0.0;
            }
        };
        fetch(count - 1, a);
        for (int ih = count - 1; ih >= 0; ih -= 4) {
            fetch(ih - 4, b);                             // next group's entries land under this group's steps
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = ih - q;
                if (i >= 0) {                             // uniform
                    if (k == i) { al_k = ro * (-sg - t); al_[i] = al_k; }
                    __syncthreads();
                    if (k < i) t += al_[i] * a[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = b[q];
        }
    }
    __syncthreads();
    // ---- v_k = sum_j al_j y_k.y_j : every row streams its own Gram row (no dependence between rows)
    double v = 0.0;
    if (live) {
#pragma unroll 8
        for (int j = 0; j < count; ++j) v += al_[j] * A.YY[sk * M1 + slot_[j]];
    }
    // ---- second loop, i = 0 .. count-1: row i of S^T Y
    {
        double a[4], b[4];
        auto fetch = [&](int i_lo, double (&x)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i_lo + q;
                x[q] = (i < count && k > i && live) ? A.SY[slot_[i] * M1 + sk] : 0.0;
            }
        };
        fetch(0, a);
        for (int il = 0; il < count; il += 4) {
            fetch(il + 4, b);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = il + q;
                if (i < count) {                          // uniform
                    if (k == i) { c_k = al_k - ro * (H * (-yg - v) + w); c_[i] = c_k; }
                    __syncthreads();
                    if (k > i && live) w += c_[i] * a[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = b[q];
        }
    }
    // ---- coefficients by slot, g.d, step, break flag
    double part = 0.0;
    if (live) {
        const double cyj = -H * al_k;
        A.al[sk] = al_k; A.cy[sk] = cyj; A.cs[sk] = c_k;
        part = cyj * yg + c_k * sg;
    }
    const double sum = block_sum(part, red);
    if (k == 0) {
        S.cg = -H;
        S.gtd = -H * S.gg + sum;
        S.t = S.n_iter == 1 ? fmin(1.0, 1.0 / S.g_abssum) * lr : lr;
        S.stop_gtd = S.gtd > -tol_change ? 1 : 0;
    }
}

// ---- the same recursion with the two sequential loops on ONE wavefront (round 4).  The form above pays a barrier and an LDS
// round trip per sequential step (2 x count steps: 52 us at 100 pairs -- a fifth of a sharded iteration, and it does not shrink
// with the number of ranks).  Here the 256 threads first stage S^T Y (logical order, odd row stride: both the column reads of
// the first loop and the row reads of the second are bank-conflict-free) into LDS -- a workgroup may take all 160 KB, 128 pairs
// need 129 KB --, then wave 0 alone walks the two loops: lane l owns rows l, l + 64 (R per lane), the running sums stay in
// registers, the value a step publishes travels by v_readlane (the publishing lane is wave-uniform) -- no barrier, no memory
// round trip on the critical path; the Gram entries of the next P steps are already in registers.  The row-parallel middle
// part (v_k) runs on all 256 threads as before.  Same operations in the same order per row, and the final g.d sum adds the
// per-64-row wave sums in block_sum's order: BIT-IDENTICAL coefficients, g.d and step to recursion_rank1_kernel
// (scripts/lbfgs_recursion_bits.py: same parameter bits after 160 iterations for histories 100 and 7).  Measured at 100 pairs
// (s_memrealtime stamps, -DHFEM_REC_DEBUG): first loop 7.6-12 us, v_k 2.5 us (its Gram column is requested into registers
// before the first loop), second loop + coefficients 12 us; 52 -> ~28 us for the launch, a sharded inner iteration of T1M / 8
// 0.268-0.272 -> 0.250-0.259 ms.  What did NOT work: one wave with the Gram entries prefetched from global memory 8 steps
// ahead (130 us: one wave cannot cover the load latency); publishing al_i through LDS inside the loop or guarding every
// prefetched element (a wait / a branch per step: 15 us per loop); splitting H (..) + w_k out of the second loop's chain
// (the product no longer contracts into the fused multiply-add of the reference form: different bits).
__device__ __forceinline__ double readlane_f64(double v, int uniform_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), uniform_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), uniform_lane);
    return __hiloint2double(hi, lo);
}

constexpr int kRecWaveMax = 128;                          // pairs: (128 | 1) * 128 * 8 B = 129 KB of LDS for S^T Y
template <int R>
__global__ __launch_bounds__(kRecT) void recursion_wave_kernel(LbfgsArrays A, int M1, double lr, double tol_change, int stride) {
    static_assert(R == 1 || R == 2, "rows per lane: 64 or 128 pairs");
    constexpr int P = 8;                                  // steps whose Gram entries are held in registers ahead of time
    extern __shared__ double SYl[];                       // [count][stride]: SYl[i * stride + k] = s_i . y_k, logical indices
    __shared__ double al_[kRecMax], v_[kRecMax];
    LbfgsState &S = *A.st;
    if (S.skip || S.halt) return;
    const int count = S.count, head = S.head, ns = S.new_slot, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto slot_of = [&](int l) { const int x = head + l; return x >= M1 ? x - M1 : x; };      // l < M1, head < M1: no division
    if (ns >= 0 && tid < count) {                         // Gram row / column of the new pair
        const int j = slot_of(tid);
        A.SY[ns * M1 + j] = A.dots[j * 5 + 2];
        A.SY[j * M1 + ns] = A.dots[j * 5 + 4];
        A.YY[ns * M1 + j] = A.dots[j * 5 + 3];
        A.YY[j * M1 + ns] = A.dots[j * 5 + 3];
    }
    __syncthreads();                                      // orders the Gram writes above before the reads below (one block)
    __threadfence_block();
    // S^T Y into LDS: wave w takes rows w, w + 4, ...; lanes take the columns; four rows' loads in flight per lane
    for (int i0 = wave; i0 < count; i0 += 16) {
        double x[4][R];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 4 * u, si = slot_of(min(i, count - 1));
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int k = lane + 64 * j;
                x[u][j] = (i < count && k < count) ? A.SY[si * M1 + slot_of(k)] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 4 * u;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int k = lane + 64 * j;
                if (i < count && k < count) SYl[i * stride + k] = x[u][j];
            }
        }
    }
    const double H = S.H_diag;
    bool live[R];
    int sk[R], rowc[R];
    double ro[R], sg[R], yg[R], al_k[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int k = lane + 64 * j;
        live[j] = k < count;
        sk[j] = live[j] ? slot_of(k) : 0;
        ro[j] = live[j] ? A.ro[sk[j]] : 0.0;
        sg[j] = live[j] ? A.dots[sk[j] * 5 + 1] : 0.0;
        yg[j] = live[j] ? A.dots[sk[j] * 5] : 0.0;
        al_k[j] = 0.0;
        rowc[j] = min(k, count > 0 ? count - 1 : 0);       // a staged row for every lane (rows past count read row count - 1, masked)
    }
    __syncthreads();                                      // S^T Y staged
#ifdef HFEM_REC_DEBUG
    const unsigned long long T0 = __builtin_amdgcn_s_memrealtime();
#endif
    // column tid of Y^T Y (stored symmetric: = row tid, but the 64 lanes of a load read consecutive addresses) requested NOW (after the barrier: nothing in the first loop waits for memory), into
    // registers: the loads land under the first loop, and v_k below is a chain of fused multiply-adds without a memory round trip
    double yyc[64 * R];
    {
        const int skk = slot_of(min(tid, count > 0 ? count - 1 : 0));
#pragma unroll
        for (int j = 0; j < 64 * R; ++j) yyc[j] = j < count ? A.YY[slot_of(j) * M1 + skk] : 0.0;
    }
    double a[R][P], b[R][P];
    // ---- first loop (wave 0), i = count-1 .. 0: column i of S^T Y.  Row k takes t_k += al_i s_k.y_i for every i > k and is
    //      final from step k on, so al_k = ro_k (-s_k.g - t_k) is what step k publishes AND what the row holds after the loop;
    //      entries a row must not take are zeroed when they are fetched (P steps ahead, off the chain): the step itself is
    //      one add, one multiply, two v_readlane and R fused multiply-adds
    if (wave == 0) {
        double t[R];
#pragma unroll
        for (int j = 0; j < R; ++j) t[j] = 0.0;
        auto fetch = [&](int i_hi, double (&x)[R][P]) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int i = i_hi - q, ic = max(i, 0);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const double e = SYl[rowc[j] * stride + ic];          // unguarded LDS read
                    x[j][q] = (lane + 64 * j < i) ? e : 0.0;
                }
            }
        };
        fetch(count - 1, a);
        for (int ih = count - 1; ih >= 0; ih -= P) {
            fetch(ih - P, b);                             // the next group's entries, read under this group's steps
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int i = ih - q;
                if (i >= 0) {                             // uniform
                    double cand;
                    if (R == 1 || i < 64) cand = ro[0] * (-sg[0] - t[0]);           // uniform: the publishing row's register
                    else cand = ro[R - 1] * (-sg[R - 1] - t[R - 1]);
                    const double al_i = readlane_f64(cand, i & 63);
#pragma unroll
                    for (int j = 0; j < R; ++j) t[j] += al_i * a[j][q];
                }
            }
#pragma unroll
            for (int q = 0; q < P; ++q)
#pragma unroll
                for (int j = 0; j < R; ++j) a[j][q] = b[j][q];
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            al_k[j] = ro[j] * (-sg[j] - t[j]);
            if (live[j]) al_[lane + 64 * j] = al_k[j];
        }
    }
    __syncthreads();                                      // al_[] complete
#ifdef HFEM_REC_DEBUG
    const unsigned long long T1 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- v_k = sum_j al_j y_k.y_j (all threads, thread k = row k), from the registers filled before the first loop
    if (tid < count) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 64 * R; ++j)
            if (j < count) v += al_[j] * yyc[j];          // uniform guard; the sum in recursion_rank1_kernel's order
        v_[tid] = v;
    }
    __syncthreads();
#ifdef HFEM_REC_DEBUG
    const unsigned long long T2 = __builtin_amdgcn_s_memrealtime();
#endif
    if (wave != 0) return;
    // ---- second loop (wave 0), i = 0 .. count-1: row i of S^T Y.  Row k takes w_k += c_i s_i.y_k for every i < k and is
    //      final from step k on: c_k = al_k - ro_k (H (-y_k.g - v_k) + w_k)
    double w[R], c_k[R], v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { w[j] = 0.0; c_k[j] = 0.0; v[j] = live[j] ? v_[lane + 64 * j] : 0.0; }
    {
        auto fetch = [&](int i_lo, double (&x)[R][P]) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int i = i_lo + q, ic = max(min(i, count - 1), 0);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const double e = SYl[ic * stride + rowc[j]];
                    x[j][q] = (i < count && lane + 64 * j > i && live[j]) ? e : 0.0;
                }
            }
        };
        fetch(0, a);
        for (int il = 0; il < count; il += P) {
            fetch(il + P, b);
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int i = il + q;
                if (i < count) {                          // uniform
                    double cand;
                    // (the expression as recursion_rank1_kernel writes it: H * (...) + w contracts to one fused multiply-add)
                    if (R == 1 || i < 64) cand = al_k[0] - ro[0] * (H * (-yg[0] - v[0]) + w[0]);
                    else cand = al_k[R - 1] - ro[R - 1] * (H * (-yg[R - 1] - v[R - 1]) + w[R - 1]);
                    const double c_i = readlane_f64(cand, i & 63);
#pragma unroll
                    for (int j = 0; j < R; ++j) w[j] += c_i * a[j][q];
                }
            }
#pragma unroll
            for (int q = 0; q < P; ++q)
#pragma unroll
                for (int j = 0; j < R; ++j) a[j][q] = b[j][q];
        }
    }
    // ---- coefficients by slot, g.d (the per-64-row sums added in block_sum's wave order), step, break flag
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        double part = 0.0;
        if (live[j]) {
            c_k[j] = al_k[j] - ro[j] * (H * (-yg[j] - v[j]) + w[j]);
            const double cyj = -H * al_k[j];
            A.al[sk[j]] = al_k[j]; A.cy[sk[j]] = cyj; A.cs[sk[j]] = c_k[j];
            part = cyj * yg[j] + c_k[j] * sg[j];
        }
        sum += wave_sum(part);                            // valid in lane 0
    }
    if (lane == 0) {
        S.cg = -H;
        S.gtd = -H * S.gg + sum;
        S.t = S.n_iter == 1 ? fmin(1.0, 1.0 / S.g_abssum) * lr : lr;
        S.stop_gtd = S.gtd > -tol_change ? 1 : 0;
#ifdef HFEM_REC_DEBUG
        const unsigned long long T3 = __builtin_amdgcn_s_memrealtime();
        if (count >= 100 && S.n_iter % 20 == 0)
            printf("rec: count %d  loop1 %.2f us  v %.2f us  loop2+tail %.2f us\n", count, (T1 - T0) * 0.01, (T2 - T1) * 0.01, (T3 - T2) * 0.01);
#endif
    }
}

// ---- d = cg g + sum_j cy_j Y_j + cs_j S_j (one pass over the history), max|d| partials [nb]
// PER elements per thread: 8 for long vectors (fewer, fatter workgroups), 1 for short ones (more workgroups); the slot
// loop is unrolled so that several slots' loads are in flight (it is latency-bound otherwise).  Same sums either way.
// g_prev != NULL (sharded flow): prev_flat_grad = g is taken HERE -- i.e. only when a direction is really computed, as torch
// does (a break after the closure leaves prev_flat_grad alone) -- instead of in the pair pass.
template <typename T, int PER>
__global__ __launch_bounds__(kLb) void direction_kernel(LbfgsArrays A, const T *__restrict__ g, const T *__restrict__ Sring,
                                                        const T *__restrict__ Yring, T *__restrict__ d, int64_t n, int M1,
                                                        T *__restrict__ g_prev = nullptr) {
    const LbfgsState &S = *A.st;
    if (S.skip || S.halt) return;
    const int count = S.count, head = S.head;
    const int64_t base = (int64_t)blockIdx.x * (kLb * PER) + threadIdx.x;
    const double cg = S.cg;
    double acc[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int64_t i = base + (int64_t)k * kLb;
        const T gi = i < n ? g[i] : (T)0;
        acc[k] = cg * (double)gi;
        if (g_prev && i < n) g_prev[i] = gi;
    }
#pragma unroll 4
    for (int l = 0; l < count; ++l) {
        const int slot = (head + l) % M1;
        const double cy = A.cy[slot], cs = A.cs[slot];
        const T *Yj = Yring + (int64_t)slot * n, *Sj = Sring + (int64_t)slot * n;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t i = base + (int64_t)k * kLb;
            if (i < n) acc[k] += cy * (double)Yj[i] + cs * (double)Sj[i];
        }
    }
    double mx = 0.0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int64_t i = base + (int64_t)k * kLb;
        if (i < n) {
            const T v = (T)acc[k];
            d[i] = v;
            const double a = fabs((double)v);
            mx = (a > mx || a != a) ? a : mx;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(mx, off);
        mx = (o > mx || o != o) ? o : mx;
    }
    __shared__ double mred[kLb / 64];
    if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = mred[0];
        for (int w = 1; w < kLb / 64; ++w) m = (mred[w] > m || mred[w] != mred[w]) ? mred[w] : m;
        A.part[blockIdx.x] = m;
    }
}

// status != NULL (sharded flow): the status record is written HERE, after the recursion -- g.d, t and the g.d break flag
// (bit 3) are those of the direction just computed; bit 4: no direction was computed (break tests / the host's wish)
__global__ __launch_bounds__(kLb) void dmax_reduce_kernel(LbfgsArrays A, int nb, double *__restrict__ status = nullptr) {
    if (A.st->halt) return;                               // the status record stays that of the iteration that ended the step
    if (A.st->skip) {
        if (status && threadIdx.x == 0) {
            const LbfgsState &S = *A.st;
            status[0] = S.loss; status[1] = S.flags + 16.0; status[2] = S.g_absmax; status[3] = S.gtd; status[4] = S.t;
            status[5] = (double)S.count; status[6] = (double)S.n_iter; status[7] = S.H_diag;
        }
        return;
    }
    double mx = 0.0;
    for (int b = threadIdx.x; b < nb; b += kLb) {
        const double m = A.part[b];
        mx = (m > mx || m != m) ? m : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(mx, off);
        mx = (o > mx || o != o) ? o : mx;
    }
    __shared__ double mred[kLb / 64];
    if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = mred[0];
        for (int w = 1; w < kLb / 64; ++w) m = (mred[w] > m || mred[w] != mred[w]) ? mred[w] : m;
        A.st->d_absmax = m;
        if (status) {
            const LbfgsState &S = *A.st;
            status[0] = S.loss; status[1] = S.flags + (S.stop_gtd ? 8.0 : 0.0); status[2] = S.g_absmax; status[3] = S.gtd;
            status[4] = S.t; status[5] = (double)S.count; status[6] = (double)S.n_iter; status[7] = S.H_diag;
        }
    }
}

// =====================================================================================================================
// NODE-SHARDED L-BFGS (hfem_lbfgs_shard_*; hidenn_fem_amd/optim.py ShardedLBFGS): every rank keeps the history, the
// gradient and the direction of the parameter rows ITS tiles own (n = local length); what crosses ranks per inner
// iteration is ONE small payload per rank -- the per-slot dots, y.s, y.y, the gradient statistics, max|d| and the rank's
// partial energy -- gathered to every rank and summed there in RANK ORDER, so that all ranks take bit-identical decisions
// (accept / reject the pair, every break test) without another message.  To get there in one exchange the pair and its dots
// are computed SPECULATIVELY: y = g - g_prev, s = t d go into the ring's spare slot and the one pass over the history also
// takes the dots with that slot; the one-block finish kernel then applies torch's break tests, and only if none fires (and the
// host wants a direction) accepts the pair, advances the ring and lets the recursion and the direction pass run (S.skip).
// Payload (doubles): [5 x M1 dots by slot | y.s, y.y, max|g|, sum|g|, g.g, max|d| of the previous direction, the rank's
// energy, 0].
constexpr int kShardTail = 8;

// y, s into the spare slot (speculative) + five partial sums per block: y.s, y.y, max|g|, sum|g|, g.g
template <typename T>
__global__ __launch_bounds__(kLb) void shard_pair_kernel(LbfgsArrays A, const T *__restrict__ g, const T *__restrict__ g_prev,
                                                         const T *__restrict__ d, T *__restrict__ Sring, T *__restrict__ Yring,
                                                         int64_t n, int M1, int first) {
    __shared__ double red[kLb / 64];
    const LbfgsState &S = *A.st;
    const int spare = (S.head + S.count) % M1;
    const double t = S.t;
    T *ys_ = Yring + (int64_t)spare * n, *ss_ = Sring + (int64_t)spare * n;
    double a = 0.0, b = 0.0, mx = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kLb + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLb) {
        const T gi = g[i];
        const double v = (double)gi, av = fabs(v);
        mx = av > mx || av != av ? av : mx;
        s1 += av;
        s2 += v * v;
        if (!first) {
            const T y = gi - g_prev[i];
            const T sv = (T)((double)d[i] * t);
            ys_[i] = y; ss_[i] = sv;
            a += (double)y * (double)sv;
            b += (double)y * (double)y;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(mx, off);
        mx = (o > mx || o != o) ? o : mx;
    }
    __shared__ double mred[kLb / 64];
    if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = mx;
    const double ta = block_sum(a, red);
    __syncthreads();
    const double tb = block_sum(b, red);
    __syncthreads();
    const double t1 = block_sum(s1, red);
    __syncthreads();
    const double t2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        double m = mred[0];
        for (int w = 1; w < kLb / 64; ++w) m = (mred[w] > m || mred[w] != mred[w]) ? mred[w] : m;
        double *o = A.part + 5 * (int64_t)blockIdx.x;
        o[0] = ta; o[1] = tb; o[2] = m; o[3] = t1; o[4] = t2;
    }
}

// the block partials of shard_pair_kernel -> the payload's tail (+ the rank's energy and the max|d| of its last direction)
__global__ __launch_bounds__(kLb) void shard_tail_kernel(LbfgsArrays A, int nb, int M1, const double *__restrict__ loss_local,
                                                         double *__restrict__ payload) {
    __shared__ double red[kLb / 64];
    double mx = 0.0, v[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < nb; b += kLb) {
        const double *p = A.part + 5 * (int64_t)b;
        v[0] += p[0]; v[1] += p[1]; v[2] += p[3]; v[3] += p[4];
        mx = (p[2] > mx || p[2] != p[2]) ? p[2] : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(mx, off);
        mx = (o > mx || o != o) ? o : mx;
    }
    __shared__ double mred[kLb / 64];
    if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = mx;
    double r[4];
    for (int q = 0; q < 4; ++q) {
        r[q] = block_sum(v[q], red);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double m = mred[0];
        for (int w = 1; w < kLb / 64; ++w) m = (mred[w] > m || mred[w] != mred[w]) ? mred[w] : m;
        double *t = payload + 5 * M1;
        t[0] = r[0]; t[1] = r[1]; t[2] = m; t[3] = r[2]; t[4] = r[3];
        t[5] = A.st->d_absmax;                                   // of this rank's rows, from its last direction pass
        t[6] = loss_local ? loss_local[0] : 0.0;
        t[7] = 0.0;
    }
    for (int i = threadIdx.x; i < 5 * M1; i += kLb) payload[i] = 0.0;      // slots the multidot reduce does not visit stay 0
}

// One block: sum the ranks' payloads in rank order, torch's break tests, accept / reject the speculative pair.
__global__ __launch_bounds__(kLb) void shard_finish_kernel(LbfgsArrays A, const double *__restrict__ gathered, int world, int P,
                                                           int M1, int first, int after_update, int want_direction,
                                                           double tol_grad, double tol_change) {
    for (int i = threadIdx.x; i < 5 * M1; i += kLb) {
        double v = 0.0;
        for (int r = 0; r < world; ++r) v += gathered[(int64_t)r * P + i];
        A.dots[i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        LbfgsState &S = *A.st;
        // an iteration replayed AFTER the one that ended the step (graphs of several iterations, optim.ShardedLBFGS): the host
        // loop would have left before it -- nothing of it may count.  Its apply did nothing already (skip / stop_gtd).
        if (after_update && (S.skip || S.stop_gtd)) { S.halt = 1; return; }
        S.halt = 0;
        double ys = 0.0, yy = 0.0, gsum = 0.0, gg = 0.0, loss = 0.0, gmax = 0.0, dmax = 0.0;
        for (int r = 0; r < world; ++r) {
            const double *t = gathered + (int64_t)r * P + 5 * M1;
            ys += t[0]; yy += t[1]; gsum += t[3]; gg += t[4]; loss += t[6];
            gmax = (t[2] > gmax || t[2] != t[2]) ? t[2] : gmax;
            dmax = (t[5] > dmax || t[5] != t[5]) ? t[5] : dmax;
        }
        S.g_absmax = gmax; S.g_abssum = gsum; S.gg = gg; S.loss = loss; S.d_absmax = dmax;
        int f = 0;
        if (gmax <= tol_grad) f |= 1;
        if (after_update) {
            if (dmax * fabs(S.t) <= tol_change) f |= 2;
            if (fabs(S.loss - S.prev_loss) < tol_change) f |= 4;
        }
        S.flags = (double)f;
        S.skip = (f != 0 || !want_direction) ? 1 : 0;
        if (!S.skip) {                                           // pair_reduce_kernel's bookkeeping on the GLOBAL y.s, y.y
            S.n_iter += 1;
            S.new_slot = -1;
            if (first) { S.count = 0; S.head = 0; S.H_diag = 1.0; }
            else {
                S.ys = ys; S.yy = yy;
                if (ys > 1e-10) {
                    const int m = M1 - 1, slot = (S.head + S.count) % M1;
                    if (S.count == m) S.head = (S.head + 1) % M1;
                    else S.count += 1;
                    A.ro[slot] = 1.0 / ys;
                    S.H_diag = ys / yy;
                    S.new_slot = slot;
                }
            }
            S.prev_loss = S.loss;
        }
    }
}

// gather the rows this rank owns into its flat local vector: [x rows | u rows], two values per row
template <typename T>
__global__ __launch_bounds__(kLb) void shard_gather_kernel(const T *__restrict__ gx, const int32_t *__restrict__ rx, int64_t nx,
                                                           const T *__restrict__ gu, const int32_t *__restrict__ ru, int64_t nu,
                                                           T *__restrict__ out) {
    typedef T pair_t __attribute__((ext_vector_type(2)));
    for (int64_t i = (int64_t)blockIdx.x * kLb + threadIdx.x; i < nx + nu; i += (int64_t)gridDim.x * kLb) {
        const pair_t v = i < nx ? reinterpret_cast<const pair_t *>(gx)[rx[i]] : reinterpret_cast<const pair_t *>(gu)[ru[i - nx]];
        reinterpret_cast<pair_t *>(out)[i] = v;
    }
}

// p[row] += t d on the rows this rank owns (nothing if no direction was computed or it stopped on g.d)
template <typename T>
__global__ __launch_bounds__(kLb) void shard_apply_kernel(LbfgsArrays A, T *__restrict__ x, const int32_t *__restrict__ rx, int64_t nx,
                                                          T *__restrict__ u, const int32_t *__restrict__ ru, int64_t nu,
                                                          const T *__restrict__ d) {
    const LbfgsState &S = *A.st;
    if (S.stop_gtd || S.skip) return;
    const double t = S.t;
    for (int64_t i = (int64_t)blockIdx.x * kLb + threadIdx.x; i < nx + nu; i += (int64_t)gridDim.x * kLb) {
        T *p = i < nx ? x + 2 * (int64_t)rx[i] : u + 2 * (int64_t)ru[i - nx];
        p[0] = (T)((double)p[0] + t * (double)d[2 * i]);
        p[1] = (T)((double)p[1] + t * (double)d[2 * i + 1]);
    }
}

// ---- p += t d on one parameter segment, unless the iteration stopped on g.d
template <typename T>
__global__ __launch_bounds__(kLb) void apply_kernel(LbfgsArrays A, T *__restrict__ p, const T *__restrict__ d, int64_t n) {
    const LbfgsState &S = *A.st;
    if (S.stop_gtd) return;
    const double t = S.t;
    for (int64_t i = (int64_t)blockIdx.x * kLb + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLb)
        p[i] = (T)((double)p[i] + t * (double)d[i]);
}

}  // namespace hfem

using namespace hfem;

struct hfem_lbfgs {
    int device = -1, M1 = 0, dtype = 0, nb_stream = 0, nb_chunk = 0;
    int per = kLbPer, nb_md = 0;                    // multidot: elements per thread (8 measured best of 2 / 4 / 8), its chunk count
    int64_t n = 0;
    bool first = true;              // no pair yet (torch state["n_iter"] == 0)
    void *Sring = nullptr, *Yring = nullptr, *g_prev = nullptr, *d = nullptr;
    double *scal = nullptr, *status = nullptr;
    double *status_pinned = nullptr;  // host-pinned copy of the status record (hfem_lbfgs_shard_finish without a host pointer: capturable)
    LbfgsArrays A{};
};

template <typename T>
static int lb_malloc(T **p, size_t count) {
    HFEM_HIP_CHECK(hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
    HFEM_HIP_CHECK(hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(T)));
    return 0;
}

extern "C" int hfem_lbfgs_destroy(hfem_lbfgs *o) {
    if (!o) return 0;
    if (o->device >= 0) {
        (void)hipSetDevice(o->device);
        (void)hipFree(o->Sring); (void)hipFree(o->Yring); (void)hipFree(o->g_prev); (void)hipFree(o->d);
        (void)hipFree(o->scal); (void)hipFree(o->status); (void)hipFree(o->A.part); (void)hipFree(o->A.st);
        if (o->status_pinned) (void)hipHostFree(o->status_pinned);
    }
    delete o;
    return 0;
}

extern "C" int hfem_lbfgs_create(int device, int64_t n, int32_t history, int32_t dtype, hfem_lbfgs **out) {
    HFEM_ARG_CHECK(out, "null out pointer");
    *out = nullptr;
    HFEM_ARG_CHECK(n >= 1 && history >= 1 && history <= 1024, "need n >= 1 and 1 <= history <= 1024");
    HFEM_ARG_CHECK(dtype == 0 || dtype == 1, "dtype: 0 = fp64, 1 = fp32");
    if (int rc = use_device(device)) return rc;
    hfem_lbfgs *o = new (std::nothrow) hfem_lbfgs();
    HFEM_ARG_CHECK(o, "out of host memory");
    o->device = device; o->n = n; o->M1 = history + 1; o->dtype = dtype;
    const size_t esz = dtype == 0 ? 8 : 4, M1 = (size_t)o->M1;
    o->nb_chunk = (int)((n + kLbChunk - 1) / kLbChunk);
    o->nb_stream = (int)std::min<int64_t>((n + kLb - 1) / kLb, 2048);
    // fp32 histories of long vectors: 16 elements per thread as 8 pair loads (the bytes in flight per thread of the fp64 pass;
    // 0.87 -> 0.79 ms per iteration on 2 x 10^6 parameters with 100 pairs; 8 elements as 4 pair loads: no gain)
    if (dtype == 1 && n % 2 == 0 && n >= (1 << 20)) o->per = 16;
    o->nb_md = (int)((n + (int64_t)kLb * o->per - 1) / ((int64_t)kLb * o->per));
    int rc = 0;
    char *ring = nullptr;
    if (!rc) rc = lb_malloc(&ring, M1 * (size_t)n * esz); o->Sring = ring; ring = nullptr;
    if (!rc) rc = lb_malloc(&ring, M1 * (size_t)n * esz); o->Yring = ring; ring = nullptr;
    if (!rc) rc = lb_malloc(&ring, (size_t)n * esz); o->g_prev = ring; ring = nullptr;
    if (!rc) rc = lb_malloc(&ring, (size_t)n * esz); o->d = ring;
    const size_t nscal = 4 * M1 + 5 * M1 + 2 * M1 * M1;
    if (!rc) rc = lb_malloc(&o->scal, nscal);
    if (!rc) rc = lb_malloc(&o->status, 8);
    if (!rc && hipHostMalloc((void **)&o->status_pinned, 8 * sizeof(double)) != hipSuccess) {      // for the enqueue-only finish
        set_error("hfem_lbfgs_create: hipHostMalloc failed");
        rc = -1;
    }
    const size_t npart = std::max<size_t>({(size_t)o->nb_md * M1 * 5, (size_t)o->nb_stream * 5, (size_t)o->nb_chunk,
                                           (size_t)((n + kLb - 1) / kLb) < 8192 ? (size_t)((n + kLb - 1) / kLb) : 0});
    if (!rc) rc = lb_malloc(&o->A.part, npart);
    if (!rc) rc = lb_malloc(&o->A.st, 1);
    if (rc) { hfem_lbfgs_destroy(o); return rc; }
    double *s = o->scal;
    o->A.ro = s; s += M1; o->A.al = s; s += M1; o->A.cy = s; s += M1; o->A.cs = s; s += M1;
    o->A.dots = s; s += 5 * M1; o->A.SY = s; s += M1 * M1; o->A.YY = s;
    *out = o;
    return 0;
}

// Statistics of the gradient just evaluated + torch's break tests (opt_cond; after an update also the
// "lack of progress" tests and the g.d break).  loss: device scalar (fp64) of that evaluation.
// status (host, 8 doubles): loss, flags (bit0 opt_cond, 1 small step, 2 small loss change, 3 g.d break),
// max|g|, g.d, t, history count, n_iter, H_diag.  Synchronises the stream.
extern "C" int hfem_lbfgs_check(hfem_lbfgs *o, const void *g, const double *loss, int32_t after_update, double tol_grad,
                                double tol_change, double *status_host, void *stream) {
    HFEM_ARG_CHECK(o && g && status_host, "null pointer");
    if (int rc = use_device(o->device)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (o->dtype == 0) hipLaunchKernelGGL(gstats_kernel<double>, dim3(o->nb_stream), dim3(kLb), 0, s, (const double *)g, o->n, o->A.part);
    else hipLaunchKernelGGL(gstats_kernel<float>, dim3(o->nb_stream), dim3(kLb), 0, s, (const float *)g, o->n, o->A.part);
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(kLb), 0, s, o->A, o->nb_stream, loss, (int)after_update, tol_grad, tol_change,
                       o->status);
    if (int rc = launch_status("hfem_lbfgs_check")) return rc;
    HFEM_HIP_CHECK(hipMemcpyAsync(status_host, o->status, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    HFEM_HIP_CHECK(hipStreamSynchronize(s));
    return 0;
}

// the two loops in coefficient space: one wavefront (histories up to 256 pairs; HFEM_LBFGS_RECURSION=1 at build time keeps the
// 256-thread form for A/B runs), else the reduction form
#ifndef HFEM_LBFGS_RECURSION
#define HFEM_LBFGS_RECURSION 0
#endif
static void launch_recursion(hfem_lbfgs *o, int M1, double lr, double tol_change, hipStream_t s) {
    const int pairs = M1 - 1;
    if (pairs > kRecMax) { hipLaunchKernelGGL(recursion_kernel, dim3(1), dim3(64), 0, s, o->A, M1, lr, tol_change); return; }
    if (HFEM_LBFGS_RECURSION == 1 || pairs > kRecWaveMax) {
        hipLaunchKernelGGL(recursion_rank1_kernel, dim3(1), dim3(kRecT), 0, s, o->A, M1, lr, tol_change);
        return;
    }
    const int stride = pairs | 1;                         // odd: column reads (stride apart) and row reads both hit distinct banks
    const size_t lds = (size_t)pairs * stride * sizeof(double);
    static std::once_flag big_lds;                        // up to 129 KB of dynamic LDS: say so once (64 KB is the default ceiling)
    std::call_once(big_lds, [] {
        (void)hipFuncSetAttribute((const void *)recursion_wave_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
        (void)hipFuncSetAttribute((const void *)recursion_wave_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    });
    if (pairs <= 64) hipLaunchKernelGGL(recursion_wave_kernel<1>, dim3(1), dim3(kRecT), lds, s, o->A, M1, lr, tol_change, stride);
    else hipLaunchKernelGGL(recursion_wave_kernel<2>, dim3(1), dim3(kRecT), lds, s, o->A, M1, lr, tol_change, stride);
}

// Memory update + direction for the gradient `g` (the one hfem_lbfgs_check saw last): d, t, g.d and the g.d break
// flag stay on the device.
extern "C" int hfem_lbfgs_direction(hfem_lbfgs *o, const void *g, double lr, double tol_change, void *stream) {
    HFEM_ARG_CHECK(o && g, "null pointer");
    if (int rc = use_device(o->device)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int first = o->first ? 1 : 0, M1 = o->M1;
    if (o->dtype == 0) hipLaunchKernelGGL(pair_kernel<double>, dim3(o->nb_stream), dim3(kLb), 0, s, o->A, (const double *)g, (double *)o->g_prev, (const double *)o->d, (double *)o->Sring, (double *)o->Yring, o->n, M1, first);
    else hipLaunchKernelGGL(pair_kernel<float>, dim3(o->nb_stream), dim3(kLb), 0, s, o->A, (const float *)g, (float *)o->g_prev, (const float *)o->d, (float *)o->Sring, (float *)o->Yring, o->n, M1, first);
    hipLaunchKernelGGL(pair_reduce_kernel, dim3(1), dim3(kLb), 0, s, o->A, o->nb_stream, M1, first);
    if (!first) {
        int gy = 2048 / o->nb_chunk;                      // slot classes: fill the chip when the vectors are short
        gy = gy < 1 ? 1 : (gy > 16 ? 16 : gy);
#define HFEM_MD(T, P, V) hipLaunchKernelGGL((multidot_kernel<T, P, V>), dim3(o->nb_md, gy), dim3(kLb), 0, s, o->A, (const T *)g, (const T *)o->Sring, (const T *)o->Yring, o->n, M1)
        if (o->dtype == 0 && (HFEM_LBFGS_VARIANT & 8) && o->n % 2 == 0) HFEM_MD(double, kLbPer / 2, 2);
        else if (o->dtype == 0) HFEM_MD(double, kLbPer, 1);
        else if (o->per == 16) HFEM_MD(float, 8, 2);
        else HFEM_MD(float, kLbPer, 1);
#undef HFEM_MD
        hipLaunchKernelGGL(multidot_reduce_kernel, dim3(M1 - 1), dim3(kLb), 0, s, o->A, o->nb_md, M1);
    }
    launch_recursion(o, M1, lr, tol_change, s);
    // elements per thread of the direction pass: 8 for very long vectors (fewer, fatter workgroups), 1 for short ones (more
    // workgroups), in between 2 (fp64) / 4 (fp32) -- measured on 2 x 10^6 parameters with 100 pairs: fp64 1.29 -> 1.26 ms, fp32
    // 0.79 -> 0.77 ms per iteration; 8 x 10^6 parameters: 8 is best (2.08 against 2.13 / 2.17 ms with 2 / 1)
    const int per_dir = o->nb_chunk >= 1024 ? kLbPer : (o->n >= (1 << 20) ? (o->dtype == 0 ? 2 : 4) : 1);
    const int nb_dir = (int)((o->n + (int64_t)per_dir * kLb - 1) / ((int64_t)per_dir * kLb));
#define HFEM_DIR(T, P) hipLaunchKernelGGL((direction_kernel<T, P>), dim3(nb_dir), dim3(kLb), 0, s, o->A, (const T *)g, (const T *)o->Sring, (const T *)o->Yring, (T *)o->d, o->n, M1)
    if (o->dtype == 0) {
        if (per_dir == kLbPer) HFEM_DIR(double, kLbPer);
        else if (per_dir == 2) HFEM_DIR(double, 2);
        else HFEM_DIR(double, 1);
    } else {
        if (per_dir == kLbPer) HFEM_DIR(float, kLbPer);
        else if (per_dir == 4) HFEM_DIR(float, 4);
        else HFEM_DIR(float, 1);
    }
#undef HFEM_DIR
    hipLaunchKernelGGL(dmax_reduce_kernel, dim3(1), dim3(kLb), 0, s, o->A, nb_dir);
    o->first = false;
    return launch_status("hfem_lbfgs_direction");
}

// p[0..numel) += t * d[offset .. offset+numel) for one parameter tensor (nothing if the iteration stopped on g.d)
extern "C" int hfem_lbfgs_apply(hfem_lbfgs *o, void *p, int64_t offset, int64_t numel, void *stream) {
    HFEM_ARG_CHECK(o && (p || numel == 0), "null pointer");
    HFEM_ARG_CHECK(offset >= 0 && numel >= 0 && offset + numel <= o->n, "segment outside the flat vector");
    if (numel == 0) return 0;
    if (int rc = use_device(o->device)) return rc;
    const int grid = (int)std::min<int64_t>((numel + kLb - 1) / kLb, 4096);
    if (o->dtype == 0) hipLaunchKernelGGL(apply_kernel<double>, dim3(grid), dim3(kLb), 0, (hipStream_t)stream, o->A, (double *)p, (const double *)o->d + offset, numel);
    else hipLaunchKernelGGL(apply_kernel<float>, dim3(grid), dim3(kLb), 0, (hipStream_t)stream, o->A, (float *)p, (const float *)o->d + offset, numel);
    return launch_status("hfem_lbfgs_apply");
}

// ---- node-sharded flow (see the block comment above shard_pair_kernel) ------------------------------------------------
extern "C" int64_t hfem_lbfgs_shard_payload_doubles(const hfem_lbfgs *o) {
    if (!o) { set_error("hfem_lbfgs_shard_payload_doubles: null optimiser"); return -1; }
    return 5 * (int64_t)o->M1 + kShardTail;
}

// Phase 1 (local): g = this rank's flat gradient (n values); speculative pair into the spare slot, ONE pass over the local
// history, everything another rank needs into payload_dev (hfem_lbfgs_shard_payload_doubles doubles).  loss_local_dev: the
// rank's partial energy (device scalar) or NULL.  Enqueues only.
extern "C" int hfem_lbfgs_shard_local(hfem_lbfgs *o, const void *g, const double *loss_local_dev, double *payload_dev, void *stream) {
    HFEM_ARG_CHECK(o && g && payload_dev, "null pointer");
    if (int rc = use_device(o->device)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int first = o->first ? 1 : 0, M1 = o->M1;
    if (o->dtype == 0) hipLaunchKernelGGL(shard_pair_kernel<double>, dim3(o->nb_stream), dim3(kLb), 0, s, o->A, (const double *)g, (const double *)o->g_prev, (const double *)o->d, (double *)o->Sring, (double *)o->Yring, o->n, M1, first);
    else hipLaunchKernelGGL(shard_pair_kernel<float>, dim3(o->nb_stream), dim3(kLb), 0, s, o->A, (const float *)g, (const float *)o->g_prev, (const float *)o->d, (float *)o->Sring, (float *)o->Yring, o->n, M1, first);
    hipLaunchKernelGGL(shard_tail_kernel, dim3(1), dim3(kLb), 0, s, o->A, o->nb_stream, M1, loss_local_dev, payload_dev);
    if (!first) {
        int gy = 2048 / o->nb_chunk;
        gy = gy < 1 ? 1 : (gy > 16 ? 16 : gy);
#define HFEM_MD(T, P, V) hipLaunchKernelGGL((multidot_kernel<T, P, V>), dim3(o->nb_md, gy), dim3(kLb), 0, s, o->A, (const T *)g, (const T *)o->Sring, (const T *)o->Yring, o->n, M1, 1)
        if (o->dtype == 0 && (HFEM_LBFGS_VARIANT & 8) && o->n % 2 == 0) HFEM_MD(double, kLbPer / 2, 2);
        else if (o->dtype == 0) HFEM_MD(double, kLbPer, 1);
        else if (o->per == 16) HFEM_MD(float, 8, 2);
        else HFEM_MD(float, kLbPer, 1);
#undef HFEM_MD
        hipLaunchKernelGGL(multidot_reduce_kernel, dim3(M1), dim3(kLb), 0, s, o->A, o->nb_md, M1, payload_dev);
    }
    return launch_status("hfem_lbfgs_shard_local");
}

// Phase 2 (after the payloads of all `world` ranks were gathered to gathered_dev [world][payload doubles], rank order): the
// global sums, torch's break tests (after_update: also the lack-of-progress tests), and -- unless one fired or
// want_direction == 0 -- the memory update, the recursion and this rank's part of the direction (prev_flat_grad = g is taken
// then, as in torch).  status_host[8] as hfem_lbfgs_check, g.d / t / bit 3 of the direction JUST computed, bit 4 (16): no
// direction was computed.  Identical on every rank by construction.  Synchronises the stream.
extern "C" int hfem_lbfgs_shard_finish(hfem_lbfgs *o, const void *g, const double *gathered_dev, int32_t world, int32_t after_update,
                                       int32_t want_direction, double lr, double tol_grad, double tol_change, double *status_host,
                                       void *stream) {
    HFEM_ARG_CHECK(o && g && gathered_dev, "null pointer");
    HFEM_ARG_CHECK(world >= 1, "world must be >= 1");
    HFEM_ARG_CHECK(status_host || o->status_pinned, "no pinned status buffer");
    if (int rc = use_device(o->device)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int first = o->first ? 1 : 0, M1 = o->M1;
    hipLaunchKernelGGL(shard_finish_kernel, dim3(1), dim3(kLb), 0, s, o->A, gathered_dev, (int)world, 5 * M1 + kShardTail, M1, first,
                       (int)after_update, (int)want_direction, tol_grad, tol_change);
    launch_recursion(o, M1, lr, tol_change, s);
    const int per_dir = o->nb_chunk >= 1024 ? kLbPer : (o->n >= (1 << 20) ? (o->dtype == 0 ? 2 : 4) : 1);
    const int nb_dir = (int)((o->n + (int64_t)per_dir * kLb - 1) / ((int64_t)per_dir * kLb));
#define HFEM_DIR(T, P) hipLaunchKernelGGL((direction_kernel<T, P>), dim3(nb_dir), dim3(kLb), 0, s, o->A, (const T *)g, (const T *)o->Sring, (const T *)o->Yring, (T *)o->d, o->n, M1, (T *)o->g_prev)
    if (o->dtype == 0) {
        if (per_dir == kLbPer) HFEM_DIR(double, kLbPer);
        else if (per_dir == 2) HFEM_DIR(double, 2);
        else HFEM_DIR(double, 1);
    } else {
        if (per_dir == kLbPer) HFEM_DIR(float, kLbPer);
        else if (per_dir == 4) HFEM_DIR(float, 4);
        else HFEM_DIR(float, 1);
    }
#undef HFEM_DIR
    // enqueue only (status_host NULL; legal inside a hipGraph capture): the last kernel writes the record straight into the
    // pinned, device-visible host buffer hfem_lbfgs_shard_status reads after its stream synchronisation -- no copy node behind it
    hipLaunchKernelGGL(dmax_reduce_kernel, dim3(1), dim3(kLb), 0, s, o->A, nb_dir, status_host ? o->status : o->status_pinned);
    if (int rc = launch_status("hfem_lbfgs_shard_finish")) return rc;
    if (!status_host) return 0;
    HFEM_HIP_CHECK(hipMemcpyAsync(status_host, o->status, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    HFEM_HIP_CHECK(hipStreamSynchronize(s));
    if (!((int)status_host[1] & 16)) o->first = false;          // a direction exists from now on
    return 0;
}

// Status record of the last hfem_lbfgs_shard_finish that was given no host pointer (a captured / replayed iteration):
// synchronises the stream, copies the pinned record out.
extern "C" int hfem_lbfgs_shard_status(hfem_lbfgs *o, double *status_host, void *stream) {
    HFEM_ARG_CHECK(o && status_host && o->status_pinned, "null pointer / no enqueue-only finish has run");
    if (int rc = use_device(o->device)) return rc;
    HFEM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < 8; ++i) status_host[i] = o->status_pinned[i];
    if (!((int)status_host[1] & 16)) o->first = false;
    return 0;
}

// this rank's flat gradient from the gradient ROW arrays: out = [gx rows rows_x | gu rows rows_u] (two values per row; the
// optimiser's dtype); out NULL = into the optimiser's own buffer (returned by hfem_lbfgs_shard_grad_ptr)
extern "C" int hfem_lbfgs_shard_gather(hfem_lbfgs *o, const void *gx, const int32_t *rows_x, int64_t nx, const void *gu,
                                       const int32_t *rows_u, int64_t nu, void *out, void *stream) {
    HFEM_ARG_CHECK(o && out && (gx || nx == 0) && (gu || nu == 0), "null pointer");
    HFEM_ARG_CHECK(2 * (nx + nu) == o->n, "row counts do not match the optimiser's length (two values per row)");
    if (int rc = use_device(o->device)) return rc;
    const int grid = (int)std::min<int64_t>((nx + nu + kLb - 1) / kLb, 4096);
    if (o->dtype == 0) hipLaunchKernelGGL(shard_gather_kernel<double>, dim3(grid), dim3(kLb), 0, (hipStream_t)stream, (const double *)gx, rows_x, nx, (const double *)gu, rows_u, nu, (double *)out);
    else hipLaunchKernelGGL(shard_gather_kernel<float>, dim3(grid), dim3(kLb), 0, (hipStream_t)stream, (const float *)gx, rows_x, nx, (const float *)gu, rows_u, nu, (float *)out);
    return launch_status("hfem_lbfgs_shard_gather");
}

// x[rows_x], u[rows_u] += t d on the rows this rank owns (nothing if no direction was computed or it stopped on g.d)
extern "C" int hfem_lbfgs_shard_apply(hfem_lbfgs *o, void *x, const int32_t *rows_x, int64_t nx, void *u, const int32_t *rows_u,
                                      int64_t nu, void *stream) {
    HFEM_ARG_CHECK(o && (x || nx == 0) && (u || nu == 0), "null pointer");
    HFEM_ARG_CHECK(2 * (nx + nu) == o->n, "row counts do not match the optimiser's length (two values per row)");
    if (int rc = use_device(o->device)) return rc;
    const int grid = (int)std::min<int64_t>((nx + nu + kLb - 1) / kLb, 4096);
    if (o->dtype == 0) hipLaunchKernelGGL(shard_apply_kernel<double>, dim3(grid), dim3(kLb), 0, (hipStream_t)stream, o->A, (double *)x, rows_x, nx, (double *)u, rows_u, nu, (const double *)o->d);
    else hipLaunchKernelGGL(shard_apply_kernel<float>, dim3(grid), dim3(kLb), 0, (hipStream_t)stream, o->A, (float *)x, rows_x, nx, (float *)u, rows_u, nu, (const float *)o->d);
    return launch_status("hfem_lbfgs_shard_apply");
}

// device pointer of the direction d (flat, n elements of the optimiser's dtype): for tests / callers that
// want to inspect it
extern "C" void *hfem_lbfgs_direction_ptr(hfem_lbfgs *o) { return o ? o->d : nullptr; }

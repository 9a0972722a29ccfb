// Dev tool (GPU box): what read rate does the L-BFGS direction pass's access pattern allow on this chip?  d[i] = sum_l c_l V_l[i]
// over NV = 200 vectors of n = 1 999 002 doubles (3.2 GB, far beyond the 256 MB Infinity Cache), one output write.  Variants:
// elements per load (8 / 16 bytes), loads per thread, slot-loop unroll, non-temporal loads, block size, and a form where a
// workgroup walks SEVERAL chunks (fewer, longer-lived workgroups).  HIP events, median of 5, prints TB/s.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/history_stream_bench.hip -o /tmp/history_stream_bench && /tmp/history_stream_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } \
    } while (0)

template <int VEC, bool NT>
__device__ __forceinline__ void ld(const double *p, int64_t i, int64_t n, double *out) {
    if constexpr (VEC == 2) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        if (i < n) {
            const d2 *q = reinterpret_cast<const d2 *>(p + i);
            const d2 v = NT ? __builtin_nontemporal_load(q) : *q;
            out[0] = v.x; out[1] = v.y;
        } else { out[0] = 0; out[1] = 0; }
    } else {
        out[0] = i < n ? (NT ? __builtin_nontemporal_load(p + i) : p[i]) : 0.0;
    }
}

template <int BLOCK, int PER, int VEC, int UNR, bool NT>
__global__ __launch_bounds__(BLOCK) void dir_kernel(const double *__restrict__ V, const double *__restrict__ c, double *__restrict__ d,
                                                    int64_t n, int nv, int chunks_per_wg) {
    constexpr int E = PER * VEC;
    for (int cc = 0; cc < chunks_per_wg; ++cc) {
        const int64_t chunk = (int64_t)blockIdx.x * chunks_per_wg + cc;
        const int64_t base = chunk * (BLOCK * E) + (int64_t)threadIdx.x * VEC;
        if (base - (int64_t)threadIdx.x * VEC >= n) return;
        double acc[E];
#pragma unroll
        for (int k = 0; k < E; ++k) acc[k] = 0.0;
#pragma unroll UNR
        for (int l = 0; l < nv; ++l) {
            const double cl = c[l];
            const double *Vl = V + (int64_t)l * n;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                double v[VEC];
                ld<VEC, NT>(Vl, base + (int64_t)k * BLOCK * VEC, n, v);
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[k * VEC + q] += cl * v[q];
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t i = base + (int64_t)k * BLOCK * VEC;
            if (i < n)
                for (int q = 0; q < VEC; ++q) d[i + q] = acc[k * VEC + q];
        }
    }
}

struct Case { const char *name; void (*launch)(const double *, const double *, double *, int64_t, int, hipStream_t); };

template <int BLOCK, int PER, int VEC, int UNR, bool NT, int CPW>
void launch(const double *V, const double *c, double *d, int64_t n, int nv, hipStream_t s) {
    const int64_t per_wg = (int64_t)BLOCK * PER * VEC * CPW;
    const int grid = (int)((n + per_wg - 1) / per_wg);
    hipLaunchKernelGGL((dir_kernel<BLOCK, PER, VEC, UNR, NT>), dim3(grid), dim3(BLOCK), 0, s, V, c, d, n, nv, CPW);
}

int main() {
    const int64_t n = 1999002;
    const int nv = 200;
    double *V, *c, *d;
    CK(hipMalloc(&V, sizeof(double) * n * nv));
    CK(hipMalloc(&c, sizeof(double) * nv));
    CK(hipMalloc(&d, sizeof(double) * n));
    CK(hipMemset(V, 0, sizeof(double) * n * nv));
    std::vector<double> hc(nv, 1.0);
    CK(hipMemcpy(c, hc.data(), sizeof(double) * nv, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
#define CASE(B, P, VV, U, NT, CPW) Case{"block " #B " per " #P " vec " #VV " unroll " #U " nt " #NT " chunks/wg " #CPW, launch<B, P, VV, U, NT, CPW>}
    const Case cases[] = {
        CASE(256, 2, 1, 4, false, 1), CASE(256, 2, 1, 4, true, 1), CASE(256, 1, 2, 4, false, 1), CASE(256, 1, 2, 4, true, 1),
        CASE(256, 2, 1, 8, false, 1), CASE(256, 1, 2, 8, false, 1), CASE(256, 2, 2, 4, false, 1), CASE(256, 2, 2, 8, false, 1),
        CASE(256, 4, 2, 4, false, 1), CASE(256, 1, 1, 8, false, 1), CASE(256, 1, 1, 16, false, 1), CASE(512, 1, 2, 4, false, 1),
        CASE(1024, 1, 2, 4, false, 1), CASE(256, 1, 2, 4, false, 2), CASE(256, 1, 2, 4, false, 4), CASE(256, 1, 2, 8, true, 1),
        CASE(256, 2, 2, 4, true, 1), CASE(128, 1, 2, 4, false, 1), CASE(64, 1, 2, 8, false, 1),
    };
    const double bytes = (double)n * 8.0 * (nv + 1);
    for (const Case &k : cases) {
        for (int w = 0; w < 3; ++w) k.launch(V, c, d, n, nv, s);
        CK(hipStreamSynchronize(s));
        std::vector<float> ms;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, s));
            k.launch(V, c, d, n, nv, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-62s %8.1f us  %6.3f TB/s\n", k.name, ms[2] * 1e3, bytes / (ms[2] * 1e-3) / 1e12);
        fflush(stdout);
    }
    CK(hipGetLastError());
    return 0;
}

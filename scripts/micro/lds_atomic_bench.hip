// Dev tool (GPU box): throughput of LDS atomic adds by operand type on gfx950 -- what the slot loop of the tiled energy kernels
// is bound by.  256 threads per block, 4 blocks per CU, every lane adds to its own slot (conflict-free: slot = lane-dependent,
// distinct within a wave), ITER x 16 atomics per thread, timed with HIP events.  Prints ns per wave-instruction per CU.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/micro/lds_atomic_bench.hip -o /tmp/lds_atomic_bench && /tmp/lds_atomic_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

constexpr int ITER = 64;

template <typename T>
__global__ __launch_bounds__(256) void k_atomic(T *out, int stride) {
    __shared__ T acc[4 * 640];
    for (int i = threadIdx.x; i < 4 * 640; i += 256) acc[i] = (T)0;
    __syncthreads();
    const int l = (threadIdx.x * stride) % 560;          // stride 1: consecutive slots; other strides: spread
    T v = (T)(threadIdx.x + 1);
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (sizeof(T) == 8 && !__is_floating_point(T)) {
                atomicAdd((unsigned long long *)&acc[0 * 640 + l], (unsigned long long)v);
                atomicAdd((unsigned long long *)&acc[1 * 640 + l], (unsigned long long)v);
                atomicAdd((unsigned long long *)&acc[2 * 640 + l], (unsigned long long)v);
                atomicAdd((unsigned long long *)&acc[3 * 640 + l], (unsigned long long)v);
            } else if constexpr (sizeof(T) == 4 && !__is_floating_point(T)) {
                atomicAdd((unsigned *)&acc[0 * 640 + l], (unsigned)v);
                atomicAdd((unsigned *)&acc[1 * 640 + l], (unsigned)v);
                atomicAdd((unsigned *)&acc[2 * 640 + l], (unsigned)v);
                atomicAdd((unsigned *)&acc[3 * 640 + l], (unsigned)v);
            } else {
                unsafeAtomicAdd(&acc[0 * 640 + l], v);
                unsafeAtomicAdd(&acc[1 * 640 + l], v);
                unsafeAtomicAdd(&acc[2 * 640 + l], v);
                unsafeAtomicAdd(&acc[3 * 640 + l], v);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0] + acc[641];
}

template <typename T>
double run(const char *name, int stride) {
    T *out;
    hipMalloc(&out, 4096 * sizeof(T));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grid = 1024;                               // 4 blocks per CU on 256 CUs: one resident round
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_atomic<T>, dim3(grid), dim3(256), 0, 0, out, stride);
    hipEventRecord(a);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k_atomic<T>, dim3(grid), dim3(256), 0, 0, out, stride);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps;
    // per CU: 4 blocks x 4 waves x ITER x 16 wave-instructions
    const double winst = 4.0 * 4.0 * ITER * 16.0;
    printf("%-6s stride %2d: %8.2f us per launch, %6.2f ns per wave-instruction per CU (%.1f cycles at 2.4 GHz)\n", name, stride, us,
           us * 1e3 / winst, us * 1e3 / winst * 2.4);
    hipFree(out);
    return us;
}

int main() {
    for (int stride : {1, 17}) {
        run<double>("f64", stride);
        run<unsigned long long>("u64", stride);
        run<float>("f32", stride);
        run<unsigned>("u32", stride);
    }
    return 0;
}

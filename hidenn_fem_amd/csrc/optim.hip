// Fused Adam step on a flat parameter tensor (the first "next" row after the
// energy path): the reference drives its models with torch.optim.Adam
// (/root/reference/examples/example1.py:31, example2.py:37, example3.py:89), which costs ~10 small
// kernels per parameter tensor per step; once the energy is one launch the optimiser dominates.
// One launch here reads p,g,m,v and writes p,m,v (56 B/param fp64) -- bandwidth-bound, no reuse.
// Arithmetic follows torch.optim.Adam (betas, eps, bias corrections, no weight decay / amsgrad)
// operation for operation:  m += (g-m)(1-b1);  v = v b2 + (1-b2) g g;
//                           p -= (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps).
#include <hip/hip_runtime.h>

#include <cmath>

#include "hfem_device.h"

namespace hfem {

template <typename T>
__global__ __launch_bounds__(256) void adam_step_kernel(T *__restrict__ p, const T *__restrict__ g,
                                                        T *__restrict__ m, T *__restrict__ v, int64_t n,
                                                        double w1, double b2, double w2, double step_size,
                                                        double inv_sqrt_bc2, double eps) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const T gi = g[i];
        const T mi = m[i] + (T)w1 * (gi - m[i]);                 // lerp_(grad, 1 - beta1)
        const T vi = v[i] * (T)b2 + (T)w2 * gi * gi;             // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const T denom = (T)sqrt((double)vi) * (T)inv_sqrt_bc2 + (T)eps;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - (T)step_size * (mi / denom);
    }
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
    const double step_size = lr / bc1, inv_sqrt_bc2 = 1.0 / std::sqrt(bc2);
    int64_t grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (dtype == 0)
        hipLaunchKernelGGL(adam_step_kernel<double>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (double *)p,
                           (const double *)g, (double *)m, (double *)v, n, 1.0 - beta1, beta2, 1.0 - beta2, step_size,
                           inv_sqrt_bc2, eps);
    else
        hipLaunchKernelGGL(adam_step_kernel<float>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (float *)p,
                           (const float *)g, (float *)m, (float *)v, n, 1.0 - beta1, beta2, 1.0 - beta2, step_size,
                           inv_sqrt_bc2, eps);
    return launch_status("hfem_adam_step");
}

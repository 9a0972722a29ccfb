// Interface exchange of the owner-sharded multi-GPU mode (the reference has no
// distributed code).  Elements are sharded by contiguous tile ranges; owner-computes tiles with halo
// recompute give every rank COMPLETE gradient rows for the nodes its tiles own, so gradients never cross
// ranks.  What does cross, once per step, is (a) each rank's partial energy and (b) the parameter rows
// (x, u) of the nodes a rank owns that other ranks' tiles read as halo -- O(sqrt(elements per rank)) rows.
// Both travel in ONE all_gather of a fixed-size payload per rank:
//     [ x rows of my interface | u rows of my interface | padding ]  [ loss partial, 0 ]      (double2 units)
// iface_pack fills the payload from the local parameters (the energy kernel writes the loss slot itself),
// iface_unpack copies the rows this rank needs out of the gathered payloads into its local parameter
// arrays and adds the partial energies in rank order (deterministic).  Pure data movement, a few KB..MB.
#include <hip/hip_runtime.h>

#include "hfem_plan_dev.h"

namespace hfem {

__global__ __launch_bounds__(256) void iface_pack_kernel(const double2 *__restrict__ x_free,
                                                         const double2 *__restrict__ u_free,
                                                         const int32_t *__restrict__ rows, int n_x, int n_u,
                                                         double2 *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_x) out[i] = x_free[rows[i]];
    else if (i < n_x + n_u) out[i] = u_free[rows[i]];
}

template <typename V>
__global__ __launch_bounds__(256) void iface_unpack_kernel(const double2 *__restrict__ recv,
                                                           const int32_t *__restrict__ src,
                                                           const int32_t *__restrict__ dst, int n_x, int n_u,
                                                           V *__restrict__ x_free, V *__restrict__ u_free,
                                                           int world, int64_t stride, int64_t loss_slot,
                                                           double *__restrict__ loss_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_x) x_free[dst[i]] = row_narrow<V>(recv[src[i]]);
    else if (i < n_x + n_u) u_free[dst[i]] = row_narrow<V>(recv[src[i]]);
    if (i == 0 && loss_out) {
        double tot = 0.0;
        for (int r = 0; r < world; ++r) tot += recv[(int64_t)r * stride + loss_slot].x;   // fixed order
        loss_out[0] = tot;
    }
}

// pack + the rank's energy + the step counter, one launch (hfem_plan_iface_pack): block 0 also reduces the tile energies
// exactly as sum_partials_kernel does (256 adders, shuffle tree, wave sums in wave order: the same bits)
template <typename V>
__global__ __launch_bounds__(256) void iface_pack_sum_kernel(const V *__restrict__ x_free,
                                                             const V *__restrict__ u_free,
                                                             const int32_t *__restrict__ rows, int n_x, int n_u,
                                                             double2 *__restrict__ out, int64_t loss_slot,
                                                             const double *__restrict__ partials, int n_partials,
                                                             int64_t *__restrict__ counter, double beta1, double beta2,
                                                             double *__restrict__ bc_next) {
    __shared__ double red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_x) out[i] = row_widen(x_free[rows[i]]);
    else if (i < n_x + n_u) out[i] = row_widen(u_free[rows[i]]);
    if (blockIdx.x == 0) {
        double v = 0.0;
        for (int k = threadIdx.x; k < n_partials; k += 256) v += partials[k];
        const double tot = block_sum(v, red);
        if (threadIdx.x == 0) {
            out[loss_slot] = make_double2(tot, 0.0);
            if (counter) {
                const int64_t c = counter[0] + 1;
                counter[0] = c;
                if (bc_next) {                              // bias corrections of step c + 1 (hfem_adam_prep's scalars)
                    bc_next[0] = 1.0 - pow(beta1, (double)(c + 1));
                    bc_next[1] = sqrt(1.0 - pow(beta2, (double)(c + 1)));
                }
            }
        }
    }
}

int launch_iface_pack_sum(int dtype, const void *x_free, const void *u_free, const int32_t *rows, int n_x, int n_u, double *out,
                          int64_t loss_slot, const double *partials, int n_partials, int64_t *counter, double beta1,
                          double beta2, double *bc_next, hipStream_t s) {
    const int n = n_x + n_u > 0 ? n_x + n_u : 1;
    if (dtype == 0)
        hipLaunchKernelGGL(iface_pack_sum_kernel<double2>, dim3((n + 255) / 256), dim3(256), 0, s, (const double2 *)x_free,
                           (const double2 *)u_free, rows, n_x, n_u, (double2 *)out, loss_slot, partials, n_partials, counter,
                           beta1, beta2, bc_next);
    else
        hipLaunchKernelGGL(iface_pack_sum_kernel<float2>, dim3((n + 255) / 256), dim3(256), 0, s, (const float2 *)x_free,
                           (const float2 *)u_free, rows, n_x, n_u, (double2 *)out, loss_slot, partials, n_partials, counter,
                           beta1, beta2, bc_next);
    return launch_status("hfem_plan_iface_pack");
}

}  // namespace hfem

using namespace hfem;

extern "C" int hfem_iface_pack(int device, const double *x_free, const double *u_free, const int32_t *rows,
                               int32_t n_x, int32_t n_u, double *out, void *stream) {
    HFEM_ARG_CHECK(n_x >= 0 && n_u >= 0, "negative row count");
    if (n_x + n_u == 0) return 0;
    HFEM_ARG_CHECK(rows && out && (n_x == 0 || x_free) && (n_u == 0 || u_free), "null pointer");
    if (int rc = use_device(device)) return rc;
    hipLaunchKernelGGL(iface_pack_kernel, dim3((n_x + n_u + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const double2 *)x_free, (const double2 *)u_free, rows, n_x, n_u, (double2 *)out);
    return launch_status("hfem_iface_pack");
}

static int iface_unpack_any(int dtype, int device, const double *recv, const int32_t *src, const int32_t *dst, int32_t n_x,
                            int32_t n_u, void *x_free, void *u_free, int32_t world, int64_t stride, int64_t loss_slot,
                            double *loss_out, void *stream) {
    HFEM_ARG_CHECK(n_x >= 0 && n_u >= 0 && world >= 1 && stride >= 1 && loss_slot >= 0 && loss_slot < stride,
                   "bad sizes");
    HFEM_ARG_CHECK(recv && (n_x + n_u == 0 || (src && dst)) && (n_x == 0 || x_free) && (n_u == 0 || u_free),
                   "null pointer");
    if (int rc = use_device(device)) return rc;
    const int n = n_x + n_u > 0 ? n_x + n_u : 1;
    if (dtype == 0)
        hipLaunchKernelGGL(iface_unpack_kernel<double2>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const double2 *)recv, src, dst, n_x, n_u, (double2 *)x_free, (double2 *)u_free, world, stride,
                           loss_slot, loss_out);
    else
        hipLaunchKernelGGL(iface_unpack_kernel<float2>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const double2 *)recv, src, dst, n_x, n_u, (float2 *)x_free, (float2 *)u_free, world, stride,
                           loss_slot, loss_out);
    return launch_status("hfem_iface_unpack");
}

extern "C" int hfem_iface_unpack(int device, const double *recv, const int32_t *src, const int32_t *dst, int32_t n_x,
                                 int32_t n_u, double *x_free, double *u_free, int32_t world, int64_t stride,
                                 int64_t loss_slot, double *loss_out, void *stream) {
    return iface_unpack_any(0, device, recv, src, dst, n_x, n_u, x_free, u_free, world, stride, loss_slot, loss_out, stream);
}

// float rows (an fp32 model, the reference's default dtype): the payload stays double2, rows are rounded back on the way in
extern "C" int hfem_iface_unpack_f32(int device, const double *recv, const int32_t *src, const int32_t *dst, int32_t n_x,
                                     int32_t n_u, float *x_free, float *u_free, int32_t world, int64_t stride,
                                     int64_t loss_slot, double *loss_out, void *stream) {
    return iface_unpack_any(1, device, recv, src, dst, n_x, n_u, x_free, u_free, world, stride, loss_slot, loss_out, stream);
}

// Deterministic TRI3 + EDGE2 energy and gradients, gfx950: the cross-check variant of SURVEY section 5
// (HFEM_FLAG_DETERMINISTIC on hfem_tri3_energy_plan).
//
// Replaces the same reference code as the tiled kernels (/root/reference/src/loss.py:55-116 over
// /root/reference/src/models.py:292-376) but accumulates in a FIXED order, so two launches on the same inputs give
// bit-identical gradients and loss (the tiled kernels add with ds_add_f64, whose order varies in the last bits):
//   * gradients: one thread per NODE walks the node -> (element, corner) adjacency (CSR, ascending element id),
//     re-evaluates each adjacent element with the closed forms of hfem_device.h and keeps that corner's row -- every
//     element is evaluated by each of its three corners (3x the flops, free at this arithmetic intensity), nothing
//     is shared between threads, no atomics; the node's Neumann edges follow, then one store per row;
//   * loss: one thread per element / edge, shuffle-tree + wave-order block sums, a one-block sum of the block partials.
// Four dependent gather levels (adjacency -> connectivity -> row map -> row): L2-latency bound, ~3-4x the time of the
// tiled kernel -- a checker, not the fast path.  Either gradient convention (PHYS, hfem_device.h).
#include <hip/hip_runtime.h>

#include <vector>

#include "hfem_device.h"
#include "hfem_plan_dev.h"

namespace hfem {

constexpr int kDetBlock = 256;

__device__ __forceinline__ double2 row_of(const double2 *__restrict__ free_rows, const double2 *__restrict__ fixed_rows,
                                          int32_t src) {
    return src >= 0 ? free_rows[src] : fixed_rows[~src];
}

template <bool PHYS>
__global__ __launch_bounds__(kDetBlock) void tri3_det_grad_kernel(
    int32_t nn, const int32_t *__restrict__ conn, const int32_t *__restrict__ x_src, const int32_t *__restrict__ u_src,
    const int32_t *__restrict__ adj_ptr, const int32_t *__restrict__ adj, const int32_t *__restrict__ edges,
    const int32_t *__restrict__ eadj_ptr, const int32_t *__restrict__ eadj, const double2 *__restrict__ x_free,
    const double2 *__restrict__ x_fixed, const double2 *__restrict__ u_free, const double2 *__restrict__ u_fixed,
    Tri3Consts k, const double4 *__restrict__ T_edge, double4 Tconst, int skip_edges, double2 *__restrict__ gx_free,
    double2 *__restrict__ gu_free) {
    const int32_t n = blockIdx.x * kDetBlock + threadIdx.x;
    if (n >= nn) return;
    double2 sx = make_double2(0.0, 0.0), su = make_double2(0.0, 0.0);
    for (int32_t i = adj_ptr[n]; i < adj_ptr[n + 1]; ++i) {
        const int32_t ec = adj[i], e = ec >> 2, c = ec & 3;
        const int32_t n0 = conn[3 * (int64_t)e], n1 = conn[3 * (int64_t)e + 1], n2 = conn[3 * (int64_t)e + 2];
        double2 gx[3], gu[3];
        tri3_element<true, true, PHYS>(row_of(x_free, x_fixed, x_src[n0]), row_of(x_free, x_fixed, x_src[n1]),
                                       row_of(x_free, x_fixed, x_src[n2]), row_of(u_free, u_fixed, u_src[n0]),
                                       row_of(u_free, u_fixed, u_src[n1]), row_of(u_free, u_fixed, u_src[n2]), k, gx, gu);
        // this node's corner by 0 / 1 weights, not by a select chain: the compiler turns `c == 0 ? g[0] : ...` into an indexed
        // load of a SCRATCH copy of the arrays.  Exact (x * 1 + y * 0 + z * 0) for finite rows; an element whose rows are not
        // finite (det = 0) has no finite corner anyway.
        const double m0 = c == 0 ? 1.0 : 0.0, m1 = c == 1 ? 1.0 : 0.0, m2 = c == 2 ? 1.0 : 0.0;
        sx.x += m0 * gx[0].x + m1 * gx[1].x + m2 * gx[2].x; sx.y += m0 * gx[0].y + m1 * gx[1].y + m2 * gx[2].y;
        su.x += m0 * gu[0].x + m1 * gu[1].x + m2 * gu[2].x; su.y += m0 * gu[0].y + m1 * gu[1].y + m2 * gu[2].y;
    }
    if (!skip_edges)
        for (int32_t i = eadj_ptr[n]; i < eadj_ptr[n + 1]; ++i) {
            const int32_t ge = eadj[i], g = ge >> 1, end = ge & 1;
            const int32_t ni = edges[2 * (int64_t)g], nj = edges[2 * (int64_t)g + 1];
            const double4 tt = T_edge ? T_edge[g] : Tconst;
            double2 gx[2], gu[2];
            edge2_element<true>(row_of(x_free, x_fixed, x_src[ni]), row_of(x_free, x_fixed, x_src[nj]),
                                row_of(u_free, u_fixed, u_src[ni]), row_of(u_free, u_fixed, u_src[nj]), tt, gx, gu);
            const double2 px = end ? gx[1] : gx[0], pu = end ? gu[1] : gu[0];      // selects, not a dynamic index (that puts the arrays in scratch)
            sx.x += px.x; sx.y += px.y;
            su.x += pu.x; su.y += pu.y;
        }
    const int32_t rx = x_src[n], ru = u_src[n];
    if (gx_free && rx >= 0) gx_free[rx] = sx;
    if (gu_free && ru >= 0) gu_free[ru] = su;
}

template <bool PHYS>
__global__ __launch_bounds__(kDetBlock) void tri3_det_loss_kernel(
    int32_t ne, int32_t ned, const int32_t *__restrict__ conn, const int32_t *__restrict__ x_src,
    const int32_t *__restrict__ u_src, const int32_t *__restrict__ edges, const double2 *__restrict__ x_free,
    const double2 *__restrict__ x_fixed, const double2 *__restrict__ u_free, const double2 *__restrict__ u_fixed,
    Tri3Consts k, const double4 *__restrict__ T_edge, double4 Tconst, double *__restrict__ partials) {
    __shared__ double red[kDetBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kDetBlock + threadIdx.x;
    double v = 0.0;
    double2 gx[3], gu[3];
    if (i < ne) {
        const int32_t n0 = conn[3 * i], n1 = conn[3 * i + 1], n2 = conn[3 * i + 2];
        v = tri3_element<false, true, PHYS>(row_of(x_free, x_fixed, x_src[n0]), row_of(x_free, x_fixed, x_src[n1]),
                                            row_of(x_free, x_fixed, x_src[n2]), row_of(u_free, u_fixed, u_src[n0]),
                                            row_of(u_free, u_fixed, u_src[n1]), row_of(u_free, u_fixed, u_src[n2]), k, gx, gu);
    } else if (i < (int64_t)ne + ned) {
        const int64_t g = i - ne;
        const int32_t ni = edges[2 * g], nj = edges[2 * g + 1];
        double2 ex[2], eu[2];
        v = -edge2_element<false>(row_of(x_free, x_fixed, x_src[ni]), row_of(x_free, x_fixed, x_src[nj]),
                                  row_of(u_free, u_fixed, u_src[ni]), row_of(u_free, u_fixed, u_src[nj]),
                                  T_edge ? T_edge[g] : Tconst, ex, eu);
    }
    const double tot = block_sum(v, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kDetBlock) void det_sum_kernel(const double *__restrict__ partials, int n,
                                                            double *__restrict__ out) {
    __shared__ double red[kDetBlock / 64];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += kDetBlock) v += partials[i];
    const double tot = block_sum(v, red);
    if (threadIdx.x == 0) out[0] = tot;
}

template <typename T>
static int det_upload(T **dst, const std::vector<T> &src, size_t min_count = 1) {
    const size_t n = std::max(src.size(), min_count);
    HFEM_HIP_CHECK(hipMalloc((void **)dst, n * sizeof(T)));
    if (!src.empty()) HFEM_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static int det_prepare_impl(hfem_plan *plan);

// First deterministic launch on a plan (TRI3 or QUAD4): build and upload the node -> element adjacency (hipMalloc + synchronous copies).
// Never inside a stream capture (the copies would invalidate it): a capturing caller gets an error and runs ONE eager
// deterministic evaluation first.  A failure part-way releases what was allocated.
int det_prepare(hfem_plan *plan, hipStream_t s) {
    if (plan->det.ready) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        set_error("HFEM_FLAG_DETERMINISTIC: the first deterministic launch on a plan allocates and uploads its adjacency and "
                  "cannot run inside a hipGraph capture; evaluate once eagerly before capturing");
        return -1;
    }
    const int rc = det_prepare_impl(plan);
    if (rc) free_tri3_det(plan);
    return rc;
}

static int det_prepare_impl(hfem_plan *plan) {
    hfem_plan::Det &D = plan->det;
    const HostPlan &h = plan->host;
    const int64_t ne = h.ne, nn = h.nn, ned = h.ned;
    const int npe = h.npe;
    HFEM_ARG_CHECK(ne < (1 << 29), "deterministic path: fewer than 2^29 elements");
    // node -> (element, corner) and node -> (edge, end) adjacency, ascending element / edge id: the fixed order
    std::vector<int32_t> aptr(nn + 1, 0), adj(npe * ne), eptr(nn + 1, 0), eadj(2 * ned);
    for (int64_t i = 0; i < npe * ne; ++i) aptr[h.conn32[i] + 1]++;
    for (int64_t i = 0; i < 2 * ned; ++i) eptr[h.edges32[i] + 1]++;
    for (int64_t n = 0; n < nn; ++n) { aptr[n + 1] += aptr[n]; eptr[n + 1] += eptr[n]; }
    {
        std::vector<int32_t> f(aptr.begin(), aptr.end() - 1), fe(eptr.begin(), eptr.end() - 1);
        for (int64_t e = 0; e < ne; ++e)
            for (int c = 0; c < npe; ++c) adj[f[h.conn32[npe * e + c]]++] = (int32_t)(e << 2 | c);
        for (int64_t g = 0; g < ned; ++g)
            for (int c = 0; c < 2; ++c) eadj[fe[h.edges32[2 * g + c]]++] = (int32_t)(g << 1 | c);
    }
    if (int rc = det_upload(&D.conn, h.conn32)) return rc;
    if (int rc = det_upload(&D.x_src, h.x_src_g)) return rc;
    if (int rc = det_upload(&D.u_src, h.u_src_g)) return rc;
    if (int rc = det_upload(&D.edges, h.edges32)) return rc;
    if (int rc = det_upload(&D.adj_ptr, aptr)) return rc;
    if (int rc = det_upload(&D.adj, adj)) return rc;
    if (int rc = det_upload(&D.eadj_ptr, eptr)) return rc;
    if (int rc = det_upload(&D.eadj, eadj)) return rc;
    D.n_blocks = (int)((ne + ned + kDetBlock - 1) / kDetBlock);
    HFEM_HIP_CHECK(hipMalloc((void **)&D.partials, std::max(D.n_blocks, 1) * sizeof(double)));
    D.ready = true;
    return 0;
}

void free_tri3_det(hfem_plan *plan) {
    hfem_plan::Det &D = plan->det;
    (void)hipFree(D.conn); (void)hipFree(D.x_src); (void)hipFree(D.u_src); (void)hipFree(D.edges);
    (void)hipFree(D.adj_ptr); (void)hipFree(D.adj); (void)hipFree(D.eadj_ptr); (void)hipFree(D.eadj);
    (void)hipFree(D.partials);
    D = hfem_plan::Det();
}

int launch_tri3_det(hfem_plan *plan, const double *x_free, const double *x_fixed, const double *u_free,
                    const double *u_fixed, const Tri3Consts &kc, const double *T_edge, double4 tc, double *loss_out,
                    double *gx_free, double *gu_free, int skip_edges, bool phys, hipStream_t s) {
    HFEM_ARG_CHECK(plan->host.npe == 3, "launch_tri3_det: TRI3 plans");
    if (int rc = det_prepare(plan, s)) return rc;
    const hfem_plan::Det &D = plan->det;
    const HostPlan &h = plan->host;
    const int32_t nn = (int32_t)h.nn, ne = (int32_t)h.ne, ned = skip_edges ? 0 : (int32_t)h.ned;
    if (nn > 0 && (gx_free || gu_free)) {
        const dim3 grid((nn + kDetBlock - 1) / kDetBlock);
#define HFEM_DET_GRAD(P)                                                                                              \
    hipLaunchKernelGGL((tri3_det_grad_kernel<P>), grid, dim3(kDetBlock), 0, s, nn, D.conn, D.x_src, D.u_src, D.adj_ptr, \
                       D.adj, D.edges, D.eadj_ptr, D.eadj, (const double2 *)x_free, (const double2 *)x_fixed,         \
                       (const double2 *)u_free, (const double2 *)u_fixed, kc, (const double4 *)T_edge, tc, skip_edges, \
                       (double2 *)gx_free, (double2 *)gu_free)
        if (phys) HFEM_DET_GRAD(true); else HFEM_DET_GRAD(false);
#undef HFEM_DET_GRAD
        if (int rc = launch_status("hfem_tri3_energy_plan(deterministic gradients)")) return rc;
    }
    const int nb = (int)(((int64_t)ne + ned + kDetBlock - 1) / kDetBlock);
    if (nb > 0) {
#define HFEM_DET_LOSS(P)                                                                                            \
    hipLaunchKernelGGL((tri3_det_loss_kernel<P>), dim3(nb), dim3(kDetBlock), 0, s, ne, ned, D.conn, D.x_src, D.u_src, \
                       D.edges, (const double2 *)x_free, (const double2 *)x_fixed, (const double2 *)u_free,         \
                       (const double2 *)u_fixed, kc, (const double4 *)T_edge, tc, D.partials)
        if (phys) HFEM_DET_LOSS(true); else HFEM_DET_LOSS(false);
#undef HFEM_DET_LOSS
    }
    hipLaunchKernelGGL(det_sum_kernel, dim3(1), dim3(kDetBlock), 0, s, D.partials, nb, loss_out);
    return launch_status("hfem_tri3_energy_plan(deterministic loss)");
}

}  // namespace hfem

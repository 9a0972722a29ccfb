// Fused TRI3 + EDGE2 energy, forward + hand-derived backward, gfx950 (MI355X).
//
// Replaces EnergyLoss2D.__call__ + loss.backward() of the reference
// (/root/reference/src/loss.py:55-116 over /root/reference/src/models.py:292-376,
// ~308 ATen ops per evaluation) with ONE element kernel (+ a tiny reduction).
//
//  * tri3_energy_atomic_kernel / edge2_energy_atomic_kernel: planless,
//    one thread per element, fp64 global atomics.  Simple, any mesh order.
//  * tri3_energy_tiled_kernel: the fast path.  One workgroup per tile of the
//    owner-computes plan (plan.cpp): node data gathered once into LDS through
//    the free/fixed row maps, elements read packed 10-bit local indices, gradients
//    accumulate in LDS (ds_add_f64), every owned gradient row leaves with one plain
//    16-B store.  Bandwidth-bound, no MFMA (2x2 / 2x3 contractions).
#include <hip/hip_runtime.h>

#include <cstring>
#include <memory>

#include "hfem_device.h"

namespace hfem {

constexpr int kBlock = 256;   // 4 wavefronts

// ------------------------------------------------------------------ planless
__global__ __launch_bounds__(kBlock) void tri3_energy_atomic_kernel(
    const double2 *__restrict__ X, const double2 *__restrict__ U, const int32_t *__restrict__ conn,
    int64_t e_begin, int64_t e_end, Tri3Consts k, double *__restrict__ loss_acc,
    double *__restrict__ gX, double *__restrict__ gU) {
    __shared__ double red[kBlock / 64];
    double e_loc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = e_begin + (int64_t)blockIdx.x * kBlock + threadIdx.x; e < e_end; e += stride) {
        const int32_t n0 = conn[3 * e], n1 = conn[3 * e + 1], n2 = conn[3 * e + 2];
        double2 gx[3], gu[3];
        if (gX) {
            e_loc += tri3_element<true>(X[n0], X[n1], X[n2], U[n0], U[n1], U[n2], k, gx, gu);
            const int32_t n[3] = {n0, n1, n2};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                unsafeAtomicAdd(&gX[2 * (int64_t)n[j]], gx[j].x);
                unsafeAtomicAdd(&gX[2 * (int64_t)n[j] + 1], gx[j].y);
                unsafeAtomicAdd(&gU[2 * (int64_t)n[j]], gu[j].x);
                unsafeAtomicAdd(&gU[2 * (int64_t)n[j] + 1], gu[j].y);
            }
        } else {
            e_loc += tri3_element<false>(X[n0], X[n1], X[n2], U[n0], U[n1], U[n2], k, gx, gu);
        }
    }
    const double tot = block_sum(e_loc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_acc, tot);
}

__global__ __launch_bounds__(kBlock) void edge2_energy_atomic_kernel(
    const double2 *__restrict__ X, const double2 *__restrict__ U, const int32_t *__restrict__ edges,
    int64_t ned, const double4 *__restrict__ T, double4 Tconst, double *__restrict__ loss_acc,
    double *__restrict__ gX, double *__restrict__ gU) {
    __shared__ double red[kBlock / 64];
    double w_loc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < ned; g += stride) {
        const int32_t i = edges[2 * g], j = edges[2 * g + 1];
        const double4 t = T ? T[g] : Tconst;
        double2 gx[2], gu[2];
        if (gX) {
            w_loc += edge2_element<true>(X[i], X[j], U[i], U[j], t, gx, gu);
            const int32_t n[2] = {i, j};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                unsafeAtomicAdd(&gX[2 * (int64_t)n[q]], gx[q].x);
                unsafeAtomicAdd(&gX[2 * (int64_t)n[q] + 1], gx[q].y);
                unsafeAtomicAdd(&gU[2 * (int64_t)n[q]], gu[q].x);
                unsafeAtomicAdd(&gU[2 * (int64_t)n[q] + 1], gu[q].y);
            }
        } else {
            w_loc += edge2_element<false>(X[i], X[j], U[i], U[j], t, gx, gu);
        }
    }
    const double tot = block_sum(w_loc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_acc, -tot);
}

// ------------------------------------------------------------------ tiled plan
struct PlanDev {
    const TileDesc *tiles;
    const uint32_t *elem_pack;
    const int2 *node_src;
    const uint32_t *edge_pack;
    const int32_t *edge_gid;
};

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an
// L2).  Map block -> tile so that each XCD walks one contiguous run of the
// Morton-ordered tiles: neighbouring tiles share halo nodes, which then hit the
// same L2.  Bijective for any grid size; placement only affects speed.
__device__ __forceinline__ int xcd_tile(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// LDS: xy[cap_nodes] double2 | uv[cap_nodes] double2 | acc[4][cap_owned] double | red[4]
__global__ __launch_bounds__(kBlock) void tri3_energy_tiled_kernel(
    PlanDev pd, int tile_begin, const double2 *__restrict__ x_free,
    const double2 *__restrict__ x_fixed, const double2 *__restrict__ u_free,
    const double2 *__restrict__ u_fixed, Tri3Consts k, const double4 *__restrict__ T_edge,
    double4 Tconst, double *__restrict__ partials, double2 *__restrict__ gx_free,
    double2 *__restrict__ gu_free, int cap_nodes, int cap_owned, int skip_edges) {
    extern __shared__ double2 lds[];
    double2 *nd_xy = lds;
    double2 *nd_uv = lds + cap_nodes;
    double *acc0 = reinterpret_cast<double *>(lds + 2 * cap_nodes);
    double *acc1 = acc0 + cap_owned, *acc2 = acc1 + cap_owned, *acc3 = acc2 + cap_owned;
    double *red = acc3 + cap_owned;

    const int tid = threadIdx.x;
    const int slot = xcd_tile(blockIdx.x, gridDim.x);
    const TileDesc d = pd.tiles[tile_begin + slot];

    // ---- phase 1: gather node data through the free/fixed maps, clear accumulators
    const int2 *src = pd.node_src + d.node_off;
    for (int l = tid; l < d.n_node; l += kBlock) {
        const int2 s = src[l];
        nd_xy[l] = s.x >= 0 ? x_free[s.x] : x_fixed[~s.x];
        nd_uv[l] = s.y >= 0 ? u_free[s.y] : u_fixed[~s.y];
    }
    for (int l = tid; l < d.n_owned; l += kBlock) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
    __syncthreads();

    // ---- phase 2: elements (home + halo), then the tile's Neumann edges
    double e_loc = 0.0;
    const uint32_t *ep = pd.elem_pack + d.elem_off;
    const int n_owned = d.n_owned;
    for (int i = tid; i < d.n_elem; i += kBlock) {
        const uint32_t p = ep[i];
        const int l[3] = {(int)(p & kLocalMask), (int)((p >> kLocalBits) & kLocalMask),
                          (int)((p >> (2 * kLocalBits)) & kLocalMask)};
        double2 gx[3], gu[3];
        const double e = tri3_element<true>(nd_xy[l[0]], nd_xy[l[1]], nd_xy[l[2]], nd_uv[l[0]],
                                            nd_uv[l[1]], nd_uv[l[2]], k, gx, gu);
        if (p & kHomeBit) e_loc += e;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (l[j] < n_owned) {
                unsafeAtomicAdd(&acc0[l[j]], gx[j].x);
                unsafeAtomicAdd(&acc1[l[j]], gx[j].y);
                unsafeAtomicAdd(&acc2[l[j]], gu[j].x);
                unsafeAtomicAdd(&acc3[l[j]], gu[j].y);
            }
    }
    const int n_edge = skip_edges ? 0 : d.n_edge;
    for (int i = tid; i < n_edge; i += kBlock) {
        const uint32_t p = pd.edge_pack[d.edge_off + i];
        const int l[2] = {(int)(p & kLocalMask), (int)((p >> kLocalBits) & kLocalMask)};
        const double4 t = T_edge ? T_edge[pd.edge_gid[d.edge_off + i]] : Tconst;
        double2 gx[2], gu[2];
        const double w = edge2_element<true>(nd_xy[l[0]], nd_xy[l[1]], nd_uv[l[0]], nd_uv[l[1]], t, gx, gu);
        if (p & kHomeBit) e_loc -= w;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (l[j] < n_owned) {
                unsafeAtomicAdd(&acc0[l[j]], gx[j].x);
                unsafeAtomicAdd(&acc1[l[j]], gx[j].y);
                unsafeAtomicAdd(&acc2[l[j]], gu[j].x);
                unsafeAtomicAdd(&acc3[l[j]], gu[j].y);
            }
    }
    __syncthreads();

    // ---- phase 3: every owned gradient row is written exactly once
    for (int l = tid; l < n_owned; l += kBlock) {
        const int2 s = src[l];
        if (gx_free && s.x >= 0) gx_free[s.x] = make_double2(acc0[l], acc1[l]);
        if (gu_free && s.y >= 0) gu_free[s.y] = make_double2(acc2[l], acc3[l]);
    }
    const double tot = block_sum(e_loc, red);
    if (tid == 0) partials[slot] = tot;
}

// Deterministic sum of the per-tile partial energies (fixed order).
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const double *__restrict__ partials, int n,
                                                             double *__restrict__ out) {
    __shared__ double red[kBlock / 64];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) v += partials[i];
    const double tot = block_sum(v, red);
    if (threadIdx.x == 0) out[0] = tot;
}

}  // namespace hfem

// ====================================================================== C ABI
using namespace hfem;

struct hfem_plan {
    HostPlan host;
    int device = -1;
    // device mirrors
    TileDesc *d_tiles = nullptr;
    uint32_t *d_elem_pack = nullptr;
    int2 *d_node_src = nullptr;
    uint32_t *d_edge_pack = nullptr;
    int32_t *d_edge_gid = nullptr;
    double *d_partials = nullptr;
    int64_t device_bytes = 0;
    int32_t lds_bytes = 0;
};

static Tri3Consts make_consts(const double mat[4], double W, const double Bk[6]) {
    Tri3Consts k;
    k.c11 = mat[0]; k.c12 = mat[1]; k.c22 = mat[2]; k.c33 = mat[3];
    k.W = W;
    for (int i = 0; i < 6; ++i) k.Bk[i] = Bk ? Bk[i] : 0.0;
    return k;
}

static int grid_for(int64_t n, int cap = 256 * 8) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

extern "C" int hfem_tri3_energy_atomic(int device, const double *X, const double *U, const int32_t *conn,
                                       int64_t e_begin, int64_t e_end, int64_t nn, const double mat[4],
                                       double W, const double Bk[6], double *loss_acc, double *gX,
                                       double *gU, void *stream) {
    HFEM_ARG_CHECK(X && U && conn && mat && loss_acc, "null pointer");
    HFEM_ARG_CHECK(e_begin >= 0 && e_end >= e_begin && nn >= 0, "bad element range");
    HFEM_ARG_CHECK((gX == nullptr) == (gU == nullptr), "gX and gU must both be given or both NULL");
    if (int rc = use_device(device)) return rc;
    if (e_end == e_begin) return 0;
    hipLaunchKernelGGL(tri3_energy_atomic_kernel, dim3(grid_for(e_end - e_begin)), dim3(kBlock), 0,
                       (hipStream_t)stream, (const double2 *)X, (const double2 *)U, conn, e_begin, e_end,
                       make_consts(mat, W, Bk), loss_acc, gX, gU);
    return launch_status("hfem_tri3_energy_atomic");
}

extern "C" int hfem_edge2_energy_atomic(int device, const double *X, const double *U, const int32_t *edges,
                                        int64_t ned, const double *T, const double Tconst[4],
                                        double *loss_acc, double *gX, double *gU, void *stream) {
    HFEM_ARG_CHECK(X && U && loss_acc && (edges || ned == 0), "null pointer");
    HFEM_ARG_CHECK(T || Tconst, "need a per-edge traction table or a constant one");
    HFEM_ARG_CHECK(ned >= 0, "negative edge count");
    HFEM_ARG_CHECK((gX == nullptr) == (gU == nullptr), "gX and gU must both be given or both NULL");
    if (int rc = use_device(device)) return rc;
    if (ned == 0) return 0;
    const double4 tc = Tconst ? make_double4(Tconst[0], Tconst[1], Tconst[2], Tconst[3]) : make_double4(0, 0, 0, 0);
    hipLaunchKernelGGL(edge2_energy_atomic_kernel, dim3(grid_for(ned)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const double2 *)X, (const double2 *)U, edges, ned, (const double4 *)T, tc, loss_acc,
                       gX, gU);
    return launch_status("hfem_edge2_energy_atomic");
}

template <typename T>
static int upload(T **dst, const void *src, size_t count, int64_t &bytes) {
    const size_t nb = std::max<size_t>(count, 1) * sizeof(T);
    HFEM_HIP_CHECK(hipMalloc((void **)dst, nb));
    if (count && src) HFEM_HIP_CHECK(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    if (!src) HFEM_HIP_CHECK(hipMemset(*dst, 0, nb));
    bytes += (int64_t)nb;
    return 0;
}

extern "C" int hfem_plan_destroy(hfem_plan *plan) {
    if (!plan) return 0;
    if (plan->device >= 0) {
        (void)hipSetDevice(plan->device);
        (void)hipFree(plan->d_tiles);
        (void)hipFree(plan->d_elem_pack);
        (void)hipFree(plan->d_node_src);
        (void)hipFree(plan->d_edge_pack);
        (void)hipFree(plan->d_edge_gid);
        (void)hipFree(plan->d_partials);
    }
    delete plan;
    return 0;
}

extern "C" int hfem_plan_create(int device, const int64_t *conn, int64_t ne, int64_t nn,
                                const double *coords_hint, const int32_t *x_src, const int32_t *u_src,
                                const int64_t *edges, int64_t ned, int32_t tile_elems, hfem_plan **out) {
    HFEM_ARG_CHECK(out, "null out pointer");
    *out = nullptr;
    std::unique_ptr<hfem_plan> p(new hfem_plan);
    if (build_host_plan(conn, ne, nn, coords_hint, x_src, u_src, edges, ned, tile_elems, p->host)) return -1;
    const HostPlan &h = p->host;
    p->lds_bytes = h.max_nodes * 32 + h.max_owned * 32 + 64;
    if (device >= 0) {
        if (int rc = use_device(device)) return rc;
        p->device = device;
        hfem_plan *raw = p.get();
        int rc = 0;
        if (!rc) rc = upload(&raw->d_tiles, h.tiles.data(), h.tiles.size(), raw->device_bytes);
        if (!rc) rc = upload(&raw->d_elem_pack, h.elem_pack.data(), h.elem_pack.size(), raw->device_bytes);
        if (!rc) rc = upload(&raw->d_node_src, h.node_src.data(), h.node_src.size() / 2, raw->device_bytes);
        if (!rc) rc = upload(&raw->d_edge_pack, h.edge_pack.data(), h.edge_pack.size(), raw->device_bytes);
        if (!rc) rc = upload(&raw->d_edge_gid, h.edge_gid.data(), h.edge_gid.size(), raw->device_bytes);
        if (!rc) rc = upload(&raw->d_partials, nullptr, h.tiles.size(), raw->device_bytes);
        if (!rc) {
            hipError_t e = hipFuncSetAttribute((const void *)tri3_energy_tiled_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, p->lds_bytes);
            if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); rc = (int)e; }
        }
        if (rc) { hfem_plan_destroy(p.release()); return rc; }
    }
    *out = p.release();
    return 0;
}

extern "C" int hfem_plan_get_stats(const hfem_plan *plan, hfem_plan_stats *out) {
    HFEM_ARG_CHECK(plan && out, "null pointer");
    const HostPlan &h = plan->host;
    std::memset(out, 0, sizeof(*out));
    out->n_elems = h.ne; out->n_nodes = h.nn; out->n_edges = h.ned;
    out->n_tiles = (int32_t)h.tiles.size();
    out->tile_elems = h.tile_elems;
    out->tile_elem_total = (int64_t)h.elem_pack.size();
    out->tile_node_total = (int64_t)h.node_src.size() / 2;
    out->max_tile_nodes = h.max_nodes; out->max_tile_owned = h.max_owned;
    out->max_tile_elems = h.max_elems; out->max_tile_edges = h.max_edges;
    out->device_bytes = plan->device_bytes;
    out->lds_bytes = plan->lds_bytes;
    return 0;
}

extern "C" int64_t hfem_plan_export(const hfem_plan *plan, int which, void *buf, int64_t cap_elems) {
    if (!plan) { set_error("hfem_plan_export: null plan"); return -1; }
    const HostPlan &h = plan->host;
    const void *src = nullptr;
    int64_t n = 0;
    size_t esz = 4;
    switch (which) {
        case 0: src = h.tiles.data(); n = (int64_t)h.tiles.size() * 8; break;
        case 1: src = h.elem_pack.data(); n = (int64_t)h.elem_pack.size(); break;
        case 2: src = h.node_src.data(); n = (int64_t)h.node_src.size(); break;
        case 3: src = h.edge_pack.data(); n = (int64_t)h.edge_pack.size(); break;
        case 4: src = h.edge_gid.data(); n = (int64_t)h.edge_gid.size(); break;
        case 5: src = h.elem_gid.data(); n = (int64_t)h.elem_gid.size(); break;
        default: set_error("hfem_plan_export: unknown array id"); return -1;
    }
    if (buf) {
        if (cap_elems < n) { set_error("hfem_plan_export: buffer too small"); return -1; }
        if (n) std::memcpy(buf, src, (size_t)n * esz);
    }
    return n;
}

extern "C" int hfem_tri3_energy_plan(hfem_plan *plan, const double *x_free, const double *x_fixed,
                                     const double *u_free, const double *u_fixed, const double mat[4],
                                     double W, const double Bk[6], const double *T_edge,
                                     const double Tconst[4], int32_t tile_begin, int32_t tile_end,
                                     double *loss_out, double *gx_free, double *gu_free, int32_t flags,
                                     void *stream) {
    HFEM_ARG_CHECK(plan && mat && loss_out, "null pointer");
    HFEM_ARG_CHECK(plan->device >= 0, "host-only plan (created with device < 0) cannot launch");
    HFEM_ARG_CHECK(x_free && u_free, "x_free / u_free must be given");
    const int32_t nt = (int32_t)plan->host.tiles.size();
    if (tile_end < 0) tile_end = nt;
    HFEM_ARG_CHECK(tile_begin >= 0 && tile_begin <= tile_end && tile_end <= nt, "bad tile range");
    HFEM_ARG_CHECK(plan->host.ned == 0 || T_edge || Tconst, "plan has Neumann edges: need a traction table");
    if (int rc = use_device(plan->device)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int n = tile_end - tile_begin;
    if (n > 0) {
        PlanDev pd{plan->d_tiles, plan->d_elem_pack, plan->d_node_src, plan->d_edge_pack, plan->d_edge_gid};
        const double4 tc = Tconst ? make_double4(Tconst[0], Tconst[1], Tconst[2], Tconst[3]) : make_double4(0, 0, 0, 0);
        hipLaunchKernelGGL(tri3_energy_tiled_kernel, dim3(n), dim3(kBlock), (size_t)plan->lds_bytes, s, pd,
                           (int)tile_begin, (const double2 *)x_free, (const double2 *)x_fixed,
                           (const double2 *)u_free, (const double2 *)u_fixed, make_consts(mat, W, Bk),
                           (const double4 *)T_edge, tc, plan->d_partials + tile_begin,
                           (flags & HFEM_FLAG_NO_GX) ? nullptr : (double2 *)gx_free,
                           (flags & HFEM_FLAG_NO_GU) ? nullptr : (double2 *)gu_free, plan->host.max_nodes,
                           plan->host.max_owned, (flags & HFEM_FLAG_NO_EDGES) ? 1 : 0);
        if (int rc = launch_status("hfem_tri3_energy_plan")) return rc;
    }
    if (flags & HFEM_FLAG_NO_LOSS_SUM) return 0;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, s, plan->d_partials + tile_begin, n, loss_out);
    return launch_status("hfem_tri3_energy_plan(sum)");
}

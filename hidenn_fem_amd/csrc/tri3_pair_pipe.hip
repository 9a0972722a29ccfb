#ifdef HFEM_LAB   // lab build only (see the result paragraph in DESIGN.md section 4.1)
// Software-pipelined paired-slot TRI3 + EDGE2 energy kernel, gfx950 (MI355X): tri3_energy_pair_kernel (tri3_pair.hip) with
// SEVERAL TILES PER WORKGROUP, the next tile's memory phase running under the current tile's element loop.
//
// Same contract, closed forms, plan and outputs as tri3_pair.hip (replaces EnergyLoss2D.__call__ + loss.backward() of
// /root/reference/src/loss.py:55-116 over /root/reference/src/models.py:292-376).  Why it exists: the round-2 ablation
// ladder of the one-tile kernel (profiles/r02/r2_lab19.jsonl, r2_lab20.jsonl) is ADDITIVE -- 1.6 us boundary, ~1.8 us of
// index + gather loads (Infinity-Cache bandwidth), 0.6 us LDS fill, 3.4 us element loop (fp64 VALU and the LDS unit both
// ~90 % busy), 0.5-1.5 us write-out -- because the resident workgroups of a CU start together and walk the phases in
// lockstep.  Here a workgroup owns `tpw` consecutive tiles and, while it walks tile t's slots,
//   * tile t+1's row maps and slot records are loaded into registers (plain loads; uniform strides: addresses follow from
//     the tile index),
//   * tile t+1's node rows are gathered by LDS-DMA (global_load_lds_dwordx4, per-lane source row through the row map) into
//     the OTHER half of a double-buffered node array -- no staging registers, nothing the compiler counts,
//   * tile t's gradient stores (branch-free: a static count) drain under tile t+1's loop.
// Every wait on the prefetch is a counted vmcnt(N) that leaves the youngest stores in flight.  Exposed: the first tile's
// loads and the last tile's store drain.
// HBM-bound accounting unchanged: algorithmic bytes per launch 12 Ne + 64 Nn + 8.  No MFMA (2x2 / 2x3 contractions).
#include <hip/hip_runtime.h>

#include "hfem_device.h"
#include "hfem_plan_dev.h"

namespace hfem {

typedef __attribute__((address_space(3))) void lds_void_pp_t;

// One LDS-DMA piece: lane i of the wave copies 16 B from its own global address to lds_dst + 16 i (active lanes only).
// M0 carries the LDS base and is compiler-reserved, so it is saved and restored inside the statement.
__device__ __forceinline__ void pp_glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned pp_lds_addr(const void *p) { return (unsigned)(size_t)(lds_void_pp_t *)p; }

// BLOCK threads; NPT >= ceil(max nodes / BLOCK); EPT >= slots per thread (rows of the slot array).
// LDS: node buffers 0 and 1 (xy[cap_nodes] | uv[cap_nodes] double2 each) | acc[4][cap_owned] | red[8] | descriptors [tpw][8]
template <int BLOCK, int NPT, int EPT, int WPS>
__global__ __launch_bounds__(BLOCK, WPS) void tri3_energy_pair_pipe_kernel(
    PlanDev pd, int tile_begin, int n_tiles, int tpw, const double2 *__restrict__ x_free, const double2 *__restrict__ x_fixed,
    const double2 *__restrict__ u_free, const double2 *__restrict__ u_fixed, Tri3Consts k,
    const double4 *__restrict__ T_edge, double4 Tconst, double *__restrict__ partials,
    double2 *__restrict__ gx_free, double2 *__restrict__ gu_free, int cap_nodes, int cap_owned, int skip_edges,
    LagSum lag, int col_stride) {
    extern __shared__ double2 lds[];
    double *acc0 = reinterpret_cast<double *>(lds + 4 * cap_nodes);
    double *acc1 = acc0 + cap_owned, *acc2 = acc1 + cap_owned, *acc3 = acc2 + cap_owned;
    double *red = acc3 + cap_owned;
    int *dcache = reinterpret_cast<int *>(red + 8);

    const int tid = threadIdx.x, lane = tid & 63;
    const int n_wg = (int)gridDim.x - (lag.prev ? 1 : 0);
    if (lag.prev && (int)blockIdx.x == n_wg) {          // HFEM_FLAG_SUM_PREVIOUS: reduce the previous launch's tile energies
        double v = 0.0;
        if (tid < 256)
            for (int i = tid; i < lag.prev_n; i += 256) v += lag.prev[i];
        const double tot = block_sum(v, red);
        if (tid == 0) lag.out[0] = tot;
        return;
    }
    const int wg = xcd_tile(blockIdx.x, n_wg);          // consecutive workgroups (and their tiles) share an XCD's L2
    const int t_first = wg * tpw, t_end = min(t_first + tpw, n_tiles);
    if (t_first >= t_end) return;

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)gx_free, 0, 0x7FFFFFF0, 0x00020000);
    __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void *)gu_free, 0, 0x7FFFFFF0, 0x00020000);

    int2 s[NPT], sn[NPT];
    uint32_t w0[EPT], w1[EPT], wn0[EPT], wn1[EPT];
    int n_node, n_owned, n_elem, n_edge, edge_off;      // current tile (workgroup-uniform)

#define HFEM_PP_INDEX(T, S, A, B)                                                                    \
    {                                                                                                 \
        const int2 *src_ = pd.node_src + (size_t)(tile_begin + (T)) * pd.node_stride;                 \
        const size_t rec_ = (size_t)(tile_begin + (T)) * pd.elem_stride;                              \
        _Pragma("unroll") for (int j = 0; j < NPT; ++j) S[j] = src_[min(tid + j * BLOCK, pd.node_stride - 1)]; \
        _Pragma("unroll") for (int j = 0; j < EPT; ++j) {                                             \
            const size_t i_ = rec_ + min(tid + j * col_stride, pd.elem_stride - 1);                   \
            A[j] = pd.elem_pack[i_];                                                                  \
            B[j] = pd.elem_pack_hi[i_];                                                               \
        }                                                                                             \
    }
    // gather of a tile with NN nodes into node buffer BUF: each wave copies the rows of its own 64-id pieces
#define HFEM_PP_GATHER(S, NN, BUF)                                                                    \
    _Pragma("unroll") for (int j = 0; j < NPT; ++j) {                                                \
        const int l_ = tid + j * BLOCK;                                                               \
        if (l_ < (NN)) {                                                                              \
            double2 *bx_ = lds + (BUF) * 2 * cap_nodes;                                               \
            const unsigned dx_ = __builtin_amdgcn_readfirstlane(pp_lds_addr(bx_ + (l_ - lane)));            \
            const unsigned du_ = __builtin_amdgcn_readfirstlane(pp_lds_addr(bx_ + cap_nodes + (l_ - lane))); \
            pp_glds16(S[j].x >= 0 ? x_free + S[j].x : x_fixed + ~S[j].x, dx_);                        \
            pp_glds16(S[j].y >= 0 ? u_free + S[j].y : u_fixed + ~S[j].y, du_);                        \
        }                                                                                             \
    }
    auto add_row = [&](int l, const double2 gx, const double2 gu) {
        unsafeAtomicAdd(&acc0[l], gx.x); unsafeAtomicAdd(&acc1[l], gx.y);
        unsafeAtomicAdd(&acc2[l], gu.x); unsafeAtomicAdd(&acc3[l], gu.y);
    };

    // ---- prologue: first tile
    HFEM_PP_INDEX(t_first, s, w0, w1)
    if (tid < 8 * (t_end - t_first)) dcache[tid] = reinterpret_cast<const int *>(pd.tiles + tile_begin + t_first)[tid];
    {
        const TileDesc d = pd.tiles[tile_begin + t_first];
        n_node = d.n_node; n_owned = d.n_owned; n_elem = d.n_elem; n_edge = skip_edges ? 0 : d.n_edge; edge_off = d.edge_off;
    }
    HFEM_PP_GATHER(s, n_node, 0)
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int l = tid + j * BLOCK;
        if (l < n_owned) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (!(tid < col_stride && tid + j * col_stride < n_elem)) { w0[j] = kSkipBit; w1[j] = 0u; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = t_first; t < t_end; ++t) {
        const int cur = (t - t_first) & 1;
        const double2 *nd_xy = lds + cur * 2 * cap_nodes, *nd_uv = nd_xy + cap_nodes;
        const bool more = t + 1 < t_end;                // workgroup-uniform
        int nn_node = 0, nn_owned = 0, nn_elem = 0, nn_edge = 0, nn_edge_off = 0;
        if (more) {
            const int *dc = dcache + 8 * (t + 1 - t_first);
            nn_elem = __builtin_amdgcn_readfirstlane(dc[1]); nn_node = __builtin_amdgcn_readfirstlane(dc[3]);
            nn_owned = __builtin_amdgcn_readfirstlane(dc[4]); nn_edge_off = __builtin_amdgcn_readfirstlane(dc[5]);
            nn_edge = skip_edges ? 0 : __builtin_amdgcn_readfirstlane(dc[6]);
            HFEM_PP_INDEX(t + 1, sn, wn0, wn1)
        }
        uint32_t edge_rec = 0u;
        int edge_id = 0;
        if (tid < n_edge) {                              // boundary tiles only
            edge_rec = pd.edge_pack[edge_off + tid];
            if (T_edge) edge_id = pd.edge_gid[edge_off + tid];
        }
        // ---- slots: registers + LDS only; the next tile's gather goes out after the first row (its row maps have landed)
        double e_loc = 0.0;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const uint32_t p = w0[j], q = w1[j];
            if (!(p & kSkipBit)) {
                const int ln = (int)(p & kLocalMask), lb = (int)((p >> kLocalBits) & kLocalMask),
                          lc = (int)((p >> (2 * kLocalBits)) & kLocalMask);
                const double2 Xn = nd_xy[ln], Un = nd_uv[ln], Xc = nd_xy[lc], Uc = nd_uv[lc];
                double2 sxn, sun, sxc, suc;             // running rows of the shared nodes n and c
                {
                    double2 gx[3], gu[3];
                    const double e = tri3_element<true, false, false>(Xn, nd_xy[lb], Xc, Un, nd_uv[lb], Uc, k, gx, gu);
                    if (p & kHomeBit) e_loc += e;
                    if (lb < n_owned) add_row(lb, gx[1], gu[1]);
                    sxn = gx[0]; sun = gu[0]; sxc = gx[2]; suc = gu[2];
                }
                if (q & (1u << 10)) {                   // B = (n, c, d)
                    const int ld = (int)(q & kLocalMask);
                    double2 gx[3], gu[3];
                    const double e = tri3_element<true, false, false>(Xn, Xc, nd_xy[ld], Un, Uc, nd_uv[ld], k, gx, gu);
                    if (q & (1u << 11)) e_loc += e;
                    if (ld < n_owned) add_row(ld, gx[2], gu[2]);
                    sxn.x += gx[0].x; sxn.y += gx[0].y; sun.x += gu[0].x; sun.y += gu[0].y;
                    sxc.x += gx[1].x; sxc.y += gx[1].y; suc.x += gu[1].x; suc.y += gu[1].y;
                }
                if (ln < n_owned) add_row(ln, sxn, sun);
                if (lc < n_owned) add_row(lc, sxc, suc);
            }
            if (j == 0 && more) { HFEM_PP_GATHER(sn, nn_node, cur ^ 1) }
        }
        for (int i = tid; i < n_edge; i += BLOCK) {      // boundary tiles only
            const uint32_t p = i == tid ? edge_rec : pd.edge_pack[edge_off + i];
            const int l0 = (int)(p & kLocalMask), l1 = (int)((p >> kLocalBits) & kLocalMask);
            const double4 tt = T_edge ? T_edge[i == tid ? edge_id : pd.edge_gid[edge_off + i]] : Tconst;
            double2 gx[2], gu[2];
            const double wk = edge2_element<true>(nd_xy[l0], nd_xy[l1], nd_uv[l0], nd_uv[l1], tt, gx, gu);
            if (p & kHomeBit) e_loc -= wk;
            if (l0 < n_owned) add_row(l0, gx[0], gu[0]);
            if (l1 < n_owned) add_row(l1, gx[1], gu[1]);
        }
        {
            const double w = wave_sum(e_loc);
            if ((tid & 63) == 0) red[tid >> 6] = w;
        }
        __syncthreads();

        // ---- write-out, branch-free: an idle lane stores past the buffer's range (dropped), so the store COUNT is static and
        //      the counted wait below skips exactly these stores; they drain under the next tile's loop
        if (tid == 0) {                                  // fixed order: the tile energy is bit-reproducible
            double tile_e = 0.0;
#pragma unroll
            for (int w = 0; w < BLOCK / 64; ++w) tile_e += red[w];
            partials[t] = tile_e;
        }
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int l = tid + j * BLOCK;
            const int lc = min(l, cap_owned - 1);
            const bool own = l < n_owned;
            double2 v;
            v.x = acc0[lc]; v.y = acc1[lc];
            const unsigned ox = (own && gx_free && s[j].x >= 0) ? (unsigned)s[j].x * 16u : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), rx, ox, 0, 16);
            v.x = acc2[lc]; v.y = acc3[lc];
            const unsigned ou = (own && gu_free && s[j].y >= 0) ? (unsigned)s[j].y * 16u : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), ru, ou, 0, 16);
        }
        if (more) {
            __syncthreads();                             // the accumulators and `red` are read: clear them for the next tile
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                const int l = tid + j * BLOCK;
                if (l < nn_owned) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
                s[j] = sn[j];
            }
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const bool live = tid < col_stride && tid + j * col_stride < nn_elem;
                w0[j] = live ? wn0[j] : kSkipBit;
                w1[j] = live ? wn1[j] : 0u;
            }
            n_node = nn_node; n_owned = nn_owned; n_elem = nn_elem; n_edge = nn_edge; edge_off = nn_edge_off;
            // everything older than this tile's 2 NPT gradient stores has landed: the next tile's records and its DMA pieces
            if (NPT == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (NPT == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __syncthreads();
        }
    }
#undef HFEM_PP_INDEX
#undef HFEM_PP_GATHER
}

template <int NPT, int EPT, int WPS>
static void launch_pipe_inst(const PairLaunch &A, const LagSum &lag, int n_tiles, int tpw, int grid) {
    const size_t lds = (size_t)A.max_nodes * 64 + (size_t)A.max_owned * 32 + 64 + 32 * kPipeMaxTiles;
    hipLaunchKernelGGL((tri3_energy_pair_pipe_kernel<256, NPT, EPT, WPS>), dim3(grid), dim3(256), lds, A.s, A.pd, A.tile_begin,
                       n_tiles, tpw, (const double2 *)A.x_free, (const double2 *)A.x_fixed, (const double2 *)A.u_free,
                       (const double2 *)A.u_fixed, A.k, A.T_edge, A.tc, A.partials, (double2 *)A.gx, (double2 *)A.gu,
                       A.max_nodes, A.max_owned, A.skip_edges, lag, A.col_stride);
}

// Pipelined launch over n_tiles tiles starting at A.tile_begin, tpw tiles per workgroup.  fp64, reference convention, zero
// body force, plain (unchained) paired plans.  1 = launched, 0 = the plan's tile shape has no instance.
int launch_tri3_pair_pipe(const hfem_plan *plan, PairLaunch A, int n_tiles, int tpw, const LagSum &lag) {
    const HostPlan &h = plan->host;
    if (!h.paired || h.n_chained > 0 || !plan->d_elem_pack_hi || tpw < 1 || tpw > kPipeMaxTiles || n_tiles < 1) return 0;
    A.pd = plan_dev(plan);
    A.max_nodes = (h.max_nodes + 63) / 64 * 64;          // whole 64-id DMA pieces
    A.max_owned = h.max_owned;
    A.col_stride = h.col_stride;
    const int npt = (h.max_nodes + 255) / 256, ept = h.max_rows;
    const int grid = (n_tiles + tpw - 1) / tpw + (lag.prev ? 1 : 0);
    if (npt > 3 || ept > 3 || ept < 1) return 0;
    if ((size_t)A.max_nodes * 64 + (size_t)A.max_owned * 32 + 64 + 32 * kPipeMaxTiles > 64 * 1024) return 0;
    const bool w2 = plan->tune.pair_pipe_wps <= 3;       // register budget: 2 or 4 waves per SIMD
    if (npt <= 2 && ept <= 2) { if (w2) launch_pipe_inst<2, 2, 2>(A, lag, n_tiles, tpw, grid); else launch_pipe_inst<2, 2, 4>(A, lag, n_tiles, tpw, grid); }
    else if (npt <= 2) { if (w2) launch_pipe_inst<2, 3, 2>(A, lag, n_tiles, tpw, grid); else launch_pipe_inst<2, 3, 4>(A, lag, n_tiles, tpw, grid); }
    else { if (w2) launch_pipe_inst<3, 3, 2>(A, lag, n_tiles, tpw, grid); else launch_pipe_inst<3, 3, 4>(A, lag, n_tiles, tpw, grid); }
    return 1;
}

}  // namespace hfem
#endif  // HFEM_LAB

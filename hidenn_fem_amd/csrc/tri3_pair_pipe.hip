// Software-pipelined paired-slot TRI3 + EDGE2 energy kernel, gfx950 (MI355X): tri3_energy_pair_kernel (tri3_pair.hip) with
// SEVERAL TILES PER WORKGROUP, the next tile's memory phase running under the current tile's element loop.
//
// Same contract, closed forms, plan and outputs as tri3_pair.hip (replaces EnergyLoss2D.__call__ + loss.backward() of
// /root/reference/src/loss.py:55-116 over /root/reference/src/models.py:292-376).  Why it exists: the round-2 ablation
// ladder of the one-tile kernel (profiles/r02/r2_lab19.jsonl, r2_lab20.jsonl) is ADDITIVE -- 1.6 us boundary, ~1.9 us of
// index + gather loads, 0.6 us LDS fill, 3.4 us element loop (fp64 VALU and the LDS unit both ~90 % busy), 0.5-1.5 us
// write-out -- because the four resident workgroups of a CU start together and walk the phases in lockstep: nothing
// overlaps the loads.  Here a workgroup owns `tpw` consecutive tiles (smaller ones: plan_node_cap), keeps tile t+1's row
// maps, slot records and gathered rows in REGISTERS (uniform node / slot strides: all addresses follow from the tile index,
// no descriptor round trip), and fills LDS from them after tile t's write-out; tile t's gradient stores drain under tile
// t+1's loop.  Exposed: the first tile's loads and the last tile's store drain.
// HBM-bound accounting unchanged: algorithmic bytes per launch 12 Ne + 64 Nn + 8.  No MFMA (2x2 / 2x3 contractions).
#ifdef HFEM_LAB   // lab build only: measured slower than the one-tile kernel, kept as the evidence
#include <hip/hip_runtime.h>

#include "hfem_device.h"
#include "hfem_plan_dev.h"

namespace hfem {

// BLOCK threads; NPT >= ceil(max nodes / BLOCK); EPT >= slots per thread (rows of the slot array); CAPO > 0: compile-time
// stride of the accumulator arrays.  LDS layout as tri3_energy_pair_kernel.
template <int BLOCK, int NPT, int EPT, int WPS, int CAPO>
__global__ __launch_bounds__(BLOCK, WPS) void tri3_energy_pair_pipe_kernel(
    PlanDev pd, int tile_begin, int n_tiles, int tpw, const double2 *__restrict__ x_free, const double2 *__restrict__ x_fixed,
    const double2 *__restrict__ u_free, const double2 *__restrict__ u_fixed, Tri3Consts k,
    const double4 *__restrict__ T_edge, double4 Tconst, double *__restrict__ partials,
    double2 *__restrict__ gx_free, double2 *__restrict__ gu_free, int cap_nodes, int cap_owned_rt, int skip_edges,
    LagSum lag, int col_stride) {
    const int cap_owned = CAPO > 0 ? CAPO : cap_owned_rt;
    extern __shared__ double2 lds[];
    double2 *nd_xy = lds;
    double2 *nd_uv = lds + cap_nodes;
    double *acc0 = reinterpret_cast<double *>(lds + 2 * cap_nodes);
    double *acc1 = acc0 + cap_owned, *acc2 = acc1 + cap_owned, *acc3 = acc2 + cap_owned;
    double *red = acc3 + cap_owned;
    int *dcache = reinterpret_cast<int *>(red + 8);     // [tpw][8] tile descriptors of this workgroup (<= kPipeMaxTiles)

    const int tid = threadIdx.x;
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

    // ---- registers of the tile being worked on, and of the next one
    int2 s[NPT], sn[NPT];
    uint32_t w0[EPT], w1[EPT], wn0[EPT], wn1[EPT];
    double vxx[NPT], vxy[NPT], vux[NPT], vuy[NPT], nxx[NPT], nxy[NPT], nux[NPT], nuy[NPT];   // plain doubles: double2 arrays went to scratch
    TileDesc d;

    auto load_index = [&](int t, int2 (&S)[NPT], uint32_t (&A)[EPT], uint32_t (&B)[EPT]) {
        const int2 *src = pd.node_src + (size_t)(tile_begin + t) * pd.node_stride;
        const size_t rec0 = (size_t)(tile_begin + t) * pd.elem_stride;
#pragma unroll
        for (int j = 0; j < NPT; ++j) S[j] = src[tid + j * BLOCK];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            A[j] = pd.elem_pack[rec0 + tid + j * col_stride];
            B[j] = pd.elem_pack_hi[rec0 + tid + j * col_stride];
        }
    };
#define HFEM_PIPE_GATHER(S, XX, XY, UX, UY)                                                   \
    _Pragma("unroll") for (int j = 0; j < NPT; ++j) {                                        \
        const double2 *px = S[j].x >= 0 ? x_free + S[j].x : x_fixed + ~S[j].x;                \
        const double2 *pu = S[j].y >= 0 ? u_free + S[j].y : u_fixed + ~S[j].y;                \
        const double2 tx = *px, tu = *pu;                                                     \
        XX[j] = tx.x; XY[j] = tx.y; UX[j] = tu.x; UY[j] = tu.y;                               \
    }
    auto add_row = [&](int l, const double2 gx, const double2 gu) {
        unsafeAtomicAdd(&acc0[l], gx.x); unsafeAtomicAdd(&acc1[l], gx.y);
        unsafeAtomicAdd(&acc2[l], gu.x); unsafeAtomicAdd(&acc3[l], gu.y);
    };

    load_index(t_first, s, w0, w1);
    if (tid < 8 * (t_end - t_first)) dcache[tid] = reinterpret_cast<const int *>(pd.tiles + tile_begin + t_first)[tid];
    d = pd.tiles[tile_begin + t_first];
    HFEM_PIPE_GATHER(s, vxx, vxy, vux, vuy)

    for (int t = t_first; t < t_end; ++t) {
        const int n_owned = d.n_owned;
        // ---- LDS fill from the registers (the gather was issued one tile ago, or in the prologue)
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int l = tid + j * BLOCK;
            if (l < d.n_node) { nd_xy[l] = make_double2(vxx[j], vxy[j]); nd_uv[l] = make_double2(vux[j], vuy[j]); }
            if (l < n_owned) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
        }
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            if (!(tid < col_stride && tid + j * col_stride < d.n_elem)) { w0[j] = kSkipBit; w1[j] = 0u; }
        const int n_edge = skip_edges ? 0 : d.n_edge;
        uint32_t edge_rec = 0u;
        int edge_id = 0;
        if (tid < n_edge) {
            edge_rec = pd.edge_pack[d.edge_off + tid];
            if (T_edge) edge_id = pd.edge_gid[d.edge_off + tid];
        }
        __syncthreads();

        const bool more = t + 1 < t_end;                // workgroup-uniform
        if (more) load_index(t + 1, sn, wn0, wn1);
        // ---- slots: registers + LDS only; the next tile's gather is issued after the first row (its row maps have landed)
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
            if (j == 0 && more) { HFEM_PIPE_GATHER(sn, nxx, nxy, nux, nuy) }
        }
        for (int i = tid; i < n_edge; i += BLOCK) {      // boundary tiles only
            const uint32_t p = i == tid ? edge_rec : pd.edge_pack[d.edge_off + i];
            const int l0 = (int)(p & kLocalMask), l1 = (int)((p >> kLocalBits) & kLocalMask);
            const double4 tt = T_edge ? T_edge[i == tid ? edge_id : pd.edge_gid[d.edge_off + i]] : Tconst;
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

        // ---- every owned gradient row is written exactly once (write-through); the stores drain under the next tile
#pragma unroll
        for (int j = 0; j < NPT; ++j) {                  // branch-free: an idle lane stores past the buffer's range (dropped), so
            const int l = tid + j * BLOCK;               // the store COUNT is static and the next fill's vmcnt(N) skips the stores
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
        if (tid == 0) {                                  // fixed order: the tile energy is bit-reproducible
            double tile_e = 0.0;
#pragma unroll
            for (int w = 0; w < BLOCK / 64; ++w) tile_e += red[w];
            partials[t] = tile_e;
        }
        if (more) {
            __syncthreads();                             // the accumulators and `red` are read; LDS may be refilled
#pragma unroll
            for (int j = 0; j < NPT; ++j) { s[j] = sn[j]; vxx[j] = nxx[j]; vxy[j] = nxy[j]; vux[j] = nux[j]; vuy[j] = nuy[j]; }
#pragma unroll
            for (int j = 0; j < EPT; ++j) { w0[j] = wn0[j]; w1[j] = wn1[j]; }
            {
                const int *dc = dcache + 8 * (t + 1 - t_first);
                d.elem_off = __builtin_amdgcn_readfirstlane(dc[0]); d.n_elem = __builtin_amdgcn_readfirstlane(dc[1]);
                d.node_off = __builtin_amdgcn_readfirstlane(dc[2]); d.n_node = __builtin_amdgcn_readfirstlane(dc[3]);
                d.n_owned = __builtin_amdgcn_readfirstlane(dc[4]); d.edge_off = __builtin_amdgcn_readfirstlane(dc[5]);
                d.n_edge = __builtin_amdgcn_readfirstlane(dc[6]); d.pad = __builtin_amdgcn_readfirstlane(dc[7]);
            }
        }
    }
}

#undef HFEM_PIPE_GATHER

template <int NPT, int EPT, int CAPO, int WPS>
static void launch_pipe_inst(const PairLaunch &A, const LagSum &lag, int n_tiles, int tpw, int grid) {
    const size_t lds = (CAPO > 0 ? (size_t)(A.max_nodes * 32 + CAPO * 32 + 128) : A.lds) + 32 * kPipeMaxTiles;
    hipLaunchKernelGGL((tri3_energy_pair_pipe_kernel<256, NPT, EPT, WPS, CAPO>), dim3(grid), dim3(256), lds, A.s, A.pd, A.tile_begin,
                       n_tiles, tpw, (const double2 *)A.x_free, (const double2 *)A.x_fixed, (const double2 *)A.u_free,
                       (const double2 *)A.u_fixed, A.k, A.T_edge, A.tc, A.partials, (double2 *)A.gx, (double2 *)A.gu,
                       A.max_nodes, CAPO > 0 ? CAPO : A.max_owned, A.skip_edges, lag, A.col_stride);
}

// Pipelined launch over n_tiles tiles starting at A.tile_begin, tpw tiles per workgroup.  fp64, reference convention, zero
// body force, plain (unchained) paired plans.  1 = launched, 0 = the plan's tile shape has no instance.
int launch_tri3_pair_pipe(const hfem_plan *plan, PairLaunch A, int n_tiles, int tpw, const LagSum &lag) {
    const HostPlan &h = plan->host;
    if (!h.paired || h.n_chained > 0 || !plan->d_elem_pack_hi || tpw < 1 || tpw > kPipeMaxTiles || n_tiles < 1) return 0;
    A.pd = plan_dev(plan);
    A.max_nodes = h.max_nodes; A.max_owned = h.max_owned; A.lds = (size_t)plan->lds_bytes;
    A.col_stride = h.col_stride;
    const int npt = (h.max_nodes + 255) / 256, ept = h.max_rows;
    const int grid = (n_tiles + tpw - 1) / tpw + (lag.prev ? 1 : 0);
    if (npt > 3 || ept > 3 || ept < 1) return 0;
    if (plan->tune.pair_pipe_wps == 3) {
        if (npt <= 2 && ept <= 2) launch_pipe_inst<2, 2, 0, 3>(A, lag, n_tiles, tpw, grid);
        else if (npt <= 2) launch_pipe_inst<2, 3, 0, 3>(A, lag, n_tiles, tpw, grid);
        else launch_pipe_inst<3, 3, 0, 3>(A, lag, n_tiles, tpw, grid);
    } else {
        if (npt <= 2 && ept <= 2) launch_pipe_inst<2, 2, 0, 4>(A, lag, n_tiles, tpw, grid);
        else if (npt <= 2) launch_pipe_inst<2, 3, 0, 4>(A, lag, n_tiles, tpw, grid);
        else launch_pipe_inst<3, 3, 0, 4>(A, lag, n_tiles, tpw, grid);
    }
    return 1;
}

}  // namespace hfem
#endif  // HFEM_LAB

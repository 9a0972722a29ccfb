// Streamed TRI3 + EDGE2 energy kernel, gfx950 (MI355X): the tiled owner-computes pass of tri3_energy.hip with
// its three phases OVERLAPPED inside every workgroup instead of run one after the other.
//
// Replaces EnergyLoss2D.__call__ + loss.backward() of the reference (/root/reference/src/loss.py:55-116 over
// /root/reference/src/models.py:292-376) exactly as tri3_energy_fast_kernel does -- same closed forms
// (hfem_device.h), same tile plan, same outputs -- but needs a CHUNKED plan (plan_elem_order 4, plan.cpp): a
// tile's element slots are three spatial strips, and the owned / halo local ids are sorted by the first strip
// that touches them.  Then
//   * the node gather is LDS-DMA (global_load_lds_dwordx4: per-lane source row through the free/fixed row map,
//     destination = 64 consecutive LDS slots): no VGPR staging, no ds_write, and it stays in flight while the
//     wave computes;
//   * the DMA pieces are issued in the order the strips need them, and strip c starts as soon as ITS pieces have
//     landed (counted s_waitcnt vmcnt + s_barrier): the rest of the gather streams in under the element math of
//     the earlier strips;
//   * gradients accumulate in LDS (ds_add_f64) and every owned row leaves with one write-through 16-B store.
// HBM-bound, no MFMA (2x2 / 2x3 contractions).  Algorithmic bytes per launch: 12 Ne + 64 Nn + 8.
// LAB BUILD ONLY (-DHFEM_LAB, libhidenn_hip_lab.so): measured slower than the register-prefetched kernel (round 2:
// 11.9 vs 11.1 us on T1M) -- kept as the evidence behind that section, not shipped.
#ifdef HFEM_LAB
#include <hip/hip_runtime.h>

#include "hfem_device.h"
#include "hfem_plan_dev.h"

namespace hfem {

typedef __attribute__((address_space(3))) void lds_void_t;

// One LDS-DMA piece: lane i of the wave copies 16 B from its own global address to lds_dst + 16 i.  M0 carries
// the LDS base and is compiler-reserved, so it is saved and restored inside the statement; the compiler does not
// count this load (its completion is waited for with dma_wait below).
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// Wait until at most `pending` (0, 2 or 4; wave-uniform) of this wave's youngest vector-memory operations are
// outstanding.  vmcnt counts loads, stores and LDS-DMA together in issue order, so this retires every older piece.
__device__ __forceinline__ void dma_wait(int pending) {
    if (pending <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (pending <= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (pending <= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}

__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(size_t)(lds_void_t *)p;
}

// BLOCK threads walk a tile in kChunks (= 3) strips of <= BLOCK element slots each.  MAXP = DMA pieces (64 local
// ids each, both node arrays) a wave may own: the launcher guarantees pieces(tile) <= MAXP * BLOCK / 64.
// CAPO > 0: compile-time stride of the four accumulator arrays (>= the plan's max owned nodes per tile).
// SP: cache policy of the gradient stores (16 = sc1 write-through, 0 = plain).
// LDS: xy[cap_nodes] double2 | uv[cap_nodes] double2 | acc[4][cap_owned] double | red[BLOCK/64]
template <int BLOCK, int MAXP, int CAPO, int SP, int ABL = 0>
__global__ __launch_bounds__(BLOCK, BLOCK >= 512 ? 8 : 1) void tri3_energy_stream_kernel(
    PlanDev pd, int tile_begin, const double2 *__restrict__ x_free, const double2 *__restrict__ x_fixed,
    const double2 *__restrict__ u_free, const double2 *__restrict__ u_fixed, Tri3Consts k,
    const double4 *__restrict__ T_edge, double4 Tconst, double *__restrict__ partials,
    double2 *__restrict__ gx_free, double2 *__restrict__ gu_free, int cap_nodes, int cap_owned_rt, int skip_edges,
    LagSum lag, unsigned long long *__restrict__ stamps) {
    static_assert(kChunks == 3, "the strip loop below is written out for three strips");
    constexpr int NW = BLOCK / 64;
    const int cap_owned = CAPO > 0 ? CAPO : cap_owned_rt;
    extern __shared__ double2 lds[];
    double2 *nd_xy = lds;
    double2 *nd_uv = lds + cap_nodes;
    double *acc0 = reinterpret_cast<double *>(lds + 2 * cap_nodes);
    double *acc1 = acc0 + cap_owned, *acc2 = acc1 + cap_owned, *acc3 = acc2 + cap_owned;
    double *red = acc3 + cap_owned;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // lab (ABL & 256): s_memrealtime stamps (100 MHz) of thread 0 -> stamps[16 * blockIdx.x + I]
    unsigned long long st_[12];
#define HFEM_SSTAMP(I) if (ABL & 256) st_[I] = __builtin_amdgcn_s_memrealtime();   // wave-uniform: stays in SGPRs
    HFEM_SSTAMP(0)
    const int n_launch = (int)gridDim.x - (lag.prev ? 1 : 0);
    if (lag.prev && (int)blockIdx.x == n_launch) {
        // the launch's one extra workgroup: sum the tile energies the PREVIOUS launch left (HFEM_FLAG_SUM_PREVIOUS;
        // same order as sum_partials_kernel: bit-identical)
        double v = 0.0;
        if (tid < 256)
            for (int i = tid; i < lag.prev_n; i += 256) v += lag.prev[i];
        const double tot = block_sum(v, red);
        if (tid == 0) lag.out[0] = tot;
        return;
    }
    const int slot = xcd_tile(blockIdx.x, n_launch);
    if (ABL & 32) { if (tid == 0 && cap_nodes < 0) partials[slot] = 0.0; return; }      // lab: launch cost only
    const TileDesc d = pd.tiles[tile_begin + slot];
    const int4 ck = pd.tile_chunks[tile_begin + slot];
    const int n_owned = d.n_owned, n_node = d.n_node;
    if ((ABL & 256) && n_node >= 0) HFEM_SSTAMP(1)
    if (ABL & 16) { if (tid == 0) partials[slot] = (double)(n_owned + ck.x); return; }  // lab: + the descriptor round trip

    // ---- element records of the three strips (one slot per thread and strip)
    const uint32_t *ep = pd.elem_pack + d.elem_off;
    uint32_t pk0 = kSkipBit, pk1 = kSkipBit, pk2 = kSkipBit;
    if (!(ABL & 64)) {
        if (tid < ck.x) pk0 = ep[tid];
        if (ck.x + tid < ck.y) pk1 = ep[ck.x + tid];
        if (ck.y + tid < d.n_elem) pk2 = ep[ck.y + tid];
    }

    // ---- DMA pieces in need order: [owned: strip 0][halo: strip 0][owned: +strip 1][halo: +strip 1][owned: rest][halo: rest]
    const int po0 = ck.z & 255, po1 = (ck.z >> 8) & 255, ph0 = (ck.z >> 16) & 255, ph1 = (ck.z >> 24) & 255;
    const int po2 = (n_owned + 63) >> 6, ph2 = (n_node - n_owned + 63) >> 6;
    const int b0 = po0, b1 = b0 + ph0, b2 = b1 + (po1 - po0), b3 = b2 + (ph1 - ph0), b4 = b3 + (po2 - po1),
              b5 = b4 + (ph2 - ph1);
    const int2 *src = pd.node_src + d.node_off;
    int2 m[MAXP];            // row-map entries of this lane's nodes (kept for the write-out)
    int lid[MAXP];           // local id of this lane's node in piece i, -1: none
    bool own[MAXP];          // piece i lies in the owned id range (wave-uniform)
    int n_mine = 0, q0 = 0, q1 = 0;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int s = wv + NW * i;                      // wave-uniform position in the need order
        int base;
        if (s < b0) { own[i] = true; base = 64 * s; }
        else if (s < b1) { own[i] = false; base = n_owned + 64 * (s - b0); }
        else if (s < b2) { own[i] = true; base = 64 * (po0 + s - b1); }
        else if (s < b3) { own[i] = false; base = n_owned + 64 * (ph0 + s - b2); }
        else if (s < b4) { own[i] = true; base = 64 * (po1 + s - b3); }
        else { own[i] = false; base = n_owned + 64 * (ph1 + s - b4); }
        const int l = base + lane;
        const bool valid = s < b5 && l < (own[i] ? n_owned : n_node);
        lid[i] = valid ? l : -1;
        m[i] = make_int2(0, 0);
        if (valid) m[i] = (ABL & 128) ? make_int2(l, l) : src[l];
        n_mine += s < b5;
        q0 += s < b1;
        q1 += s < b3;
    }
    // every index load has landed before the first DMA goes out: nothing the compiler counts is in flight while
    // the pieces are (its own waits would otherwise drain them)
    asm volatile("" : "+v"(pk0), "+v"(pk1), "+v"(pk2));
    if ((ABL & 256) && (pk0 | 1u)) HFEM_SSTAMP(2)
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        if (lid[i] >= 0) {
            const int l = lid[i];
            if (ABL & 2) {       // lab: no gather (synthetic node data)
                nd_xy[l] = make_double2(0.001 * l + 1e-4 * (l & 7), 0.002 * (l & 15) + 1e-4 * (l & 3) * (l & 5));
                nd_uv[l] = make_double2(1e-5, 2e-5 * (l & 3));
                if (own[i]) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
                continue;
            }
            const unsigned dst_xy = __builtin_amdgcn_readfirstlane(lds_addr(nd_xy + (l - lane)));
            const unsigned dst_uv = __builtin_amdgcn_readfirstlane(lds_addr(nd_uv + (l - lane)));
            glds16(m[i].x >= 0 ? x_free + m[i].x : x_fixed + ~m[i].x, dst_xy);
            glds16(m[i].y >= 0 ? u_free + m[i].y : u_fixed + ~m[i].y, dst_uv);
            if (own[i]) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
        }
    }

    HFEM_SSTAMP(3)
    double e_loc = 0.0;
    auto strip = [&](const uint32_t p) {
        if ((ABL & 1) == 0 && !(p & kSkipBit)) {
            const int l0 = (int)(p & kLocalMask), l1 = (int)((p >> kLocalBits) & kLocalMask),
                      l2 = (int)((p >> (2 * kLocalBits)) & kLocalMask);
            double2 gx[3], gu[3];
            const double e = tri3_element<true, false>(nd_xy[l0], nd_xy[l1], nd_xy[l2], nd_uv[l0], nd_uv[l1],
                                                       nd_uv[l2], k, gx, gu);
            if (p & kHomeBit) e_loc += e;
            if (ABL & 8) {       // lab: math without the LDS accumulation
                asm volatile("" ::"v"(gx[0].x), "v"(gx[0].y), "v"(gx[1].x), "v"(gx[1].y), "v"(gx[2].x), "v"(gx[2].y),
                             "v"(gu[0].x), "v"(gu[0].y), "v"(gu[1].x), "v"(gu[1].y), "v"(gu[2].x), "v"(gu[2].y));
                return;
            }
            if (l0 < n_owned) {
                unsafeAtomicAdd(&acc0[l0], gx[0].x); unsafeAtomicAdd(&acc1[l0], gx[0].y);
                unsafeAtomicAdd(&acc2[l0], gu[0].x); unsafeAtomicAdd(&acc3[l0], gu[0].y);
            }
            if (l1 < n_owned) {
                unsafeAtomicAdd(&acc0[l1], gx[1].x); unsafeAtomicAdd(&acc1[l1], gx[1].y);
                unsafeAtomicAdd(&acc2[l1], gu[1].x); unsafeAtomicAdd(&acc3[l1], gu[1].y);
            }
            if (l2 < n_owned) {
                unsafeAtomicAdd(&acc0[l2], gx[2].x); unsafeAtomicAdd(&acc1[l2], gx[2].y);
                unsafeAtomicAdd(&acc2[l2], gu[2].x); unsafeAtomicAdd(&acc3[l2], gu[2].y);
            }
        }
    };
    // ---- strip c: this wave's pieces of strips 0..c have landed, then the workgroup's (barrier), then the math
    dma_wait((ABL & 512) ? 0 : 2 * (n_mine - q0));            // lab 512: one wait for the whole gather, no strip barriers
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the accumulator clears
    HFEM_SSTAMP(4)
    __builtin_amdgcn_s_barrier();
    HFEM_SSTAMP(5)
    strip(pk0);
    if (!(ABL & 512)) {
        dma_wait(2 * (n_mine - q1));
        __builtin_amdgcn_s_barrier();
    }
    HFEM_SSTAMP(6)
    strip(pk1);
    if (!(ABL & 512)) {
        dma_wait(0);
        __builtin_amdgcn_s_barrier();
    }
    HFEM_SSTAMP(7)
    strip(pk2);
    HFEM_SSTAMP(8)

    const int n_edge = skip_edges ? 0 : d.n_edge;
    for (int i = tid; i < n_edge; i += BLOCK) {          // boundary tiles only
        const uint32_t p = pd.edge_pack[d.edge_off + i];
        const int l0 = (int)(p & kLocalMask), l1 = (int)((p >> kLocalBits) & kLocalMask);
        const double4 tt = T_edge ? T_edge[pd.edge_gid[d.edge_off + i]] : Tconst;
        double2 gx[2], gu[2];
        const double wk = edge2_element<true>(nd_xy[l0], nd_xy[l1], nd_uv[l0], nd_uv[l1], tt, gx, gu);
        if (p & kHomeBit) e_loc -= wk;
        if (l0 < n_owned) {
            unsafeAtomicAdd(&acc0[l0], gx[0].x); unsafeAtomicAdd(&acc1[l0], gx[0].y);
            unsafeAtomicAdd(&acc2[l0], gu[0].x); unsafeAtomicAdd(&acc3[l0], gu[0].y);
        }
        if (l1 < n_owned) {
            unsafeAtomicAdd(&acc0[l1], gx[1].x); unsafeAtomicAdd(&acc1[l1], gx[1].y);
            unsafeAtomicAdd(&acc2[l1], gu[1].x); unsafeAtomicAdd(&acc3[l1], gu[1].y);
        }
    }
    {   // tile energy: wave shuffle reduction, one LDS slot per wave -- rides on the barrier below
        const double w = wave_sum(e_loc);
        if (lane == 0) red[tid >> 6] = w;
    }
    __syncthreads();
    HFEM_SSTAMP(9)

    // ---- every owned gradient row is written exactly once, by the lane that fetched the node
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t rx, ru;
    if (SP != 0) {
        rx = __builtin_amdgcn_make_buffer_rsrc((void *)gx_free, 0, 0x7FFFFFF0, 0x00020000);
        ru = __builtin_amdgcn_make_buffer_rsrc((void *)gu_free, 0, 0x7FFFFFF0, 0x00020000);
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        if (!(ABL & 4) && own[i] && lid[i] >= 0) {
            const int l = lid[i];
            if (gx_free && m[i].x >= 0) {
                double2 v;
                v.x = acc0[l]; v.y = acc1[l];
                if (SP == 0) gx_free[m[i].x] = v;
                else __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), rx, m[i].x * 16, 0, SP);
            }
            if (gu_free && m[i].y >= 0) {
                double2 v;
                v.x = acc2[l]; v.y = acc3[l];
                if (SP == 0) gu_free[m[i].y] = v;
                else __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), ru, m[i].y * 16, 0, SP);
            }
        }
    }
    if (tid == 0) {                                     // fixed order: the tile energy is bit-reproducible
        double tile_e = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tile_e += red[w];
        partials[slot] = tile_e;
    }
    HFEM_SSTAMP(10)
    if ((ABL & 256) && tid == 0) {
#pragma unroll
        for (int i = 0; i < 11; ++i) stamps[16 * (size_t)blockIdx.x + i] = st_[i];
    }
#undef HFEM_SSTAMP
}

// Launch on a chunked plan.  Returns 1 when launched, 0 when the plan's shape is outside what the kernel holds
// (the caller then takes the generic tiled kernel), < 0 never.
int launch_tri3_stream(const hfem_plan *plan, int n_grid, int tile_begin, const double *x_free, const double *x_fixed,
                       const double *u_free, const double *u_fixed, const Tri3Consts &kc, const double *T_edge,
                       double4 tc, double *partials, double *gx_free, double *gu_free, int skip_edges, int store_policy,
                       const LagSum &lag, hipStream_t s, int ablate) {
    const HostPlan &h = plan->host;
    if (h.npe != 3 || h.max_chunk_elems <= 0 || h.max_chunk_elems > 512 || !plan->d_tile_chunks) return 0;
    if (h.max_nodes + 128 > 2 * 512) return 0;               // pieces(tile) <= nodes/64 + 2 <= MAXP * 8
    if (store_policy != 16) return 0;
    PlanDev pd = plan_dev(plan);
#define HFEM_STREAM_ABL(A)                                                                                          \
    case A:                                                                                                         \
        hipLaunchKernelGGL((tri3_energy_stream_kernel<512, 2, 560, 16, A>), dim3(n_grid), dim3(512),                \
                           (size_t)(h.max_nodes * 32 + 560 * 32 + 128), s, pd, tile_begin, (const double2 *)x_free, \
                           (const double2 *)x_fixed, (const double2 *)u_free, (const double2 *)u_fixed, kc,         \
                           (const double4 *)T_edge, tc, partials, (double2 *)gx_free, (double2 *)gu_free,           \
                           h.max_nodes, 560, skip_edges, lag, plan->d_stamps);                                      \
        break;
    if (h.max_owned <= 560 && h.max_nodes * 32 + 560 * 32 + 128 <= 38912) {
        switch (ablate) {
            HFEM_STREAM_ABL(0) HFEM_STREAM_ABL(1) HFEM_STREAM_ABL(2) HFEM_STREAM_ABL(4) HFEM_STREAM_ABL(5)
            HFEM_STREAM_ABL(6) HFEM_STREAM_ABL(7) HFEM_STREAM_ABL(16) HFEM_STREAM_ABL(32) HFEM_STREAM_ABL(199) HFEM_STREAM_ABL(512) HFEM_STREAM_ABL(768) HFEM_STREAM_ABL(256) HFEM_STREAM_ABL(263) HFEM_STREAM_ABL(455) HFEM_STREAM_ABL(8) HFEM_STREAM_ABL(3) HFEM_STREAM_ABL(10) HFEM_STREAM_ABL(14)
            default: return 0;
        }
    } else {
        hipLaunchKernelGGL((tri3_energy_stream_kernel<512, 2, 0, 16>), dim3(n_grid), dim3(512),
                           (size_t)plan->lds_bytes, s, pd, tile_begin, (const double2 *)x_free,
                           (const double2 *)x_fixed, (const double2 *)u_free, (const double2 *)u_fixed, kc,
                           (const double4 *)T_edge, tc, partials, (double2 *)gx_free, (double2 *)gu_free, h.max_nodes,
                           h.max_owned, skip_edges, lag, plan->d_stamps);
    }
    return 1;
}

}  // namespace hfem
#endif  // HFEM_LAB

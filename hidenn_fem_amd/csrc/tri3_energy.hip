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

#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <type_traits>

#include "hfem_device.h"
#include "hfem_plan_dev.h"

#ifdef HFEM_LAB
#define HFEM_LAB_SECTION 0           // in-kernel lab hooks (start staggers)
#include "tri3_energy_lab.inc"
#undef HFEM_LAB_SECTION
#else
#define HFEM_LAB_TILED_STAGGER(stagger_ticks, stagger_mode)
#define HFEM_LAB_FAST_STAGGER(stagger_ticks, stagger_cfg)
#endif

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
// Ablation bits (ABL) exist for the lab build only (hfem_set_option("tiled_ablate", bits) of libhidenn_hip_lab.so);
// the product library instantiates ABL = 0, so every `if (ABL & ...)` below is compiled out of it.
//   1: LDS atomics -> plain LDS stores    2: skip the element/edge phase
//   4: skip the global node gather        8: skip the gradient write-out
//  16: no LDS node reads in the element loop (synthetic operands)   32: no LDS accumulate at all
//
// LDS: xy[cap_nodes] double2 | uv[cap_nodes] double2 | acc[4][cap_owned] double | red[BLOCK/64]
template <int BLOCK, int ABL>
__global__ __launch_bounds__(BLOCK) void tri3_energy_tiled_kernel(
    PlanDev pd, int tile_begin, const double2 *__restrict__ x_free,
    const double2 *__restrict__ x_fixed, const double2 *__restrict__ u_free,
    const double2 *__restrict__ u_fixed, Tri3Consts k, const double4 *__restrict__ T_edge,
    double4 Tconst, double *__restrict__ partials, double2 *__restrict__ gx_free,
    double2 *__restrict__ gu_free, int cap_nodes, int cap_owned, int skip_edges, int stagger_ticks,
    int stagger_mode, unsigned long long *__restrict__ stamps) {
    extern __shared__ double2 lds[];
    double2 *nd_xy = lds;
    double2 *nd_uv = lds + cap_nodes;
    double *acc0 = reinterpret_cast<double *>(lds + 2 * cap_nodes);
    double *acc1 = acc0 + cap_owned, *acc2 = acc1 + cap_owned, *acc3 = acc2 + cap_owned;
    double *red = acc3 + cap_owned;

    const int tid = threadIdx.x;
    const int slot = xcd_tile(blockIdx.x, gridDim.x);
#define HFEM_STAMP(I)                                                                              \
    if ((ABL & 64) && tid == 0) stamps[16 * (size_t)blockIdx.x + (I)] = __builtin_amdgcn_s_memrealtime();
    HFEM_STAMP(0)
    HFEM_LAB_TILED_STAGGER(stagger_ticks, stagger_mode)
    const TileDesc d = pd.tiles[tile_begin + slot];
    if ((ABL & 64) && tid == 0 && d.n_node >= 0) stamps[16 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();

    // ---- phase 1: gather node data through the free/fixed maps, clear accumulators
    const int2 *src = pd.node_src + d.node_off;
    for (int l = tid; l < d.n_node; l += BLOCK) {
        if (ABL & 4) {
            nd_xy[l] = make_double2(0.001 * l, 0.002 * (l % 7));
            nd_uv[l] = make_double2(1e-5, 2e-5 * (l % 3));
        } else {
            const int2 s = src[l];
            nd_xy[l] = s.x >= 0 ? x_free[s.x] : x_fixed[~s.x];
            nd_uv[l] = s.y >= 0 ? u_free[s.y] : u_fixed[~s.y];
        }
    }
    for (int l = tid; l < d.n_owned; l += BLOCK) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
    HFEM_STAMP(2)
    __syncthreads();
    HFEM_STAMP(3)

    // ---- phase 2: elements (home + halo), then the tile's Neumann edges
    double e_loc = 0.0;
    const uint32_t *ep = pd.elem_pack + d.elem_off;
    const int n_owned = d.n_owned;
    auto accumulate = [&](int l, const double2 gx, const double2 gu) {
        if (ABL & 128) {   // lab: synthetic address patterns (timing only)
            const int lane = tid & 63;
            int pat = lane;                                                    // lane-linear: conflict-free under any model
            if (ABL & 256) pat = ((lane & 15) << 1) + ((lane >> 4) & 1) + (lane & 32);   // A: distinct mod 32 per 32-lane half, 2-way mod 16 per 16-lane group
            if (ABL & 512) pat = (lane & 15) + ((lane >> 4) << 5);                       // B: distinct mod 16 per 16-lane group, 2-way mod 32 per half
            l = pat + ((l >> 7) << 7) < n_owned ? pat + ((l >> 7) << 7) : pat;
        }
        if (ABL & 32) {
            asm volatile("" ::"v"(gx.x), "v"(gx.y), "v"(gu.x), "v"(gu.y));
        } else if (ABL & 1) {
            acc0[l] = gx.x; acc1[l] = gx.y; acc2[l] = gu.x; acc3[l] = gu.y;
        } else {
            unsafeAtomicAdd(&acc0[l], gx.x);
            unsafeAtomicAdd(&acc1[l], gx.y);
            unsafeAtomicAdd(&acc2[l], gu.x);
            unsafeAtomicAdd(&acc3[l], gu.y);
        }
    };
    if (!(ABL & 2)) {
        for (int i = tid; i < d.n_elem; i += BLOCK) {
            const uint32_t p = ep[i];
            if (p & kSkipBit) continue;
            const int l[3] = {(int)(p & kLocalMask), (int)((p >> kLocalBits) & kLocalMask),
                              (int)((p >> (2 * kLocalBits)) & kLocalMask)};
            double2 gx[3], gu[3];
            double e;
            if (ABL & 16) {
                const double f = 1e-3 * (double)(p & 0xFFFF);
                e = tri3_element<true>(make_double2(1.0 + f, f), make_double2(f, 1.0 - f), make_double2(-f, f * f),
                                       make_double2(f, 2 * f), make_double2(3 * f, f), make_double2(f, -f), k, gx, gu);
            } else {
                e = tri3_element<true>(nd_xy[l[0]], nd_xy[l[1]], nd_xy[l[2]], nd_uv[l[0]], nd_uv[l[1]],
                                       nd_uv[l[2]], k, gx, gu);
            }
            if (p & kHomeBit) e_loc += e;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (l[j] < n_owned) accumulate(l[j], gx[j], gu[j]);
        }
        const int n_edge = skip_edges ? 0 : d.n_edge;
        for (int i = tid; i < n_edge; i += BLOCK) {
            const uint32_t p = pd.edge_pack[d.edge_off + i];
            const int l[2] = {(int)(p & kLocalMask), (int)((p >> kLocalBits) & kLocalMask)};
            const double4 t = T_edge ? T_edge[pd.edge_gid[d.edge_off + i]] : Tconst;
            double2 gx[2], gu[2];
            const double w = edge2_element<true>(nd_xy[l[0]], nd_xy[l[1]], nd_uv[l[0]], nd_uv[l[1]], t, gx, gu);
            if (p & kHomeBit) e_loc -= w;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (l[j] < n_owned) accumulate(l[j], gx[j], gu[j]);
        }
    }
    HFEM_STAMP(4)
    __syncthreads();
    HFEM_STAMP(5)

    // ---- phase 3: every owned gradient row is written exactly once
    if (!(ABL & 8)) {
        for (int l = tid; l < n_owned; l += BLOCK) {
            const int2 s = src[l];
            if (gx_free && s.x >= 0) gx_free[s.x] = make_double2(acc0[l], acc1[l]);
            if (gu_free && s.y >= 0) gu_free[s.y] = make_double2(acc2[l], acc3[l]);
        }
    }
    HFEM_STAMP(6)
    const double tot = block_sum(e_loc, red);
    if (tid == 0) partials[slot] = tot;
    HFEM_STAMP(7)
#undef HFEM_STAMP
}

// ------------------------------------------------------------------ tiled, register-prefetched
// Production variant of the tiled kernel: same phases, but everything a thread will need is
// requested up front -- its row-map entries AND its packed element records go out with the first
// loads (so the element loop starts on registers instead of a global-load latency per iteration),
// and the row maps stay in registers for the write-out (no re-load).  HASB compiles the body-force
// table out.  NPT >= ceil(max nodes/BLOCK), EPT >= ceil(max element slots/BLOCK); the launcher
// falls back to tri3_energy_tiled_kernel when a plan exceeds them.
// __launch_bounds__(BLOCK, 8) on the default instances (no body force, >= 512 threads): 8 waves per
// SIMD (<= 64 VGPRs) so that four 512-thread workgroups (32 waves) are resident per CU -- the
// residency the balanced tile plan is sized for.  The other instances keep the compiler's budget
// (forcing 64 VGPRs on them spills).
// V2 = storage type of the parameter / gradient rows: double2, or float2 for fp32 models (the reference's default
// dtype) -- rows are widened on load and rounded once on store, all arithmetic stays fp64.
// CAPN / CAPO > 0: compile-time LDS array strides (nodes / owned nodes per tile, >= the plan's maxima): every LDS
// address is then ONE scaled local id plus an immediate offset instead of a runtime base add per array.
// ADAM: the write-out applies the Adam update instead of storing the gradient (struct AdamFuse below).
template <int BLOCK, int NPT, int EPT, bool HASB, bool STAMP = false, int SP = 0, typename V2 = double2, int CAPN = 0,
          int CAPO = 0, bool ADAM = false, bool PHYS = false, bool PG = false>
__global__ __launch_bounds__(BLOCK, (!HASB && BLOCK >= 512) ? 8 : 1) void tri3_energy_fast_kernel(
    PlanDev pd, int tile_begin, typename RowArg<V2, PG>::type x_free,
    const V2 *__restrict__ x_fixed, typename RowArg<V2, PG>::type u_free,
    const V2 *__restrict__ u_fixed, Tri3Consts k, const double4 *__restrict__ T_edge,
    double4 Tconst, double *__restrict__ partials, V2 *__restrict__ gx_free,
    V2 *__restrict__ gu_free, int cap_nodes_rt, int cap_owned_rt, int skip_edges, int stagger_ticks, int stagger_cfg,
    unsigned long long *__restrict__ stamps, AdamFuse af, LagSum lag) {
#define HFEM_FSTAMP(I)                                                                             \
    if (STAMP && threadIdx.x == 0) stamps[16 * (size_t)blockIdx.x + (I)] = __builtin_amdgcn_s_memrealtime();
    HFEM_FSTAMP(0)
    const int cap_nodes = CAPN > 0 ? CAPN : cap_nodes_rt, cap_owned = CAPO > 0 ? CAPO : cap_owned_rt;
    extern __shared__ double2 lds[];
    double2 *nd_xy = lds;
    double2 *nd_uv = lds + cap_nodes;
    double *acc0 = reinterpret_cast<double *>(lds + 2 * cap_nodes);
    double *acc1 = acc0 + cap_owned, *acc2 = acc1 + cap_owned, *acc3 = acc2 + cap_owned;
    double *red = acc3 + cap_owned;

    const int tid = threadIdx.x;
    // PG instances (HFEM_FLAG_PEER_GET launches only): the first lag.pg_blocks workgroups are the peer-window get (peer.hip,
    // as in tri3_pair.hip); every other launch runs the PG = false code
    int bid = (int)blockIdx.x;
    if constexpr (PG) {
        if (bid < lag.pg_blocks) {
            peer_get_block<V2>(*lag.pg, bid, lag.pg_blocks, x_free, u_free);      // RowArg<V2, true>: writable, not restrict
            return;
        }
        bid -= lag.pg_blocks;
    }
    const int n_launch = (int)gridDim.x - (lag.prev ? 1 : 0) - (PG ? lag.pg_blocks : 0);
    if (lag.prev && bid == n_launch) {
        // The launch's one extra workgroup: sum the tile energies the PREVIOUS launch left (other partials bank) --
        // the 1-block reduction and the kernel boundary in front of it leave the critical path.  Same order as
        // sum_partials_kernel (256 adders, shuffle tree, wave sums in wave order): bit-identical result.
        double v = 0.0;
        if (tid < 256)
            for (int i = tid; i < lag.prev_n; i += 256) v += lag.prev[i];
        const double tot = block_sum(v, red);
        if (tid == 0) lag.out[0] = tot;
        return;
    }
    const int slot = xcd_tile(bid, n_launch);
    HFEM_LAB_FAST_STAGGER(stagger_ticks, stagger_cfg)
    // ---- row maps and element records from the tile index alone (uniform node / slot strides, plan.cpp), in flight together
    //      with the descriptor; unguarded loads (padding = valid rows / skip records), then the gather, all issued back to back.
    //      Round-2 ISA reading: guarded loads are basic blocks of their own and made desc -> maps -> rows three dependent trips.
    int2 s[NPT];
    uint32_t pk[EPT];
    const int2 *src = pd.node_src + (size_t)(tile_begin + slot) * pd.node_stride;
    const uint32_t *ep = pd.elem_pack + (size_t)(tile_begin + slot) * pd.elem_stride;
#pragma unroll
    for (int j = 0; j < NPT; ++j) s[j] = src[min(tid + j * BLOCK, pd.node_stride - 1)];      // lanes past the stride repeat its last record
#pragma unroll
    for (int j = 0; j < EPT; ++j) pk[j] = ep[min(tid + j * BLOCK, pd.elem_stride - 1)];
    const TileDesc d = pd.tiles[tile_begin + slot];
    const int n_owned = d.n_owned;
    if (STAMP && threadIdx.x == 0 && n_owned >= 0) stamps[16 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (tid + j * BLOCK >= d.n_elem) pk[j] = kSkipBit;
    if constexpr (PG) {
        if (tile_begin + slot >= lag.wait_begin && tile_begin + slot < lag.wait_end)
            peer_wait_unpacked(*lag.pg);                // a boundary tile: the rows it reads from other ranks are being copied in
    }
    // ---- gather (unguarded loads: the compiler batches as many as the register budget of the instance allows) into LDS,
    //      clear the accumulators
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int l = tid + j * BLOCK;
        const V2 *px = s[j].x >= 0 ? x_free + s[j].x : x_fixed + ~s[j].x;
        const V2 *pu = s[j].y >= 0 ? u_free + s[j].y : u_fixed + ~s[j].y;
        const V2 vx = *px, vu = *pu;
        if (l < d.n_node) {
            nd_xy[l] = make_double2((double)vx.x, (double)vx.y);
            nd_uv[l] = make_double2((double)vu.x, (double)vu.y);
        }
        if (l < n_owned) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
    }
    HFEM_FSTAMP(2)
    __syncthreads();
    HFEM_FSTAMP(3)

    // ---- elements: registers + LDS only
    double e_loc = 0.0;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const uint32_t p = pk[j];
        if (!(p & kSkipBit)) {
            const int l0 = (int)(p & kLocalMask), l1 = (int)((p >> kLocalBits) & kLocalMask),
                      l2 = (int)((p >> (2 * kLocalBits)) & kLocalMask);
            double2 gx[3], gu[3];
            const double e = tri3_element<true, HASB, PHYS>(nd_xy[l0], nd_xy[l1], nd_xy[l2], nd_uv[l0], nd_uv[l1],
                                                            nd_uv[l2], k, gx, gu);
            if (p & kHomeBit) e_loc += e;
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
    }
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
    // tile energy: wave shuffle reduction, one LDS slot per wave -- rides on the barrier below
    {
        const double w = wave_sum(e_loc);
        if ((tid & 63) == 0) red[tid >> 6] = w;          // one slot per wave: summed in wave order below
    }
    HFEM_FSTAMP(4)
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): all loads returned long ago; keeps per-store vmcnt waits out of the write-out
    __syncthreads();
    HFEM_FSTAMP(5)

    // ---- every owned gradient row is written exactly once (row maps are still in registers)
    // SP: cache policy of the gradient stores.  0 plain; 16 = sc1 (write-through, the line is dropped
    // from the XCD's L2, so the once-written gradients do not evict the re-read inputs / plan arrays);
    // 2 = nt.  (aux bits of the buffer store: sc0 = 1, nt = 2, sc1 = 16.)
    if (ADAM) {
        // Fused optimiser step (hfem_tri3_energy_adam_step): every free row is owned by exactly one tile, which holds
        // the row's complete gradient (acc*) and its current value (nd_xy / nd_uv) in LDS -- so the tile applies
        // torch.optim.Adam's update right here: m, v read and written in place, the NEW parameter row written to the
        // OTHER parameter buffer (tiles still gathering this launch must keep seeing the old one: ping-pong), and the
        // gradient never goes to memory.  Arithmetic = optim.hip's adam_step_dev_kernel, operation for operation.
        const double bc1 = af.bc[0], sqrt_bc2 = af.bc[1];   // 1 - b1^step, sqrt(1 - b2^step): hfem_adam_prep (no pow() in here)
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int l = tid + j * BLOCK;
            if (l < n_owned) {
                const int2 rows = src[l];                  // re-read (L2 hit) rather than kept: the element stage has no VGPR to spare
                if (rows.x >= 0) adam_fused_row<V2>(af, 0, rows.x, acc0[l], acc1[l], nd_xy[l], bc1, sqrt_bc2);
                if (rows.y >= 0) adam_fused_row<V2>(af, 1, rows.y, acc2[l], acc3[l], nd_uv[l], bc1, sqrt_bc2);
            }
        }
    } else {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    constexpr bool kWide = sizeof(V2) == 16;
    __amdgpu_buffer_rsrc_t rx, ru;
    if (SP != 0) {
        rx = __builtin_amdgcn_make_buffer_rsrc((void *)gx_free, 0, 0x7FFFFFF0, 0x00020000);
        ru = __builtin_amdgcn_make_buffer_rsrc((void *)gu_free, 0, 0x7FFFFFF0, 0x00020000);
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int l = tid + j * BLOCK;
        if (l < n_owned) {                          // (accumulators are read inside each branch: 64-VGPR budget)
            if (gx_free && s[j].x >= 0) {
                V2 v;
                v.x = acc0[l]; v.y = acc1[l];           // rounds once for float2
                if (SP == 0) gx_free[s[j].x] = v;
                else if (kWide) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), rx, s[j].x * 16, 0, SP);
                else __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), rx, s[j].x * 8, 0, SP);
            }
            if (gu_free && s[j].y >= 0) {
                V2 v;
                v.x = acc2[l]; v.y = acc3[l];
                if (SP == 0) gu_free[s[j].y] = v;
                else if (kWide) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), ru, s[j].y * 16, 0, SP);
                else __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), ru, s[j].y * 8, 0, SP);
            }
        }
    }
    }   // !ADAM
    HFEM_FSTAMP(6)
    if (tid == 0) {                                     // fixed order: the tile energy is bit-reproducible
        double tile_e = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) tile_e += red[w];
        partials[slot] = tile_e;
    }
    HFEM_FSTAMP(7)
#undef HFEM_FSTAMP
}

#ifdef HFEM_LAB
#define HFEM_LAB_SECTION 1           // persistent pipelined variant (lab build only)
#include "tri3_energy_lab.inc"
#undef HFEM_LAB_SECTION
#endif

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

// ---- tuning options.  hfem_set_option changes the DEFAULTS (atomics: no data race); hfem_plan_create captures them into
// the plan (hfem_plan::tune), and a plan keeps the values it was created with -- so an option change never alters what
// a concurrent launch on an existing plan does.  Lab knobs (ablations, stamps, staggers, pipelined / streamed variants)
// exist in the -DHFEM_LAB build only (libhidenn_hip_lab.so, hidenn_fem_amd/csrc/build.py --lab).
namespace {
struct Defaults {
    std::atomic<int> tiled_block{512};      // threads per tile (measured best on T1M: 512)
    std::atomic<int> store_policy{-1};      // gradient stores: 16 = sc1 (write-through: the line is dropped from the XCD's
                                            // L2 and does not evict the re-read inputs / plan arrays), 2 = nt (non-temporal),
                                            // 0 = plain; -1 = by mesh size (hfem_plan_create: nt from 750 k nodes)
    std::atomic<int> tiled_fast{1};         // register-prefetched kernel (0: the generic loop kernel)
    std::atomic<int> fast_const_caps{1};    // default tile shape: instance with compile-time accumulator strides
    std::atomic<int> pair_pipe_wps{4};      // pipelined kernel: register budget sized for this many waves per SIMD (3 or 4)
    std::atomic<int> pair_tiles_per_wg{1};  // paired plans: > 1 = software-pipelined kernel, that many tiles per workgroup
    std::atomic<int> plan_elem_order{-1};   // TRI3: -1 = auto (paired when the pair slots cover >= 0.7 x the elements, else 3; measured
                                            // crossover), 5 = paired slots (two fan-adjacent elements per slot, tri3_pair.hip; the default),
                                            // 3 = one element per slot in LDS-bank-aware 16-lane groups, 4 = the same inside three
                                            // strips (lab), 0..2 legacy orders; QUAD4 plans always use 3
    std::atomic<int> plan_node_cap{-1};     // max distinct nodes among a tile's own elements; 0: cut by element count only;
                                            // -1 (auto): 557 when tile_elems is left to the library, else 0.  557 nodes keep
                                            // (n_node + n_owned) * 32 B <= 38.9 KB: four 512-thread workgroups per CU
    std::atomic<int> plan_read_pack{2};     // paired slots packed against ds_read_b128 bank conflicts too (partner rows examined; 0 off)
    std::atomic<int> plan_snap{0};          // tile cuts snap to coarse curve cells (percent of a tile they may move back)
    std::atomic<int> plan_chunk_cap{512};   // chunked plans: longest strip of a tile (slots)
    std::atomic<int> plan_shards{1};        // ranks the plan's tiles will be sharded over (contiguous tile ranges): sizes the tiles
                                            // for elements PER RANK and orders every rank's boundary tiles first (plan.cpp)
    std::atomic<int> plan_pair_block{-1};   // paired plans: threads per tile, 256 or 512; -1 = by the shard-aware policy below
} g_def;

#ifdef HFEM_LAB
#define HFEM_LAB_SECTION 2           // lab knobs
#include "tri3_energy_lab.inc"
#undef HFEM_LAB_SECTION
#endif

hfem_plan::Tune current_tune() {
    hfem_plan::Tune t;
    t.tiled_block = g_def.tiled_block.load();
    t.store_policy = g_def.store_policy.load();          // -1 (auto) is resolved by hfem_plan_create, which knows the mesh
    t.tiled_fast = g_def.tiled_fast.load();
    t.fast_const_caps = g_def.fast_const_caps.load();
    t.pair_tiles_per_wg = g_def.pair_tiles_per_wg.load();
    t.pair_pipe_wps = g_def.pair_pipe_wps.load();
    return t;
}

int grid_for(int64_t n, int cap = 256 * 8) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// everything a tiled launch passes to its kernel, whatever the instance
struct Tri3Launch {
    PlanDev pd;
    int tile_begin = 0;
    const void *x_free = nullptr, *x_fixed = nullptr, *u_free = nullptr, *u_fixed = nullptr;
    Tri3Consts k;
    const double4 *T_edge = nullptr;
    double4 tc;
    double *partials = nullptr;
    void *gx = nullptr, *gu = nullptr;
    int max_nodes = 0, max_owned = 0, skip_edges = 0;
    unsigned long long *stamps = nullptr;
    size_t lds = 0;
    hipStream_t s = nullptr;
    int stagger = 0, stagger_cfg = 0x108;          // lab only (0 in the product build)
};

template <int BLK, int NPT, int EPT, bool HB, int SP, typename V2, int CO, bool ADAM, bool PHYS, bool STAMP = false>
void launch_fast(const Tri3Launch &A, int grid, const AdamFuse &af, const LagSum &lag) {
    const size_t lds = CO > 0 ? (size_t)(A.max_nodes * 32 + CO * 32 + 128) : A.lds;
    if constexpr (!HB && !PHYS && !STAMP && BLK == 512 && (SP == 16 || SP == 2)) {
        if (lag.pg_blocks) {                               // HFEM_FLAG_PEER_GET: the instance with the in-launch get
            hipLaunchKernelGGL((tri3_energy_fast_kernel<BLK, NPT, EPT, HB, STAMP, SP, V2, 0, CO, ADAM, PHYS, true>), dim3(grid), dim3(BLK),
                               lds, A.s, A.pd, A.tile_begin, (V2 *)const_cast<void *>(A.x_free), (const V2 *)A.x_fixed, (V2 *)const_cast<void *>(A.u_free),
                               (const V2 *)A.u_fixed, A.k, A.T_edge, A.tc, A.partials, (V2 *)A.gx, (V2 *)A.gu, A.max_nodes,
                               CO > 0 ? CO : A.max_owned, A.skip_edges, A.stagger, A.stagger_cfg, A.stamps, af, lag);
            return;
        }
    }
    hipLaunchKernelGGL((tri3_energy_fast_kernel<BLK, NPT, EPT, HB, STAMP, SP, V2, 0, CO, ADAM, PHYS>), dim3(grid), dim3(BLK),
                       lds, A.s, A.pd, A.tile_begin, (const V2 *)A.x_free, (const V2 *)A.x_fixed, (const V2 *)A.u_free,
                       (const V2 *)A.u_fixed, A.k, A.T_edge, A.tc, A.partials, (V2 *)A.gx, (V2 *)A.gu, A.max_nodes,
                       CO > 0 ? CO : A.max_owned, A.skip_edges, A.stagger, A.stagger_cfg, A.stamps, af, lag);
}

template <int BLK>
void launch_generic(const Tri3Launch &A, int grid) {
    hipLaunchKernelGGL((tri3_energy_tiled_kernel<BLK, 0>), dim3(grid), dim3(BLK), A.lds, A.s, A.pd, A.tile_begin,
                       (const double2 *)A.x_free, (const double2 *)A.x_fixed, (const double2 *)A.u_free,
                       (const double2 *)A.u_fixed, A.k, A.T_edge, A.tc, A.partials, (double2 *)A.gx, (double2 *)A.gu,
                       A.max_nodes, A.max_owned, A.skip_edges, 0, 0, A.stamps);
}

// fp64 rows, reference convention: the instance of the register-prefetched kernel that holds this plan's tile shape
// (false: none does -- the caller takes the generic loop kernel).  `lag` only reaches the default instances
// (512 threads, no body force, write-through stores); the entry point has checked that before asking for it.
bool launch_fast_f64(const hfem_plan *plan, const Tri3Launch &A, int n, bool hasb, const LagSum &lag) {
    const HostPlan &h = plan->host;
    const int blk = plan->tune.tiled_block, sp = plan->tune.store_policy;
    const int n_lag = n + (lag.prev ? 1 : 0) + lag.pg_blocks;
    if (lag.pg_blocks && !(blk == 512 && !hasb && (sp == 16 || sp == 2))) return false;   // no instance with the in-launch get
    // default shape (auto tile policy: <= 557 owned nodes): compile-time stride of the four accumulator arrays (12 of an
    // element's 18 LDS addresses); the footprint must stay <= 38 912 B (four workgroups per CU)
    if (plan->tune.fast_const_caps && blk == 512 && !hasb && (sp == 16 || sp == 2) && h.max_nodes > 512 && h.max_owned <= 560 &&
        h.max_nodes * 32 + 560 * 32 + 128 <= 38912 && h.max_elems <= 3 * 512) {
        if (sp == 2) launch_fast<512, 2, 3, false, 2, double2, 560, false, false>(A, n_lag, AdamFuse{}, lag);   // nt stores (big meshes)
        else launch_fast<512, 2, 3, false, 16, double2, 560, false, false>(A, n_lag, AdamFuse{}, lag);
        return true;
    }
    if (sp == 2 && blk == 512 && !hasb && h.max_nodes <= 2 * 512 && h.max_elems <= 4 * 512) {      // nt stores, general 512-thread shapes
        if (h.max_elems <= 3 * 512) launch_fast<512, 2, 3, false, 2, double2, 0, false, false>(A, n_lag, AdamFuse{}, lag);
        else launch_fast<512, 2, 4, false, 2, double2, 0, false, false>(A, n_lag, AdamFuse{}, lag);
        return true;
    }
#define HFEM_FAST_HB(BLK, NPT, EPT)                                                                                       \
    {                                                                                                                     \
        if (hasb) launch_fast<BLK, NPT, EPT, true, 0, double2, 0, false, false>(A, n, AdamFuse{}, LagSum{});             \
        else if (sp == 0) launch_fast<BLK, NPT, EPT, false, 0, double2, 0, false, false>(A, n, AdamFuse{}, LagSum{});    \
        else launch_fast<BLK, NPT, EPT, false, 16, double2, 0, false, false>(A, n_lag, AdamFuse{}, lag);                 \
        return true;                                                                                                      \
    }
    if (blk == 512 && h.max_nodes <= 512 && h.max_elems <= 2 * 512) HFEM_FAST_HB(512, 1, 2)
    if (blk == 256 && h.max_nodes <= 2 * 256 && h.max_elems <= 4 * 256) HFEM_FAST_HB(256, 2, 4)
    if (blk == 512 && h.max_nodes <= 2 * 512 && h.max_elems <= 3 * 512) HFEM_FAST_HB(512, 2, 3)
    if (blk == 512 && h.max_nodes <= 2 * 512 && h.max_elems <= 4 * 512) HFEM_FAST_HB(512, 2, 4)
    if (blk == 256 && h.max_nodes <= 4 * 256 && h.max_elems <= 6 * 256) HFEM_FAST_HB(256, 4, 6)
    if (blk == 1024 && h.max_nodes <= 1024 && h.max_elems <= 2 * 1024) HFEM_FAST_HB(1024, 1, 2)
#undef HFEM_FAST_HB
    return false;
}

#ifdef HFEM_LAB
#define HFEM_LAB_SECTION 3           // lab launches + the hooks used below
#include "tri3_energy_lab.inc"
#undef HFEM_LAB_SECTION
#else
#define HFEM_LAB_UPLOAD_STAMPS(raw, h, rc)
#define HFEM_LAB_REFRESH_TUNE(plan)
#define HFEM_LAB_TRI3_LAUNCH(plan, A, n, hasb, phys, lag, n_partials, launched)
#define HFEM_LAB_PAIR_LAUNCH(plan, P, n, hasb, phys, lag, rc_pair)
#endif

// Host-side launch state of a plan (partials bank of the lagged loss sum).  One plan = one stream at a time: the state
// is guarded by the plan's mutex, and a HFEM_FLAG_SUM_PREVIOUS launch must come on the stream that left the partials.
struct PlanLock {
    explicit PlanLock(hfem_plan *p) : g(p->mu) {}
    std::lock_guard<std::mutex> g;
};

template <typename T>
int hfem_upload(T **dst, const void *src, size_t count, int64_t &bytes) {
    const size_t nb = std::max<size_t>(count, 1) * sizeof(T);
    HFEM_HIP_CHECK(hipMalloc((void **)dst, nb));
    if (count && src) HFEM_HIP_CHECK(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    if (!src) HFEM_HIP_CHECK(hipMemset(*dst, 0, nb));
    bytes += (int64_t)nb;
    return 0;
}
}  // namespace

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

extern "C" int hfem_plan_destroy(hfem_plan *plan) {
    if (!plan) return 0;
    if (plan->device >= 0) {
        (void)hipSetDevice(plan->device);
        (void)hipFree(plan->d_tiles);
        (void)hipFree(plan->d_elem_pack);
        (void)hipFree(plan->d_elem_pack_hi);
        (void)hipFree(plan->d_node_src);
        (void)hipFree(plan->d_edge_pack);
        (void)hipFree(plan->d_edge_gid);
        (void)hipFree(plan->d_tile_chunks);
        free_tri3_det(plan);
        (void)hipFree(plan->d_partials);
        (void)hipFree(plan->d_stamps);
    }
    delete plan;
    return 0;
}

namespace {
// LDS footprint + device mirrors of a complete host plan (shared by hfem_plan_create_ex and hfem_plan_deserialize).
int finalize_plan(std::unique_ptr<hfem_plan> &p, int device) {
    const HostPlan &h = p->host;
    p->lds_bytes = h.max_nodes * 32 + h.max_owned * 32 + 128;
    p->lds_bytes_pipe = p->lds_bytes + 32 * kPipeMaxTiles + (h.npe == 4 ? 8 : 4) * ((h.max_elems + 3) & ~3);
    if (device >= 0) {
        if (int rc = use_device(device)) return rc;
        p->device = device;
        hfem_plan *raw = p.get();
        int rc = 0;
        if (!rc) rc = hfem_upload(&raw->d_tiles, h.tiles.data(), h.tiles.size(), raw->device_bytes);
        if (!rc) rc = hfem_upload(&raw->d_elem_pack, h.elem_pack.data(), h.elem_pack.size(), raw->device_bytes);
        if (!rc && (h.npe == 4 || h.paired)) rc = hfem_upload(&raw->d_elem_pack_hi, h.elem_pack_hi.data(), h.elem_pack_hi.size(), raw->device_bytes);
        if (!rc) rc = hfem_upload(&raw->d_node_src, h.node_src.data(), h.node_src.size() / 2, raw->device_bytes);
        if (!rc) rc = hfem_upload(&raw->d_edge_pack, h.edge_pack.data(), h.edge_pack.size(), raw->device_bytes);
        if (!rc) rc = hfem_upload(&raw->d_edge_gid, h.edge_gid.data(), h.edge_gid.size(), raw->device_bytes);
        if (!rc && !h.tile_chunks.empty()) rc = hfem_upload(&raw->d_tile_chunks, h.tile_chunks.data(), h.tile_chunks.size() / 4, raw->device_bytes);
        if (!rc) rc = hfem_upload(&raw->d_partials, nullptr, 2 * h.tiles.size(), raw->device_bytes);   // two banks
        HFEM_LAB_UPLOAD_STAMPS(raw, h, rc)
        if (!rc && p->lds_bytes > 64 * 1024) {
            set_error("plan: tile needs more than 64 KiB of LDS");
            rc = -1;
        }
        if (rc) { hfem_plan_destroy(p.release()); return rc; }
    }
    return 0;
}
}  // namespace

extern "C" int hfem_plan_create(int device, const int64_t *conn, int64_t ne, int64_t nn,
                                const double *coords_hint, const int32_t *x_src, const int32_t *u_src,
                                const int64_t *edges, int64_t ned, int32_t tile_elems, hfem_plan **out) {
    return hfem_plan_create_ex(device, conn, ne, nn, 3, coords_hint, x_src, u_src, edges, ned, tile_elems, out);
}

extern "C" int hfem_plan_create_ex(int device, const int64_t *conn, int64_t ne, int64_t nn, int32_t nodes_per_elem,
                                   const double *coords_hint, const int32_t *x_src, const int32_t *u_src,
                                   const int64_t *edges, int64_t ned, int32_t tile_elems, hfem_plan **out) {
    HFEM_ARG_CHECK(out, "null out pointer");
    *out = nullptr;
    std::unique_ptr<hfem_plan> p(new hfem_plan);
    p->tune = current_tune();
    // Gradient-store policy by mesh size (scripts/store_policy_sweep.py, profiles/r03/store_policy_sweep.jsonl; fraction of the
    // 8 TB/s roofline, same buffers every launch / rotating sets that exceed the Infinity Cache):
    //   10^6 elements    sc1 0.60 / 0.46    nt 0.55 / 0.51
    //   2 10^6           sc1 0.57 / 0.48    nt 0.56 / 0.54
    //   4 10^6           sc1 0.66 / 0.56    nt 0.66 / 0.65
    // Write-through stores of 16-byte rows are partial-line writes; on lines the Infinity Cache does not hold they cost a fill.
    // nt stores are merged in L2 and leave at the kernel's end: regime-independent, 9 % slower only where everything is
    // cache-resident.  A training loop's working set (x, u, gradients, two moments each, the plan: ~190 B per node) stays in
    // the 256 MB Infinity Cache up to ~10^6 nodes; from 750 k nodes the plan takes nt stores.
    const bool auto_store = p->tune.store_policy < 0;
    if (auto_store) p->tune.store_policy = nn >= 750000 ? 2 : 16;
    int32_t node_cap = g_def.plan_node_cap.load();
    const int32_t shards = std::max(1, g_def.plan_shards.load());
    int32_t pair_block = g_def.plan_pair_block.load();
    // Shard-aware tile policy (TRI3, tile size left to the library).  A launch over E elements is one resident round of
    // workgroups whatever E is, so what a small launch costs is the critical path of ONE tile: fewer slot rows per thread
    // and, below ~256 tiles, smaller tiles.  Measured on one MI355X, kernel only, per-rank tile ranges of T1M and whole
    // meshes of the same sizes (profiles/r03/shard_sweep_*.jsonl; us per launch, old -> new):
    //   E > 600 k           557 home nodes per tile, 256 threads, three slot rows per thread      (1 M: 8.96; 512 threads: 11.0)
    //   200 k < E <= 600 k  the same tiles, 512 threads, two slot rows                            (500 k: 6.4 -> 6.1, 250 k: 5.3 -> 4.8)
    //   100 k < E <= 200 k  430 home nodes, 512 threads, ONE slot row, <= 1 tile per CU            (125 k: 5.2 -> 4.2)
    //   E <= 100 k          208 home nodes, 256 threads, one slot row                             (62 k: 5.1 -> 3.7, 10 k: 5.2 -> 3.7)
    const int64_t per_shard = ne / shards;
    if (node_cap < 0) {
        node_cap = 0;
        if (tile_elems <= 0) {
            node_cap = 557;
            if (nodes_per_elem == 3 && per_shard <= 600000) {
                if (pair_block < 0) pair_block = per_shard > 100000 ? 512 : 256;
                if (per_shard <= 200000) node_cap = pair_block == 512 ? 430 : 208;
            }
        }
    }
    if (pair_block < 0) pair_block = 256;
    if (tile_elems <= 0) tile_elems = node_cap > 0 ? 1200 : 1024;
    int order = g_def.plan_elem_order.load();
#ifndef HFEM_LAB
    // the strip order's carrying slot loop measured slower and ships in the lab build only; the PLANNER keeps
    // the order (host-only plans: tests/test_plan_host.py, scripts/plan_bank_stats.py)
    HFEM_ARG_CHECK(!(order == 6 && device >= 0), "plan_elem_order 6 (pairs chained into strips): device plans in the lab build only (libhidenn_hip_lab.so)");
#endif
    const bool auto_order = order < 0;
    if (auto_order) order = nodes_per_elem == 3 ? 5 : 3;
    if (build_host_plan(conn, nodes_per_elem, ne, nn, coords_hint, x_src, u_src, edges, ned, tile_elems, node_cap, order,
                        g_def.plan_chunk_cap.load(), p->host, pair_block, shards))
        return -1;
    if (auto_order && p->host.paired && 2 * p->host.n_pairs * 10 < 7 * ne) {
        // few fan-adjacent partners (local node orders are what they are -- they cannot be rotated): the
        // one-element-per-slot order with the 512-thread kernel is faster there.  Crossover measured after the round-2 prologue
        // and packing work (profiles/r02/r2_lab32_*.jsonl; the ratio counts halo slots, ~1.13 x the element coverage): random
        // diagonals (73 % of the elements pair, ratio 0.82) paired 10.35 vs 10.65 us; zigzag (50 %, 0.57) 11.67 vs 11.18 us;
        // Delaunay (39 %, 0.44) 42.4-43.0 vs 41.9 us
        p->host = HostPlan();
        if (build_host_plan(conn, nodes_per_elem, ne, nn, coords_hint, x_src, u_src, edges, ned, tile_elems, node_cap, 3,
                            g_def.plan_chunk_cap.load(), p->host, 256, shards))
            return -1;
    }
    const HostPlan &h = p->host;
    // Locality figure of the caller's row numbering as the tiles see it: distinct 128-byte lines (8 rows of 16 bytes) among a
    // tile's coordinate rows / the minimum, averaged over the tiles.  ~1.3 for stored-along-the-curve or row-major structured
    // numberings, up to 8 for a random one (every gathered row pulls its own line; hidenn_fem_amd/models.py repairs that
    // with reorder="auto").  nt stores are merged line-wise in L2 -- on scattered rows there is nothing to merge and they
    // lose (cfg5 as numbered: 169 vs 128 us), so such plans keep the write-through stores.
    {
        double acc = 0.0;
        int64_t cnt = 0;
        std::vector<int32_t> lines;
        for (const TileDesc &d : h.tiles) {
            lines.clear();
            for (int32_t l = 0; l < d.n_node; ++l) {
                const int32_t r = h.node_src[2 * ((size_t)d.node_off + l)];
                if (r >= 0) lines.push_back(r >> 3);
            }
            if (lines.empty()) continue;
            const size_t rows = lines.size();
            std::sort(lines.begin(), lines.end());
            const size_t distinct = std::unique(lines.begin(), lines.end()) - lines.begin();
            acc += (double)distinct / (double)((rows + 7) / 8);
            ++cnt;
        }
        p->row_line_factor = cnt ? acc / (double)cnt : 1.0;
    }
    if (auto_store && p->tune.store_policy == 2 && p->row_line_factor > 2.5) p->tune.store_policy = 16;
    if (int rc = finalize_plan(p, device)) return rc;
    *out = p.release();
    return 0;
}

namespace {
struct PlanTrailer {   // what hfem_plan_create decided besides the HostPlan: travels in the blob's trailer
    int32_t tiled_block, store_policy, tiled_fast, fast_const_caps, pair_tiles_per_wg, pair_pipe_wps;
    int32_t hfem_version, reserved;
    double row_line_factor;
};
}  // namespace

extern "C" int64_t hfem_plan_serialize(const hfem_plan *plan, void *buf, int64_t cap_bytes) {
    if (!plan) { set_error("hfem_plan_serialize: null plan"); return -1; }
    PlanTrailer t{plan->tune.tiled_block, plan->tune.store_policy, plan->tune.tiled_fast, plan->tune.fast_const_caps,
                  plan->tune.pair_tiles_per_wg, plan->tune.pair_pipe_wps, HFEM_VERSION, 0, plan->row_line_factor};
    std::vector<unsigned char> blob;
    serialize_host_plan(plan->host, &t, sizeof(t), blob);
    if (buf) {
        if (cap_bytes < (int64_t)blob.size()) { set_error("hfem_plan_serialize: buffer too small"); return -1; }
        std::memcpy(buf, blob.data(), blob.size());
    }
    return (int64_t)blob.size();
}

extern "C" int hfem_plan_deserialize(int device, const void *blob, int64_t n_bytes, hfem_plan **out) {
    HFEM_ARG_CHECK(out, "null out pointer");
    *out = nullptr;
    HFEM_ARG_CHECK(blob && n_bytes > 0, "empty blob");
    std::unique_ptr<hfem_plan> p(new hfem_plan);
    PlanTrailer t{};
    if (deserialize_host_plan(blob, (size_t)n_bytes, p->host, &t, sizeof(t))) return -1;
    HFEM_ARG_CHECK(t.hfem_version == HFEM_VERSION, "plan blob written by another library version: rebuild the plan");
    p->tune.tiled_block = t.tiled_block; p->tune.store_policy = t.store_policy; p->tune.tiled_fast = t.tiled_fast;
    p->tune.fast_const_caps = t.fast_const_caps; p->tune.pair_tiles_per_wg = t.pair_tiles_per_wg; p->tune.pair_pipe_wps = t.pair_pipe_wps;
    p->row_line_factor = t.row_line_factor;
    if (int rc = finalize_plan(p, device)) return rc;
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
    out->tile_elem_total = h.elem_records;
    out->tile_node_total = h.node_records;
    out->max_tile_nodes = h.max_nodes; out->max_tile_owned = h.max_owned;
    out->max_tile_elems = h.max_elems; out->max_tile_edges = h.max_edges;
    out->device_bytes = plan->device_bytes;
    out->lds_bytes = plan->lds_bytes;
    out->shards = h.shards;
    out->threads_per_tile = h.paired ? h.pair_block : (h.npe == 4 ? 256 : plan->tune.tiled_block);
    out->paired = h.paired ? 1 : 0;
    out->slot_rows = h.paired ? h.max_rows : 0;
    out->store_policy = plan->tune.store_policy;
    out->nodes_per_elem = h.npe;
    out->row_line_factor = plan->row_line_factor;
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
        case 7: src = h.elem_pack_hi.data(); n = (int64_t)h.elem_pack_hi.size(); break;
        case 8: src = h.tile_chunks.data(); n = (int64_t)h.tile_chunks.size(); break;
        case 9: src = h.elem_gid_b.data(); n = (int64_t)h.elem_gid_b.size(); break;
        case 10: src = h.shard_desc.data(); n = (int64_t)h.shard_desc.size(); break;
        case 11: src = h.owned_gid.data(); n = (int64_t)h.owned_gid.size(); break;
        case 6: {   // lab build: device stamps, 16 x uint64 per tile, returned as 32 x int32 per tile
            n = (int64_t)h.tiles.size() * 32;
            if (buf) {
                if (cap_elems < n || plan->device < 0 || !plan->d_stamps) { set_error("hfem_plan_export: stamps need the lab build, a device plan and a buffer"); return -1; }
                (void)hipSetDevice(plan->device);
                if (hipMemcpy(buf, plan->d_stamps, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_error("hfem_plan_export: stamp copy failed"); return -1; }
            }
            return n;
        }
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
    HFEM_ARG_CHECK(plan->host.npe == 3, "this plan was built for QUAD4: use hfem_quad4_energy_plan");
    HFEM_ARG_CHECK(x_free && u_free, "x_free / u_free must be given");
    const HostPlan &h = plan->host;
    const int32_t nt = (int32_t)h.tiles.size();
    if (tile_end < 0) tile_end = nt;
    HFEM_ARG_CHECK(tile_begin >= 0 && tile_begin <= tile_end && tile_end <= nt, "bad tile range");
    HFEM_ARG_CHECK(h.ned == 0 || T_edge || Tconst, "plan has Neumann edges: need a traction table");
    if (int rc = use_device(plan->device)) return rc;
    PlanLock lock(plan);
    HFEM_LAB_REFRESH_TUNE(plan)
    hipStream_t s = (hipStream_t)stream;
    const int n = tile_end - tile_begin;
    const double4 tc = Tconst ? make_double4(Tconst[0], Tconst[1], Tconst[2], Tconst[3]) : make_double4(0, 0, 0, 0);
    bool hasb = false;
    for (int i = 0; i < 6; ++i) hasb = hasb || (Bk && Bk[i] != 0.0);
    const bool phys = (flags & HFEM_FLAG_PHYSICAL_GRAD) != 0;
    const bool phys_fast = phys && !(flags & HFEM_FLAG_DETERMINISTIC) &&
                           (h.paired || (plan->tune.tiled_block == 512 && h.max_nodes <= 2 * 512 && h.max_elems <= 4 * 512));
    if ((flags & HFEM_FLAG_DETERMINISTIC) || (phys && !phys_fast)) {
        // fixed-order node-centric path (tri3_det.hip); also carries the physical convention for plan shapes the
        // tiled PHYS instance does not hold
        HFEM_ARG_CHECK(tile_begin == 0 && tile_end == nt, "HFEM_FLAG_DETERMINISTIC / physical fallback: whole plan only");
        HFEM_ARG_CHECK(!(flags & (HFEM_FLAG_NO_LOSS_SUM | HFEM_FLAG_SUM_PREVIOUS | HFEM_FLAG_PEER_GET)), "HFEM_FLAG_DETERMINISTIC always delivers the loss and has no in-launch get");
        return launch_tri3_det(plan, x_free, x_fixed, u_free, u_fixed, make_consts(mat, W, Bk), T_edge, tc, loss_out,
                               (flags & HFEM_FLAG_NO_GX) ? nullptr : gx_free, (flags & HFEM_FLAG_NO_GU) ? nullptr : gu_free,
                               (flags & HFEM_FLAG_NO_EDGES) ? 1 : 0, phys, s);
    }
    // partials banks: a launch that leaves its tile energies unsummed writes the OTHER bank, so that the next launch
    // can reduce them with its one extra workgroup (HFEM_FLAG_SUM_PREVIOUS) while it fills this one
    const bool lag_consume = (flags & HFEM_FLAG_SUM_PREVIOUS) != 0;
    HFEM_ARG_CHECK(!lag_consume || (flags & HFEM_FLAG_NO_LOSS_SUM), "HFEM_FLAG_SUM_PREVIOUS needs HFEM_FLAG_NO_LOSS_SUM");
    HFEM_ARG_CHECK(!lag_consume || (plan->prev_n > 0 && n > 0), "HFEM_FLAG_SUM_PREVIOUS: no previous unsummed launch on this plan");
    HFEM_ARG_CHECK(!lag_consume || plan->prev_stream == stream,
                   "HFEM_FLAG_SUM_PREVIOUS: the previous unsummed launch went to another stream (one plan = one stream)");
    // HFEM_FLAG_SAME_BANK: this launch is another tile range of the SAME evaluation as the previous unsummed launch (boundary
    // tiles after interior tiles, hidenn_fem_amd/sharded.py): its tile energies go to the bank that launch wrote
    const bool same_bank = (flags & HFEM_FLAG_SAME_BANK) != 0;
    HFEM_ARG_CHECK(!same_bank || ((flags & HFEM_FLAG_NO_LOSS_SUM) && !lag_consume), "HFEM_FLAG_SAME_BANK needs HFEM_FLAG_NO_LOSS_SUM and excludes HFEM_FLAG_SUM_PREVIOUS");
    const int wbank = (flags & HFEM_FLAG_NO_LOSS_SUM) ? (same_bank ? plan->bank : (plan->bank ^ 1)) : plan->bank;
    double *pbase = plan->d_partials + (size_t)wbank * nt;
    LagSum lag;
    if (lag_consume) {
        HFEM_ARG_CHECK(h.paired || (plan->tune.tiled_fast && plan->tune.tiled_block == 512 && !hasb && !phys && plan->tune.store_policy != 0 &&
                       h.max_nodes <= 1024 && h.max_elems <= 2048),
                       "HFEM_FLAG_SUM_PREVIOUS: only the default (register-prefetched, 512-thread) kernel path implements it");
        lag.prev = plan->d_partials + (size_t)plan->bank * nt + plan->prev_begin;
        lag.prev_n = plan->prev_n;
        lag.out = loss_out;
    }
    if (flags & HFEM_FLAG_PEER_GET) {
        HFEM_ARG_CHECK(plan->peer_get, "HFEM_FLAG_PEER_GET: hfem_plan_set_peer_get has not been called");
        HFEM_ARG_CHECK(!hasb && !phys && n > 0, "HFEM_FLAG_PEER_GET: default forces and convention, a non-empty tile range");
        lag.pg = plan->peer_get; lag.pg_blocks = kPeerGetBlocks;
        lag.wait_begin = plan->peer_wait_begin; lag.wait_end = plan->peer_wait_end;
    }
    int n_partials = n;
    if (n > 0) {
        Tri3Launch A;
        A.pd = plan_dev(plan); A.tile_begin = (int)tile_begin;
        A.x_free = x_free; A.x_fixed = x_fixed; A.u_free = u_free; A.u_fixed = u_fixed;
        A.k = make_consts(mat, W, Bk); A.T_edge = (const double4 *)T_edge; A.tc = tc; A.partials = pbase + tile_begin;
        A.gx = (flags & HFEM_FLAG_NO_GX) ? nullptr : gx_free; A.gu = (flags & HFEM_FLAG_NO_GU) ? nullptr : gu_free;
        A.max_nodes = h.max_nodes; A.max_owned = h.max_owned; A.skip_edges = (flags & HFEM_FLAG_NO_EDGES) ? 1 : 0;
        A.stamps = plan->d_stamps; A.lds = (size_t)plan->lds_bytes; A.s = s;
        bool launched = false;
        HFEM_LAB_TRI3_LAUNCH(plan, A, n, hasb, phys, lag, n_partials, launched)
        if (!launched && h.paired) {      // paired plan (plan_elem_order 5, the default): its records are the pair kernel's
            PairLaunch P;
            P.grid = n + (lag.prev ? 1 : 0) + lag.pg_blocks; P.tile_begin = (int)tile_begin;
            P.x_free = x_free; P.x_fixed = x_fixed; P.u_free = u_free; P.u_fixed = u_fixed;
            P.k = A.k; P.T_edge = A.T_edge; P.tc = tc; P.partials = A.partials; P.gx = A.gx; P.gu = A.gu;
            P.skip_edges = A.skip_edges; P.s = s;

            HFEM_ARG_CHECK(!(lag.prev && (hasb || phys)), "HFEM_FLAG_SUM_PREVIOUS: default forces and convention only");
            if (plan->span_buf) P.span = plan->span_buf + (size_t)(plan->span_next++ % plan->span_slots) * 2 * (size_t)nt;
            int rc_pair = 0;
            HFEM_LAB_PAIR_LAUNCH(plan, P, n, hasb, phys, lag, rc_pair)
            if (!rc_pair) rc_pair = launch_tri3_pair(plan, P, (hasb || phys) ? 1 : 0, hasb, phys, lag, AdamFuse{});
            HFEM_ARG_CHECK(rc_pair == 1, "paired plan: tile shape outside the pair kernel's instances");
            launched = true;
        }
        if (!launched && phys_fast && !h.paired) {     // opt-in physical convention: one general instance of the register-prefetched kernel
            launch_fast<512, 2, 4, true, 16, double2, 0, false, true>(A, n, AdamFuse{}, LagSum{});
            launched = true;
        }
        if (!launched && plan->tune.tiled_fast) launched = launch_fast_f64(plan, A, n, hasb, lag);
        if (!launched) {
            HFEM_ARG_CHECK(!lag_consume, "HFEM_FLAG_SUM_PREVIOUS: this plan's tile shape has no register-prefetched instance");
            HFEM_ARG_CHECK(!lag.pg_blocks, "HFEM_FLAG_PEER_GET: this plan's tile shape has no instance with the in-launch get");
            switch (plan->tune.tiled_block) {
                case 512: launch_generic<512>(A, n); break;
                case 1024: launch_generic<1024>(A, n); break;
                default: launch_generic<256>(A, n); break;
            }
        }
        if (int rc = launch_status("hfem_tri3_energy_plan")) return rc;
    }
    if (flags & HFEM_FLAG_NO_LOSS_SUM) {
        if (same_bank && plan->prev_n > 0 && n > 0 && n_partials == n &&
            (tile_end == plan->prev_begin || tile_begin == plan->prev_begin + plan->prev_n)) {
            plan->prev_begin = std::min(plan->prev_begin, (int)tile_begin);      // adjacent ranges of one evaluation: their union
            plan->prev_n += n;
        } else if (!same_bank || n > 0) {
            plan->prev_begin = tile_begin; plan->prev_n = n > 0 ? n_partials : 0;
        }
        plan->bank = wbank; plan->prev_stream = stream;
        return 0;
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, s, pbase + tile_begin, n_partials, loss_out);
    return launch_status("hfem_tri3_energy_plan(sum)");
}

// fp32-storage form: x/u rows and the gradient rows are float2 (an fp32 model -- the reference's default dtype --
// without widening copies); arithmetic, LDS staging and the loss are fp64 exactly as above.  Only the
// register-prefetched kernel at 512 threads per tile implements it; other plan shapes return an error and the
// caller widens (hidenn_fem_amd/ops.py does).
extern "C" int hfem_tri3_energy_plan_f32(hfem_plan *plan, const float *x_free, const float *x_fixed,
                                         const float *u_free, const float *u_fixed, const double mat[4], double W,
                                         const double Bk[6], const double *T_edge, const double Tconst[4],
                                         int32_t tile_begin, int32_t tile_end, double *loss_out, float *gx_free,
                                         float *gu_free, int32_t flags, void *stream) {
    HFEM_ARG_CHECK(plan && mat && loss_out, "null pointer");
    HFEM_ARG_CHECK(plan->device >= 0, "host-only plan (created with device < 0) cannot launch");
    HFEM_ARG_CHECK(plan->host.npe == 3, "this plan was built for QUAD4");
    HFEM_ARG_CHECK(x_free && u_free, "x_free / u_free must be given");
    const bool f32math = (flags & HFEM_FLAG_FP32_MATH) != 0;
    HFEM_ARG_CHECK(!(flags & (HFEM_FLAG_PHYSICAL_GRAD | HFEM_FLAG_DETERMINISTIC)), "fp32-storage path: reference convention, atomic accumulation");
    HFEM_ARG_CHECK(f32math || !(flags & HFEM_FLAG_SUM_PREVIOUS), "fp32 rows with fp64 arithmetic: no lagged loss sum (HFEM_FLAG_FP32_MATH has it)");
    const HostPlan &h = plan->host;
    const int32_t nt = (int32_t)h.tiles.size();
    if (tile_end < 0) tile_end = nt;
    HFEM_ARG_CHECK(tile_begin >= 0 && tile_begin <= tile_end && tile_end <= nt, "bad tile range");
    HFEM_ARG_CHECK(h.ned == 0 || T_edge || Tconst, "plan has Neumann edges: need a traction table");
    bool hasb = false;
    for (int i = 0; i < 6; ++i) hasb = hasb || (Bk && Bk[i] != 0.0);
    if (f32math) {
        // fp32 ARITHMETIC (tri3_pair_f32.hip): paired-slot plans without chained records; body force, tile ranges, the lagged
        // loss sum and the partials banks as hfem_tri3_energy_plan; no in-launch get
        HFEM_ARG_CHECK(h.paired && h.n_chained == 0, "HFEM_FLAG_FP32_MATH: paired-slot plans only (this mesh's plan keeps one element per slot: drop the flag)");
        HFEM_ARG_CHECK(!(flags & HFEM_FLAG_PEER_GET), "HFEM_FLAG_FP32_MATH: no in-launch get (the sharded steps run the fp64-arithmetic float-row instances)");
        if (int rc = use_device(plan->device)) return rc;
        PlanLock lock(plan);
        const int n = tile_end - tile_begin;
        const bool lag_consume = (flags & HFEM_FLAG_SUM_PREVIOUS) != 0, same_bank = (flags & HFEM_FLAG_SAME_BANK) != 0;
        HFEM_ARG_CHECK(!lag_consume || (flags & HFEM_FLAG_NO_LOSS_SUM), "HFEM_FLAG_SUM_PREVIOUS needs HFEM_FLAG_NO_LOSS_SUM");
        HFEM_ARG_CHECK(!lag_consume || (plan->prev_n > 0 && n > 0), "HFEM_FLAG_SUM_PREVIOUS: no previous unsummed launch on this plan");
        HFEM_ARG_CHECK(!lag_consume || plan->prev_stream == stream, "HFEM_FLAG_SUM_PREVIOUS: the previous unsummed launch went to another stream (one plan = one stream)");
        HFEM_ARG_CHECK(!same_bank || ((flags & HFEM_FLAG_NO_LOSS_SUM) && !lag_consume), "HFEM_FLAG_SAME_BANK needs HFEM_FLAG_NO_LOSS_SUM and excludes HFEM_FLAG_SUM_PREVIOUS");
        const int wbank = (flags & HFEM_FLAG_NO_LOSS_SUM) ? (same_bank ? plan->bank : (plan->bank ^ 1)) : plan->bank;
        double *pb = plan->d_partials + (size_t)wbank * nt;
        LagSum lag;
        if (lag_consume) {
            lag.prev = plan->d_partials + (size_t)plan->bank * nt + plan->prev_begin;
            lag.prev_n = plan->prev_n;
            lag.out = loss_out;
        }
        if (n > 0) {
            PairLaunch P;
            P.grid = n + (lag.prev ? 1 : 0); P.tile_begin = (int)tile_begin;
            P.x_free = x_free; P.x_fixed = x_fixed; P.u_free = u_free; P.u_fixed = u_fixed;
            P.k = make_consts(mat, W, Bk); P.T_edge = (const double4 *)T_edge;
            P.tc = Tconst ? make_double4(Tconst[0], Tconst[1], Tconst[2], Tconst[3]) : make_double4(0, 0, 0, 0);
            P.partials = pb + tile_begin;
            P.gx = (flags & HFEM_FLAG_NO_GX) ? nullptr : gx_free; P.gu = (flags & HFEM_FLAG_NO_GU) ? nullptr : gu_free;
            P.skip_edges = (flags & HFEM_FLAG_NO_EDGES) ? 1 : 0; P.s = (hipStream_t)stream;
            HFEM_ARG_CHECK(launch_tri3_pair_f32(plan, P, hasb, lag) == 1, "HFEM_FLAG_FP32_MATH: tile shape outside the fp32 pair kernel's instances");
            if (int rc = launch_status("hfem_tri3_energy_plan_f32(fp32 arithmetic)")) return rc;
        }
        if (flags & HFEM_FLAG_NO_LOSS_SUM) {              // the partials-bank bookkeeping of hfem_tri3_energy_plan
            if (same_bank && plan->prev_n > 0 && n > 0 && (tile_end == plan->prev_begin || tile_begin == plan->prev_begin + plan->prev_n)) {
                plan->prev_begin = std::min(plan->prev_begin, (int)tile_begin);
                plan->prev_n += n;
            } else if (!same_bank || n > 0) {
                plan->prev_begin = tile_begin; plan->prev_n = n;
            }
            plan->bank = wbank; plan->prev_stream = stream;
            return 0;
        }
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, pb + tile_begin, n, loss_out);
        return launch_status("hfem_tri3_energy_plan_f32(sum)");
    }
    HFEM_ARG_CHECK(!hasb && h.max_nodes <= 2 * 512 && h.max_elems <= 4 * 512,
                   "fp32-storage path: needs a zero body force and tiles of <= 1024 nodes / 2048 slots");
    if (int rc = use_device(plan->device)) return rc;
    PlanLock lock(plan);
    hipStream_t s = (hipStream_t)stream;
    const int n = tile_end - tile_begin;
    double *pbase = plan->d_partials + (size_t)plan->bank * nt + tile_begin;
    LagSum lag;
    if (flags & HFEM_FLAG_PEER_GET) {
        HFEM_ARG_CHECK(plan->peer_get && n > 0, "HFEM_FLAG_PEER_GET: hfem_plan_set_peer_get first, and a non-empty tile range");
        lag.pg = plan->peer_get; lag.pg_blocks = kPeerGetBlocks;
        lag.wait_begin = plan->peer_wait_begin; lag.wait_end = plan->peer_wait_end;
    }
    if (n > 0 && h.paired) {
        PairLaunch P;
        P.grid = n + lag.pg_blocks; P.tile_begin = (int)tile_begin;
        P.x_free = x_free; P.x_fixed = x_fixed; P.u_free = u_free; P.u_fixed = u_fixed;
        P.k = make_consts(mat, W, Bk); P.T_edge = (const double4 *)T_edge;
        P.tc = Tconst ? make_double4(Tconst[0], Tconst[1], Tconst[2], Tconst[3]) : make_double4(0, 0, 0, 0);
        P.partials = pbase;
        P.gx = (flags & HFEM_FLAG_NO_GX) ? nullptr : gx_free; P.gu = (flags & HFEM_FLAG_NO_GU) ? nullptr : gu_free;
        P.skip_edges = (flags & HFEM_FLAG_NO_EDGES) ? 1 : 0; P.s = s;
        HFEM_ARG_CHECK(launch_tri3_pair(plan, P, 2, false, false, lag, AdamFuse{}) == 1,
                       "paired plan: tile shape outside the pair kernel's instances");
        if (int rc = launch_status("hfem_tri3_energy_plan_f32")) return rc;
    } else if (n > 0) {
        Tri3Launch A;
        A.pd = plan_dev(plan); A.tile_begin = (int)tile_begin;
        A.x_free = x_free; A.x_fixed = x_fixed; A.u_free = u_free; A.u_fixed = u_fixed;
        A.k = make_consts(mat, W, Bk); A.T_edge = (const double4 *)T_edge;
        A.tc = Tconst ? make_double4(Tconst[0], Tconst[1], Tconst[2], Tconst[3]) : make_double4(0, 0, 0, 0);
        A.partials = pbase;
        A.gx = (flags & HFEM_FLAG_NO_GX) ? nullptr : gx_free; A.gu = (flags & HFEM_FLAG_NO_GU) ? nullptr : gu_free;
        A.max_nodes = h.max_nodes; A.max_owned = h.max_owned; A.skip_edges = (flags & HFEM_FLAG_NO_EDGES) ? 1 : 0;
        A.stamps = plan->d_stamps; A.lds = (size_t)plan->lds_bytes; A.s = s;
        if (h.max_elems <= 3 * 512) launch_fast<512, 2, 3, false, 16, float2, 0, false, false>(A, n + lag.pg_blocks, AdamFuse{}, lag);
        else launch_fast<512, 2, 4, false, 16, float2, 0, false, false>(A, n + lag.pg_blocks, AdamFuse{}, lag);
        if (int rc = launch_status("hfem_tri3_energy_plan_f32")) return rc;
    }
    if (flags & HFEM_FLAG_NO_LOSS_SUM) return 0;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, s, pbase, n, loss_out);
    return launch_status("hfem_tri3_energy_plan_f32(sum)");
}

// One training step in one launch: energy + gradients as hfem_tri3_energy_plan, but the write-out applies
// torch.optim.Adam's update (betas, eps, bias correction; no weight decay / amsgrad; one learning rate per parameter
// tensor, as the reference's commented example-4 variant sets them) instead of storing the gradient:
//   m, v   [rows][2] moments of node_coords_free / u_free, updated in place
//   x_out, u_out   the NEW parameter rows -- buffers different from x_free / u_free (tiles that are still gathering
//                  must see the old values), to be swapped with them by the caller after the launch (ping-pong)
//   bc_dev         device {1 - beta1^step, sqrt(1 - beta2^step)} of this step: hfem_adam_prep(step counter, betas) writes it
// loss_out = the energy at the OLD parameters, as `loss = closure(); optimizer.step()` reports it.  Whole-plan launches
// only (every free row must be owned by a tile of the launch); default tile shape; zero body force.
extern "C" int hfem_tri3_energy_adam_step(hfem_plan *plan, const double *x_free, const double *x_fixed,
                                          const double *u_free, const double *u_fixed, const double mat[4], double W,
                                          const double *T_edge, const double Tconst[4], double *x_out, double *u_out,
                                          double *m_x, double *v_x, double *m_u, double *v_u, double lr_x, double lr_u,
                                          double beta1, double beta2, double eps, const double *bc_dev,
                                          double *loss_out, int32_t flags, void *stream) {
    return hfem_tri3_energy_adam_step_ex(plan, 0, x_free, x_fixed, u_free, u_fixed, mat, W, nullptr, T_edge, Tconst, x_out,
                                         u_out, m_x, v_x, m_u, v_u, lr_x, lr_u, beta1, beta2, eps, bc_dev, 0, -1, loss_out,
                                         flags, stream);
}

// General form: dtype 0 = fp64 rows, 1 = fp32 rows (parameters, fixed rows, moments and new rows all float: an fp32 model,
// the reference's default dtype, trains in one launch per iteration without widening copies; element arithmetic and the
// loss stay fp64, the update is torch's fp32 arithmetic); Bk = the body-force table of hfem_tri3_energy_plan (NULL / zeros:
// none).
extern "C" int hfem_tri3_energy_adam_step_ex(hfem_plan *plan, int32_t dtype, const void *x_free, const void *x_fixed,
                                             const void *u_free, const void *u_fixed, const double mat[4], double W,
                                             const double Bk[6], const double *T_edge, const double Tconst[4], void *x_out,
                                             void *u_out, void *m_x, void *v_x, void *m_u, void *v_u, double lr_x,
                                             double lr_u, double beta1, double beta2, double eps, const double *bc_dev,
                                             int32_t tile_begin, int32_t tile_end, double *loss_out, int32_t flags,
                                             void *stream) {
    HFEM_ARG_CHECK(plan && mat && loss_out && x_free && u_free, "null pointer");
    HFEM_ARG_CHECK(dtype == 0 || dtype == 1, "dtype: 0 = fp64 rows, 1 = fp32 rows");
    HFEM_ARG_CHECK(x_out && u_out && m_x && v_x && m_u && v_u && bc_dev, "null optimiser buffer");
    HFEM_ARG_CHECK(x_out != x_free && u_out != u_free, "x_out / u_out must not alias the input parameters (ping-pong)");
    HFEM_ARG_CHECK(plan->device >= 0, "host-only plan (created with device < 0) cannot launch");
    HFEM_ARG_CHECK(plan->host.npe == 3, "this plan was built for QUAD4");
    const HostPlan &h = plan->host;
    HFEM_ARG_CHECK(h.ned == 0 || T_edge || Tconst, "plan has Neumann edges: need a traction table");
    HFEM_ARG_CHECK(h.max_nodes <= 2 * 512 && h.max_elems <= 4 * 512,
                   "fused Adam step: needs tiles of <= 1024 nodes / 2048 element slots");
    HFEM_ARG_CHECK(!(flags & (HFEM_FLAG_NO_GX | HFEM_FLAG_NO_GU)), "fused Adam step updates both parameter tensors");
    HFEM_ARG_CHECK(!(flags & (HFEM_FLAG_PHYSICAL_GRAD | HFEM_FLAG_DETERMINISTIC)), "fused Adam step: reference convention, atomic accumulation");
    if (flags & HFEM_FLAG_FP32_MATH) {
        HFEM_ARG_CHECK(dtype == 1, "HFEM_FLAG_FP32_MATH: fp32 rows (dtype 1)");
        HFEM_ARG_CHECK(plan->host.paired && plan->host.n_chained == 0, "HFEM_FLAG_FP32_MATH: paired-slot plans only (this mesh's plan keeps one element per slot: drop the flag)");
        HFEM_ARG_CHECK(!(flags & (HFEM_FLAG_PEER_GET | HFEM_FLAG_PEER_PUT)), "HFEM_FLAG_FP32_MATH: no in-launch get / put (the sharded steps run the fp64-arithmetic float-row instances)");
    }
    bool hasb = false;
    for (int i = 0; i < 6; ++i) hasb = hasb || (Bk && Bk[i] != 0.0);
    if (int rc = use_device(plan->device)) return rc;
    PlanLock lock(plan);
    hipStream_t s = (hipStream_t)stream;
    // a tile RANGE updates exactly the rows its tiles own (element sharding: hidenn_fem_amd/sharded.py, fused steps); the rows
    // of the other tiles are not touched in x_out / u_out -- the caller keeps both parameter buffers complete
    const int nt = (int)h.tiles.size();
    if (tile_end < 0) tile_end = nt;
    HFEM_ARG_CHECK(tile_begin >= 0 && tile_begin <= tile_end && tile_end <= nt, "bad tile range");
    const int n = tile_end - tile_begin;
    const bool lag_consume = (flags & HFEM_FLAG_SUM_PREVIOUS) != 0;
    const bool same_bank = (flags & HFEM_FLAG_SAME_BANK) != 0;
    HFEM_ARG_CHECK(!lag_consume || (flags & HFEM_FLAG_NO_LOSS_SUM), "HFEM_FLAG_SUM_PREVIOUS needs HFEM_FLAG_NO_LOSS_SUM");
    HFEM_ARG_CHECK(!same_bank || ((flags & HFEM_FLAG_NO_LOSS_SUM) && !lag_consume), "HFEM_FLAG_SAME_BANK needs HFEM_FLAG_NO_LOSS_SUM and excludes HFEM_FLAG_SUM_PREVIOUS");
    HFEM_ARG_CHECK(!lag_consume || (plan->prev_n > 0 && n > 0), "HFEM_FLAG_SUM_PREVIOUS: no previous unsummed launch on this plan");
    HFEM_ARG_CHECK(!lag_consume || plan->prev_stream == stream,
                   "HFEM_FLAG_SUM_PREVIOUS: the previous unsummed launch went to another stream (one plan = one stream)");
    HFEM_ARG_CHECK(!lag_consume || !hasb, "HFEM_FLAG_SUM_PREVIOUS: zero body force only");
    const int wbank = (flags & HFEM_FLAG_NO_LOSS_SUM) ? (same_bank ? plan->bank : (plan->bank ^ 1)) : plan->bank;
    double *pbase = plan->d_partials + (size_t)wbank * nt + tile_begin;
    LagSum lag;
    if (lag_consume) {
        lag.prev = plan->d_partials + (size_t)plan->bank * nt + plan->prev_begin;
        lag.prev_n = plan->prev_n;
        lag.out = loss_out;
    }
    if (flags & HFEM_FLAG_PEER_GET) {
        HFEM_ARG_CHECK(plan->peer_get, "HFEM_FLAG_PEER_GET: hfem_plan_set_peer_get has not been called");
        HFEM_ARG_CHECK(!hasb && n > 0, "HFEM_FLAG_PEER_GET: zero body force, a non-empty tile range");
        lag.pg = plan->peer_get; lag.pg_blocks = kPeerGetBlocks;
        lag.wait_begin = plan->peer_wait_begin; lag.wait_end = plan->peer_wait_end;
    }
    if (flags & HFEM_FLAG_PEER_PUT) {
        HFEM_ARG_CHECK((flags & HFEM_FLAG_PEER_GET) && plan->peer_put, "HFEM_FLAG_PEER_PUT: with HFEM_FLAG_PEER_GET, after hfem_plan_set_peer_put");
        HFEM_ARG_CHECK(h.paired && (flags & HFEM_FLAG_NO_LOSS_SUM) && !lag_consume && !same_bank,
                       "HFEM_FLAG_PEER_PUT: paired-slot plans, with HFEM_FLAG_NO_LOSS_SUM, one launch per evaluation");
        HFEM_ARG_CHECK(lag.wait_begin < lag.wait_end && tile_begin <= lag.wait_begin && lag.wait_end <= tile_end,
                       "HFEM_FLAG_PEER_PUT: the launch must cover the rank's (non-empty) boundary range");
        HFEM_ARG_CHECK(bc_dev == plan->put_bc[0] || bc_dev == plan->put_bc[1], "HFEM_FLAG_PEER_PUT: bc_dev must be one of the two buffers given to hfem_plan_set_peer_put");
        lag.put = plan->peer_put;
        lag.put_pos_x = plan->put_pos[0]; lag.put_pos_u = plan->put_pos[1];
        lag.put_bc_next = bc_dev == plan->put_bc[0] ? plan->put_bc[1] : plan->put_bc[0];
        if (plan->prev_n > 0 && plan->prev_stream == stream) {      // the previous evaluation's tile energies (the other bank)
            lag.put_prev = plan->d_partials + (size_t)plan->bank * nt + plan->prev_begin;
            lag.put_prev_n = plan->prev_n;
        }
    }
    if (n > 0) {
        AdamFuse af;
        af.x_out = x_out; af.u_out = u_out;
        af.mx = m_x; af.vx = v_x; af.mu = m_u; af.vu = v_u;
        af.bc = bc_dev; af.lr_x = lr_x; af.lr_u = lr_u; af.b1 = beta1; af.b2 = beta2; af.eps = eps;
        Tri3Launch A;
        A.pd = plan_dev(plan); A.tile_begin = tile_begin;
        A.x_free = x_free; A.x_fixed = x_fixed; A.u_free = u_free; A.u_fixed = u_fixed;
        A.k = make_consts(mat, W, hasb ? Bk : nullptr); A.T_edge = (const double4 *)T_edge;
        A.tc = Tconst ? make_double4(Tconst[0], Tconst[1], Tconst[2], Tconst[3]) : make_double4(0, 0, 0, 0);
        A.partials = pbase;
        A.max_nodes = h.max_nodes; A.max_owned = h.max_owned; A.skip_edges = (flags & HFEM_FLAG_NO_EDGES) ? 1 : 0;
        A.stamps = plan->d_stamps; A.lds = (size_t)plan->lds_bytes; A.s = s;
        const int grid = n + (lag_consume ? 1 : 0) + lag.pg_blocks;
        if (h.paired) {
            PairLaunch P;
            P.grid = grid; P.tile_begin = tile_begin;
            P.x_free = x_free; P.x_fixed = x_fixed; P.u_free = u_free; P.u_fixed = u_fixed;
            P.k = A.k; P.T_edge = A.T_edge; P.tc = A.tc; P.partials = pbase; P.skip_edges = A.skip_edges; P.s = s;
            if (flags & HFEM_FLAG_FP32_MATH) {          // fp32 rows AND fp32 arithmetic (csrc/tri3_pair_f32.hip, ADAM instances)
                HFEM_ARG_CHECK(launch_tri3_pair_f32(plan, P, hasb, lag, &af) == 1,
                               "HFEM_FLAG_FP32_MATH: tile shape outside the fp32 pair kernel's instances");
            } else
            HFEM_ARG_CHECK(launch_tri3_pair(plan, P, dtype == 0 ? 3 : 4, hasb, false, lag, af) == 1,
                           "paired plan: tile shape outside the pair kernel's instances");
        } else {
            // one element per slot (meshes whose elements do not pair): one general instance per row type and force
            const bool e3 = h.max_elems <= 3 * 512;
#define HFEM_ADAM_FAST(HB, V)                                                                            \
    {                                                                                                    \
        if (e3) launch_fast<512, 2, 3, HB, 16, V, 0, true, false>(A, grid, af, lag);                     \
        else launch_fast<512, 2, 4, HB, 16, V, 0, true, false>(A, grid, af, lag);                        \
    }
            if (dtype == 0 && !hasb) HFEM_ADAM_FAST(false, double2)
            else if (dtype == 0) HFEM_ADAM_FAST(true, double2)
            else if (!hasb) HFEM_ADAM_FAST(false, float2)
            else HFEM_ADAM_FAST(true, float2)
#undef HFEM_ADAM_FAST
        }
        if (int rc = launch_status("hfem_tri3_energy_adam_step")) return rc;
    }
    if (flags & HFEM_FLAG_NO_LOSS_SUM) {
        if (same_bank && plan->prev_n > 0 && n > 0 &&
            (tile_end == plan->prev_begin || tile_begin == plan->prev_begin + plan->prev_n)) {
            plan->prev_begin = std::min(plan->prev_begin, (int)tile_begin);      // adjacent ranges of one evaluation: their union
            plan->prev_n += n;
        } else if (!same_bank || n > 0) {
            plan->prev_begin = tile_begin; plan->prev_n = n;
        }
        plan->bank = wbank; plan->prev_stream = stream;
        return 0;
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, s, pbase, n, loss_out);
    return launch_status("hfem_tri3_energy_adam_step(sum)");
}

// Sum, in tile order, of the per-tile partial energies a launch with HFEM_FLAG_NO_LOSS_SUM left in the plan
// (what the energy entry points do themselves unless that flag is set): for gradient-driven callers that
// want the scalar only now and then.
extern "C" int hfem_plan_loss_sum(hfem_plan *plan, int32_t tile_begin, int32_t tile_end, double *loss_out,
                                  void *stream) {
    HFEM_ARG_CHECK(plan && loss_out, "null pointer");
    HFEM_ARG_CHECK(plan->device >= 0, "host-only plan (created with device < 0) cannot launch");
    const int32_t nt = (int32_t)plan->host.tiles.size();
    if (tile_end < 0) tile_end = nt;
    HFEM_ARG_CHECK(tile_begin >= 0 && tile_begin <= tile_end && tile_end <= nt, "bad tile range");
    if (int rc = use_device(plan->device)) return rc;
    PlanLock lock(plan);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       plan->d_partials + (size_t)plan->bank * nt + tile_begin, tile_end - tile_begin, loss_out);
    return launch_status("hfem_plan_loss_sum");
}

// Span stamps: a measurement aid for callers that need the duration of the energy kernel INSIDE a launch sequence
// (bench.py's cache regimes; HIP events only see whole sequences and a profiler is not always at hand).  While a buffer is
// set, launch i of the paired-slot kernel through hfem_tri3_energy_plan writes, for every tile it evaluates, the
// s_memrealtime tick (100 MHz) at which the tile's workgroup started and the tick at which its first wave had issued and
// drained its gradient stores, into dev_buf[(i % n_slots) * n_tiles * 2 + 2 * tile + {0, 1}].  max(end) - min(start) over
// the tiles of a launch is its first-workgroup-start to last-workgroup-end span.  NULL switches it off (the default; the
// kernel then pays one scalar compare).  Other kernels ignore the setting.
extern "C" int hfem_plan_set_span_stamps(hfem_plan *plan, uint64_t *dev_buf, int64_t n_slots) {
    HFEM_ARG_CHECK(plan, "null pointer");
    HFEM_ARG_CHECK(dev_buf == nullptr || n_slots >= 1, "need n_slots >= 1 with a buffer");
    PlanLock lock(plan);
    plan->span_buf = (unsigned long long *)dev_buf;
    plan->span_slots = dev_buf ? n_slots : 0;
    plan->span_next = 0;
    return 0;
}

// Owner-sharded step, one launch: pack this rank's interface rows into its all_gather payload (hfem_iface_pack), put
// the sum of the tile energies that the HFEM_FLAG_NO_LOSS_SUM launch(es) over [tile_begin, tile_end) left in the plan into
// out[loss_slot] (same order and bits as hfem_plan_loss_sum), and -- when `counter` is given -- bump that device counter
// (the optimiser's step count: hfem_adam_step_rows2_dev reads it BEFORE this launch with step_offset = 1) and, when bc_next is
// given too, write the bias-correction scalars {1 - beta1^(c + 1), sqrt(1 - beta2^(c + 1))} of the NEXT step (c = the bumped
// count) there -- what the fused energy + Adam launch of that step reads (hfem_tri3_energy_adam_step_ex), so no hfem_adam_prep.
static int plan_iface_pack_any(int dtype, hfem_plan *plan, int32_t tile_begin, int32_t tile_end, const void *x_free,
                               const void *u_free, const int32_t *rows, int32_t n_x, int32_t n_u, double *out,
                               int64_t loss_slot, int64_t *counter, double beta1, double beta2, double *bc_next,
                               void *stream) {
    HFEM_ARG_CHECK(plan && out, "null pointer");
    HFEM_ARG_CHECK(plan->device >= 0, "host-only plan (created with device < 0) cannot launch");
    HFEM_ARG_CHECK(n_x >= 0 && n_u >= 0 && loss_slot >= n_x + n_u, "bad sizes");
    HFEM_ARG_CHECK((n_x + n_u == 0 || rows) && (n_x == 0 || x_free) && (n_u == 0 || u_free), "null pointer");
    const int32_t nt = (int32_t)plan->host.tiles.size();
    if (tile_end < 0) tile_end = nt;
    HFEM_ARG_CHECK(tile_begin >= 0 && tile_begin <= tile_end && tile_end <= nt, "bad tile range");
    if (int rc = use_device(plan->device)) return rc;
    PlanLock lock(plan);
    return launch_iface_pack_sum(dtype, x_free, u_free, rows, n_x, n_u, out, loss_slot,
                                 plan->d_partials + (size_t)plan->bank * nt + tile_begin, tile_end - tile_begin, counter,
                                 beta1, beta2, bc_next, (hipStream_t)stream);
}

extern "C" int hfem_plan_iface_pack(hfem_plan *plan, int32_t tile_begin, int32_t tile_end, const double *x_free,
                                    const double *u_free, const int32_t *rows, int32_t n_x, int32_t n_u, double *out,
                                    int64_t loss_slot, int64_t *counter, double beta1, double beta2, double *bc_next,
                                    void *stream) {
    return plan_iface_pack_any(0, plan, tile_begin, tile_end, x_free, u_free, rows, n_x, n_u, out, loss_slot, counter, beta1, beta2,
                               bc_next, stream);
}

// float rows (an fp32 model): widened into the double2 payload
extern "C" int hfem_plan_iface_pack_f32(hfem_plan *plan, int32_t tile_begin, int32_t tile_end, const float *x_free,
                                        const float *u_free, const int32_t *rows, int32_t n_x, int32_t n_u, double *out,
                                        int64_t loss_slot, int64_t *counter, double beta1, double beta2, double *bc_next,
                                        void *stream) {
    return plan_iface_pack_any(1, plan, tile_begin, tile_end, x_free, u_free, rows, n_x, n_u, out, loss_slot, counter, beta1, beta2,
                               bc_next, stream);
}

// The in-launch get: launches of this plan with HFEM_FLAG_PEER_GET start with kPeerGetBlocks service workgroups that wait for
// the peers' flags and copy the interface rows in, while tiles [wait_begin, wait_end) -- the rank's boundary tiles -- wait
// for them and every other tile runs (csrc/peer.hip, hfem_peer_attach_get).  peer == NULL detaches.
extern "C" int hfem_plan_set_peer_get(hfem_plan *plan, hfem_peer *peer, int32_t wait_begin, int32_t wait_end) {
    HFEM_ARG_CHECK(plan, "null pointer");
    PlanLock lock(plan);
    if (!peer) { plan->peer_get = nullptr; return 0; }
    HFEM_ARG_CHECK(peer->connected && peer->get_dev, "hfem_peer_connect / hfem_peer_attach_get first");
    HFEM_ARG_CHECK(peer->device == plan->device, "plan and peer windows live on different devices");
    const int32_t nt = (int32_t)plan->host.tiles.size();
    HFEM_ARG_CHECK(wait_begin >= 0 && wait_begin <= wait_end && wait_end <= nt, "bad tile range");
    HFEM_ARG_CHECK(plan->host.paired ? plan->host.n_chained == 0
                                     : (plan->tune.tiled_fast && plan->tune.tiled_block == 512 &&
                                        (plan->tune.store_policy == 16 || plan->tune.store_policy == 2)),
                   "the in-launch get needs the paired-slot kernel's plain slot loop (no chained records) or the 512-thread "
                   "one-element-per-slot kernel with write-through / nt stores");
    plan->peer_get = peer->get_dev; plan->peer_wait_begin = wait_begin; plan->peer_wait_end = wait_end;
    return 0;
}

// The in-launch put: fused energy + Adam launches of this plan with HFEM_FLAG_PEER_PUT | HFEM_FLAG_PEER_GET also PUBLISH -- the
// boundary tiles store the new rows of their interface nodes into every rank's window at write-out and the last of them
// completes the put (csrc/peer.hip, hfem_peer_attach_put): an owner-sharded training step is ONE launch.  bc_a / bc_b: the
// two bias-correction buffers the steps alternate between (a launch reads one as bc_dev, its put writes the next step's
// into the other).  peer == NULL detaches.
extern "C" int hfem_plan_set_peer_put(hfem_plan *plan, hfem_peer *peer, double *bc_a, double *bc_b) {
    HFEM_ARG_CHECK(plan, "null pointer");
    PlanLock lock(plan);
    if (!peer) { plan->peer_put = nullptr; plan->put_bc[0] = plan->put_bc[1] = nullptr; plan->put_pos[0] = plan->put_pos[1] = nullptr; return 0; }
    HFEM_ARG_CHECK(peer->connected && peer->put_dev, "hfem_peer_connect / hfem_peer_attach_put first");
    HFEM_ARG_CHECK(peer->device == plan->device, "plan and peer windows live on different devices");
    HFEM_ARG_CHECK(plan->host.paired && plan->host.n_chained == 0, "the in-launch put needs a paired-slot plan (plain slot loop)");
    HFEM_ARG_CHECK(bc_a && bc_b && bc_a != bc_b, "need two distinct bias-correction buffers");
    plan->peer_put = peer->put_dev; plan->put_bc[0] = bc_a; plan->put_bc[1] = bc_b;
    plan->put_pos[0] = peer->put_pos[0]; plan->put_pos[1] = peer->put_pos[1];
    return 0;
}

// hfem_plan_iface_pack whose payload goes straight into every rank's receive window (csrc/peer.hip): pack + energy sum +
// step count + the stores over xGMI + the arrival flags, one launch, no collective
static int plan_iface_put_any(int dtype, hfem_plan *plan, hfem_peer *peer, int32_t tile_begin, int32_t tile_end,
                              const void *x_free, const void *u_free, const int32_t *rows, int32_t n_x, int32_t n_u,
                              int64_t loss_slot, int64_t *counter, double beta1, double beta2, double *bc_next, void *stream) {
    HFEM_ARG_CHECK(plan && peer, "null pointer");
    HFEM_ARG_CHECK(plan->device >= 0, "host-only plan (created with device < 0) cannot launch");
    HFEM_ARG_CHECK(peer->connected, "hfem_peer_connect has not been called");
    HFEM_ARG_CHECK(peer->device == plan->device, "plan and peer windows live on different devices");
    HFEM_ARG_CHECK(n_x >= 0 && n_u >= 0 && loss_slot >= n_x + n_u && loss_slot < peer->stride, "bad sizes");
    HFEM_ARG_CHECK((n_x + n_u == 0 || rows) && (n_x == 0 || x_free) && (n_u == 0 || u_free), "null pointer");
    const int32_t nt = (int32_t)plan->host.tiles.size();
    if (tile_end < 0) tile_end = nt;
    HFEM_ARG_CHECK(tile_begin >= 0 && tile_begin <= tile_end && tile_end <= nt, "bad tile range");
    if (int rc = use_device(plan->device)) return rc;
    PlanLock lock(plan);
    return launch_iface_put(peer, dtype, x_free, u_free, rows, n_x, n_u, loss_slot,
                            plan->d_partials + (size_t)plan->bank * nt + tile_begin, tile_end - tile_begin, counter, beta1,
                            beta2, bc_next, (hipStream_t)stream);
}

extern "C" int hfem_plan_iface_put(hfem_plan *plan, hfem_peer *peer, int32_t tile_begin, int32_t tile_end,
                                   const double *x_free, const double *u_free, const int32_t *rows, int32_t n_x, int32_t n_u,
                                   int64_t loss_slot, int64_t *counter, double beta1, double beta2, double *bc_next,
                                   void *stream) {
    return plan_iface_put_any(0, plan, peer, tile_begin, tile_end, x_free, u_free, rows, n_x, n_u, loss_slot, counter, beta1, beta2,
                              bc_next, stream);
}

extern "C" int hfem_plan_iface_put_f32(hfem_plan *plan, hfem_peer *peer, int32_t tile_begin, int32_t tile_end,
                                       const float *x_free, const float *u_free, const int32_t *rows, int32_t n_x, int32_t n_u,
                                       int64_t loss_slot, int64_t *counter, double beta1, double beta2, double *bc_next,
                                       void *stream) {
    return plan_iface_put_any(1, plan, peer, tile_begin, tile_end, x_free, u_free, rows, n_x, n_u, loss_slot, counter, beta1, beta2,
                              bc_next, stream);
}

// Options.  Product: tiled_block (256 / 512 / 1024 threads per tile), store_policy (0 plain, 16 sc1 write-through),
// tiled_fast, fast_const_caps, quad4_const_caps, plan_elem_order (0..4), plan_node_cap, plan_chunk_cap, plan_curve
// (0 Morton, 1 Hilbert): DEFAULTS that the next hfem_plan_create captures (a plan keeps what it was created with).
// Everything else (ablations, stamps, staggers, pipelined / streamed kernels) exists in the lab build only.
extern "C" int hfem_set_option(const char *name, int value) {
    HFEM_ARG_CHECK(name, "null option name");
    const std::string n(name);
    if (n == "tiled_block") {
        HFEM_ARG_CHECK(value == 256 || value == 512 || value == 1024, "tiled_block must be 256, 512 or 1024");
        g_def.tiled_block = value;
    } else if (n == "store_policy") {
        HFEM_ARG_CHECK(value == -1 || value == 0 || value == 2 || value == 16 || value == 17 || value == 18,
                       "store_policy: -1 (by mesh size, default), 16 (sc1 write-through), 2 (nt), 0 (plain), 17 (sc0 sc1), 18 (sc1 nt)");
        g_def.store_policy = value;
    } else if (n == "tiled_fast") {
        g_def.tiled_fast = value ? 1 : 0;
    } else if (n == "fast_const_caps") {
        g_def.fast_const_caps = value ? 1 : 0;
    } else if (n == "quad4_const_caps") {
        g_quad4_const_caps = value ? 1 : 0;
    } else if (n == "plan_elem_order") {
        HFEM_ARG_CHECK(value >= -1 && value <= 6, "plan_elem_order must be -1 (auto) or 0..6");
        g_def.plan_elem_order = value;
    } else if (n == "plan_node_cap") {
        HFEM_ARG_CHECK(value >= -1 && value <= 1024, "plan_node_cap must be -1 (auto), 0 (off) or 1..1024");
        g_def.plan_node_cap = value;
    } else if (n == "plan_chunk_cap") {
        HFEM_ARG_CHECK(value >= 0 && value <= 4096, "plan_chunk_cap: 0 (no limit) .. 4096 slots");
        g_def.plan_chunk_cap = value;
    } else if (n == "plan_shards") {
        HFEM_ARG_CHECK(value >= 1 && value <= 4096, "plan_shards: 1 .. 4096 ranks");
        g_def.plan_shards = value;
    } else if (n == "plan_pair_block") {
        HFEM_ARG_CHECK(value == -1 || value == 256 || value == 512, "plan_pair_block: -1 (auto), 256 or 512 threads per tile");
        g_def.plan_pair_block = value;
    } else if (n == "plan_curve") {
        HFEM_ARG_CHECK(value == 0 || value == 1, "plan_curve: 0 Morton, 1 Hilbert");
        set_plan_curve(value);
    } else if (n == "plan_read_pack") {
        g_def.plan_read_pack = value;
        set_plan_read_pack(value);
    } else if (n == "plan_snap") {
        HFEM_ARG_CHECK(value >= 0 && value <= 50, "plan_snap: 0..50 (percent of a tile)");
        g_def.plan_snap = value;
        set_plan_snap(value);
#ifdef HFEM_LAB
#define HFEM_LAB_SECTION 4
#include "tri3_energy_lab.inc"
#undef HFEM_LAB_SECTION
#endif
    } else {
        set_error("hfem_set_option: unknown option '" + n + "' (lab knobs need libhidenn_hip_lab.so)");
        return -1;
    }
    return 0;
}

extern "C" int hfem_get_option(const char *name) {
    if (!name) return -1;
    const std::string n(name);
    if (n == "tiled_block") return g_def.tiled_block.load();
    if (n == "store_policy") return g_def.store_policy.load();
    if (n == "tiled_fast") return g_def.tiled_fast.load();
    if (n == "fast_const_caps") return g_def.fast_const_caps.load();
    if (n == "pair_tiles_per_wg") return g_def.pair_tiles_per_wg.load();
    if (n == "pair_pipe_wps") return g_def.pair_pipe_wps.load();
    if (n == "quad4_const_caps") return g_quad4_const_caps;
    if (n == "plan_elem_order") return g_def.plan_elem_order.load();
    if (n == "plan_node_cap") return g_def.plan_node_cap.load();
    if (n == "plan_chunk_cap") return g_def.plan_chunk_cap.load();
    if (n == "plan_shards") return g_def.plan_shards.load();
    if (n == "plan_pair_block") return g_def.plan_pair_block.load();
    if (n == "plan_snap") return g_def.plan_snap.load();
    if (n == "plan_read_pack") return g_def.plan_read_pack.load();
    if (n == "lab_build") {
#ifdef HFEM_LAB
        return 1;
#else
        return 0;
#endif
    }
#ifdef HFEM_LAB
#define HFEM_LAB_SECTION 5
#include "tri3_energy_lab.inc"
#undef HFEM_LAB_SECTION
#endif
    return -1;
}

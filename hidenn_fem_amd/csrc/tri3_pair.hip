// Paired-slot TRI3 + EDGE2 energy kernel, gfx950 (MI355X): the tiled owner-computes pass of tri3_energy.hip with FEWER
// LDS ATOMICS per element.
//
// Replaces EnergyLoss2D.__call__ + loss.backward() of the reference (/root/reference/src/loss.py:55-116 over
// /root/reference/src/models.py:292-376) exactly as tri3_energy_fast_kernel does -- same closed forms (hfem_device.h),
// same owner-computes tiling, same outputs -- on a PAIRED plan (plan_elem_order 5, plan.cpp): a slot holds element
// A = (n, b, c) and, when the planner found one, the next element of the fan around n, B = (n, c, d) (A's corner 0 is B's
// corner 0, A's corner 2 is B's corner 1: e.g. the two triangles of a split quad; each element keeps ITS OWN local node
// order -- the reference energy depends on it).  The thread evaluates A, then B, and adds the contributions to
// the two shared nodes in registers: a full pair costs 16 ds_add_f64 and 8 ds_read_b128 instead of 24 and 12.  The round-2
// decomposition put the LDS atomics at 2.9 us of an 11 us launch, the largest single item.
// Elements without a partner are slots with hasB = 0 (12 atomics, as before).
// HBM-bound, no MFMA (2x2 / 2x3 contractions).  Algorithmic bytes per launch: 12 Ne + 64 Nn + 8.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "hfem_device.h"
#include "hfem_plan_dev.h"

namespace hfem {

// BLOCK threads per tile; NPT >= ceil(max nodes / BLOCK), EPT >= ceil(max slots / BLOCK).  WPS = waves per SIMD the
// register budget is sized for.  Measured best on T1M (round 2, profiles/r02): 256 threads, three slots per thread,
// 86 VGPRs, four workgroups per CU -- 9.5 us against 11.2 us for the one-element-per-slot kernel at 512 threads.
// CAPO > 0: compile-time stride of the four accumulator arrays.  LDS layout as tri3_energy_fast_kernel.
// SP: cache policy (aux bits) of the gradient stores: 16 = sc1 write-through (default), 0 plain, 2 nt, 17 sc0|sc1, 18 sc1|nt.
// HASB: body-force table; PHYS: opt-in physical gradient convention (hfem_device.h); V2: row storage type (double2, or
// float2 for fp32 models: widened on load, rounded once on store, fp64 arithmetic); ADAM: the write-out applies
// torch.optim.Adam's update instead of storing the gradient (AdamFuse, hfem_tri3_energy_adam_step).
template <int BLOCK, int NPT, int EPT, int WPS, int CAPO, bool HASB = false, bool PHYS = false, typename V2 = double2,
          bool ADAM = false, bool CHAIN = false, int CAPN = 0, int SP = 16, bool PG = false>
__global__ __launch_bounds__(BLOCK, WPS) void tri3_energy_pair_kernel(
    PlanDev pd, int tile_begin, typename RowArg<V2, PG>::type x_free, const V2 *__restrict__ x_fixed,
    typename RowArg<V2, PG>::type u_free, const V2 *__restrict__ u_fixed, Tri3Consts k,
    const double4 *__restrict__ T_edge, double4 Tconst, double *__restrict__ partials,
    V2 *__restrict__ gx_free, V2 *__restrict__ gu_free, int cap_nodes, int cap_owned_rt, int skip_edges,
    LagSum lag, AdamFuse af, int col_stride) {
    const int cap_owned = CAPO > 0 ? CAPO : cap_owned_rt;
    const int cap_n = CAPN > 0 ? CAPN : cap_nodes;       // CAPN > 0: the uv array's offset folds into the ds_read immediates
    extern __shared__ double2 lds[];
    double2 *nd_xy = lds;
    double2 *nd_uv = lds + cap_n;
    double *acc0 = reinterpret_cast<double *>(lds + 2 * cap_n);
    double *acc1 = acc0 + cap_owned, *acc2 = acc1 + cap_owned, *acc3 = acc2 + cap_owned;
    double *red = acc3 + cap_owned;

    const int tid = threadIdx.x;
    // PG instances (HFEM_FLAG_PEER_GET launches only -- every other launch runs the PG = false code, which knows nothing of
    // this): the first lag.pg_blocks workgroups are the peer-window get (peer.hip)
    int bid = (int)blockIdx.x;
    if constexpr (PG) {
        if (bid < lag.pg_blocks) {
            peer_get_block<V2>(*lag.pg, bid, lag.pg_blocks, x_free, u_free);      // RowArg<V2, true>: writable, not restrict
            return;
        }
        bid -= lag.pg_blocks;                           // a multiple of 8: the block -> XCD mapping of the tiles is unchanged
    }
    const int n_launch = (int)gridDim.x - (lag.prev ? 1 : 0) - (PG ? lag.pg_blocks : 0);
    if (lag.prev && bid == n_launch) {                  // HFEM_FLAG_SUM_PREVIOUS: reduce the previous launch's tile energies
        double v = 0.0;
        if (tid < 256)
            for (int i = tid; i < lag.prev_n; i += 256) v += lag.prev[i];
        const double tot = block_sum(v, red);
        if (tid == 0) lag.out[0] = tot;
        return;
    }
    const int slot = xcd_tile(bid, n_launch);
    mem_phase_begin();                                  // prologue at raised wave priority (hfem_plan_dev.h)
    // span stamps (hfem_plan_set_span_stamps, off by default): when this workgroup started -- scalar registers only
    unsigned long long t_start = 0;
    if (pd.span) t_start = __builtin_amdgcn_s_memrealtime();
    // ---- row maps first, from the tile index alone (uniform node stride, plan.cpp): these loads and the descriptor's are in
    //      flight together.  Unguarded: lanes past n_node read padding / the next tile's records -- valid rows, never stored.
    int2 s[NPT];
    const int2 *src = pd.node_src + (size_t)(tile_begin + slot) * pd.node_stride;
#pragma unroll
    for (int j = 0; j < NPT; ++j) s[j] = src[min(tid + j * BLOCK, pd.node_stride - 1)];      // lanes past the stride repeat its last record
    const TileDesc d = pd.tiles[tile_begin + slot];     // scalar loads, in flight with the row maps
    // ---- slot records, from the tile index alone as well (uniform slot stride; column stride `col_stride` is the plan's):
    //      thread t walks column t, row j at j * col_stride + t.  Unguarded; what lies past n_elem is masked below.
    uint32_t w0[EPT], w1[EPT];
    const size_t rec0 = (size_t)(tile_begin + slot) * pd.elem_stride;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const size_t i = rec0 + min(tid + j * col_stride, pd.elem_stride - 1);
        w0[j] = pd.elem_pack[i];
        w1[j] = pd.elem_pack_hi[i];
    }
    // HFEM_FLAG_PEER_PUT: a boundary tile will publish the new rows of its interface nodes at write-out -- where each row sits
    // in the payload is looked up NOW, behind the row maps and under the wait below, not as a late dependent load
    int put_px[NPT], put_pu[NPT];
    bool putting = false;
    if constexpr (PG && ADAM) {
        putting = lag.put && tile_begin + slot >= lag.wait_begin && tile_begin + slot < lag.wait_end;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            put_px[j] = (putting && s[j].x >= 0) ? lag.put_pos_x[s[j].x] : -1;
            put_pu[j] = (putting && s[j].y >= 0) ? lag.put_pos_u[s[j].y] : -1;
        }
    }
    if constexpr (PG) {
        if (tile_begin + slot >= lag.wait_begin && tile_begin + slot < lag.wait_end) {
            mem_phase_end();                            // a spinning wave must not outrank the service workgroups it waits for
            peer_wait_unpacked(*lag.pg);                // a boundary tile: the rows it reads from other ranks are being copied in
            mem_phase_begin();
        }
    }                                                   // (its row maps and slot records are already on their way)
    // ---- gather through the row maps: issued before anything that needs the descriptor (program order = vmcnt order)
    V2 vx[NPT], vu[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const V2 *px = s[j].x >= 0 ? x_free + s[j].x : x_fixed + ~s[j].x;
        const V2 *pu = s[j].y >= 0 ? u_free + s[j].y : u_fixed + ~s[j].y;
        vx[j] = *px;
        vu[j] = *pu;
    }
    __builtin_amdgcn_sched_barrier(0);                  // all gather loads are issued before the first is waited for
    const int n_owned = d.n_owned;
    // boundary tiles: their Neumann-edge records now, not after the slot loop (a late dependent load on the critical path)
    const int n_edge = skip_edges ? 0 : d.n_edge;
    uint32_t edge_rec = 0u;
    int edge_id = 0;
    if (tid < n_edge) {
        edge_rec = pd.edge_pack[d.edge_off + tid];
        if (T_edge) edge_id = pd.edge_gid[d.edge_off + tid];
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (!(tid < col_stride && tid + j * col_stride < d.n_elem)) { w0[j] = kSkipBit; w1[j] = 0u; }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int l = tid + j * BLOCK;
        if (l < d.n_node) {
            nd_xy[l] = make_double2((double)vx[j].x, (double)vx[j].y);
            nd_uv[l] = make_double2((double)vu[j].x, (double)vu[j].y);
        }
        if (l < n_owned) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
    }
    __syncthreads();
    mem_phase_end();

    auto add_row = [&](int l, const double2 gx, const double2 gu) {
        unsafeAtomicAdd(&acc0[l], gx.x); unsafeAtomicAdd(&acc1[l], gx.y);
        unsafeAtomicAdd(&acc2[l], gu.x); unsafeAtomicAdd(&acc3[l], gu.y);
    };
    double e_loc = 0.0;
    if (!CHAIN) {
    // ---- slots: registers + LDS only
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const uint32_t p = w0[j], q = w1[j];
        if (!(p & kSkipBit)) {
            const int ln = (int)(p & kLocalMask), lb = (int)((p >> kLocalBits) & kLocalMask),
                      lc = (int)((p >> (2 * kLocalBits)) & kLocalMask);
            const double2 Xn = nd_xy[ln], Un = nd_uv[ln], Xc = nd_xy[lc], Uc = nd_uv[lc];
            double2 sxn, sun, sxc, suc;                 // running rows of the shared nodes n and c
            {
                double2 gx[3], gu[3];
                const double e = tri3_element<true, HASB, PHYS>(Xn, nd_xy[lb], Xc, Un, nd_uv[lb], Uc, k, gx, gu);
                if (p & kHomeBit) e_loc += e;
                if (lb < n_owned) add_row(lb, gx[1], gu[1]);
                sxn = gx[0]; sun = gu[0]; sxc = gx[2]; suc = gu[2];
            }
            if (q & (1u << 10)) {                       // B = (n, c, d)
                const int ld = (int)(q & kLocalMask);
                double2 gx[3], gu[3];
                const double e = tri3_element<true, HASB, PHYS>(Xn, Xc, nd_xy[ld], Un, Uc, nd_uv[ld], k, gx, gu);
                if (q & (1u << 11)) e_loc += e;
                if (ld < n_owned) add_row(ld, gx[2], gu[2]);
                sxn.x += gx[0].x; sxn.y += gx[0].y; sun.x += gu[0].x; sun.y += gu[0].y;
                sxc.x += gx[1].x; sxc.y += gx[1].y; suc.x += gu[1].x; suc.y += gu[1].y;
            }
            if (ln < n_owned) add_row(ln, sxn, sun);
            if (lc < n_owned) add_row(lc, sxc, suc);
        }
    }
    } else {
    // ---- slots: registers + LDS only.  A CHAINED slot (strip order, plan.cpp) hands its rows of b and c -- and the node
    //      values -- to the next slot of the column, whose n and d they are: 8 instead of 16 atomics, 4 instead of 8 reads.
    double2 kxb, kub, kxc, kuc;                         // carried rows
    double2 pXb, pUb, pXc, pUc;                         // carried node values
    bool chained = false;
    constexpr bool kCarryVals = false;                  // also carry the node VALUES (saves 4 LDS reads per chained slot; +16 VGPRs: spills)
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const uint32_t p = w0[j], q = w1[j];
        if (!(p & kSkipBit)) {
            const int ln = (int)(p & kLocalMask), lb = (int)((p >> kLocalBits) & kLocalMask),
                      lc = (int)((p >> (2 * kLocalBits)) & kLocalMask);
            double2 Xn, Un;
            if (kCarryVals && j > 0 && chained) { Xn = pXb; Un = pUb; } else { Xn = nd_xy[ln]; Un = nd_uv[ln]; }
            const double2 Xc = nd_xy[lc], Uc = nd_uv[lc];
            double2 sxn = make_double2(0.0, 0.0), sun = sxn, sxc = sxn, suc = sxn;   // running rows of the shared nodes n and c
            if (q & (1u << 10)) {                       // B = (n, c, d) first: the carried rows and node values die here
                const int ld = (int)(q & kLocalMask);
                double2 Xd, Ud;
                if (kCarryVals && j > 0 && chained) { Xd = pXc; Ud = pUc; } else { Xd = nd_xy[ld]; Ud = nd_uv[ld]; }
                double2 gx[3], gu[3];
                const double e = tri3_element<true, HASB, PHYS>(Xn, Xc, Xd, Un, Uc, Ud, k, gx, gu);
                if (q & (1u << 11)) e_loc += e;
                if (j > 0 && chained) {
                    gx[0].x += kxb.x; gx[0].y += kxb.y; gu[0].x += kub.x; gu[0].y += kub.y;
                    gx[2].x += kxc.x; gx[2].y += kxc.y; gu[2].x += kuc.x; gu[2].y += kuc.y;
                }
                if (ld < n_owned) add_row(ld, gx[2], gu[2]);
                sxn = gx[0]; sun = gu[0]; sxc = gx[1]; suc = gu[1];
            }
            {
                const double2 Xb = nd_xy[lb], Ub = nd_uv[lb];
                double2 gx[3], gu[3];
                const double e = tri3_element<true, HASB, PHYS>(Xn, Xb, Xc, Un, Ub, Uc, k, gx, gu);
                if (p & kHomeBit) e_loc += e;
                sxn.x += gx[0].x; sxn.y += gx[0].y; sun.x += gu[0].x; sun.y += gu[0].y;
                sxc.x += gx[2].x; sxc.y += gx[2].y; suc.x += gu[2].x; suc.y += gu[2].y;
                if (ln < n_owned) add_row(ln, sxn, sun);
                chained = (q & (1u << 12)) != 0;
                if (j + 1 < EPT && chained) {
                    kxb = gx[1]; kub = gu[1]; kxc = sxc; kuc = suc;
                    if (kCarryVals) { pXb = Xb; pUb = Ub; pXc = Xc; pUc = Uc; }
                } else {
                    if (lb < n_owned) add_row(lb, gx[1], gu[1]);
                    if (lc < n_owned) add_row(lc, sxc, suc);
                }
            }
        } else {
            chained = false;
        }
    }
    }   // CHAIN
    for (int i = tid; i < n_edge; i += BLOCK) {          // boundary tiles only
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
        if ((tid & 63) == 0) red[tid >> 6] = w;          // one slot per wave: summed in wave order below
    }
    // every load has long returned; saying so keeps the compiler from guarding each write-out store with a vmcnt(0) of its
    // own (gfx9 counts stores in vmcnt: the stores would wait for one another -- seen in the ISA of the carrying slot loop)
    mem_phase_begin();                                  // write-out
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    if (ADAM) {
        // Fused optimiser step (hfem_tri3_energy_adam_step): the tile that owns a row holds its complete gradient (acc*) and
        // its current value (nd_xy / nd_uv) in LDS, so it applies torch.optim.Adam's update here: m, v read-modify-written,
        // the NEW row to the OTHER parameter buffer (tiles still gathering must see the old one: ping-pong); the gradient
        // never goes to memory.  Arithmetic = optim.hip's adam_step_dev_kernel, operation for operation.
        const double bc1 = af.bc[0], sqrt_bc2 = af.bc[1];
        // HFEM_FLAG_PEER_PUT (PG instances): a boundary tile also PUBLISHES -- the new rows of its interface nodes go straight
        // into every rank's receive window (the separate put launch of a sharded step disappears)
        uint64_t put_seq = 0;
        size_t put_off = 0;
        if constexpr (PG) {
            if (putting) {
                put_seq = __hip_atomic_load((uint64_t *)lag.put->pv.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                put_off = kPeerData + ((size_t)(put_seq & 1) * lag.put->pv.world + lag.put->pv.rank) * (size_t)lag.put->stride * sizeof(double2);
            }
        }
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int l = tid + j * BLOCK;
            if (l < n_owned) {
                if (s[j].x >= 0) {
                    const V2 pn = adam_fused_row<V2>(af, 0, s[j].x, acc0[l], acc1[l], nd_xy[l], bc1, sqrt_bc2);
                    if constexpr (PG) {
                        if (putting && put_px[j] >= 0) peer_put_row<V2>(*lag.put, put_off, put_px[j], pn);
                    }
                }
                if (s[j].y >= 0) {
                    const V2 pn = adam_fused_row<V2>(af, 1, s[j].y, acc2[l], acc3[l], nd_uv[l], bc1, sqrt_bc2);
                    if constexpr (PG) {
                        if (putting && put_pu[j] >= 0) peer_put_row<V2>(*lag.put, put_off, put_pu[j], pn);
                    }
                }
            }
        }
        if constexpr (PG) {
            if (putting) {
                if (tid == 0) {                                 // the tile energy first: this block may return inside the put
                    double tile_e = 0.0;
#pragma unroll
                    for (int w = 0; w < BLOCK / 64; ++w) tile_e += red[w];
                    partials[slot] = tile_e;
                }
                __syncthreads();                                // red[] is reused by the put's block sum
                peer_put_finish(*lag.put, lag, put_seq, put_off, lag.wait_end - lag.wait_begin, red);
                return;
            }
        }
    } else {
    // ---- every owned gradient row is written exactly once (write-through: the line leaves the XCD's L2)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    constexpr bool kWide = sizeof(V2) == 16;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)gx_free, 0, 0x7FFFFFF0, 0x00020000);
    __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void *)gu_free, 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int l = tid + j * BLOCK;
        if (l < n_owned) {
            if (gx_free && s[j].x >= 0) {
                V2 v;
                v.x = acc0[l]; v.y = acc1[l];           // rounds once for float2
                if (kWide) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), rx, s[j].x * 16, 0, SP);
                else __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), rx, s[j].x * 8, 0, SP);
            }
            if (gu_free && s[j].y >= 0) {
                V2 v;
                v.x = acc2[l]; v.y = acc3[l];
                if (kWide) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), ru, s[j].y * 16, 0, SP);
                else __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), ru, s[j].y * 8, 0, SP);
            }
        }
    }
    }   // !ADAM
    if (tid == 0) {                                     // fixed order: the tile energy is bit-reproducible
        double tile_e = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) tile_e += red[w];
        partials[slot] = tile_e;
    }
    if (pd.span && tid == 0) {                          // ... and when its first wave's stores had left (100 MHz ticks)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        pd.span[2 * (size_t)(tile_begin + slot)] = t_start;
        pd.span[2 * (size_t)(tile_begin + slot) + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

constexpr int kPairCapN = 656, kPairCapO = 560;          // compile-time LDS strides of the default tile shape (557 owned nodes)
template <int BLK, int NPT, int EPT, int CAPO, bool HASB, bool PHYS, typename V2, bool ADAM, bool CHAIN, int SP = 16>
static void launch_pair_inst2(const PairLaunch &A, const LagSum &lag, const AdamFuse &af) {
    constexpr int CAPN = CAPO > 0 ? kPairCapN : 0;
    const size_t lds = CAPO > 0 ? (size_t)(CAPN * 32 + CAPO * 32 + 128) : A.lds;
    if constexpr (!HASB && !PHYS && !CHAIN) {
        if (lag.pg_blocks) {                               // HFEM_FLAG_PEER_GET: the instance with the in-launch get
            hipLaunchKernelGGL((tri3_energy_pair_kernel<BLK, NPT, EPT, 4, CAPO, HASB, PHYS, V2, ADAM, CHAIN, CAPN, SP, true>), dim3(A.grid), dim3(BLK), lds, A.s,
                               A.pd, A.tile_begin, (V2 *)const_cast<void *>(A.x_free), (const V2 *)A.x_fixed, (V2 *)const_cast<void *>(A.u_free),
                               (const V2 *)A.u_fixed, A.k, A.T_edge, A.tc, A.partials, (V2 *)A.gx, (V2 *)A.gu, A.max_nodes,
                               CAPO > 0 ? CAPO : A.max_owned, A.skip_edges, lag, af, A.col_stride);
            return;
        }
    }
    hipLaunchKernelGGL((tri3_energy_pair_kernel<BLK, NPT, EPT, 4, CAPO, HASB, PHYS, V2, ADAM, CHAIN, CAPN, SP>), dim3(A.grid), dim3(BLK), lds, A.s,
                       A.pd, A.tile_begin, (const V2 *)A.x_free, (const V2 *)A.x_fixed, (const V2 *)A.u_free,
                       (const V2 *)A.u_fixed, A.k, A.T_edge, A.tc, A.partials, (V2 *)A.gx, (V2 *)A.gu, A.max_nodes,
                       CAPO > 0 ? CAPO : A.max_owned, A.skip_edges, lag, af, A.col_stride);
}
template <int BLK, int NPT, int EPT, int CAPO, bool HASB, bool PHYS, typename V2, bool ADAM>
static void launch_pair_inst(const PairLaunch &A, const LagSum &lag, const AdamFuse &af) {
#ifdef HFEM_LAB
    if (A.chain) { launch_pair_inst2<BLK, NPT, EPT, CAPO, HASB, PHYS, V2, ADAM, true>(A, lag, af); return; }   // strip order: lab evidence only
#endif
    launch_pair_inst2<BLK, NPT, EPT, CAPO, HASB, PHYS, V2, ADAM, false>(A, lag, af);
}

// Launch on a paired plan: picks the instance that holds the plan's tile shape.  1 = launched, 0 = none does.
// mode: 0 fp64 reference convention (zero body force), 1 general fp64 (body force and / or physical convention),
//       2 fp32 rows, 3 fused Adam write-out on fp64 rows, 4 the same on fp32 rows (3, 4: hasb selects the body-force instance).
// Instance matrix (round 4: 92 instances in the product, 209 before; tests/test_build_resources.py holds the count and
// "no scratch"): the hot modes (0, 2, 3, 4 without a body force) get the exact shapes -- NPT 3 | 4 nodes per thread, 3 | 4 | 6
// slot rows (a five-row plan takes the six-row instance: one masked row), the compile-time-stride instance of the default
// tile, nt stores for the big-mesh policy, and the in-launch get (PG) of each; body force and the physical convention -- not
// the reference's defaults (src/loss.py:43-45, src/models.py:351) -- share ONE generic shape per block size (<256, 4, 6> /
// <512, 2, 2>).  Chained records (strip order, plan_elem_order 6) and the store-policy A/B instances exist in the lab build only.
int launch_tri3_pair(const hfem_plan *plan, PairLaunch A, int mode, bool hasb, bool phys, const LagSum &lag,
                     const AdamFuse &af) {
    const HostPlan &h = plan->host;
    if (!h.paired || !plan->d_elem_pack_hi) return 0;
    A.pd = plan_dev(plan);
    A.pd.span = A.span;
    A.max_nodes = h.max_nodes; A.max_owned = h.max_owned; A.lds = (size_t)plan->lds_bytes;
    A.col_stride = h.col_stride;
    if (A.chain < 0) A.chain = h.n_chained > 0 ? 1 : 0;   // chained records need the carrying slot loop
#ifndef HFEM_LAB
    if (A.chain) return 0;                                // product: no carrying slot loop (it measured slower)
#endif
    if (h.pair_block == 512) {
        // 512 threads per tile (shard-aware tile policy, hfem_plan_create: launches of 100 k - 600 k elements): NPT = 2
        // (<= 1024 nodes), one or two slot rows; the plain slot loop only
        if (h.max_nodes > 2 * 512 || h.max_rows > 2 || A.chain) return 0;
        if (mode == 0) {
            if (h.max_owned <= kPairCapO && h.max_nodes <= kPairCapN) launch_pair_inst2<512, 2, 2, 560, false, false, double2, false, false>(A, lag, af);
            else launch_pair_inst2<512, 2, 2, 0, false, false, double2, false, false>(A, lag, af);
            return 1;
        }
        if (mode == 1) {
            if (phys) launch_pair_inst2<512, 2, 2, 0, true, true, double2, false, false>(A, lag, af);
            else launch_pair_inst2<512, 2, 2, 0, true, false, double2, false, false>(A, lag, af);
            return 1;
        }
        if (mode == 2) { launch_pair_inst2<512, 2, 2, 0, false, false, float2, false, false>(A, lag, af); return 1; }
        if (mode == 3) {
            if (hasb) launch_pair_inst2<512, 2, 2, 0, true, false, double2, true, false>(A, lag, af);
            else launch_pair_inst2<512, 2, 2, 0, false, false, double2, true, false>(A, lag, af);
            return 1;
        }
        if (mode == 4) {
            if (hasb) launch_pair_inst2<512, 2, 2, 0, true, false, float2, true, false>(A, lag, af);
            else launch_pair_inst2<512, 2, 2, 0, false, false, float2, true, false>(A, lag, af);
            return 1;
        }
        return 0;
    }
    const bool cc = h.max_owned <= kPairCapO && h.max_nodes <= kPairCapN;   // (656 + 560) * 32 + 128 = 39040 B: four workgroups per CU
    const int npt = h.max_nodes <= 3 * 256 ? 3 : 4;
    const int ept = h.max_rows > 0 ? h.max_rows : 1;     // slots per thread (0: a plan of element-less tiles)
    if (h.max_nodes > 4 * 256 || ept > 6) return 0;
#define HFEM_PAIR_EPT(NPT, CO, HB, PH, V, AD)                                                    \
    switch (ept) {                                                                               \
        case 1: case 2: case 3: launch_pair_inst<256, NPT, 3, CO, HB, PH, V, AD>(A, lag, af); return 1; \
        case 4: launch_pair_inst<256, NPT, 4, CO, HB, PH, V, AD>(A, lag, af); return 1;           \
        default: launch_pair_inst<256, NPT, 6, CO, HB, PH, V, AD>(A, lag, af); return 1;          \
    }
    // nt (non-temporal) gradient stores -- the plan's store policy for meshes whose gradient arrays cannot stay in the
    // Infinity Cache (hfem_plan_create, "store_policy" -1): instances for the plain slot loop and up to four slot rows; other
    // shapes keep the write-through stores
    const bool nt = plan->tune.store_policy == 2 && !A.chain && ept <= 4;
#define HFEM_PAIR_NT(NPT, CO, V)                                                                              \
    {                                                                                                         \
        if (ept <= 3) launch_pair_inst2<256, NPT, 3, CO, false, false, V, false, false, 2>(A, lag, af);      \
        else launch_pair_inst2<256, NPT, 4, CO, false, false, V, false, false, 2>(A, lag, af);               \
        return 1;                                                                                             \
    }
    if (mode == 0 && nt) {
        if (cc && npt == 3) HFEM_PAIR_NT(3, 560, double2)
        if (npt == 3) HFEM_PAIR_NT(3, 0, double2)
        HFEM_PAIR_NT(4, 0, double2)
    }
    if (mode == 2 && nt) {
        if (npt == 3) HFEM_PAIR_NT(3, 0, float2)
        HFEM_PAIR_NT(4, 0, float2)
    }
#undef HFEM_PAIR_NT
    if (mode == 0) {
#ifdef HFEM_LAB
        const int sp = plan->tune.store_policy;
        if (cc && npt == 3 && ept <= 3 && sp != 16 && !A.chain && !lag.pg_blocks) {   // "store_policy" A/B instances of the default tile shape
            const size_t lds = (size_t)(kPairCapN * 32 + kPairCapO * 32 + 128);
#define HFEM_PAIR_SP(SPV)                                                                                                  \
    hipLaunchKernelGGL((tri3_energy_pair_kernel<256, 3, 3, 4, 560, false, false, double2, false, false, kPairCapN, SPV>),  \
                       dim3(A.grid), dim3(256), lds, A.s, A.pd, A.tile_begin, (const double2 *)A.x_free,                   \
                       (const double2 *)A.x_fixed, (const double2 *)A.u_free, (const double2 *)A.u_fixed, A.k, A.T_edge,   \
                       A.tc, A.partials, (double2 *)A.gx, (double2 *)A.gu, A.max_nodes, 560, A.skip_edges, lag, af,         \
                       A.col_stride)
            switch (sp) {
                case 0: HFEM_PAIR_SP(0); return 1;
                case 17: HFEM_PAIR_SP(17); return 1;
                case 18: HFEM_PAIR_SP(18); return 1;
                default: break;
            }
#undef HFEM_PAIR_SP
        }
#endif
        if (cc && npt == 3) HFEM_PAIR_EPT(3, 560, false, false, double2, false)
        if (npt == 3) HFEM_PAIR_EPT(3, 0, false, false, double2, false)
        HFEM_PAIR_EPT(4, 0, false, false, double2, false)
    } else if (mode == 1) {                      // body force and / or physical convention: the generic shape
        if (phys) launch_pair_inst<256, 4, 6, 0, true, true, double2, false>(A, lag, af);
        else launch_pair_inst<256, 4, 6, 0, true, false, double2, false>(A, lag, af);
        return 1;
    } else if (mode == 2) {
        if (npt == 3) HFEM_PAIR_EPT(3, 0, false, false, float2, false)
        HFEM_PAIR_EPT(4, 0, false, false, float2, false)
    } else if (mode == 3) {                      // fused Adam write-out: fp64 rows
        if (hasb) { launch_pair_inst<256, 4, 6, 0, true, false, double2, true>(A, lag, af); return 1; }
        if (npt == 3) HFEM_PAIR_EPT(3, 0, false, false, double2, true)
        HFEM_PAIR_EPT(4, 0, false, false, double2, true)
    } else if (mode == 4) {                      // fused Adam write-out: fp32 rows (parameters, moments, new rows all float)
        if (hasb) { launch_pair_inst<256, 4, 6, 0, true, false, float2, true>(A, lag, af); return 1; }
        if (npt == 3) HFEM_PAIR_EPT(3, 0, false, false, float2, true)
        HFEM_PAIR_EPT(4, 0, false, false, float2, true)
    }
#undef HFEM_PAIR_EPT
    return 0;
}

}  // namespace hfem

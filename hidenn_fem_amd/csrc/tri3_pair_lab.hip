// LAB BUILD ONLY (-DHFEM_LAB, libhidenn_hip_lab.so): the instrumented copy of the paired-slot kernel -- the ablation bits
// (HFEM_PAIR_LAB), the lane-chain cost model and the forced slot loops that scripts/ drives (hfem_set_option "pair_ablate",
// "pair_chain").  The product kernel (tri3_pair.hip) carries none of it; this copy is what the ablation ladders of DESIGN.md
// section 4.1 were measured with, fp64 reference-convention instances only.  Nothing in this file is compiled into
// libhidenn_hip.so.
#ifdef HFEM_LAB
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

#include "hfem_device.h"
#include "hfem_plan_dev.h"

namespace hfem {

#ifdef HFEM_LAB
#define HFEM_PAIR_LAB(bit) (lab_bits & (bit))          /* ablations: 1 no atomics, 2 no slot phase, 4 no write-out, 8 no gather loads, 16 return at once, 32 return after the row-map loads, 64 after the record loads, 128 after the first barrier, 256 no tile-energy store, 512 no edges, 1024 no record loads, 2048 no LDS fill, 4096 lane-chain cost model (8 atomics + DPP), 8192 8 atomics only */
#else
#define HFEM_PAIR_LAB(bit) false
#endif

// BLOCK threads per tile; NPT >= ceil(max nodes / BLOCK), EPT >= ceil(max slots / BLOCK).  WPS = waves per SIMD the
// register budget is sized for.  Measured best on T1M (round 2, profiles/r02): 256 threads, three slots per thread,
// 86 VGPRs, four workgroups per CU -- 9.5 us against 11.2 us for the one-element-per-slot kernel at 512 threads.
// CAPO > 0: compile-time stride of the four accumulator arrays.  LDS layout as tri3_energy_fast_kernel.
// HASB: body-force table; PHYS: opt-in physical gradient convention (hfem_device.h); V2: row storage type (double2, or
// float2 for fp32 models: widened on load, rounded once on store, fp64 arithmetic); ADAM: the write-out applies
// torch.optim.Adam's update instead of storing the gradient (AdamFuse, hfem_tri3_energy_adam_step).
template <int BLOCK, int NPT, int EPT, int WPS, int CAPO, bool HASB = false, bool PHYS = false, typename V2 = double2,
          bool ADAM = false, bool CHAIN = false, int CAPN = 0>
__global__ __launch_bounds__(BLOCK, WPS) void tri3_energy_pair_lab_kernel(
    PlanDev pd, int tile_begin, const V2 *__restrict__ x_free, const V2 *__restrict__ x_fixed,
    const V2 *__restrict__ u_free, const V2 *__restrict__ u_fixed, Tri3Consts k,
    const double4 *__restrict__ T_edge, double4 Tconst, double *__restrict__ partials,
    V2 *__restrict__ gx_free, V2 *__restrict__ gu_free, int cap_nodes, int cap_owned_rt, int skip_edges,
    LagSum lag, AdamFuse af, int col_stride, int lab_bits) {
    const int cap_owned = CAPO > 0 ? CAPO : cap_owned_rt;
    const int cap_n = CAPN > 0 ? CAPN : cap_nodes;       // CAPN > 0: the uv array's offset folds into the ds_read immediates
    extern __shared__ double2 lds[];
    double2 *nd_xy = lds;
    double2 *nd_uv = lds + cap_n;
    double *acc0 = reinterpret_cast<double *>(lds + 2 * cap_n);
    double *acc1 = acc0 + cap_owned, *acc2 = acc1 + cap_owned, *acc3 = acc2 + cap_owned;
    double *red = acc3 + cap_owned;

    const int tid = threadIdx.x;
    const int n_launch = (int)gridDim.x - (lag.prev ? 1 : 0);
    if (lag.prev && (int)blockIdx.x == n_launch) {      // HFEM_FLAG_SUM_PREVIOUS: reduce the previous launch's tile energies
        double v = 0.0;
        if (tid < 256)
            for (int i = tid; i < lag.prev_n; i += 256) v += lag.prev[i];
        const double tot = block_sum(v, red);
        if (tid == 0) lag.out[0] = tot;
        return;
    }
    if (HFEM_PAIR_LAB(16)) return;                      // lab: dispatch cost of this grid shape alone
    const int slot = xcd_tile(blockIdx.x, n_launch);
    // lab bits 16384 / 65536 / 131072: wave priority 3 / 1 / 2 through the prologue (index loads, gather, LDS fill) and, unless bit
    // 32768 is set, again through the write-out; the slot loop runs at priority 0.  262144: the plain lab kernel (no change).
    const bool prio_on = HFEM_PAIR_LAB(16384 | 65536 | 131072);
    if (HFEM_PAIR_LAB(16384)) __builtin_amdgcn_s_setprio(3);
    else if (HFEM_PAIR_LAB(65536)) __builtin_amdgcn_s_setprio(1);
    else if (HFEM_PAIR_LAB(131072)) __builtin_amdgcn_s_setprio(2);
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
    if (HFEM_PAIR_LAB(32)) {                            // lab: + one round of index loads
        int acc = d.n_node;
        for (int j = 0; j < NPT; ++j) acc += s[j].x;
        if (acc == 0x7fffffff) partials[slot] = 1.0;
        return;
    }
    // ---- slot records, from the tile index alone as well (uniform slot stride; column stride `col_stride` is the plan's):
    //      thread t walks column t, row j at j * col_stride + t.  Unguarded; what lies past n_elem is masked below.
    uint32_t w0[EPT], w1[EPT];
    const size_t rec0 = (size_t)(tile_begin + slot) * pd.elem_stride;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        if (HFEM_PAIR_LAB(1024)) { w0[j] = kSkipBit | tid; w1[j] = 0u; continue; }
        const size_t i = rec0 + min(tid + j * col_stride, pd.elem_stride - 1);
        w0[j] = pd.elem_pack[i];
        w1[j] = pd.elem_pack_hi[i];
    }
    // ---- gather through the row maps: issued before anything that needs the descriptor (program order = vmcnt order)
    V2 vx[NPT], vu[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const V2 *px = s[j].x >= 0 ? x_free + s[j].x : x_fixed + ~s[j].x;
        const V2 *pu = s[j].y >= 0 ? u_free + s[j].y : u_fixed + ~s[j].y;
        if (HFEM_PAIR_LAB(8)) { vx[j].x = vx[j].y = (decltype(vx[j].x))(0.001 * tid); vu[j] = vx[j]; continue; }
        vx[j] = *px;
        vu[j] = *pu;
    }
    __builtin_amdgcn_sched_barrier(0);                  // all gather loads are issued before the first is waited for
    const int n_owned = d.n_owned;
    // boundary tiles: their Neumann-edge records now, not after the slot loop (a late dependent load on the critical path)
    const int n_edge = (skip_edges || HFEM_PAIR_LAB(512)) ? 0 : d.n_edge;
    uint32_t edge_rec = 0u;
    int edge_id = 0;
    if (tid < n_edge) {
        edge_rec = pd.edge_pack[d.edge_off + tid];
        if (T_edge) edge_id = pd.edge_gid[d.edge_off + tid];
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (!(tid < col_stride && tid + j * col_stride < d.n_elem)) { w0[j] = kSkipBit; w1[j] = 0u; }
    if (HFEM_PAIR_LAB(64)) {                            // lab: + the slot-record loads (second dependent round)
        uint32_t acc = 0;
        for (int j = 0; j < EPT; ++j) acc += w0[j] ^ w1[j];
        for (int j = 0; j < NPT; ++j) acc += (uint32_t)s[j].x;
        if (acc == 0x7fffffffu) partials[slot] = 1.0;
        return;
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int l = tid + j * BLOCK;
        if (HFEM_PAIR_LAB(2048)) continue;
        if (l < d.n_node) {
            nd_xy[l] = make_double2((double)vx[j].x, (double)vx[j].y);
            nd_uv[l] = make_double2((double)vu[j].x, (double)vu[j].y);
        }
        if (l < n_owned) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
    }
    __syncthreads();
    if (prio_on) __builtin_amdgcn_s_setprio(0);
    if (HFEM_PAIR_LAB(128)) {                           // lab: + LDS fill and the first barrier
        uint32_t acc = 0;
        for (int j = 0; j < EPT; ++j) acc += w0[j] ^ w1[j];
        if (acc == 0x7fffffffu || nd_xy[tid].x == 1.2345) partials[slot] = 1.0;
        return;
    }

    auto add_row = [&](int l, const double2 gx, const double2 gu) {
        if (HFEM_PAIR_LAB(1)) return;
        unsafeAtomicAdd(&acc0[l], gx.x); unsafeAtomicAdd(&acc1[l], gx.y);
        unsafeAtomicAdd(&acc2[l], gu.x); unsafeAtomicAdd(&acc3[l], gu.y);
    };
    double e_loc = 0.0;
    if (HFEM_PAIR_LAB(2)) {
    } else if (!CHAIN) {
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
                if (lb < n_owned && !HFEM_PAIR_LAB(4096 | 8192)) add_row(lb, gx[1], gu[1]);
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
#ifdef HFEM_LAB
            if (HFEM_PAIR_LAB(4096 | 8192)) {           // lab: what a LANE chain would cost (timing only, results are not valid):
                if (HFEM_PAIR_LAB(4096)) {              // rows of b, c handed to the next lane by DPP and added to its n, d rows
                    auto shr = [](double v) {
                        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xF, 0xF, false);
                        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xF, 0xF, false);
                        return __hiloint2double(hi, lo);
                    };
                    sxn.x += shr(sxc.x); sxn.y += shr(sxc.y); sun.x += shr(suc.x); sun.y += shr(suc.y);
                    sxn.x += shr(sxc.y); sxn.y += shr(sxc.x); sun.x += shr(suc.y); sun.y += shr(suc.x);
                }
                if (ln < n_owned) add_row(ln, sxn, sun);   // n flushed; b was flushed above (stands in for d), c is not
                continue;
            }
#endif
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
    if (prio_on && !HFEM_PAIR_LAB(32768)) __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    if (ADAM) {
        // Fused optimiser step (hfem_tri3_energy_adam_step): the tile that owns a row holds its complete gradient (acc*) and
        // its current value (nd_xy / nd_uv) in LDS, so it applies torch.optim.Adam's update here: m, v read-modify-written,
        // the NEW row to the OTHER parameter buffer (tiles still gathering must see the old one: ping-pong); the gradient
        // never goes to memory.  Arithmetic = optim.hip's adam_step_dev_kernel, operation for operation.
        const double bc1 = af.bc[0], sqrt_bc2 = af.bc[1];
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int l = tid + j * BLOCK;
            if (l < n_owned) {
                if (s[j].x >= 0) adam_fused_row<V2>(af, 0, s[j].x, acc0[l], acc1[l], nd_xy[l], bc1, sqrt_bc2);
                if (s[j].y >= 0) adam_fused_row<V2>(af, 1, s[j].y, acc2[l], acc3[l], nd_uv[l], bc1, sqrt_bc2);
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
        if (l < n_owned && !HFEM_PAIR_LAB(4)) {
            if (gx_free && s[j].x >= 0) {
                V2 v;
                v.x = acc0[l]; v.y = acc1[l];           // rounds once for float2
                if (kWide) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), rx, s[j].x * 16, 0, 16);
                else __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), rx, s[j].x * 8, 0, 16);
            }
            if (gu_free && s[j].y >= 0) {
                V2 v;
                v.x = acc2[l]; v.y = acc3[l];
                if (kWide) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&v), ru, s[j].y * 16, 0, 16);
                else __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), ru, s[j].y * 8, 0, 16);
            }
        }
    }
    }   // !ADAM
    if (tid == 0) {                                     // fixed order: the tile energy is bit-reproducible
        double tile_e = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) tile_e += red[w];
        if (!HFEM_PAIR_LAB(256) || tile_e == 1.2345) partials[slot] = tile_e;
    }
    if (pd.span && tid == 0) {                          // ... and when its first wave's stores had left (100 MHz ticks)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        pd.span[2 * (size_t)(tile_begin + slot)] = t_start;
        pd.span[2 * (size_t)(tile_begin + slot) + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

constexpr int kPairLabCapN = 656, kPairLabCapO = 560;          // compile-time LDS strides of the default tile shape (557 owned nodes)
template <int BLK, int NPT, int EPT, int CAPO, bool HASB, bool PHYS, typename V2, bool ADAM, bool CHAIN>
static void launch_pair_lab_inst2(const PairLaunch &A, const LagSum &lag, const AdamFuse &af) {
    constexpr int CAPN = CAPO > 0 ? kPairLabCapN : 0;
    const size_t lds = CAPO > 0 ? (size_t)(CAPN * 32 + CAPO * 32 + 128) : A.lds;
    hipLaunchKernelGGL((tri3_energy_pair_lab_kernel<BLK, NPT, EPT, 4, CAPO, HASB, PHYS, V2, ADAM, CHAIN, CAPN>), dim3(A.grid), dim3(BLK), lds, A.s,
                       A.pd, A.tile_begin, (const V2 *)A.x_free, (const V2 *)A.x_fixed, (const V2 *)A.u_free,
                       (const V2 *)A.u_fixed, A.k, A.T_edge, A.tc, A.partials, (V2 *)A.gx, (V2 *)A.gu, A.max_nodes,
                       CAPO > 0 ? CAPO : A.max_owned, A.skip_edges, lag, af, A.col_stride, A.lab_bits);
}
template <int BLK, int NPT, int EPT, int CAPO, bool HASB, bool PHYS, typename V2, bool ADAM>
static void launch_pair_lab_inst(const PairLaunch &A, const LagSum &lag, const AdamFuse &af) {
    if (A.chain) launch_pair_lab_inst2<BLK, NPT, EPT, CAPO, HASB, PHYS, V2, ADAM, true>(A, lag, af);
    else launch_pair_lab_inst2<BLK, NPT, EPT, CAPO, HASB, PHYS, V2, ADAM, false>(A, lag, af);
}

// Launch on a paired plan: picks the instance that holds the plan's tile shape.  1 = launched, 0 = none does.
// mode: 0 fp64 reference convention (zero body force), 1 general fp64 (body force and / or physical convention),
//       2 fp32 rows, 3 fused Adam write-out on fp64 rows, 4 the same on fp32 rows (3, 4: hasb selects the body-force instance).
int launch_tri3_pair_lab(const hfem_plan *plan, PairLaunch A, int mode, bool hasb, bool phys, const LagSum &lag,
                         const AdamFuse &af) {
    const HostPlan &h = plan->host;
    if (!h.paired || !plan->d_elem_pack_hi) return 0;
    A.pd = plan_dev(plan);
    A.pd.span = A.span;
    A.max_nodes = h.max_nodes; A.max_owned = h.max_owned; A.lds = (size_t)plan->lds_bytes;
    A.col_stride = h.col_stride;
    if (A.chain < 0) A.chain = h.n_chained > 0 ? 1 : 0;   // chained records need the carrying slot loop
    if (h.pair_block != 256 || mode != 0) return 0;            // lab copy: the fp64 reference-convention instances only
    const bool cc = h.max_owned <= kPairLabCapO && h.max_nodes <= kPairLabCapN;   // (656 + 560) * 32 + 128 = 39040 B: four workgroups per CU
    const int npt = h.max_nodes <= 3 * 256 ? 3 : 4;
    const int ept = h.max_rows > 0 ? h.max_rows : 1;     // slots per thread (0: a plan of element-less tiles)
    if (h.max_nodes > 4 * 256 || ept > 6) return 0;
#define HFEM_PAIRLAB_EPT(NPT, CO, HB, PH, V, AD)                                                    \
    switch (ept) {                                                                               \
        case 1: case 2: case 3: launch_pair_lab_inst<256, NPT, 3, CO, HB, PH, V, AD>(A, lag, af); return 1; \
        case 4: launch_pair_lab_inst<256, NPT, 4, CO, HB, PH, V, AD>(A, lag, af); return 1;           \
        case 5: launch_pair_lab_inst<256, NPT, 5, CO, HB, PH, V, AD>(A, lag, af); return 1;           \
        default: launch_pair_lab_inst<256, NPT, 6, CO, HB, PH, V, AD>(A, lag, af); return 1;          \
    }
    if (mode == 0) {
        if (cc && npt == 3) HFEM_PAIRLAB_EPT(3, 560, false, false, double2, false)
        if (npt == 3) HFEM_PAIRLAB_EPT(3, 0, false, false, double2, false)
        HFEM_PAIRLAB_EPT(4, 0, false, false, double2, false)
    }
#undef HFEM_PAIRLAB_EPT
    return 0;
}

}  // namespace hfem

#endif  // HFEM_LAB

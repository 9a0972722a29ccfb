// Paired-slot TRI3 + EDGE2 energy kernel in fp32 ARITHMETIC, gfx950 (MI355X): the pass of tri3_pair.hip for models in the
// reference's DEFAULT dtype (/root/reference/src/loss.py:16 `dtype=torch.float32`, src/models.py:274; example 4 as shipped
// never calls .double()).  Same path -- EnergyLoss2D.__call__ + loss.backward() of /root/reference/src/loss.py:55-116 over
// /root/reference/src/models.py:292-376 --, same owner-computes tile plan, same closed forms (hfem_device.h), but every
// quantity the reference itself holds in fp32 is fp32 here too:
//   * rows are float2 in HBM and ONE float4 {x, y, ux, uy} per node in LDS: a node is one ds_read_b128 (the fp64 kernel reads
//     two), the node image is half the bytes;
//   * the two elements of a slot, A = (n, b, c) and B = (n, c, d), are evaluated SIDE BY SIDE in the two halves of packed fp32
//     registers (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): a pair costs the vector instructions of one element;
//   * gradient accumulators stay four DOUBLE arrays in LDS (ds_add_f64): gfx950 services ds_add_f32 lane by lane -- the
//     first version of this kernel, with float accumulators, took 46.9 us on T1M against 9.1 us for the fp64-arithmetic
//     float-row instance (profiles/r04: fp32_first_try.json) --, while ds_add_f64 goes 16 lanes per LDS cycle; each
//     contribution is widened (v_cvt_f64_f32) on the way in, rows are rounded ONCE on the way out, as float2.
//     Measured per wave-instruction per CU (scripts/micro/lds_atomic_bench.hip, profiles/r04/lds_atomic_bench.txt): ds_add_f64
//     8.6 cycles, ds_add_u64 6.5, ds_add_u32 4.5, ds_add_f32 194.  Two fixed-point variants built on that (32-bit fixed
//     point, two values per ds_add_u64: 8 atomics per pair instead of 16) were correct and SLOWER than this kernel's 7.3 us:
//     evaluate-all-slots-first / scale by the tile's maximum / add afterwards 9.3 us (every wave of a CU in the same phase),
//     one pass with a lagged per-tile scale 8.6-8.8 us (the scale / convert / pack / track-the-maximum instructions double the
//     slot loop's vector work, which then binds) -- DESIGN.md 10.2.
// The tile energy is accumulated in fp64 too (one v_cvt + one v_add_f64 per slot): loss and gradient SUMS are better than the
// reference's fp32 sums, never worse.  Accuracy contract (tests/test_gpu_tri3_f32.py): within the band the reference's own fp32 run occupies
// around exact arithmetic on the same float inputs (gradients <= 4e-6 x max|g|); the fp64-arithmetic float-row instances of
// tri3_pair.hip stay available as the accurate option (hfem_tri3_energy_plan_f32 without HFEM_FLAG_FP32_MATH).
// HBM-bound by construction, no MFMA (2x2 / 2x3 contractions).  Algorithmic bytes per launch: 12 Ne + 32 Nn + 8.
#include <hip/hip_runtime.h>

#include "hfem_device.h"
#include "hfem_plan_dev.h"

namespace hfem {

typedef float f2 __attribute__((ext_vector_type(2)));

struct Tri3ConstsF {
    float c11, c12, c22, c33, W;
    float Bk[6];
};

__device__ __forceinline__ f2 rcp2(f2 x) {               // 1/x per half: v_rcp_f32 (1 ulp) + one Newton step
    f2 r;
    r.x = __builtin_amdgcn_rcpf(x.x);
    r.y = __builtin_amdgcn_rcpf(x.y);
    const f2 e = __builtin_elementwise_fma(-x, r, (f2)(1.0f));
    return __builtin_elementwise_fma(r, e, r);
}
__device__ __forceinline__ f2 copysign2(float m, f2 s) {
    f2 r;
    r.x = __builtin_copysignf(m, s.x);
    r.y = __builtin_copysignf(m, s.y);
    return r;
}

// Two TRI3 elements at once, element A in the .x half and element B in the .y half of every operand.  The closed forms of
// tri3_element<> (hfem_device.h), reference convention, statement for statement.
// Inputs per local node k: Xkx = (A.Xk.x, B.Xk.x), ...  Outputs: gxk_ / guk_ likewise; returns (e_A, e_B).
template <bool HASB>
__device__ __forceinline__ f2 tri3_pair_f32(const f2 X0x, const f2 X0y, const f2 X1x, const f2 X1y, const f2 X2x, const f2 X2y,
                                            const f2 U0x, const f2 U0y, const f2 U1x, const f2 U1y, const f2 U2x, const f2 U2y,
                                            const Tri3ConstsF &k, f2 (&gxx)[3], f2 (&gxy)[3], f2 (&gux)[3], f2 (&guy)[3]) {
    const f2 a = X0x - X2x, d = X1y - X2y, b = X1x - X2x, c = X0y - X2y;
    const f2 det = a * d - b * c;
    const f2 inv = rcp2(det);
    const f2 ai = a * inv, bi = b * inv, ci = c * inv, di = d * inv;
    const f2 g0x = U0x - U2x, g0y = U0y - U2y, g1x = U1x - U2x, g1y = U1y - U2y;
    const f2 h00 = g0x * di - g1x * bi, h01 = g1x * ai - g0x * ci;
    const f2 h10 = g0y * di - g1y * bi, h11 = g1y * ai - g0y * ci;
    const f2 gam = h01 + h10;
    const f2 sxx = k.c11 * h00 + k.c12 * h11;
    const f2 syy = k.c12 * h00 + k.c22 * h11;
    const f2 sxy = k.c33 * gam;
    const f2 hs = h00 * sxx + h11 * syy + gam * sxy;
    const f2 sW = copysign2(k.W, det);
    const f2 aw = det * sW;
    f2 e = (0.5f * aw) * hs;
    f2 beta = (f2)(0.0f), A = (f2)(0.0f), sgn = (f2)(0.0f);
    if (HASB) {
        A = __builtin_elementwise_abs(det);
        sgn = copysign2(1.0f, det);
        beta = U0x * k.Bk[0] + U0y * k.Bk[1] + U1x * k.Bk[2] + U1y * k.Bk[3] + U2x * k.Bk[4] + U2y * k.Bk[5];
        e -= A * beta;
    }
    const f2 p00 = aw * sxx, p01 = aw * sxy, p11 = aw * syy;
    const f2 dg0x = p00 * di - p01 * ci, dg0y = p01 * di - p11 * ci;
    const f2 dg1x = p01 * ai - p00 * bi, dg1y = p11 * ai - p01 * bi;
    gux[0] = dg0x; guy[0] = dg0y;
    gux[1] = dg1x; guy[1] = dg1y;
    gux[2] = -dg0x - dg1x; guy[2] = -dg0y - dg1y;
    if (HASB) {
        gux[0] -= A * k.Bk[0]; guy[0] -= A * k.Bk[1];
        gux[1] -= A * k.Bk[2]; guy[1] -= A * k.Bk[3];
        gux[2] -= A * k.Bk[4]; guy[2] -= A * k.Bk[5];
    }
    f2 ddet = (-0.5f * sW) * hs;
    if (HASB) ddet -= sgn * beta;
    const f2 da = sW * (sxy * g1x + syy * g1y) + ddet * d;
    const f2 db = -sW * (sxx * g1x + sxy * g1y) - ddet * c;
    const f2 dc = -sW * (sxy * g0x + syy * g0y) - ddet * b;
    const f2 dd = sW * (sxx * g0x + sxy * g0y) + ddet * a;
    gxx[0] = da; gxy[0] = dc;
    gxx[1] = db; gxy[1] = dd;
    gxx[2] = -da - db; gxy[2] = -dc - dd;
    return e;
}

// BLOCK threads per tile; NPT >= ceil(max nodes / BLOCK), EPT >= ceil(max slots / BLOCK); CAPO / CAPN > 0: compile-time LDS
// strides of the default tile shape.  SP: cache policy of the gradient stores (16 sc1 write-through, 2 nt).  LDS:
// float4 nd[cap_n] | double acc[4][cap_owned] | double red[BLOCK / 64].
// ADAM: the write-out applies torch.optim.Adam's update to the rows the tile owns instead of storing the gradient
// (hfem_tri3_energy_adam_step_ex with HFEM_FLAG_FP32_MATH; arithmetic = adam_fused_row<float2>, the float-row instance's).
template <int BLOCK, int NPT, int EPT, int CAPO, int CAPN, int SP, bool HASB, bool ADAM = false>
__global__ __launch_bounds__(BLOCK, BLOCK == 512 ? 3 : (ADAM && EPT > 4 ? 4 : 5)) void tri3_energy_pair_f32_kernel(
    PlanDev pd, int tile_begin, const float2 *__restrict__ x_free, const float2 *__restrict__ x_fixed,
    const float2 *__restrict__ u_free, const float2 *__restrict__ u_fixed, Tri3ConstsF k,
    const double4 *__restrict__ T_edge, double4 Tconst, double *__restrict__ partials,
    float2 *__restrict__ gx_free, float2 *__restrict__ gu_free, int cap_nodes, int cap_owned_rt, int skip_edges,
    LagSum lag, int col_stride, AdamFuse af) {
    const int cap_owned = CAPO > 0 ? CAPO : cap_owned_rt;
    const int cap_n = CAPN > 0 ? CAPN : cap_nodes;
    extern __shared__ float4 lds4[];
    float4 *nd = lds4;
    double *acc0 = reinterpret_cast<double *>(lds4 + cap_n);
    double *acc1 = acc0 + cap_owned, *acc2 = acc1 + cap_owned, *acc3 = acc2 + cap_owned;
    double *red = acc3 + cap_owned;

    const int tid = threadIdx.x;
    const int bid = (int)blockIdx.x;
    const int n_launch = (int)gridDim.x - (lag.prev ? 1 : 0);
    if (lag.prev && bid == n_launch) {                  // HFEM_FLAG_SUM_PREVIOUS: reduce the previous launch's tile energies
        double v = 0.0;
        if (tid < 256)
            for (int i = tid; i < lag.prev_n; i += 256) v += lag.prev[i];
        const double tot = block_sum(v, red);
        if (tid == 0) lag.out[0] = tot;
        return;
    }
    const int slot = xcd_tile(bid, n_launch);
    // ---- row maps, descriptor and slot records from the tile index alone (uniform strides, plan.cpp): one round of loads
    int2 s[NPT];
    const int2 *src = pd.node_src + (size_t)(tile_begin + slot) * pd.node_stride;
#pragma unroll
    for (int j = 0; j < NPT; ++j) s[j] = src[min(tid + j * BLOCK, pd.node_stride - 1)];
    const TileDesc d = pd.tiles[tile_begin + slot];
    uint32_t w0[EPT], w1[EPT];
    const size_t rec0 = (size_t)(tile_begin + slot) * pd.elem_stride;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const size_t i = rec0 + min(tid + j * col_stride, pd.elem_stride - 1);
        w0[j] = pd.elem_pack[i];
        w1[j] = pd.elem_pack_hi[i];
    }
    // ---- gather through the row maps: all rows requested before anything is waited for
    float2 vx[NPT], vu[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const float2 *px = s[j].x >= 0 ? x_free + s[j].x : x_fixed + ~s[j].x;
        const float2 *pu = s[j].y >= 0 ? u_free + s[j].y : u_fixed + ~s[j].y;
        vx[j] = *px;
        vu[j] = *pu;
    }
    __builtin_amdgcn_sched_barrier(0);
    const int n_owned = d.n_owned;
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
        if (l < d.n_node) nd[l] = make_float4(vx[j].x, vx[j].y, vu[j].x, vu[j].y);
        if (l < n_owned) { acc0[l] = 0.0; acc1[l] = 0.0; acc2[l] = 0.0; acc3[l] = 0.0; }
    }
    __syncthreads();

    auto add_row = [&](int l, float gx, float gy, float gu, float gv) {
        unsafeAtomicAdd(&acc0[l], (double)gx); unsafeAtomicAdd(&acc1[l], (double)gy);
        unsafeAtomicAdd(&acc2[l], (double)gu); unsafeAtomicAdd(&acc3[l], (double)gv);
    };
    double e_loc = 0.0;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const uint32_t p = w0[j], q = w1[j];
        if (!(p & kSkipBit)) {
            const int ln = (int)(p & kLocalMask), lb = (int)((p >> kLocalBits) & kLocalMask),
                      lc = (int)((p >> (2 * kLocalBits)) & kLocalMask);
            const bool hasB = (q & (1u << 10)) != 0;
            const int ld = hasB ? (int)(q & kLocalMask) : lb;      // no partner: the B half re-evaluates (n, c, b) -- finite, unused
            const float4 Nn = nd[ln], Nb = nd[lb], Nc = nd[lc], Nd = nd[ld];
            f2 gxx[3], gxy[3], gux[3], guy[3];
            const f2 e = tri3_pair_f32<HASB>(
                (f2){Nn.x, Nn.x}, (f2){Nn.y, Nn.y}, (f2){Nb.x, Nc.x}, (f2){Nb.y, Nc.y}, (f2){Nc.x, Nd.x}, (f2){Nc.y, Nd.y},
                (f2){Nn.z, Nn.z}, (f2){Nn.w, Nn.w}, (f2){Nb.z, Nc.z}, (f2){Nb.w, Nc.w}, (f2){Nc.z, Nd.z}, (f2){Nc.w, Nd.w},
                k, gxx, gxy, gux, guy);
            const float mB = hasB ? 1.0f : 0.0f;                   // B's half is finite either way: a multiply masks it
            if (p & kHomeBit) e_loc += (double)e.x;
            if (q & (1u << 11)) e_loc += (double)e.y;
            if (lb < n_owned) add_row(lb, gxx[1].x, gxy[1].x, gux[1].x, guy[1].x);
            if (hasB && ld < n_owned) add_row(ld, gxx[2].y, gxy[2].y, gux[2].y, guy[2].y);
            if (ln < n_owned)
                add_row(ln, __builtin_fmaf(mB, gxx[0].y, gxx[0].x), __builtin_fmaf(mB, gxy[0].y, gxy[0].x),
                        __builtin_fmaf(mB, gux[0].y, gux[0].x), __builtin_fmaf(mB, guy[0].y, guy[0].x));
            if (lc < n_owned)
                add_row(lc, __builtin_fmaf(mB, gxx[1].y, gxx[2].x), __builtin_fmaf(mB, gxy[1].y, gxy[2].x),
                        __builtin_fmaf(mB, gux[1].y, gux[2].x), __builtin_fmaf(mB, guy[1].y, guy[2].x));
        }
    }
    for (int i = tid; i < n_edge; i += BLOCK) {          // boundary tiles only: a handful of edges, fp64 helper on the float image
        const uint32_t p = i == tid ? edge_rec : pd.edge_pack[d.edge_off + i];
        const int l0 = (int)(p & kLocalMask), l1 = (int)((p >> kLocalBits) & kLocalMask);
        const double4 tt = T_edge ? T_edge[i == tid ? edge_id : pd.edge_gid[d.edge_off + i]] : Tconst;
        const float4 N0 = nd[l0], N1 = nd[l1];
        double2 gx[2], gu[2];
        const double wk = edge2_element<true>(make_double2(N0.x, N0.y), make_double2(N1.x, N1.y), make_double2(N0.z, N0.w),
                                              make_double2(N1.z, N1.w), tt, gx, gu);
        if (p & kHomeBit) e_loc -= wk;
        if (l0 < n_owned) add_row(l0, (float)gx[0].x, (float)gx[0].y, (float)gu[0].x, (float)gu[0].y);
        if (l1 < n_owned) add_row(l1, (float)gx[1].x, (float)gx[1].y, (float)gu[1].x, (float)gu[1].y);
    }
    {
        const double w = wave_sum(e_loc);
        if ((tid & 63) == 0) red[tid >> 6] = w;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // all loads returned long ago: no per-store vmcnt waits below
    __syncthreads();

    if constexpr (ADAM) {
        // the tile owns the row: complete gradient (acc*) and current value (nd) are in LDS -- m, v read-modify-written, the NEW
        // row to the OTHER parameter buffer (ping-pong); the gradient is rounded to float once, as the store below would
        const double bc1 = af.bc[0], sqrt_bc2 = af.bc[1];
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int l = tid + j * BLOCK;
            if (l < n_owned) {
                const float4 v = nd[l];
                if (s[j].x >= 0) adam_fused_row<float2>(af, 0, s[j].x, acc0[l], acc1[l], make_double2((double)v.x, (double)v.y), bc1, sqrt_bc2);
                if (s[j].y >= 0) adam_fused_row<float2>(af, 1, s[j].y, acc2[l], acc3[l], make_double2((double)v.z, (double)v.w), bc1, sqrt_bc2);
            }
        }
    } else {
    // ---- every owned gradient row is written exactly once
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)gx_free, 0, 0x7FFFFFF0, 0x00020000);
    __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void *)gu_free, 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int l = tid + j * BLOCK;
        if (l < n_owned) {
            if (gx_free && s[j].x >= 0) {
                const float2 v = make_float2((float)acc0[l], (float)acc1[l]);       // rounded once
                __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), rx, s[j].x * 8, 0, SP);
            }
            if (gu_free && s[j].y >= 0) {
                const float2 v = make_float2((float)acc2[l], (float)acc3[l]);
                __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), ru, s[j].y * 8, 0, SP);
            }
        }
    }
    }   // !ADAM
    if (tid == 0) {                                     // fixed order: the tile energy is bit-reproducible for given gradients
        double tile_e = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) tile_e += red[w];
        partials[slot] = tile_e;
    }
}

template <int BLK, int NPT, int EPT, int CAPO, int CAPN, int SP, bool HASB, bool ADAM = false>
static void launch_pair_f32_inst(const PairLaunch &A, const Tri3ConstsF &kf, const LagSum &lag, const AdamFuse &af = AdamFuse{}) {
    const int cap_n = CAPN > 0 ? CAPN : ((A.max_nodes + 1) & ~1), cap_o = CAPO > 0 ? CAPO : ((A.max_owned + 1) & ~1);
    const size_t lds = (size_t)cap_n * 16 + (size_t)cap_o * 32 + 8 * (BLK / 64);
    hipLaunchKernelGGL((tri3_energy_pair_f32_kernel<BLK, NPT, EPT, CAPO, CAPN, SP, HASB, ADAM>), dim3(A.grid), dim3(BLK), lds, A.s, A.pd,
                       A.tile_begin, (const float2 *)A.x_free, (const float2 *)A.x_fixed, (const float2 *)A.u_free,
                       (const float2 *)A.u_fixed, kf, A.T_edge, A.tc, A.partials, (float2 *)A.gx, (float2 *)A.gu, cap_n, cap_o,
                       A.skip_edges, lag, A.col_stride, af);
}

// Launch on a paired plan without chained records; 1 = launched, 0 = no instance holds the plan's tile shape.
// adam != NULL: the fused optimiser step (instances of the plain shapes; a body force takes the generic one).
int launch_tri3_pair_f32(const hfem_plan *plan, PairLaunch A, bool hasb, const LagSum &lag, const AdamFuse *adam) {
    const HostPlan &h = plan->host;
    if (!h.paired || !plan->d_elem_pack_hi || h.n_chained > 0 || lag.pg_blocks) return 0;
    A.pd = plan_dev(plan);
    A.max_nodes = h.max_nodes; A.max_owned = h.max_owned;
    A.col_stride = h.col_stride;
    Tri3ConstsF kf;
    kf.c11 = (float)A.k.c11; kf.c12 = (float)A.k.c12; kf.c22 = (float)A.k.c22; kf.c33 = (float)A.k.c33; kf.W = (float)A.k.W;
    for (int i = 0; i < 6; ++i) kf.Bk[i] = (float)A.k.Bk[i];
    const int ept = h.max_rows > 0 ? h.max_rows : 1;
    if (adam) {
        if (h.pair_block == 512) {
            if (h.max_nodes > 2 * 512 || ept > 2) return 0;
            if (hasb) launch_pair_f32_inst<512, 2, 2, 0, 0, 16, true, true>(A, kf, lag, *adam);
            else launch_pair_f32_inst<512, 2, 2, 0, 0, 16, false, true>(A, kf, lag, *adam);
            return 1;
        }
        if (h.max_nodes > 4 * 256 || ept > 6) return 0;
        if (hasb) launch_pair_f32_inst<256, 4, 6, 0, 0, 16, true, true>(A, kf, lag, *adam);
        else if (h.max_owned <= 560 && h.max_nodes <= 656 && ept <= 3) launch_pair_f32_inst<256, 3, 3, 560, 656, 16, false, true>(A, kf, lag, *adam);
        else if (h.max_nodes <= 3 * 256 && ept <= 3) launch_pair_f32_inst<256, 3, 3, 0, 0, 16, false, true>(A, kf, lag, *adam);
        else launch_pair_f32_inst<256, 4, 6, 0, 0, 16, false, true>(A, kf, lag, *adam);
        return 1;
    }
    if (h.pair_block == 512) {
        if (h.max_nodes > 2 * 512 || ept > 2) return 0;
        if (hasb) launch_pair_f32_inst<512, 2, 2, 0, 0, 16, true>(A, kf, lag);
        else launch_pair_f32_inst<512, 2, 2, 0, 0, 16, false>(A, kf, lag);
        return 1;
    }
    if (h.max_nodes > 4 * 256 || ept > 6) return 0;
    if (hasb) { launch_pair_f32_inst<256, 4, 6, 0, 0, 16, true>(A, kf, lag); return 1; }
    const bool nt = plan->tune.store_policy == 2;
    if (h.max_owned <= 560 && h.max_nodes <= 656 && ept <= 3) {          // the default tile shape: compile-time LDS strides
        if (nt) launch_pair_f32_inst<256, 3, 3, 560, 656, 2, false>(A, kf, lag);
        else launch_pair_f32_inst<256, 3, 3, 560, 656, 16, false>(A, kf, lag);
        return 1;
    }
    if (h.max_nodes <= 3 * 256 && ept <= 3) launch_pair_f32_inst<256, 3, 3, 0, 0, 16, false>(A, kf, lag);
    else if (ept <= 4) launch_pair_f32_inst<256, 4, 4, 0, 0, 16, false>(A, kf, lag);
    else launch_pair_f32_inst<256, 4, 6, 0, 0, 16, false>(A, kf, lag);
    return 1;
}

}  // namespace hfem

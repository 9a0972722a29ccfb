// Paired-slot TRI3 + EDGE2 energy kernel in fp32 ARITHMETIC, gfx950 (MI355X): the pass of tri3_pair.hip for models in the
// reference's DEFAULT dtype (/root/reference/src/loss.py:16 `dtype=torch.float32`, src/models.py:274; example 4 as shipped
// never calls .double()).  Same path -- EnergyLoss2D.__call__ + loss.backward() of /root/reference/src/loss.py:55-116 over
// /root/reference/src/models.py:292-376 --, same owner-computes tile plan, same closed forms (hfem_device.h), but every
// quantity the reference itself holds in fp32 is fp32 here too:
//   * rows are float2 in HBM and ONE float4 {x, y, ux, uy} per node in LDS: a node is one ds_read_b128 (the fp64 kernel reads
//     two), the node image is half the bytes;
//   * the two elements of a slot, A = (n, b, c) and B = (n, c, d), are evaluated SIDE BY SIDE in the two halves of packed fp32
//     registers (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): a pair costs the vector instructions of one element;
//   * gradient rows are accumulated in 32-bit FIXED POINT, two values per ds_add_u64: gfx950 services ds_add_f32 lane by lane
//     (194 cycles per wave-instruction; the first version of this kernel, with float accumulators, took 46.9 us on T1M), where
//     ds_add_f64 takes 8.6 and ds_add_u64 6.5 (scripts/micro/lds_atomic_bench.hip, profiles/r04/lds_atomic_bench.txt): a
//     pair issues 8 ds_add_u64 instead of 16 ds_add_f64.  Every contribution is scaled by a power of two 2^k, rounded to int32,
//     and (gx, gy) / (gu, gv) are packed into one 64-bit word each (hi + borrow : lo, so that adding words adds both halves);
//     k is chosen so that valence x largest |contribution| x 2^k < 2^30 (the valence bound comes from the plan).  The largest
//     contribution of a tile is not known before the slots are evaluated, and evaluating first / adding afterwards puts
//     all waves of a CU into the same phase (measured: 9.3 us, slower than the double accumulators' 7.3) -- so the scale is
//     LAGGED: every tile remembers the maxima of its previous evaluation (plan-owned array), the slot loop runs ONCE with
//     4 x that as its bound while it tracks this evaluation's maxima, and only a tile whose bound turns out too small (the
//     first evaluation of a plan; gradients that grew more than 4 x from one evaluation to the next) clears its accumulators
//     and redoes its slots with the exact scale.  Resolution: 2^-28 / valence of the largest contribution -- the size of fp32
//     accumulation's own rounding (the reference sums in fp32).  A non-finite element energy (degenerate element) poisons
//     the tile's owned rows with NaN: NaN / Inf propagate, coarser than element by element.
//     The tile energy is accumulated in fp64 (one v_cvt + one v_add_f64 per slot): never worse than the reference's fp32 sum.
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
// float4 nd[cap_n] | u64 acc[2][cap_owned] | double red[BLOCK / 64] | float mred[2][BLOCK / 64] | int bad.
template <int BLOCK, int NPT, int EPT, int CAPO, int CAPN, int SP, bool HASB>
__global__ __launch_bounds__(BLOCK, BLOCK == 512 ? 3 : 5) void tri3_energy_pair_f32_kernel(
    PlanDev pd, int tile_begin, const float2 *__restrict__ x_free, const float2 *__restrict__ x_fixed,
    const float2 *__restrict__ u_free, const float2 *__restrict__ u_fixed, Tri3ConstsF k,
    const double4 *__restrict__ T_edge, double4 Tconst, double *__restrict__ partials,
    float2 *__restrict__ gx_free, float2 *__restrict__ gu_free, int cap_nodes, int cap_owned_rt, int skip_edges,
    LagSum lag, int col_stride, int vbits, float2 *__restrict__ tile_scale) {
    const int cap_owned = CAPO > 0 ? CAPO : cap_owned_rt;
    const int cap_n = CAPN > 0 ? CAPN : cap_nodes;
    extern __shared__ float4 lds4[];
    float4 *nd = lds4;
    unsigned long long *accX = reinterpret_cast<unsigned long long *>(lds4 + cap_n);
    unsigned long long *accU = accX + cap_owned;
    double *red = reinterpret_cast<double *>(accU + cap_owned);
    float *mred = reinterpret_cast<float *>(red + BLOCK / 64);
    int *bad_tile = reinterpret_cast<int *>(mred + 2 * (BLOCK / 64));

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
    const float2 prev_max = tile_scale[tile_begin + slot];      // this tile's maxima at its previous evaluation (0: none yet)
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
        if (l < n_owned) { accX[l] = 0ull; accU[l] = 0ull; }
    }
    if (tid == 0) *bad_tile = 0;
    __syncthreads();

    // ---- scale of this evaluation: 4 x the previous evaluation's maxima as the bound (valence x bound x 2^k < 2^30)
    int ex, eu;
    (void)frexpf(4.0f * prev_max.x, &ex);
    (void)frexpf(4.0f * prev_max.y, &eu);
    int kx = min(96, 30 - vbits - ex), ku = min(96, 30 - vbits - eu);
    float sx = ldexpf(1.0f, kx), su = ldexpf(1.0f, ku);
    float mx = 0.0f, mu = 0.0f;                          // this evaluation's largest |coordinate-row value|, |displacement-row value|
    bool bad = false;
    double e_loc = 0.0;
    // fixed point, two values per 64-bit atomic: word = (int64)hi * 2^32 + (int64)lo -- adding words adds both halves
    auto add_row = [&](int l, float gx, float gy, float gu, float gv) {
        const int qx = __float2int_rn(gx * sx), qy = __float2int_rn(gy * sx), qu = __float2int_rn(gu * su), qv = __float2int_rn(gv * su);
        atomicAdd(&accX[l], ((unsigned long long)(uint32_t)(qy + (qx >> 31)) << 32) | (uint32_t)qx);
        atomicAdd(&accU[l], ((unsigned long long)(uint32_t)(qv + (qu >> 31)) << 32) | (uint32_t)qu);
    };
    auto see = [&](float a, float b, float c, float d) {
        mx = fmaxf(mx, fmaxf(fabsf(a), fabsf(b)));
        mu = fmaxf(mu, fmaxf(fabsf(c), fabsf(d)));
    };
    // one slot: evaluate the pair, add its four rows {b, d, n, c}; first: also count the energy and track the maxima
    auto do_slot = [&](uint32_t p, uint32_t q, bool first) {
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
        const float nx = __builtin_fmaf(mB, gxx[0].y, gxx[0].x), ny = __builtin_fmaf(mB, gxy[0].y, gxy[0].x);     // n: A + B
        const float nu = __builtin_fmaf(mB, gux[0].y, gux[0].x), nv = __builtin_fmaf(mB, guy[0].y, guy[0].x);
        const float cx = __builtin_fmaf(mB, gxx[1].y, gxx[2].x), cy = __builtin_fmaf(mB, gxy[1].y, gxy[2].x);     // c: A + B
        const float cu = __builtin_fmaf(mB, gux[1].y, gux[2].x), cv = __builtin_fmaf(mB, guy[1].y, guy[2].x);
        if (first) {
            if (p & kHomeBit) e_loc += (double)e.x;
            if (q & (1u << 11)) e_loc += (double)e.y;
            bad = bad || !(fabsf(e.x) <= 3.0e38f) || (hasB && !(fabsf(e.y) <= 3.0e38f));       // a degenerate element: non-finite energy
            see(gxx[1].x, gxy[1].x, gux[1].x, guy[1].x);
            see(mB * gxx[2].y, mB * gxy[2].y, mB * gux[2].y, mB * guy[2].y);
            see(nx, ny, nu, nv);
            see(cx, cy, cu, cv);
        }
        if (lb < n_owned) add_row(lb, gxx[1].x, gxy[1].x, gux[1].x, guy[1].x);
        if (hasB && ld < n_owned) add_row(ld, gxx[2].y, gxy[2].y, gux[2].y, guy[2].y);
        if (ln < n_owned) add_row(ln, nx, ny, nu, nv);
        if (lc < n_owned) add_row(lc, cx, cy, cu, cv);
    };
    auto do_edge = [&](bool first) {                     // boundary tiles only: at most one edge per thread (host-checked), fp64 helper
        if (tid < n_edge) {
            const int l0 = (int)(edge_rec & kLocalMask), l1 = (int)((edge_rec >> kLocalBits) & kLocalMask);
            const double4 tt = T_edge ? T_edge[edge_id] : Tconst;
            const float4 N0 = nd[l0], N1 = nd[l1];
            double2 gx[2], gu[2];
            const double wk = edge2_element<true>(make_double2(N0.x, N0.y), make_double2(N1.x, N1.y), make_double2(N0.z, N0.w),
                                                  make_double2(N1.z, N1.w), tt, gx, gu);
            if (first) {
                if (edge_rec & kHomeBit) e_loc -= wk;
                see((float)gx[0].x, (float)gx[0].y, (float)gu[0].x, (float)gu[0].y);
                see((float)gx[1].x, (float)gx[1].y, (float)gu[1].x, (float)gu[1].y);
            }
            if (l0 < n_owned) add_row(l0, (float)gx[0].x, (float)gx[0].y, (float)gu[0].x, (float)gu[0].y);
            if (l1 < n_owned) add_row(l1, (float)gx[1].x, (float)gx[1].y, (float)gu[1].x, (float)gu[1].y);
        }
    };
    // ---- the slot loop, ONCE, with the lagged scale
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (!(w0[j] & kSkipBit)) do_slot(w0[j], w1[j], true);
    do_edge(true);
    // ---- this evaluation's maxima; was the bound large enough?
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        mu = fmaxf(mu, __shfl_xor(mu, off, 64));
    }
    if ((tid & 63) == 0) { mred[tid >> 6] = mx; mred[BLOCK / 64 + (tid >> 6)] = mu; }
    if (bad) *bad_tile = 1;
    __syncthreads();
    float Mx = 0.0f, Mu = 0.0f;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) { Mx = fmaxf(Mx, mred[w]); Mu = fmaxf(Mu, mred[BLOCK / 64 + w]); }
    int exn, eun;
    (void)frexpf(Mx, &exn);                              // Mx < 2^exn
    (void)frexpf(Mu, &eun);
    if (exn + kx + vbits > 30 || eun + ku + vbits > 30) {        // uniform over the tile: its accumulators may have wrapped -> redo
        for (int l = tid; l < n_owned; l += BLOCK) { accX[l] = 0ull; accU[l] = 0ull; }
        kx = min(96, 30 - vbits - exn); ku = min(96, 30 - vbits - eun);
        sx = ldexpf(1.0f, kx); su = ldexpf(1.0f, ku);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            if (!(w0[j] & kSkipBit)) do_slot(w0[j], w1[j], false);
        do_edge(false);
    }
    if (tid == 0) tile_scale[tile_begin + slot] = make_float2(Mx, Mu);      // the next evaluation's bound
    {
        const double w = wave_sum(e_loc);
        if ((tid & 63) == 0) red[tid >> 6] = w;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // all loads returned long ago: no per-store vmcnt waits below
    __syncthreads();

    // ---- every owned gradient row is written exactly once: unpack (lo signed, hi + the borrow it took), back to float
    const bool poison = *bad_tile != 0 || !(Mx <= 3.0e38f) || !(Mu <= 3.0e38f);
    const float isx = ldexpf(1.0f, -kx), isu = ldexpf(1.0f, -ku), qnan = __builtin_nanf("");
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)gx_free, 0, 0x7FFFFFF0, 0x00020000);
    __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void *)gu_free, 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int l = tid + j * BLOCK;
        if (l < n_owned) {
            if (gx_free && s[j].x >= 0) {
                const unsigned long long t = accX[l];
                const int lo = (int)(uint32_t)t, hi = (int)((uint32_t)(t >> 32) - (uint32_t)(lo >> 31));
                const float2 v = poison ? make_float2(qnan, qnan) : make_float2((float)lo * isx, (float)hi * isx);
                __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), rx, s[j].x * 8, 0, SP);
            }
            if (gu_free && s[j].y >= 0) {
                const unsigned long long t = accU[l];
                const int lo = (int)(uint32_t)t, hi = (int)((uint32_t)(t >> 32) - (uint32_t)(lo >> 31));
                const float2 v = poison ? make_float2(qnan, qnan) : make_float2((float)lo * isu, (float)hi * isu);
                __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2 *>(&v), ru, s[j].y * 8, 0, SP);
            }
        }
    }
    if (tid == 0) {                                     // fixed order: the tile energy is bit-reproducible for given gradients
        double tile_e = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) tile_e += red[w];
        partials[slot] = tile_e;
    }
}

template <int BLK, int NPT, int EPT, int CAPO, int CAPN, int SP, bool HASB>
static void launch_pair_f32_inst(const PairLaunch &A, const Tri3ConstsF &kf, const LagSum &lag) {
    const int cap_n = CAPN > 0 ? CAPN : ((A.max_nodes + 1) & ~1), cap_o = CAPO > 0 ? CAPO : ((A.max_owned + 1) & ~1);
    const size_t lds = (size_t)cap_n * 16 + (size_t)cap_o * 16 + 16 * (BLK / 64) + 16;
    hipLaunchKernelGGL((tri3_energy_pair_f32_kernel<BLK, NPT, EPT, CAPO, CAPN, SP, HASB>), dim3(A.grid), dim3(BLK), lds, A.s, A.pd,
                       A.tile_begin, (const float2 *)A.x_free, (const float2 *)A.x_fixed, (const float2 *)A.u_free,
                       (const float2 *)A.u_fixed, kf, A.T_edge, A.tc, A.partials, (float2 *)A.gx, (float2 *)A.gu, cap_n, cap_o,
                       A.skip_edges, lag, A.col_stride, A.lab_bits /* = valence bits of the plan (launch_tri3_pair_f32) */,
                       reinterpret_cast<float2 *>(const_cast<unsigned long long *>(A.span)) /* = the plan's per-tile maxima */);
}

// Launch on a paired plan without chained records; 1 = launched, 0 = no instance holds the plan's tile shape.
int launch_tri3_pair_f32(const hfem_plan *plan, PairLaunch A, bool hasb, const LagSum &lag) {
    const HostPlan &h = plan->host;
    if (!h.paired || !plan->d_elem_pack_hi || h.n_chained > 0 || lag.pg_blocks) return 0;
    // the fixed-point accumulation needs a bound on the contributions a row can receive (valence: element corners + edge
    // ends of a node, from the plan's copy of the mesh; <= 64 keeps >= 24 bits below the largest contribution) and at most one
    // Neumann edge per thread
    if (plan->f32_vbits < 0 || plan->f32_vbits > 6 || !plan->d_f32_scale || h.max_edges > (h.pair_block == 512 ? 512 : 256)) return 0;
    A.lab_bits = plan->f32_vbits;
    A.span = reinterpret_cast<unsigned long long *>(plan->d_f32_scale);      // PairLaunch fields this launcher borrows (no span stamps here)
    A.pd = plan_dev(plan);
    A.max_nodes = h.max_nodes; A.max_owned = h.max_owned;
    A.col_stride = h.col_stride;
    Tri3ConstsF kf;
    kf.c11 = (float)A.k.c11; kf.c12 = (float)A.k.c12; kf.c22 = (float)A.k.c22; kf.c33 = (float)A.k.c33; kf.W = (float)A.k.W;
    for (int i = 0; i < 6; ++i) kf.Bk[i] = (float)A.k.Bk[i];
    const int ept = h.max_rows > 0 ? h.max_rows : 1;
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

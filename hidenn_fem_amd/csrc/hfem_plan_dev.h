// Device-side view of the tile plan + the opaque plan object, shared by the TRI3 and QUAD4 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "hfem_device.h"

namespace hfem {

struct PlanDev {
    const TileDesc *tiles;
    const uint32_t *elem_pack;
    const int2 *node_src;
    const uint32_t *edge_pack;
    const int32_t *edge_gid;
    const uint32_t *elem_pack_hi;   // QUAD4 plans only: 4th local node id of every slot
    const int4 *tile_chunks;        // chunked plans only (HostPlan::tile_chunks)
    int node_stride;                // tile t's node_src records start at t * node_stride (HostPlan::node_stride)
    int elem_stride;                // tile t's slot records start at t * elem_stride
    unsigned long long *span;       // span stamps (hfem_plan_set_span_stamps): {start, end} s_memrealtime ticks per tile, or NULL
};

// peer.hip: the interface payload written straight into every rank's receive window (no collective); layout in peer.hip
constexpr int kMaxPeers = 16;
constexpr size_t kPeerFlags = 0, kPeerData = 256;
constexpr int kPeerGetBlocks = 8;    // in-launch get: service workgroups in front of the tiles (a multiple of the 8 XCDs)
struct PeerView {                    // kernel argument: every rank's window as mapped into THIS process
    char *win[kMaxPeers];
    char *ctl;                       // this rank's private control block: [0] u64 puts, [8] u32 put ticket, [12] u32 status,
    int rank, world;                 //   [16] u64 unpacked (the put count whose rows the in-launch get has copied in), [24] u32 get ticket
};
struct PeerGetDev {                  // device-resident arguments of the in-launch get (hfem_peer_attach_get)
    PeerView pv;
    const int32_t *src, *dst;        // unpack tables of hfem_iface_unpack
    int n_x, n_u;
    int64_t stride, loss_slot, timeout_ticks;
    double *loss_out;
};

struct PeerPutDev {                  // device-resident arguments of the in-launch put (hfem_peer_attach_put)
    PeerView pv;
    const int32_t *pos_x, *pos_u;    // per parameter ROW: its place in this rank's payload lane (double2 units), or -1
    int64_t stride, loss_slot;
    int64_t *counter;                // the optimiser's device step counter (bumped by the last boundary tile), or NULL
    double beta1, beta2;
};

// parameter rows of either storage type in the exchange payload (always double2: float -> double -> float is lossless)
template <typename V> __device__ __forceinline__ double2 row_widen(V v) { return make_double2((double)v.x, (double)v.y); }
template <typename V> __device__ __forceinline__ V row_narrow(double2 v) {
    V o;
    o.x = (decltype(o.x))v.x; o.y = (decltype(o.y))v.y;
    return o;
}

// Kernel-parameter type of the parameter ROW arrays (x_free, u_free): `const V2 *__restrict__` everywhere except in the PG
// instances (HFEM_FLAG_PEER_GET), whose service workgroups WRITE the foreign interface rows into those very arrays while the
// boundary tiles of the same launch read them afterwards (ordered by peer_wait_unpacked's acquire fence): there the rows are
// plain pointers to non-const data -- no write through a const __restrict__ pointer.
template <typename V2, bool PG> struct RowArg { typedef const V2 *__restrict__ type; };
template <typename V2> struct RowArg<V2, true> { typedef V2 *type; };

struct LagSum {                      // HFEM_FLAG_SUM_PREVIOUS: one extra workgroup reduces the previous launch's tile energies
    const double *prev = nullptr;    // partials bank the previous launch wrote (offset to its first tile)
    int prev_n = 0;
    double *out = nullptr;           // receives their sum (same order and bits as sum_partials_kernel)
    // HFEM_FLAG_PEER_GET (paired-slot kernel only): the first pg_blocks workgroups of the launch are the peer-window get
    // (wait for the flags, unpack into this launch's x_free / u_free, publish `unpacked`); tiles [wait_begin, wait_end) -- the
    // rank's boundary tiles, the only ones that read rows another rank owns -- wait for it before they gather
    const PeerGetDev *pg = nullptr;
    int pg_blocks = 0, wait_begin = 0, wait_end = 0;
    // HFEM_FLAG_PEER_PUT (fused-Adam instances with the in-launch get): the boundary tiles [wait_begin, wait_end) ALSO publish --
    // each stores the NEW rows of its interface nodes into every rank's window at write-out, and the last of them to finish
    // writes the rank's energy of the PREVIOUS evaluation (put_prev: its tile energies, complete since the last kernel
    // boundary), bumps the step counter, writes the next step's bias corrections (put_bc_next) and raises the flags
    const PeerPutDev *put = nullptr;     // non-NULL = the launch publishes; its fields the tiles need early travel BY VALUE below
    const int32_t *put_pos_x = nullptr, *put_pos_u = nullptr;   // (no dependent load of the struct in front of the row lookups)
    const double *put_prev = nullptr;
    int put_prev_n = 0;
    double *put_bc_next = nullptr;
};

// One service workgroup of the in-launch get (256 threads of a tile kernel's block): iface_get_kernel's wait + unpack with
// a block-stride loop, then -- rows visible device-wide -- the last of the nblk workgroups publishes unpacked = puts.
template <typename V>
__device__ __forceinline__ void peer_get_block(const PeerGetDev &G, int bid, int nblk, V *x_free, V *u_free) {
    __shared__ int pg_last;
    char *self = G.pv.win[G.pv.rank], *ctl = G.pv.ctl;
    const int tid = threadIdx.x;
    // only as many workgroups as there are rows to copy take part (the grid always carries kPeerGetBlocks of them so that the
    // tiles keep their XCDs): fewer tickets on the boundary tiles' critical path
    const int need = (G.n_x + G.n_u + 255) / 256;
    nblk = need < 1 ? 1 : (need < nblk ? need : nblk);
    if (bid >= nblk) return;
    const uint64_t want = __hip_atomic_load((uint64_t *)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (want != 0 && tid < 256) {
        const int par = (int)((want - 1) & 1);
        unsigned *status = (unsigned *)(ctl + 12);
        if (tid < G.pv.world) {
            const uint64_t *flag = (const uint64_t *)(self + kPeerFlags) + par * kMaxPeers + tid;
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            // relaxed polls (an acquire load invalidates the caches EVERY time -- other tiles of this launch are running);
            // one acquire fence after the barrier below
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
                if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) break;
                if ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) > G.timeout_ticks) {
                    atomicOr(status, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
    }
    __syncthreads();
    if (want != 0 && tid < 256) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);               // system scope, once per wave: the window's rows
        const int par = (int)((want - 1) & 1);
        const double2 *recv = (const double2 *)(self + kPeerData + (size_t)par * G.pv.world * (size_t)G.stride * sizeof(double2));
        for (int i = bid * 256 + tid; i < G.n_x + G.n_u; i += nblk * 256) {
            if (i < G.n_x) x_free[G.dst[i]] = row_narrow<V>(recv[G.src[i]]);
            else u_free[G.dst[i]] = row_narrow<V>(recv[G.src[i]]);
        }
        if (bid == 0 && tid == 0 && G.loss_out) {
            double tot = 0.0;
            for (int r = 0; r < G.pv.world; ++r) tot += recv[(int64_t)r * G.stride + G.loss_slot].x;
            G.loss_out[0] = tot;
        }
    }
    __threadfence();                                       // agent scope: the rows reach memory before the ticket
    __syncthreads();
    if (tid == 0) pg_last = atomicAdd((unsigned *)(ctl + 24), 1u) == (unsigned)(nblk - 1);
    __syncthreads();
    if (pg_last && tid == 0) {
        *(unsigned *)(ctl + 24) = 0u;
        __hip_atomic_store((uint64_t *)(ctl + 16), want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// A boundary tile of the same launch: wait (bounded) until the get workgroups have published the rows of the current put count.
__device__ __forceinline__ void peer_wait_unpacked(const PeerGetDev &G) {
    if (threadIdx.x == 0) {
        char *ctl = G.pv.ctl;
        const uint64_t want = __hip_atomic_load((uint64_t *)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load((uint64_t *)(ctl + 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) > 2 * G.timeout_ticks) {
                atomicOr((unsigned *)(ctl + 12), 2u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // once: the rows the get workgroups wrote are visible to this CU
    }
    __syncthreads();
}



struct AdamFuse {                    // arguments of the fused optimiser write-out (ADAM instances only)
    void *x_out = nullptr, *u_out = nullptr;   // new parameter rows (free rows, the kernel's row type V2); must not alias the inputs
    void *mx = nullptr, *vx = nullptr, *mu = nullptr, *vu = nullptr;   // Adam moments, free rows of the same type, updated in place
    const double *bc = nullptr;                  // device {1 - b1^step, sqrt(1 - b2^step)} of this step (hfem_adam_prep)
    double lr_x = 0, lr_u = 0, b1 = 0.9, b2 = 0.999, eps = 1e-8;
};

// The fused write-out of one owned row (c = 0: coordinates, 1: displacements): torch.optim.Adam's update in the row type's
// own arithmetic -- operation for operation optim.hip's adam_one<T> (fp64 rows: fp64; fp32 rows, the reference's default
// dtype: the gradient is rounded to float once, as the float-row kernel would store it, then torch's fp32 arithmetic).
// m, v are read and written in place, the NEW row goes to the other parameter buffer (ping-pong).
template <typename V2>
__device__ __forceinline__ V2 adam_fused_row(const AdamFuse &af, int c, int row, double gx, double gy, double2 p64,
                                             double bc1, double sqrt_bc2) {
    typedef decltype(V2().x) T;
    V2 *mp = reinterpret_cast<V2 *>(c ? af.mu : af.mx) + row, *vp = reinterpret_cast<V2 *>(c ? af.vu : af.vx) + row;
    const V2 m = *mp, v = *vp;
    const T w1 = (T)(1.0 - af.b1), w2 = (T)(1.0 - af.b2), b2 = (T)af.b2, eps = (T)af.eps;
    const T ss = (T)((c ? af.lr_u : af.lr_x) / bc1), sb = (T)sqrt_bc2;
    const T g0 = (T)gx, g1 = (T)gy, p0 = (T)p64.x, p1 = (T)p64.y;
    V2 mn, vn, pn;
    mn.x = m.x + w1 * (g0 - m.x); mn.y = m.y + w1 * (g1 - m.y);
    vn.x = v.x * b2 + w2 * (g0 * g0); vn.y = v.y * b2 + w2 * (g1 * g1);
    pn.x = p0 - ss * (mn.x / ((T)sqrt((double)vn.x) / sb + eps));
    pn.y = p1 - ss * (mn.y / ((T)sqrt((double)vn.y) / sb + eps));
    *mp = mn; *vp = vn;
    reinterpret_cast<V2 *>(c ? af.u_out : af.x_out)[row] = pn;
    return pn;
}

// The in-launch put, part 1 (a boundary tile at write-out): the NEW row of an interface node into every rank's window.
// System-scope relaxed atomic stores (global_store ... sc0 sc1: write-through, never parked in this XCD's L2): the window is
// uncached memory, and ordering comes from the counted wait in part 2 -- NOT from a system-scope fence, which from a tile
// workgroup writes back its XCD's whole L2 under the running tiles (first version: 62 us per step instead of 28).
template <typename V2>
__device__ __forceinline__ void peer_put_row(const PeerPutDev &P, size_t slot_off, int pos, V2 pn) {
    const double2 v = row_widen(pn);
    for (int p = 0; p < P.pv.world; ++p) {
        double *q = reinterpret_cast<double *>(P.pv.win[p] + slot_off) + 2 * (size_t)pos;
        __hip_atomic_store(q, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(q + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The in-launch put, part 2 (every boundary tile after its write-out): wait until its own stores are acknowledged
// (s_waitcnt vmcnt(0): write-through system-scope stores complete at their destination), then take a ticket; the LAST of the
// n_boundary tiles publishes the rank's previous energy, bumps the step counter, writes the next bias corrections, and -- its
// own stores acknowledged as well -- raises the flags in every window and completes the put (seq + 1).  A peer that sees a
// flag therefore sees every row: all of them were acknowledged before the flag store was issued.  `red`: >= 4 doubles of
// LDS.  All threads of the block.
__device__ __forceinline__ void peer_put_finish(const PeerPutDev &P, const LagSum &lag, uint64_t seq, size_t slot_off,
                                                int n_boundary, double *red) {
    __shared__ int put_last;
    char *ctl = P.pv.ctl;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this wave's window stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) put_last = atomicAdd((unsigned *)(ctl + 8), 1u) == (unsigned)(n_boundary - 1);
    __syncthreads();
    if (!put_last) return;
    double v = 0.0;
    if (lag.put_prev && threadIdx.x < 256)
        for (int k = threadIdx.x; k < lag.put_prev_n; k += 256) v += lag.put_prev[k];
    const double tot = block_sum(v, red);
    if (threadIdx.x == 0) {
        for (int p = 0; p < P.pv.world; ++p) {
            double *q = reinterpret_cast<double *>(P.pv.win[p] + slot_off) + 2 * (size_t)P.loss_slot;
            __hip_atomic_store(q, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(q + 1, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (P.counter) {                                   // device-local: the next launch reads them behind the kernel boundary
            const int64_t c = P.counter[0] + 1;
            P.counter[0] = c;
            if (lag.put_bc_next) {
                lag.put_bc_next[0] = 1.0 - pow(P.beta1, (double)(c + 1));
                lag.put_bc_next[1] = sqrt(1.0 - pow(P.beta2, (double)(c + 1)));
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    const int par = (int)(seq & 1);
    if ((int)threadIdx.x < P.pv.world)
        __hip_atomic_store((uint64_t *)(P.pv.win[threadIdx.x] + kPeerFlags) + par * kMaxPeers + P.pv.rank, seq + 1,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
        *(unsigned *)(ctl + 8) = 0u;
        __hip_atomic_store((uint64_t *)ctl, seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Wave priority through a tile's memory phases (round 4).  A workgroup that arrives on a CU whose other workgroups are in
// their fp64 slot loops has ~60 prologue instructions (index loads, address selects, the gather) to issue before its first
// byte is requested, and they queue behind the resident waves' VALU work at equal priority; the same holds for the write-out
// of a tile that finishes while its neighbours compute.  s_setprio raises the wave over the arithmetic ones for exactly those
// stretches: the memory requests go out earlier and overlap the neighbours' arithmetic.  Measured (scripts/quad4_lab.py --bits,
// scripts/ab_lib.py, profiles/r04/ab_mem_prio.jsonl): Q1M 22.3 -> 21.8 us (rotating sets 22.5 -> 21.8; 22.3 -> 21.1 in the lab sweep), the paired fp64 TRI3 kernel
// 8.82 -> 8.72 us, T2M and the fused Adam step unchanged; the fp32-arithmetic kernel (shorter slot loop) loses 1 % and the
// one-element-per-slot kernel gains nothing (zigzag 11.1 -> 11.3, Delaunay 4 M unchanged; ab_mem_prio_one_element_per_slot.jsonl): neither uses it.  HFEM_MEM_PRIO = 0 compiles it out.
#ifndef HFEM_MEM_PRIO
#define HFEM_MEM_PRIO 3
#endif
__device__ __forceinline__ void mem_phase_begin() {
    if (HFEM_MEM_PRIO) __builtin_amdgcn_s_setprio(HFEM_MEM_PRIO);
}
__device__ __forceinline__ void mem_phase_end() {
    if (HFEM_MEM_PRIO) __builtin_amdgcn_s_setprio(0);
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an
// L2).  Map block -> tile so that each XCD walks one contiguous run of the
// Morton-ordered tiles: neighbouring tiles share halo nodes, which then hit the
// same L2.  Bijective for any grid size; placement only affects speed.
__device__ __forceinline__ int xcd_tile(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

}  // namespace hfem

namespace hfem {
constexpr int kPipeMaxTiles = 16;   // tiles one persistent workgroup may walk (descriptors cached in LDS)
}

struct hfem_plan {
    hfem::HostPlan host;
    int device = -1;
    // device mirrors
    hfem::TileDesc *d_tiles = nullptr;
    uint32_t *d_elem_pack = nullptr;
    uint32_t *d_elem_pack_hi = nullptr;
    int2 *d_node_src = nullptr;
    uint32_t *d_edge_pack = nullptr;
    int32_t *d_edge_gid = nullptr;
    int4 *d_tile_chunks = nullptr;            // chunked element order only
    // deterministic node-centric path (HFEM_FLAG_DETERMINISTIC): built and uploaded on first use
    struct Det {
        int32_t *conn = nullptr, *x_src = nullptr, *u_src = nullptr, *edges = nullptr;
        int32_t *adj_ptr = nullptr, *adj = nullptr, *eadj_ptr = nullptr, *eadj = nullptr;
        double *partials = nullptr;
        int n_blocks = 0;
        bool ready = false;
    } det;
    double *d_partials = nullptr;             // two banks of [n_tiles] tile energies
    // host-side launch state of the lagged loss sum, guarded by `mu`: one plan = one stream at a time (a
    // HFEM_FLAG_SUM_PREVIOUS launch on a stream other than `prev_stream` is refused)
    std::mutex mu;
    int bank = 0;                             // bank the most recent launch wrote
    int prev_begin = 0, prev_n = 0;           // partial range of the most recent HFEM_FLAG_NO_LOSS_SUM launch
    void *prev_stream = nullptr;              // ... and the stream it went to
    // tuning options captured at creation (hfem_set_option only changes the defaults of LATER plans)
    struct Tune { int tiled_block = 512, store_policy = 16, tiled_fast = 1, fast_const_caps = 1, pair_tiles_per_wg = 1, pair_pipe_wps = 4; } tune;
    unsigned long long *d_stamps = nullptr;   // lab build only: [n_tiles][16] s_memrealtime stamps (NULL otherwise)
    // span stamps (hfem_plan_set_span_stamps; caller-owned buffer of span_slots x n_tiles x 2 uint64): launch i of the
    // paired-slot kernel writes every tile's {start, end} into slot span_next++ % span_slots.  Guarded by `mu`.
    unsigned long long *span_buf = nullptr;
    int64_t span_slots = 0, span_next = 0;
    int64_t device_bytes = 0;
    double row_line_factor = 1.0; // distinct 128-byte lines per tile's coordinate rows / the minimum (hfem_plan_create)
    const hfem::PeerGetDev *peer_get = nullptr;   // hfem_plan_set_peer_get: in-launch get of HFEM_FLAG_PEER_GET launches
    const hfem::PeerPutDev *peer_put = nullptr;   // hfem_plan_set_peer_put: in-launch put of HFEM_FLAG_PEER_PUT launches
    double *put_bc[2] = {nullptr, nullptr};       // ... and the two bias-correction buffers its steps alternate between
    const int32_t *put_pos[2] = {nullptr, nullptr};   // ... and the row -> payload position tables (by value into the launch)
    int peer_wait_begin = 0, peer_wait_end = 0;   // the tiles that wait for it (the rank's boundary tiles)
    int32_t lds_bytes = 0;        // tiled kernel: nodes + accumulators + reduction scratch
    int32_t lds_bytes_pipe = 0;   // pipelined kernel: + descriptor cache + element records
};

namespace hfem {
inline PlanDev plan_dev(const hfem_plan *p) {
    return PlanDev{p->d_tiles, p->d_elem_pack, p->d_node_src, p->d_edge_pack, p->d_edge_gid, p->d_elem_pack_hi, p->d_tile_chunks, p->host.node_stride, p->host.elem_stride, nullptr};
}
inline Tri3Consts make_consts(const double mat[4], double W, const double Bk[6]) {
    Tri3Consts k;
    k.c11 = mat[0]; k.c12 = mat[1]; k.c22 = mat[2]; k.c33 = mat[3];
    k.W = W;
    for (int i = 0; i < 6; ++i) k.Bk[i] = Bk ? Bk[i] : 0.0;
    return k;
}
// tri3_stream.hip: streamed (LDS-DMA, strip-pipelined) TRI3 kernel on a chunked plan; 1 = launched, 0 = shape not held
int launch_tri3_stream(const hfem_plan *plan, int n_grid, int tile_begin, const double *x_free, const double *x_fixed,
                       const double *u_free, const double *u_fixed, const Tri3Consts &kc, const double *T_edge,
                       double4 tc, double *partials, double *gx_free, double *gu_free, int skip_edges, int store_policy,
                       const LagSum &lag, hipStream_t s, int ablate = 0);
// tri3_pair.hip: paired-slot kernel on a paired plan (plan_elem_order 5, the default); 1 = launched, 0 = tile shape not held
struct PairLaunch {
    PlanDev pd;
    int grid = 0, tile_begin = 0;
    const void *x_free = nullptr, *x_fixed = nullptr, *u_free = nullptr, *u_fixed = nullptr;
    Tri3Consts k;
    const double4 *T_edge = nullptr;
    double4 tc;
    double *partials = nullptr;
    void *gx = nullptr, *gu = nullptr;
    int max_nodes = 0, max_owned = 0, skip_edges = 0;
    int chain = -1;               // -1: by the plan (chained records -> carrying slot loop); lab: 0 / 1 forces
    int lab_bits = 0;             // lab build: ablation bits (tri3_pair.hip)
    int col_stride = 256;         // columns of every tile's slot array (HostPlan::col_stride)
    unsigned long long *span = nullptr;   // this launch's span-stamp slot (or NULL)
    size_t lds = 0;
    hipStream_t s = nullptr;
};
int launch_tri3_pair(const hfem_plan *plan, PairLaunch A, int mode, bool hasb, bool phys, const LagSum &lag,
                     const AdamFuse &af);
// tri3_pair_f32.hip: the paired-slot pass in fp32 ARITHMETIC on float rows (HFEM_FLAG_FP32_MATH); 1 = launched, 0 = no instance
int launch_tri3_pair_f32(const hfem_plan *plan, PairLaunch A, bool hasb, const LagSum &lag, const AdamFuse *adam = nullptr);
#ifdef HFEM_LAB
// tri3_pair_lab.hip (lab build only): the instrumented copy of the paired-slot kernel (ablation bits, forced slot loops)
int launch_tri3_pair_lab(const hfem_plan *plan, PairLaunch A, int mode, bool hasb, bool phys, const LagSum &lag,
                         const AdamFuse &af);
#endif
// tri3_pair_pipe.hip: the same pass with tpw tiles per workgroup, software-pipelined; 1 = launched, 0 = no instance
int launch_tri3_pair_pipe(const hfem_plan *plan, PairLaunch A, int n_tiles, int tpw, const LagSum &lag);
// tri3_det.hip: fixed-order (bit-reproducible) energy + gradients; phys: the physical gradient convention
int launch_tri3_det(hfem_plan *plan, const double *x_free, const double *x_fixed, const double *u_free,
                    const double *u_fixed, const Tri3Consts &kc, const double *T_edge, double4 tc, double *loss_out,
                    double *gx_free, double *gu_free, int skip_edges, bool phys, hipStream_t s);
void free_tri3_det(hfem_plan *plan);
int det_prepare(hfem_plan *plan, hipStream_t s);   // adjacency of the fixed-order kernels (TRI3 and QUAD4), built on first use
// exchange.hip: interface pack + tile-energy sum + step-counter bump in one launch (hfem_plan_iface_pack)
int launch_iface_pack_sum(int dtype, const void *x_free, const void *u_free, const int32_t *rows, int n_x, int n_u, double *out,
                          int64_t loss_slot, const double *partials, int n_partials, int64_t *counter, double beta1,
                          double beta2, double *bc_next, hipStream_t s);
int launch_iface_put(const hfem_peer *peer, int dtype, const void *x_free, const void *u_free, const int32_t *rows, int n_x,
                     int n_u, int64_t loss_slot, const double *partials, int n_partials, int64_t *counter, double beta1,
                     double beta2, double *bc_next, hipStream_t s);
}  // namespace hfem
struct hfem_peer {
    int device = -1;
    int64_t stride = 0;              // double2 units per rank and slot: interface rows + the loss entry
    size_t bytes = 0;
    char *local = nullptr;           // this rank's window (hipExtMallocWithFlags, uncached)
    char *ctl = nullptr;             // puts / tickets / status / unpacked (hipMalloc)
    hfem::PeerGetDev *get_dev = nullptr;   // arguments of the in-launch get (hfem_peer_attach_get), or NULL
    hfem::PeerPutDev *put_dev = nullptr;   // arguments of the in-launch put (hfem_peer_attach_put), or NULL
    const int32_t *put_pos[2] = {nullptr, nullptr};
    std::vector<void *> opened;      // hipIpcOpenMemHandle results (the peers' windows)
    bool connected = false;
    hfem::PeerView view{};
};
namespace hfem {
extern int g_quad4_stagger, g_quad4_stagger_shift, g_quad4_stagger_groups;
extern int g_quad4_ablate, g_quad4_pipe, g_quad4_const_caps, g_quad4_bits;   // quad4.hip (lab option "quad4_ablate")

}  // namespace hfem


// Device-side view of the tile plan + the opaque plan object, shared by the TRI3 and QUAD4 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

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

struct LagSum {                      // HFEM_FLAG_SUM_PREVIOUS: one extra workgroup reduces the previous launch's tile energies
    const double *prev = nullptr;    // partials bank the previous launch wrote (offset to its first tile)
    int prev_n = 0;
    double *out = nullptr;           // receives their sum (same order and bits as sum_partials_kernel)
};


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
__device__ __forceinline__ void adam_fused_row(const AdamFuse &af, int c, int row, double gx, double gy, double2 p64,
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
int launch_iface_pack_sum(const double *x_free, const double *u_free, const int32_t *rows, int n_x, int n_u, double *out,
                          int64_t loss_slot, const double *partials, int n_partials, int64_t *counter, double beta1,
                          double beta2, double *bc_next, hipStream_t s);
// peer.hip: the interface payload written straight into every rank's receive window (no collective); layout in peer.hip
constexpr int kMaxPeers = 16;
constexpr size_t kPeerFlags = 0, kPeerData = 256;
struct PeerView {                    // kernel argument: every rank's window as mapped into THIS process
    char *win[kMaxPeers];
    char *ctl;                       // this rank's private control block (seq, ticket, status)
    int rank, world;
};
}  // namespace hfem
struct hfem_peer {
    int device = -1;
    int64_t stride = 0;              // double2 units per rank and slot: interface rows + the loss entry
    size_t bytes = 0;
    char *local = nullptr;           // this rank's window (hipExtMallocWithFlags, uncached)
    char *ctl = nullptr;             // seq / ticket / status (hipMalloc)
    std::vector<void *> opened;      // hipIpcOpenMemHandle results (the peers' windows)
    bool connected = false;
    hfem::PeerView view{};
};
namespace hfem {
int launch_iface_put(const hfem_peer *peer, const double *x_free, const double *u_free, const int32_t *rows, int n_x,
                     int n_u, int64_t loss_slot, const double *partials, int n_partials, int64_t *counter, double beta1,
                     double beta2, double *bc_next, hipStream_t s);
extern int g_quad4_stagger, g_quad4_stagger_shift, g_quad4_stagger_groups;
extern int g_quad4_ablate, g_quad4_pipe, g_quad4_const_caps;   // quad4.hip (lab option "quad4_ablate")

}  // namespace hfem


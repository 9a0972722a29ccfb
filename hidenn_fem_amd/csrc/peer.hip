// Peer-window interface exchange of the owner-sharded mode: the same payload as
// csrc/exchange.hip -- [x rows of my interface | u rows | padding][loss partial, 0] -- but WRITTEN BY THE PACK KERNEL
// ITSELF into a receive window on every rank (stores over xGMI into IPC-mapped memory of the peers) instead of an RCCL
// all_gather.  No collective, no second stream, no host in the loop: per step one `put` launch after the optimiser and one
// `get` launch before the tiles that read foreign rows; between the two the interior tiles run, and the flags have long
// arrived when `get` looks at them.  xGMI is point-to-point and the payload is a few KB: latency is what matters, and a
// remote store + flag is one link traversal, where a 16.5 KB all_gather is a whole RCCL kernel (launch + its own flags).
//
// Window of rank r (hipExtMallocWithFlags, uncached: remote writes are visible to local loads without cache games):
//     [0]    u64 flags[2][kMaxPeers]     flags[p][s] = (seq + 1) of the put of rank s whose data sits in data[p][s]
//     [256]  double2 data[2][world][stride]
// Rank-private control block (ordinary device memory -- every load of the uncached window is a trip to memory, and the put
// and the get are chains of dependent loads):
//     [0]    u64 seq      puts this rank has completed
//     [8]    u32 ticket   block counter of the running put
//     [12]   u32 status   sticky error bits (1: a get timed out waiting for a peer, 2: a boundary tile timed out waiting for
//                         the in-launch get)
//     [16]   u64 unpacked the put count whose rows the in-launch get has copied in     [24] u32 its block counter
// The in-launch get (hfem_peer_attach_get + hfem_plan_set_peer_get + HFEM_FLAG_PEER_GET; device code in hfem_plan_dev.h):
// the get runs as the first 8 workgroups of the NEXT energy launch instead of a launch of its own -- the rank's boundary tiles,
// the only ones that read rows another rank owns, wait for `unpacked` inside the kernel while every other tile runs.  One
// energy launch per step, no get launch, and the link latency is still hidden.
// Put k (seq == k) of rank s writes data[k & 1][s] of EVERY rank's window, fences at system scope, and its last workgroup
// stores k + 1 into flags[k & 1][s] of every window, then bumps its own seq.  Get k waits (bounded: timeout in 100 MHz
// ticks, then status |= 1 and it proceeds -- never a hang) until flags[k & 1][*] >= k + 1 in its OWN window, then unpacks.
// Two data slots are enough: a rank issues put k + 2 only after its get k + 1, which needs every peer's put k + 1, which
// each peer issues after its own get k -- so nobody still reads slot k & 1 when it is overwritten.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "hfem_plan_dev.h"

namespace hfem {

static int hip_fail(hipError_t e, const char *what) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return (int)e;
}

template <typename V>
__global__ __launch_bounds__(256) void iface_put_kernel(const V *__restrict__ x_free, const V *__restrict__ u_free,
                                                        const int32_t *__restrict__ rows, int n_x, int n_u, int64_t stride,
                                                        int64_t loss_slot, const double *__restrict__ partials,
                                                        int n_partials, int64_t *__restrict__ counter, double beta1,
                                                        double beta2, double *__restrict__ bc_next, PeerView pv) {
    __shared__ double red[4];
    __shared__ int last;
    char *ctl = pv.ctl;
    const uint64_t seq = __hip_atomic_load((uint64_t *)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int par = (int)(seq & 1);
    const size_t slot = kPeerData + ((size_t)par * pv.world + pv.rank) * (size_t)stride * sizeof(double2);
    const int i = blockIdx.x * 256 + threadIdx.x;
    // window stores: system-scope relaxed atomic stores (write-through, never parked in this XCD's L2) whose completion the
    // counted wait below observes -- the same ordering argument as the in-launch put (hfem_plan_dev.h, peer_put_finish), and no
    // __threadfence_system (= a write-back of the XCD's whole L2) in every workgroup
    auto put2 = [&](int p, int64_t at, double a0, double a1) {
        double *q = reinterpret_cast<double *>(pv.win[p] + slot) + 2 * (size_t)at;
        __hip_atomic_store(q, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(q + 1, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    if (i < n_x + n_u) {
        const double2 v = row_widen(i < n_x ? x_free[rows[i]] : u_free[rows[i]]);
        for (int p = 0; p < pv.world; ++p) put2(p, i, v.x, v.y);
    }
    if (blockIdx.x == 0) {          // the rank's energy: the same bits as sum_partials_kernel / iface_pack_sum_kernel
        double v = 0.0;
        for (int k = threadIdx.x; k < n_partials; k += 256) v += partials[k];
        const double tot = block_sum(v, red);
        if (threadIdx.x == 0) {
            for (int p = 0; p < pv.world; ++p) put2(p, loss_slot, tot, 0.0);
            if (counter) {
                const int64_t c = counter[0] + 1;
                counter[0] = c;
                if (bc_next) {
                    bc_next[0] = 1.0 - pow(beta1, (double)(c + 1));
                    bc_next[1] = sqrt(1.0 - pow(beta2, (double)(c + 1)));
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's window stores are acknowledged before the workgroup takes its ticket
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd((unsigned *)(ctl + 8), 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x < pv.world)
        __hip_atomic_store((uint64_t *)(pv.win[threadIdx.x] + kPeerFlags) + par * kMaxPeers + pv.rank, seq + 1,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
        *(unsigned *)(ctl + 8) = 0u;
        __hip_atomic_store((uint64_t *)ctl, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename V>
__global__ __launch_bounds__(256) void iface_get_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                        int n_x, int n_u, V *__restrict__ x_free,
                                                        V *__restrict__ u_free, int64_t stride, int64_t loss_slot,
                                                        double *__restrict__ loss_out, int64_t timeout_ticks, PeerView pv) {
    char *self = pv.win[pv.rank];
    const uint64_t want = __hip_atomic_load((uint64_t *)pv.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (want == 0) return;                                   // no put yet: nothing to copy in
    const int par = (int)((want - 1) & 1);
    unsigned *status = (unsigned *)(pv.ctl + 12);
    if ((int)threadIdx.x < pv.world) {
        const uint64_t *flag = (const uint64_t *)(self + kPeerFlags) + par * kMaxPeers + threadIdx.x;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) break;   // sticky: wait once
            if ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) > timeout_ticks) {
                atomicOr(status, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);                 // system scope: nothing of the window is served from a cache
    const double2 *recv = (const double2 *)(self + kPeerData + (size_t)par * pv.world * (size_t)stride * sizeof(double2));
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_x) x_free[dst[i]] = row_narrow<V>(recv[src[i]]);
    else if (i < n_x + n_u) u_free[dst[i]] = row_narrow<V>(recv[src[i]]);
    if (i == 0 && loss_out) {
        double tot = 0.0;
        for (int r = 0; r < pv.world; ++r) tot += recv[(int64_t)r * stride + loss_slot].x;   // rank order, as iface_unpack
        loss_out[0] = tot;
    }
}

int launch_iface_put(const hfem_peer *peer, int dtype, const void *x_free, const void *u_free, const int32_t *rows, int n_x,
                     int n_u, int64_t loss_slot, const double *partials, int n_partials, int64_t *counter, double beta1,
                     double beta2, double *bc_next, hipStream_t s) {
    const int n = n_x + n_u > 0 ? n_x + n_u : 1;
    if (dtype == 0)
        hipLaunchKernelGGL(iface_put_kernel<double2>, dim3((n + 255) / 256), dim3(256), 0, s, (const double2 *)x_free,
                           (const double2 *)u_free, rows, n_x, n_u, peer->stride, loss_slot, partials, n_partials, counter,
                           beta1, beta2, bc_next, peer->view);
    else
        hipLaunchKernelGGL(iface_put_kernel<float2>, dim3((n + 255) / 256), dim3(256), 0, s, (const float2 *)x_free,
                           (const float2 *)u_free, rows, n_x, n_u, peer->stride, loss_slot, partials, n_partials, counter,
                           beta1, beta2, bc_next, peer->view);
    return launch_status("hfem_plan_iface_put");
}

}  // namespace hfem

using namespace hfem;

extern "C" int hfem_peer_create(int device, int32_t rank, int32_t world, int64_t stride, hfem_peer **out) {
    HFEM_ARG_CHECK(out, "null pointer");
    HFEM_ARG_CHECK(world >= 1 && world <= kMaxPeers && rank >= 0 && rank < world, "rank / world out of range (at most 16 ranks)");
    HFEM_ARG_CHECK(stride >= 1, "stride must be >= 1 (double2 units: interface rows + the loss slot)");
    if (int rc = use_device(device)) return rc;
    hfem_peer *p = new hfem_peer();
    p->device = device;
    p->stride = stride;
    p->bytes = kPeerData + (size_t)2 * world * (size_t)stride * sizeof(double2);
    p->view.rank = rank;
    p->view.world = world;
    for (int i = 0; i < kMaxPeers; ++i) p->view.win[i] = nullptr;
    void *w = nullptr;
    hipError_t e = hipExtMallocWithFlags(&w, p->bytes, hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemset(w, 0, p->bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (w) (void)hipFree(w);
        delete p;
        return hip_fail(e, "hfem_peer_create");
    }
    void *c = nullptr;
    e = hipMalloc(&c, 64);
    if (e == hipSuccess) e = hipMemset(c, 0, 64);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (c) (void)hipFree(c);
        (void)hipFree(w);
        delete p;
        return hip_fail(e, "hfem_peer_create");
    }
    p->local = (char *)w;
    p->ctl = (char *)c;
    p->view.ctl = p->ctl;
    p->view.win[rank] = p->local;
    p->connected = world == 1;
    *out = p;
    return 0;
}

extern "C" int hfem_peer_ipc_handle(hfem_peer *peer, void *handle_out_64_bytes) {
    HFEM_ARG_CHECK(peer && handle_out_64_bytes, "null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (int rc = use_device(peer->device)) return rc;
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, peer->local);
    if (e != hipSuccess) return hip_fail(e, "hfem_peer_ipc_handle (hipIpcGetMemHandle; HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
    std::memcpy(handle_out_64_bytes, &h, 64);
    return 0;
}

extern "C" int hfem_peer_connect(hfem_peer *peer, const void *handles_world_x_64_bytes) {
    HFEM_ARG_CHECK(peer && handles_world_x_64_bytes, "null pointer");
    HFEM_ARG_CHECK(!peer->connected, "already connected");
    if (int rc = use_device(peer->device)) return rc;
    for (int r = 0; r < peer->view.world; ++r) {
        if (r == peer->view.rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const char *)handles_world_x_64_bytes + (size_t)r * 64, 64);
        void *w = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&w, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (void *o : peer->opened) (void)hipIpcCloseMemHandle(o);
            peer->opened.clear();
            return hip_fail(e, "hfem_peer_connect (hipIpcOpenMemHandle)");
        }
        peer->opened.push_back(w);
        peer->view.win[r] = (char *)w;
    }
    peer->connected = true;
    return 0;
}

// Arguments of the in-launch get (HFEM_FLAG_PEER_GET launches of a plan, hfem_plan_set_peer_get): the unpack tables and the
// loss slot hfem_peer_iface_get takes per call, kept in device memory so that a tile kernel needs one pointer.
extern "C" int hfem_peer_attach_get(hfem_peer *peer, const int32_t *src, const int32_t *dst, int32_t n_x, int32_t n_u,
                                    int64_t loss_slot, double *loss_out, int64_t timeout_ticks) {
    HFEM_ARG_CHECK(peer, "null pointer");
    HFEM_ARG_CHECK(peer->connected, "hfem_peer_connect has not been called");
    HFEM_ARG_CHECK(n_x >= 0 && n_u >= 0 && loss_slot >= 0 && loss_slot < peer->stride && timeout_ticks > 0, "bad sizes");
    HFEM_ARG_CHECK(n_x + n_u == 0 || (src && dst), "null pointer");
    if (int rc = use_device(peer->device)) return rc;
    PeerGetDev g;
    g.pv = peer->view; g.src = src; g.dst = dst; g.n_x = n_x; g.n_u = n_u;
    g.stride = peer->stride; g.loss_slot = loss_slot; g.timeout_ticks = timeout_ticks; g.loss_out = loss_out;
    hipError_t e = hipSuccess;
    if (!peer->get_dev) e = hipMalloc((void **)&peer->get_dev, sizeof(PeerGetDev));
    if (e == hipSuccess) e = hipMemcpy(peer->get_dev, &g, sizeof(PeerGetDev), hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail(e, "hfem_peer_attach_get");
    return 0;
}

// Arguments of the in-launch put (HFEM_FLAG_PEER_PUT launches of a plan, hfem_plan_set_peer_put): where every parameter row
// sits in this rank's payload lane (pos_x [rows of x_free], pos_u [rows of u_free]: double2 index, or -1 = not an interface
// row), the loss slot, and the optimiser's step counter + betas (the last boundary tile does what hfem_plan_iface_put's
// block 0 does).  Device arrays stay owned by the caller.
extern "C" int hfem_peer_attach_put(hfem_peer *peer, const int32_t *pos_x, const int32_t *pos_u, int64_t loss_slot,
                                    int64_t *counter, double beta1, double beta2) {
    HFEM_ARG_CHECK(peer && pos_x && pos_u, "null pointer");
    HFEM_ARG_CHECK(peer->connected, "hfem_peer_connect has not been called");
    HFEM_ARG_CHECK(loss_slot >= 0 && loss_slot < peer->stride, "bad loss slot");
    if (int rc = use_device(peer->device)) return rc;
    PeerPutDev g;
    g.pv = peer->view; g.pos_x = pos_x; g.pos_u = pos_u; g.stride = peer->stride; g.loss_slot = loss_slot;
    g.counter = counter; g.beta1 = beta1; g.beta2 = beta2;
    hipError_t e = hipSuccess;
    if (!peer->put_dev) e = hipMalloc((void **)&peer->put_dev, sizeof(PeerPutDev));
    if (e == hipSuccess) e = hipMemcpy(peer->put_dev, &g, sizeof(PeerPutDev), hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail(e, "hfem_peer_attach_put");
    peer->put_pos[0] = pos_x; peer->put_pos[1] = pos_u;
    return 0;
}

extern "C" int hfem_peer_destroy(hfem_peer *peer) {
    if (!peer) return 0;
    (void)use_device(peer->device);
    (void)hipDeviceSynchronize();
    for (void *o : peer->opened) (void)hipIpcCloseMemHandle(o);
    if (peer->local) (void)hipFree(peer->local);
    if (peer->ctl) (void)hipFree(peer->ctl);
    if (peer->get_dev) (void)hipFree(peer->get_dev);
    if (peer->put_dev) (void)hipFree(peer->put_dev);
    delete peer;
    return 0;
}

extern "C" int hfem_peer_status(hfem_peer *peer, int32_t *status_out, int64_t *puts_out) {
    HFEM_ARG_CHECK(peer, "null pointer");
    if (int rc = use_device(peer->device)) return rc;
    unsigned char head[16];
    hipError_t e = hipMemcpy(head, peer->ctl, 16, hipMemcpyDeviceToHost);        // synchronises with the device
    if (e != hipSuccess) return hip_fail(e, "hfem_peer_status");
    uint64_t seq;
    uint32_t st;
    std::memcpy(&seq, head, 8);
    std::memcpy(&st, head + 12, 4);
    if (status_out) *status_out = (int32_t)st;
    if (puts_out) *puts_out = (int64_t)seq;
    return 0;
}

static int peer_iface_get_any(int dtype, hfem_peer *peer, const int32_t *src, const int32_t *dst, int32_t n_x, int32_t n_u,
                              void *x_free, void *u_free, int64_t loss_slot, double *loss_out, int64_t timeout_ticks,
                              void *stream) {
    HFEM_ARG_CHECK(peer, "null pointer");
    HFEM_ARG_CHECK(peer->connected, "hfem_peer_connect has not been called");
    HFEM_ARG_CHECK(n_x >= 0 && n_u >= 0 && loss_slot >= 0 && loss_slot < peer->stride && timeout_ticks > 0, "bad sizes");
    HFEM_ARG_CHECK((n_x + n_u == 0 || (src && dst)) && (n_x == 0 || x_free) && (n_u == 0 || u_free), "null pointer");
    if (int rc = use_device(peer->device)) return rc;
    const int n = n_x + n_u > 0 ? n_x + n_u : 1;
    if (dtype == 0)
        hipLaunchKernelGGL(iface_get_kernel<double2>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, n_x, n_u,
                           (double2 *)x_free, (double2 *)u_free, peer->stride, loss_slot, loss_out, timeout_ticks, peer->view);
    else
        hipLaunchKernelGGL(iface_get_kernel<float2>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, n_x, n_u,
                           (float2 *)x_free, (float2 *)u_free, peer->stride, loss_slot, loss_out, timeout_ticks, peer->view);
    return launch_status("hfem_peer_iface_get");
}

extern "C" int hfem_peer_iface_get(hfem_peer *peer, const int32_t *src, const int32_t *dst, int32_t n_x, int32_t n_u,
                                   double *x_free, double *u_free, int64_t loss_slot, double *loss_out,
                                   int64_t timeout_ticks, void *stream) {
    return peer_iface_get_any(0, peer, src, dst, n_x, n_u, x_free, u_free, loss_slot, loss_out, timeout_ticks, stream);
}

extern "C" int hfem_peer_iface_get_f32(hfem_peer *peer, const int32_t *src, const int32_t *dst, int32_t n_x, int32_t n_u,
                                       float *x_free, float *u_free, int64_t loss_slot, double *loss_out,
                                       int64_t timeout_ticks, void *stream) {
    return peer_iface_get_any(1, peer, src, dst, n_x, n_u, x_free, u_free, loss_slot, loss_out, timeout_ticks, stream);
}

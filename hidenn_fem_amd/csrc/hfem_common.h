// Shared host-side helpers for libhidenn_hip.so: error reporting + plan layout.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/hidenn_fem.h"

namespace hfem {

// thread-local last-error string (hfem_last_error)
void set_error(const std::string &msg);
const char *get_error();

#define HFEM_ARG_CHECK(cond, msg)                                            \
    do {                                                                     \
        if (!(cond)) {                                                       \
            ::hfem::set_error(std::string(__func__) + ": " + (msg));         \
            return -1;                                                       \
        }                                                                    \
    } while (0)

// ---- tile plan layout -------------------------------------------------------
// Local node indices are 10 bits: a tile references at most 1024 nodes.
constexpr int kLocalBits = 10;
constexpr int kMaxLocal = 1 << kLocalBits;
constexpr uint32_t kLocalMask = kMaxLocal - 1;
constexpr uint32_t kHomeBit = 1u << 30;   // element/edge energy is counted by this tile
constexpr uint32_t kSkipBit = 1u << 31;   // padding record: the lane has no element
constexpr int32_t kNodeTailPad = 2048;    // records after the last tile's stride (unguarded loads of up to NPT x BLOCK lanes from a tile's start)
constexpr int32_t kElemTailPad = 4096;    // slot records after the last tile's stride (EPT x BLOCK lanes)
constexpr int32_t kMaxQuadSlots = 1024;   // element slots per tile the tiled QUAD4 kernel can hold in registers

// tile_desc[t] = 8 x int32
struct TileDesc {
    int32_t elem_off, n_elem;     // into elem_pack / elem_gid
    int32_t node_off, n_node;     // into node_src; owned nodes first
    int32_t n_owned;
    int32_t edge_off, n_edge;     // into edge_pack / edge_gid
    int32_t pad;                  // paired plans: column stride of the tile's slot array (slot j of thread t at j*stride + t)
};
static_assert(sizeof(TileDesc) == 32, "TileDesc must be 8 x int32");

struct HostPlan {
    int64_t ne = 0, nn = 0, ned = 0;
    int32_t tile_elems = 0;
    int32_t npe = 3;                   // nodes per element: 3 (TRI3) or 4 (QUAD4 extension)
    std::vector<TileDesc> tiles;
    std::vector<uint32_t> elem_pack;   // l0 | l1<<10 | l2<<20 | home<<30 | skip<<31
    std::vector<uint32_t> elem_pack_hi;  // QUAD4 only: l3 of the same slot
    std::vector<int32_t> elem_gid;     // global element id (tests / debugging)
    std::vector<int32_t> node_src;     // [.][2] = {x_src, u_src} of each local node; tile t's records start at t * node_stride,
                                       // padded to the stride with its last record, kNodeTailPad more after the last tile
    int32_t node_stride = 0, elem_stride = 0;   // elem_pack / elem_pack_hi / elem_gid(_b): tile t's slots start at t * elem_stride (skip-padded)
    int64_t elem_records = 0;          // real slots (padding not counted)
    int32_t col_stride = 0;            // paired plans: columns of every tile's slot array (= tile_desc.pad of every tile)
    int64_t node_records = 0;          // sum of n_node (the padding not counted)
    std::vector<uint32_t> edge_pack;   // li | lj<<10 | home<<30
    std::vector<int32_t> edge_gid;     // global edge id (row of the traction table)
    int32_t max_nodes = 0, max_owned = 0, max_elems = 0, max_edges = 0;
    // Chunked element order (elem_order 4; the streamed kernel's contract, tri3_stream.hip): a tile's slots are
    // kChunks spatial strips [0,e1) [e1,e2) [e2,n_elem), each a whole number of 16-lane groups; owned and halo
    // local ids are each sorted by the first strip that touches them, so a strip's nodes are a prefix of both
    // id ranges.  tile_chunks[t] = {e1, e2, po0 | po1<<8 | ph0<<16 | ph1<<24, 0}: po_k / ph_k = number of
    // 64-id pieces of the owned / halo id range that strips 0..k need.  Empty for the other element orders.
    std::vector<int32_t> tile_chunks;
    // Paired element order (elem_order 5; contract of tri3_pair.hip): a slot holds element A = (n, b, c) and, when
    // `hasB`, a second element B = (n, c, d) that shares A's corner-0 node and A's corner 2 as its corner 1 (two
    // fan-adjacent elements around n -- e.g. the two triangles of a split quad).  The shared nodes' contributions are
    // added in registers: 16 LDS atomics per pair instead of 24, 8 node reads instead of 12.  Records:
    //   elem_pack[s]    = l_n | l_b<<10 | l_c<<20 | homeA<<30 | skip<<31
    //   elem_pack_hi[s] = l_d | hasB<<10 | homeB<<11
    // elem_gid[s] = A's element id, elem_gid_b[s] = B's (or -1).  Elements without a partner are slots with hasB = 0.
    std::vector<int32_t> elem_gid_b;
    // Strip order (elem_order 6): as 5, plus CHAINED slots: bit 12 of elem_pack_hi[s] says that the slot one column-stride
    // further (same thread, next row; tile_desc.pad = stride) holds the pair with n' = b and d' = c; the thread carries this
    // slot's rows of b and c in registers into that slot's rows of n and d instead of adding them to LDS.
    bool paired = false;
    int64_t n_pairs = 0, n_chained = 0;
    int32_t max_rows = 0;              // paired plans: slots per thread (rows of the widest tile's slot array)
    // compact copies of the inputs (the deterministic node-centric kernel walks the mesh itself, tri3_det.hip)
    std::vector<int32_t> conn32, x_src_g, u_src_g, edges32;
    int32_t max_chunk_elems = 0;       // longest strip (slots); 0 = not chunked
    // Element sharding (plan_shards): rank r's tiles are [lo, hi) = shard_desc[4r + 0], [4r + 2]); its BOUNDARY tiles (they
    // read a node another rank's tile owns, or own a node another rank's tile reads) are [lo, mid), mid = shard_desc[4r + 1],
    // its interior tiles [mid, hi).  shards = 1: one record {0, 0, n_tiles, 0}.
    int32_t shards = 1;
    std::vector<int32_t> shard_desc;
    int32_t pair_block = 256;          // paired plans: threads per tile = columns of a tile's slot array (256 or 512)
    std::vector<int32_t> owned_gid;    // global node id of every tile's owned nodes, (tile, local) order: a permutation of the nodes
    std::vector<int32_t> owned_gid_by_slot;   // build scratch: node id of every compact node record
};
constexpr int kChunks = 3;

// Build the owner-computes tiling.  Returns 0 or -1 (message via set_error).
int build_host_plan(const int64_t *conn, int npe, int64_t ne, int64_t nn, const double *coords,
                    const int32_t *x_src, const int32_t *u_src, const int64_t *edges,
                    int64_t ned, int32_t tile_elems, int32_t node_cap, int elem_order, int32_t chunk_cap,
                    HostPlan &out, int32_t pair_block = 256, int32_t shards = 1);

// Plan blobs (hfem_plan_serialize / hfem_plan_deserialize): `trailer` = the launch options the plan captured (opaque here)
void serialize_host_plan(const HostPlan &h, const void *trailer, size_t trailer_bytes, std::vector<unsigned char> &blob);
int deserialize_host_plan(const void *blob, size_t n, HostPlan &h, void *trailer, size_t trailer_bytes);

void set_plan_curve(int c);   // 0 Morton, 1 Hilbert (default)
void set_plan_read_pack(int v);     // paired slots packed against ds_read_b128 bank conflicts as well (default 1)
void set_plan_snap(int percent);   // tile cuts snap back to coarse curve-cell boundaries by up to this share of a tile (0 off)

}  // namespace hfem

// Host-only sanitizer harness for the tile planner ("-fsanitize=address,undefined build of the
// C-ABI layer's CPU side").  Built by tests/test_plan_host.py with
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all plan_san_main.cpp ../../hidenn_fem_amd/csrc/plan.cpp
// (plan.cpp contains no HIP call) and run on mesh files the test writes.  File format (little endian):
//   int64 ne, nn, ned, npe, tile_elems, node_cap, elem_order, chunk_cap, has_maps
//   int64 conn[ne*npe]; double xy[nn*2]; int64 edges[ned*2]; int32 x_src[nn], u_src[nn] (if has_maps)
// Exit code 0 = plan built and self-checked; any ASan / UBSan finding aborts with a non-zero code.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../hidenn_fem_amd/csrc/hfem_common.h"

using namespace hfem;

template <typename T>
static bool rd(FILE *f, std::vector<T> &v, size_t n) {
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: plan_san <mesh.bin> ...\n"); return 2; }
    for (int a = 1; a < argc; ++a) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) { perror(argv[a]); return 2; }
        int64_t hdr[9];
        if (fread(hdr, sizeof(int64_t), 9, f) != 9) return 2;
        const int64_t ne = hdr[0], nn = hdr[1], ned = hdr[2], npe = hdr[3];
        std::vector<int64_t> conn, edges;
        std::vector<double> xy;
        std::vector<int32_t> xs, us;
        if (!rd(f, conn, (size_t)(ne * npe)) || !rd(f, xy, (size_t)(nn * 2)) || !rd(f, edges, (size_t)(ned * 2))) return 2;
        if (hdr[8] && (!rd(f, xs, (size_t)nn) || !rd(f, us, (size_t)nn))) return 2;
        fclose(f);
        HostPlan P;
        const int rc = build_host_plan(conn.data(), (int)npe, ne, nn, xy.empty() ? nullptr : xy.data(),
                                       hdr[8] ? xs.data() : nullptr, hdr[8] ? us.data() : nullptr,
                                       edges.empty() ? nullptr : edges.data(), ned, (int32_t)hdr[4], (int32_t)hdr[5],
                                       (int)hdr[6], (int32_t)hdr[7], P);
        if (rc != 0) { fprintf(stderr, "%s: build_host_plan failed: %s\n", argv[a], get_error()); return 1; }
        // light self-check: every element is home exactly once, every node owned exactly once
        std::vector<int> home(ne, 0), owned(nn, 0);
        for (const TileDesc &d : P.tiles) {
            if (d.n_node > kMaxLocal || d.n_owned > d.n_node) { fprintf(stderr, "bad tile\n"); return 1; }
            for (int i = 0; i < d.n_elem; ++i) {
                const uint32_t pk = P.elem_pack[d.elem_off + i];
                if (pk & kSkipBit) continue;
                for (int k = 0; k < 3; ++k)
                    if ((int)((pk >> (kLocalBits * k)) & kLocalMask) >= d.n_node) { fprintf(stderr, "local id out of range\n"); return 1; }
                if (pk & kHomeBit) home[P.elem_gid[d.elem_off + i]]++;
                if (P.paired) {
                    const uint32_t hi = P.elem_pack_hi[d.elem_off + i];
                    if ((hi >> 10) & 1u) {
                        if ((int)(hi & kLocalMask) >= d.n_node) { fprintf(stderr, "local id out of range (B)\n"); return 1; }
                        if ((hi >> 11) & 1u) home[P.elem_gid_b[d.elem_off + i]]++;
                    }
                }
            }
            for (int l = 0; l < d.n_owned; ++l) {
                const int32_t src = P.node_src[2 * (size_t)(d.node_off + l)];
                (void)src;
            }
        }
        for (int64_t e = 0; e < ne; ++e)
            if (home[e] != 1) { fprintf(stderr, "element %lld counted %d times\n", (long long)e, home[e]); return 1; }
        // plan blobs (hfem_plan_serialize / hfem_plan_deserialize): the round trip reproduces every array, and a damaged blob --
        // truncated at any of 64 cut points, a flipped byte in the header / counts / payload, counts that point outside the
        // blob (the checksum is recomputed so that the parser itself is reached) -- is rejected without touching memory it
        // does not own (ASan / UBSan watch)
        {
            const double trailer[3] = {1.0, 2.0, 3.0};
            std::vector<unsigned char> blob;
            serialize_host_plan(P, trailer, sizeof(trailer), blob);
            HostPlan Q;
            double back[3] = {0, 0, 0};
            if (deserialize_host_plan(blob.data(), blob.size(), Q, back, sizeof(back)) != 0) { fprintf(stderr, "blob round trip failed: %s\n", get_error()); return 1; }
            if (Q.elem_pack != P.elem_pack || Q.node_src != P.node_src || Q.elem_gid != P.elem_gid || Q.shard_desc != P.shard_desc ||
                Q.owned_gid != P.owned_gid || Q.conn32 != P.conn32 || Q.tiles.size() != P.tiles.size() || back[2] != 3.0 ||
                Q.paired != P.paired || Q.max_nodes != P.max_nodes || Q.elem_stride != P.elem_stride) { fprintf(stderr, "blob round trip differs\n"); return 1; }
            auto fnv = [](const unsigned char *p_, size_t n) {
                uint64_t h = 1469598103934665603ull;
                size_t i = 0;
                for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p_ + i, 8); h = (h ^ w) * 1099511628211ull; }
                for (; i < n; ++i) h = (h ^ p_[i]) * 1099511628211ull;
                return h;
            };
            int rejected = 0, tried = 0;
            for (int c = 0; c < 64; ++c) {                       // truncations
                const size_t cut = blob.size() * (size_t)c / 64;
                HostPlan R;
                ++tried;
                if (deserialize_host_plan(blob.data(), cut, R, back, sizeof(back)) != 0) ++rejected;
            }
            for (int c = 0; c < 96; ++c) {                       // a flipped byte; every other case with the checksum repaired
                std::vector<unsigned char> bad = blob;
                const size_t at = c < 48 ? (size_t)c * 4 % std::min<size_t>(bad.size() - 8, 400) : (bad.size() - 8) * (size_t)(c - 47) / 50;
                bad[at] ^= (unsigned char)(1u << (c % 8));
                if (c % 2) { const uint64_t h = fnv(bad.data(), bad.size() - 8); memcpy(bad.data() + bad.size() - 8, &h, 8); }
                HostPlan R;
                ++tried;
                const int rc2 = deserialize_host_plan(bad.data(), bad.size(), R, back, sizeof(back));
                if (rc2 != 0) ++rejected;
                else {                                           // accepted (a payload byte with the checksum repaired): must still be a
                    for (const TileDesc &d : R.tiles)            // structurally safe plan -- what a kernel indexes stays inside the arrays
                        if ((size_t)d.node_off + d.n_node > R.node_src.size() / 2 || (size_t)d.elem_off + d.n_elem > R.elem_pack.size()) {
                            fprintf(stderr, "accepted blob with a descriptor outside its arrays\n");
                            return 1;
                        }
                }
            }
            if (rejected < 64 + 48) { fprintf(stderr, "only %d of %d damaged blobs rejected\n", rejected, tried); return 1; }
        }
        printf("%s: ok tiles=%zu slots=%zu nodes=%zu max_nodes=%d max_elems=%d chunked=%d\n", argv[a], P.tiles.size(),
               P.elem_pack.size(), P.node_src.size() / 2, P.max_nodes, P.max_elems, P.max_chunk_elems);
    }
    return 0;
}

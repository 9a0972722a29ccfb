// Host-only sanitizer harness for the tile planner ("-fsanitize=address,undefined build of the
// C-ABI layer's CPU side").  Built by tests/test_plan_host.py with
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all plan_san_main.cpp ../../hidenn_fem_amd/csrc/plan.cpp
// (plan.cpp contains no HIP call) and run on mesh files the test writes.  File format (little endian):
//   int64 ne, nn, ned, npe, tile_elems, node_cap, elem_order, chunk_cap, has_maps
//   int64 conn[ne*npe]; double xy[nn*2]; int64 edges[ned*2]; int32 x_src[nn], u_src[nn] (if has_maps)
// Exit code 0 = plan built and self-checked; any ASan / UBSan finding aborts with a non-zero code.
#include <cstdio>
#include <cstdlib>
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
        printf("%s: ok tiles=%zu slots=%zu nodes=%zu max_nodes=%d max_elems=%d chunked=%d\n", argv[a], P.tiles.size(),
               P.elem_pack.size(), P.node_src.size() / 2, P.max_nodes, P.max_elems, P.max_chunk_elems);
    }
    return 0;
}

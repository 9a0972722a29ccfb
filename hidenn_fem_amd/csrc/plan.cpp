// Host-side owner-computes tiling of a TRI3 mesh (no HIP calls in this file).
//
// Input is the reference's mesh contract (connectivity [Ne][3] int64,
// neumann_edges [E][2] int64; /root/reference/src/mesh.py:261-276) plus the
// free/fixed row maps that replace the bool-mask assembly of
// /root/reference/src/models.py:292-305.  Output is the tile plan the fused
// energy kernel walks (layout: hfem_common.h, DESIGN.md "Data layout").
//
//  1. elements are sorted along a Hilbert curve of their centroids (Morton: plan_curve 0) and cut
//     into tiles of consecutive elements (the tile's HOME elements), sized by element count or by
//     an owned-node cap;
//  2. a node is OWNED by the lowest-numbered tile among its adjacent elements'
//     home tiles (nodes without elements go to extra element-less tiles);
//  3. a tile evaluates home elements + HALO elements (other tiles' elements that
//     touch one of its owned nodes): every contribution to an owned node is
//     produced inside the owning tile, so gradient rows need no atomics;
//  4. a Neumann edge is evaluated by the owner tiles of its two nodes; its work
//     is counted by the owner of its first node.
//
//  5. inside a tile the element records are laid out for the kernel that walks them: bank-aware
//     16-lane groups (order 3), or PAIRED slots (order 5): A = (n,b,c) + B = (n,c,d), two elements
//     that share the directed edge (first node, last node of A) = (first, second node of B).
//
// The local order of an element's three nodes is never changed (the reference
// energy depends on it): pairing only chooses WHICH elements share a slot.
#include <algorithm>
#include <array>
#include <atomic>
#include <unordered_map>
#include <cstring>
#include <limits>
#include <numeric>

#include "hfem_common.h"

namespace hfem {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const char *get_error() { return g_err.c_str(); }

namespace {

inline uint32_t spread16(uint32_t v) {   // 16 bits -> every other bit of 32
    v &= 0xFFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// Position of cell (x, y) along a Hilbert curve of side 2^16.  Consecutive positions are always
// edge-adjacent cells, so a run of the sorted element list is one connected, compact patch (Morton
// runs can fall apart into several pieces): smaller tile perimeter -> fewer halo elements/nodes.
inline uint32_t hilbert16(uint32_t x, uint32_t y) {
    uint32_t d = 0;
    for (uint32_t s = 1u << 15; s > 0; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {                       // rotate the quadrant
            if (rx == 1) { x = 65535u - x; y = 65535u - y; }
            const uint32_t t = x; x = y; y = t;
        }
    }
    return d;
}

static std::atomic<int> g_curve{1};   // 0 Morton, 1 Hilbert (process-wide defaults below: atomics, read once per plan build)
void set_locality_curve(int c) { g_curve = c; }
// Tile cuts snap to coarse cells of the locality curve: when the node cap ends a tile, the cut moves back (by at most
// g_snap percent of the tile) to the boundary between the two consecutive elements whose curve codes differ in the highest
// bit -- the edge of the coarsest curve cell in reach.  Tiles become unions of whole cells (straight, axis-aligned sides)
// instead of ending mid-cell on a staircase: fewer halo elements and nodes per tile.  0 = off.
static std::atomic<int> g_hw_window{2};       // rows examined as partner of a row (pack_slot_halfwaves); > 100: exhaustive split search (lab)
static std::atomic<int> g_halfwave_all{1};    // also for one-element-per-slot TRI3 / QUAD4 records (plan_read_pack >= 1000: paired plans only)
static std::atomic<int> g_halfwave_pack{1};   // paired slots packed for ds_read_b128's lane groups too (pack_slot_halfwaves); 0: atomics only
void set_halfwave_pack(int v) {
    g_halfwave_all = v < 1000;
    if (v >= 1000) v -= 1000;
    g_halfwave_pack = v ? 1 : 0;
    if (v > 1) g_hw_window = v;
}
static std::atomic<int> g_snap{0};
void set_tile_snap(int percent) { g_snap = percent < 0 ? 0 : (percent > 50 ? 50 : percent); }
static thread_local std::vector<uint32_t> g_codes;   // curve code of every element, in sorted order (empty: no coordinates)

void morton_order(const int64_t *conn, int npe, int64_t ne, int64_t nn, const double *xy,
                  std::vector<int32_t> &order) {
    order.resize(ne);
    g_codes.clear();
    if (!xy || ne == 0) {
        std::iota(order.begin(), order.end(), 0);
        return;
    }
    double lo[2] = {std::numeric_limits<double>::max(), std::numeric_limits<double>::max()};
    double hi[2] = {-lo[0], -lo[1]};
    for (int64_t n = 0; n < nn; ++n)
        for (int a = 0; a < 2; ++a) {
            const double v = xy[2 * n + a];
            if (v == v) { lo[a] = std::min(lo[a], v); hi[a] = std::max(hi[a], v); }
        }
    // one isotropic scale: keeps Morton cells square on non-square domains
    const double span = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), 1e-300);
    const double scale = 65535.0 / span;
    std::vector<uint64_t> keys(ne);
    for (int64_t e = 0; e < ne; ++e) {
        double c[2] = {0, 0};
        for (int k = 0; k < npe; ++k) {
            const int64_t n = conn[npe * e + k];
            c[0] += xy[2 * n];
            c[1] += xy[2 * n + 1];
        }
        uint32_t q[2];
        for (int a = 0; a < 2; ++a) {
            double v = (c[a] / npe - lo[a]) * scale;
            if (!(v > 0)) v = 0;
            if (v > 65535.0) v = 65535.0;
            q[a] = (uint32_t)v;
        }
        const uint64_t code = g_curve ? hilbert16(q[0], q[1]) : (spread16(q[0]) | (spread16(q[1]) << 1));
        keys[e] = (code << 32) | (uint64_t)(uint32_t)e;   // ties: element id (stable)
    }
    std::sort(keys.begin(), keys.end());
    g_codes.resize(ne);
    for (int64_t p = 0; p < ne; ++p) { order[p] = (int32_t)(keys[p] & 0xFFFFFFFFu); g_codes[p] = (uint32_t)(keys[p] >> 32); }
}

constexpr int kOrphanTileNodes = 512;

// LDS-bank-aware packing of one element list into groups of 16 (order_tile_elements mode 3, below): appends the
// packed list to `out`; holes are -1.  pad_last: also pad the final partial group (the list then ends on a
// 16-slot boundary).
void pack_bank_groups(const std::vector<int32_t> &elems, const int64_t *conn, int npe, const std::vector<int32_t> &lid,
                      int32_t n_owned, bool pad_last, bool fill_tail, std::vector<int32_t> &out) {
    constexpr int G = 16;
    struct Open { std::vector<int32_t> el; uint16_t used[4]; };
    // most-constrained first: elements with 3 owned corners, then 2, 1, 0 (stable: curve order kept
    // inside a class).  The flexible ones (halo elements, few owned corners) then fill the holes the
    // constrained ones leave; with no window limit this brings the padding from ~8 % to ~2 %.
    std::vector<int32_t> byc[5];
    for (int32_t e : elems) {
        int owned = 0;
        for (int k = 0; k < npe; ++k) owned += lid[conn[npe * (int64_t)e + k]] < n_owned;
        byc[npe - owned].push_back(e);
    }
    std::vector<Open> open;
    size_t first_open = 0;                                   // groups before this index are full
    for (int cls = 0; cls <= npe; ++cls)
        for (int32_t e : byc[cls]) {
            uint16_t bit[4] = {0, 0, 0, 0};
            for (int k = 0; k < npe; ++k) {
                const int32_t l = lid[conn[npe * (int64_t)e + k]];
                bit[k] = l < n_owned ? (uint16_t)(1u << (l & 15)) : 0;
            }
            bool placed = false;
            for (size_t j = first_open; j < open.size(); ++j) {
                Open &g = open[j];
                if ((int)g.el.size() >= G) continue;
                if ((g.used[0] & bit[0]) | (g.used[1] & bit[1]) | (g.used[2] & bit[2]) | (g.used[3] & bit[3])) continue;
                g.el.push_back(e);
                for (int k = 0; k < 4; ++k) g.used[k] |= bit[k];
                placed = true;
                break;
            }
            if (!placed) {
                Open g;
                g.el.push_back(e);
                for (int k = 0; k < 4; ++k) g.used[k] = bit[k];
                open.push_back(std::move(g));
            }
            while (first_open < open.size() && (int)open[first_open].el.size() >= G) ++first_open;
        }
    // emit: full groups first, then partial ones fullest-first.  fill_tail: the elements of the emptiest partial
    // groups are moved into the holes of the fuller ones even where that costs a bank conflict (one extra LDS
    // cycle on that group) -- cheaper than idle lanes; only the very last group can then be short.
    std::stable_sort(open.begin(), open.end(), [](const Open &a, const Open &b) { return a.el.size() > b.el.size(); });
    if (fill_tail) {
        size_t lo = 0, hi = open.size();
        while (lo < hi && (int)open[lo].el.size() >= G) ++lo;
        while (lo + 1 < hi) {
            Open &dst = open[lo], &srcg = open[hi - 1];
            while ((int)dst.el.size() < G && !srcg.el.empty()) { dst.el.push_back(srcg.el.back()); srcg.el.pop_back(); }
            if (srcg.el.empty()) --hi;
            if ((int)dst.el.size() >= G) ++lo;
        }
        open.resize(hi);
    }
    for (size_t j = 0; j < open.size(); ++j) {
        out.insert(out.end(), open[j].el.begin(), open[j].el.end());
        if ((pad_last || j + 1 < open.size()) && (int)open[j].el.size() < G) out.insert(out.end(), G - open[j].el.size(), -1);
    }
}

// Same packing for abstract slots with up to four local ids each (-1: none): returns the slot order (indices into
// `items`, -1 = padding).  Used by the paired element order, whose slots carry the node ids (n, b, c, d).
void pack_slot_groups(const std::vector<std::array<int32_t, 4>> &items, int32_t n_owned, std::vector<int32_t> &out) {
    constexpr int G = 16;
    struct Open { std::vector<int32_t> el; uint16_t used[4]; };
    std::vector<int32_t> byc[5];
    for (int32_t i = 0; i < (int32_t)items.size(); ++i) {
        int owned = 0, have = 0;
        for (int k = 0; k < 4; ++k) { have += items[i][k] >= 0; owned += items[i][k] >= 0 && items[i][k] < n_owned; }
        (void)have;
        byc[4 - owned].push_back(i);
    }
    std::vector<Open> open;
    size_t first_open = 0;
    for (int cls = 0; cls <= 4; ++cls)
        for (int32_t i : byc[cls]) {
            uint16_t bit[4];
            for (int k = 0; k < 4; ++k) {
                const int32_t l = items[i][k];
                bit[k] = (l >= 0 && l < n_owned) ? (uint16_t)(1u << (l & 15)) : 0;
            }
            bool placed = false;
            for (size_t j = first_open; j < open.size(); ++j) {
                Open &g = open[j];
                if ((int)g.el.size() >= G) continue;
                if ((g.used[0] & bit[0]) | (g.used[1] & bit[1]) | (g.used[2] & bit[2]) | (g.used[3] & bit[3])) continue;
                g.el.push_back(i);
                for (int k = 0; k < 4; ++k) g.used[k] |= bit[k];
                placed = true;
                break;
            }
            if (!placed) {
                Open g;
                g.el.push_back(i);
                for (int k = 0; k < 4; ++k) g.used[k] = bit[k];
                open.push_back(std::move(g));
            }
            while (first_open < open.size() && (int)open[first_open].el.size() >= G) ++first_open;
        }
    // fullest first; the emptiest groups are dissolved into the holes of the fuller ones even where that costs a bank
    // conflict (one extra LDS cycle on that group) -- cheaper than idle lanes.  Only the last group can be short.
    std::stable_sort(open.begin(), open.end(), [](const Open &a, const Open &b) { return a.el.size() > b.el.size(); });
    {
        size_t lo = 0, hi = open.size();
        while (lo < hi && (int)open[lo].el.size() >= G) ++lo;
        while (lo + 1 < hi) {
            Open &dst = open[lo], &srcg = open[hi - 1];
            while ((int)dst.el.size() < G && !srcg.el.empty()) { dst.el.push_back(srcg.el.back()); srcg.el.pop_back(); }
            if (srcg.el.empty()) --hi;
            if ((int)dst.el.size() >= G) ++lo;
        }
        open.resize(hi);
    }
    for (size_t j = 0; j < open.size(); ++j) out.insert(out.end(), open[j].el.begin(), open[j].el.end());
}

// Packing of paired slots for BOTH LDS instruction kinds of the pair kernel (round 2, PMC: SQ_LDS_BANK_CONFLICT was 15 % of the
// LDS-active cycles, nearly all of it ds_read_b128).  A wave's 64 lanes are serviced
//   * by ds_add_f64 in four groups of 16 CONSECUTIVE lanes (8-byte slots: conflict-free iff the owned ids are distinct mod 16),
//   * by ds_read_b128 in four groups of 16 lanes that are NOT consecutive -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same
//     + 32 (MI355X_MICROARCH.md, LDS) -- over sixteen 16-byte slots of a 256-byte row: conflict-free iff the DISTINCT ids of a
//     group are distinct mod 16 (equal ids broadcast), owned or not.
// So a half-wave (32 lanes) is four cells of 8 lanes, cell (a, r) = atomic group a x read group r:
//   (0,0) = {0-3,12-15}   (0,1) = {4-11}   (1,0) = {20-27}   (1,1) = {16-19,28-31}
// and a slot placed in cell (a, r) must not clash, at any of its four node positions, with the owned ids of row a nor with
// the ids of column r.  Greedy first fit over a window of open half-waves, most-constrained slots first; what cannot be
// placed cleanly fills the holes where it clashes least.  Returns the slot order (indices into items, -1 = hole).
void pack_slot_halfwaves(const std::vector<std::array<int32_t, 4>> &items, int32_t n_owned, std::vector<int32_t> &out) {
    static const int kCellLane[4][8] = {{0, 1, 2, 3, 12, 13, 14, 15}, {4, 5, 6, 7, 8, 9, 10, 11},
                                        {20, 21, 22, 23, 24, 25, 26, 27}, {16, 17, 18, 19, 28, 29, 30, 31}};
    const int hw_window = g_hw_window.load();
    const int kPartnerWindow = hw_window > 100 ? hw_window - 100 : hw_window;
    const bool kFull = hw_window > 100;                 // exhaustive split search + local search (slow: lab comparison only)
    // step 1: atomic rows -- groups of 16 slots whose owned ids are distinct mod 16 at every position (pack_slot_groups)
    std::vector<int32_t> o;
    pack_slot_groups(items, n_owned, o);
    const int nrow = (int)((o.size() + 15) / 16);
    auto row_slot = [&](int r, int q) { const size_t i = (size_t)r * 16 + q; return i < o.size() ? o[i] : -1; };
    // read clashes of one column (two cells of 8): distinct ids sharing a 16-byte slot, per node position
    auto col_clashes = [&](const int32_t *c0, const int32_t *c1) {
        int tot = 0;
        for (int p = 0; p < 4; ++p) {
            int32_t held[16];
            std::fill(held, held + 16, -1);
            int cnt[16] = {0};
            for (int h = 0; h < 2; ++h)
                for (int q = 0; q < 8; ++q) {
                    const int32_t sidx = (h ? c1 : c0)[q];
                    if (sidx < 0) continue;
                    const int32_t id = items[sidx][p];
                    if (id < 0) continue;
                    const int r = id & 15;
                    if (held[r] == id) continue;                 // same address: broadcast
                    if (held[r] < 0) held[r] = id;
                    ++cnt[r];
                }
            int mx = 1;
            for (int r = 0; r < 16; ++r) mx = std::max(mx, cnt[r]);
            tot += mx - 1;                                       // extra LDS cycles of this group-instruction
        }
        return tot;
    };
    // step 2: rows in pairs; every row is split into the half that shares a read group with the partner's first half and the
    // half that shares one with its second half.  The split is by the residue of the slot's first node relative to a cyclic
    // interval [k, k + 8): on a structured patch the other three positions are shifted copies, so complementary intervals in the
    // two rows are complementary at all four positions when the rows come from mesh columns of equal width; k and the
    // partner's offset are chosen by the measured clash count.
    std::vector<int> rows(nrow);
    std::iota(rows.begin(), rows.end(), 0);
    auto res0 = [&](int32_t sidx) { return sidx >= 0 && items[sidx][0] >= 0 ? (items[sidx][0] & 15) : 0; };
    // best split of the row pair (ra, rb): returns the clash count, fills cell[4][8]
    auto split_pair = [&](int ra, int rb, int32_t (&best_cell)[4][8]) {
        int32_t A0[16], A1[16];
        for (int q = 0; q < 16; ++q) { A0[q] = row_slot(ra, q); A1[q] = rb >= 0 ? row_slot(rb, q) : -1; }
        int best = 1 << 30;
        for (int k = 0; k < 16 && best > 0; ++k)
            for (int dk = 0; dk < (kFull ? 16 : 1) && best > 0; ++dk) {
                const int k1 = (k + 8 + dk) & 15;
                int32_t cell[4][8], s0[16], s1[16];
                for (int q = 0; q < 16; ++q) { s0[q] = A0[q]; s1[q] = A1[q]; }
                auto key0 = [&](int32_t v) { return v < 0 ? 99 : ((res0(v) - k) & 15); };
                auto key1 = [&](int32_t v) { return v < 0 ? 99 : ((res0(v) - k1) & 15); };
                std::stable_sort(s0, s0 + 16, [&](int32_t a, int32_t b) { return key0(a) < key0(b); });
                std::stable_sort(s1, s1 + 16, [&](int32_t a, int32_t b) { return key1(a) < key1(b); });
                for (int q = 0; q < 8; ++q) { cell[0][q] = s0[q]; cell[1][q] = s0[8 + q]; cell[2][q] = s1[q]; cell[3][q] = s1[8 + q]; }
                const int c = col_clashes(cell[0], cell[2]) + col_clashes(cell[1], cell[3]);
                if (c < best) { best = c; std::memcpy(best_cell, cell, sizeof(cell)); }
            }
        // local search: exchange two slots of one row between its two cells while that lowers the clash count
        for (int pass = 0; pass < (kFull ? 4 : 0) && best > 0; ++pass) {
            bool improved = false;
            for (int row = 0; row < 2; ++row)
                for (int x = 0; x < 8; ++x)
                    for (int y = 0; y < 8 && best > 0; ++y) {
                        std::swap(best_cell[2 * row][x], best_cell[2 * row + 1][y]);
                        const int c = col_clashes(best_cell[0], best_cell[2]) + col_clashes(best_cell[1], best_cell[3]);
                        if (c < best) { best = c; improved = true; }
                        else std::swap(best_cell[2 * row][x], best_cell[2 * row + 1][y]);
                    }
            if (!improved) break;
        }
        return best;
    };
    for (int i0 = 0; i0 < nrow; i0 += 2) {
        int32_t best_cell[4][8];
        if (i0 + 1 >= nrow) {
            split_pair(rows[i0], -1, best_cell);
        } else {
            // partner: the row among the next few that splits with the fewest clashes (the last, short row stays last)
            int bj = i0 + 1, bc = 1 << 30;
            const int lim = std::min(nrow - 1, i0 + 1 + kPartnerWindow);
            for (int j = i0 + 1; j < std::max(lim, i0 + 2) && j < nrow; ++j) {
                int32_t tmp[4][8];
                const int c = split_pair(rows[i0], rows[j], tmp);
                if (c < bc) { bc = c; bj = j; std::memcpy(best_cell, tmp, sizeof(tmp)); if (c == 0) break; }
            }
            std::swap(rows[i0 + 1], rows[bj]);
        }
        int32_t lane[32];
        std::fill(lane, lane + 32, -1);
        for (int cidx = 0; cidx < 4; ++cidx)
            for (int q = 0; q < 8; ++q) lane[kCellLane[cidx][q]] = best_cell[cidx][q];
        out.insert(out.end(), lane, lane + 32);
    }
    while (!out.empty() && out.back() < 0) out.pop_back();
}

// Same packing for COLUMNS of the strip order (elem_order 6): a column is what one thread walks, up to kMaxRows slots
// with up to four flushed local ids each (-1: no atomic at that row / position).  A 16-lane group is conflict-free iff,
// for every (row, position), the flushed owned ids are distinct mod 16.  Items are taken in the given order (the caller
// sorts them by shape, so waves stay uniform in their chain pattern); returns the column order, -1 = empty column.
constexpr int kMaxRows = 6;
void pack_column_groups(const std::vector<std::array<int32_t, 4 * kMaxRows>> &cols, int32_t n_owned, std::vector<int32_t> &out) {
    // No padding: exactly ceil(n / 16) groups, all open from the start; a column goes to the EARLIEST group where it adds
    // the fewest bank clashes (zero when possible).  With the caller's order (shape, then first node id: columns that
    // start on consecutive nodes follow one another and have consecutive local ids at every row / position) runs of 16
    // columns stacked along the node numbering are conflict-free in all their instructions at once.
    constexpr int G = 16, M = 4 * kMaxRows;
    const int n = (int)cols.size();
    const int ng = (n + G - 1) / G;
    struct Grp { std::vector<int32_t> el; uint16_t used[M]; };
    std::vector<Grp> grp(ng);
    for (auto &g : grp) std::fill(g.used, g.used + M, (uint16_t)0);
    int first_open = 0;
    for (int32_t i = 0; i < n; ++i) {
        uint16_t bit[M];
        for (int k = 0; k < M; ++k) {
            const int32_t l = cols[i][k];
            bit[k] = (l >= 0 && l < n_owned) ? (uint16_t)(1u << (l & 15)) : 0;
        }
        int best = -1, best_cost = 1 << 30;
        for (int j = first_open; j < ng && best_cost > 0; ++j) {
            if ((int)grp[j].el.size() >= G) continue;
            int cost = 0;
            for (int k = 0; k < M; ++k) cost += (grp[j].used[k] & bit[k]) != 0;
            if (cost < best_cost) { best_cost = cost; best = j; }
        }
        Grp &g = grp[best];
        g.el.push_back(i);
        for (int k = 0; k < M; ++k) g.used[k] |= bit[k];
        while (first_open < ng && (int)grp[first_open].el.size() >= G) ++first_open;
    }
    // only the last group may be short: move the tail of the last full-enough groups forward
    for (int j = 0; j < ng; ++j) out.insert(out.end(), grp[j].el.begin(), grp[j].el.end());
    // groups are emitted in order; a short group in the middle would shift the 16-lane alignment of the following ones,
    // so short groups are topped up from the very end of the list
    {
        std::vector<int32_t> res;
        res.reserve(out.size());
        std::vector<std::vector<int32_t>> gs(ng);
        for (int j = 0; j < ng; ++j) gs[j] = grp[j].el;
        int hi = ng - 1;
        for (int j = 0; j < ng; ++j) {
            while ((int)gs[j].size() < G && hi > j) {
                if (gs[hi].empty()) { --hi; continue; }
                gs[j].push_back(gs[hi].back());
                gs[hi].pop_back();
            }
        }
        for (int j = 0; j < ng; ++j) res.insert(res.end(), gs[j].begin(), gs[j].end());
        out.swap(res);
    }
}

// Order the elements of one tile so that the 64 lanes of a wave-instruction (64 consecutive
// positions) do not add into the same LDS accumulator: same-address ds_add_f64 lanes serialise.
//   mode 0: keep Morton order (neighbouring lanes share nodes: worst for atomics)
//   mode 1: transpose (lane l of batch b takes Morton position l*B + b)
//   mode 2: greedy colouring -- fill batches of 64 with elements that share no owned node
//   mode 3: LDS-bank-aware groups (default).  ds_add_f64 is serviced in groups of 16 consecutive
//           lanes over 16 eight-byte slots (measured): a group is conflict-free
//           iff, for each of the three corner positions, the owned nodes' local ids are distinct
//           mod 16.  Elements are packed greedily into such groups, most-constrained first; a group
//           that cannot be completed is padded with skip records (-1), ~2 % on T1M.
void order_tile_elements(std::vector<int32_t> &telems, const int64_t *conn, int npe,
                         const std::vector<int32_t> &lid, int32_t n_owned, int mode) {
    const int n = (int)telems.size();
    if (npe != 3 && mode != 3) mode = 3;                 // the legacy orders are TRI3-only lab variants
    if (mode == 3) {
        std::vector<int32_t> out;
        out.reserve(n + n / 8 + 16);
        if (g_halfwave_pack && g_halfwave_all) {
            // one-element-per-slot TRI3 and QUAD4 records through the same two-level packing as the paired slots: atomic
            // rows of 16, then half-waves whose ds_read_b128 lane groups are conflict-poor as well
            std::vector<std::array<int32_t, 4>> items(n);
            for (int i = 0; i < n; ++i)
                for (int k = 0; k < 4; ++k) items[i][k] = k < npe ? lid[conn[npe * (int64_t)telems[i] + k]] : -1;
            std::vector<int32_t> o;
            pack_slot_halfwaves(items, n_owned, o);
            for (int32_t v : o) out.push_back(v < 0 ? -1 : telems[v]);
        } else {
            pack_bank_groups(telems, conn, npe, lid, n_owned, false, false, out);
        }
        telems.swap(out);
        return;
    }
    if (mode == 0 || n <= 64) return;
    const int nb = (n + 63) / 64;
    std::vector<int32_t> out;
    out.reserve(n);
    if (mode == 1) {
        for (int b = 0; b < nb; ++b)
            for (int l = 0; l < 64; ++l) {
                const int p = l * nb + b;
                if (p < n) out.push_back(telems[p]);
            }
        // positions l*nb+b >= n leave short batches; the list stays a permutation
        telems.swap(out);
        return;
    }
    // mode 2
    std::vector<std::vector<int32_t>> batch(nb);
    std::vector<std::vector<uint64_t>> used(nb, std::vector<uint64_t>(kMaxLocal / 64, 0));
    std::vector<int32_t> leftover;
    int start = 0;
    for (int i = 0; i < n; ++i) {
        const int32_t e = telems[i];
        int32_t l[3];
        for (int k = 0; k < 3; ++k) l[k] = lid[conn[3 * (int64_t)e + k]];
        int placed = -1;
        for (int t = 0; t < nb; ++t) {
            const int b = (start + t) % nb;
            const int cap = (b == nb - 1) ? n - 64 * (nb - 1) : 64;
            if ((int)batch[b].size() >= cap) continue;
            bool clash = false;
            for (int k = 0; k < 3; ++k)
                if (l[k] < n_owned && (used[b][l[k] >> 6] >> (l[k] & 63)) & 1) clash = true;
            if (clash) continue;
            placed = b;
            break;
        }
        if (placed < 0) { leftover.push_back(e); continue; }
        batch[placed].push_back(e);
        for (int k = 0; k < 3; ++k)
            if (l[k] < n_owned) used[placed][l[k] >> 6] |= 1ull << (l[k] & 63);
        start = (placed + 1) % nb;      // neighbours in Morton order go to different batches
    }
    size_t lo = 0;
    for (int b = 0; b < nb; ++b) {
        const int cap = (b == nb - 1) ? n - 64 * (nb - 1) : 64;
        while ((int)batch[b].size() < cap && lo < leftover.size()) batch[b].push_back(leftover[lo++]);
        out.insert(out.end(), batch[b].begin(), batch[b].end());
    }
    telems.swap(out);
}

// returns 0 ok, 1 = a tile exceeded kMaxLocal nodes (retry smaller), -1 error
// Cut the sorted element list into tiles: at most T elements AND at most `node_cap` distinct nodes
// among a tile's own elements (node_cap <= 0: no node limit).  Cutting by nodes equalises the LDS
// footprint of the tiles, so the launch-wide maximum (which sizes every workgroup) sits near the
// median and more workgroups fit per CU.
void cut_tiles(const int64_t *conn, int npe, int64_t ne, int64_t nn, const std::vector<int32_t> &order, int32_t T,
               int32_t node_cap, std::vector<int64_t> &bounds) {
    bounds.assign(1, 0);
    if (node_cap <= 0) {
        for (int64_t p = T; p < ne; p += T) bounds.push_back(p);
        bounds.push_back(ne);
        return;
    }
    std::vector<int32_t> stamp(nn, -1);
    int32_t tile = 0, distinct = 0;
    int64_t start = 0;
    for (int64_t p = 0; p < ne; ++p) {
        const int64_t e = order[p];
        int fresh = 0;
        for (int k = 0; k < npe; ++k)
            if (stamp[conn[npe * e + k]] != tile) ++fresh;
        if (p > start && (p - start >= T || distinct + fresh > node_cap)) {
            int64_t cut = p;
            if (g_snap > 0 && (int64_t)g_codes.size() == ne) {
                const int64_t lo = std::max(start + 1, p - ((p - start) * g_snap) / 100);
                int best_lvl = -1;                                       // curve level of the boundary (two code bits per level)
                for (int64_t q = p; q >= lo; --q) {                      // the latest position wins ties: biggest tile
                    const uint32_t x = g_codes[q - 1] ^ g_codes[q];
                    const int lvl = x ? (31 - __builtin_clz(x)) / 2 : -1;
                    if (lvl > best_lvl) { best_lvl = lvl; cut = q; }
                }
            }
            bounds.push_back(cut);
            ++tile;
            start = cut;
            distinct = 0;
            if (cut != p) { p = cut - 1; continue; }                     // re-walk [cut, p) as part of the new tile
        }
        for (int k = 0; k < npe; ++k) {
            int32_t &st = stamp[conn[npe * e + k]];
            if (st != tile) { st = tile; ++distinct; }
        }
    }
    bounds.push_back(ne);
}

int try_build(const int64_t *conn, int npe, int64_t ne, int64_t nn, const double *xy, const int32_t *x_src,
              const int32_t *u_src, const int64_t *edges, int64_t ned, int32_t T, int32_t node_cap,
              const std::vector<int32_t> &order, int elem_order, int32_t chunk_cap, int32_t pair_block, int32_t shards,
              HostPlan &P) {
    std::vector<int64_t> bounds;
    cut_tiles(conn, npe, ne, nn, order, T, node_cap, bounds);
    const int32_t nt_main = ne > 0 ? (int32_t)bounds.size() - 1 : 0;
    std::vector<int32_t> tile_of(ne);                    // sorted position -> tile
    for (int32_t t = 0; t < nt_main; ++t)
        for (int64_t p = bounds[t]; p < bounds[t + 1]; ++p) tile_of[p] = t;
    std::vector<int32_t> owner(nn, std::numeric_limits<int32_t>::max());
    for (int64_t p = 0; p < ne; ++p) {
        const int32_t t = tile_of[p];
        const int64_t e = order[p];
        for (int k = 0; k < npe; ++k) {
            int32_t &o = owner[conn[npe * e + k]];
            if (t < o) o = t;
        }
    }
    int64_t n_orphan = 0;
    for (int64_t n = 0; n < nn; ++n)
        if (owner[n] == std::numeric_limits<int32_t>::max())
            owner[n] = nt_main + (int32_t)(n_orphan++ / kOrphanTileNodes);
    const int32_t nt = nt_main + (int32_t)((n_orphan + kOrphanTileNodes - 1) / kOrphanTileNodes);

    // element sharding (shards > 1): rank r evaluates the contiguous tile range [nt r / shards, nt (r + 1) / shards).  A tile is
    // a BOUNDARY tile of its rank when it reads a node that a tile of another rank owns, or owns a node that a tile of
    // another rank reads -- only those tiles take part in the interface exchange of the owner-sharded mode
    // (hidenn_fem_amd/sharded.py); the rest (interior) depend on nothing another rank produces.
    if (shards < 1) shards = 1;
    std::vector<int32_t> shard_of(nt, 0);
    std::vector<char> bnd(nt, 0);
    for (int32_t r = 0; r < shards; ++r)
        for (int64_t t = ((int64_t)nt * r) / shards; t < ((int64_t)nt * (r + 1)) / shards; ++t) shard_of[t] = r;

    // node -> element adjacency (CSR), elements listed in sorted-position order
    std::vector<int64_t> adj_ptr(nn + 1, 0);
    for (int64_t e = 0; e < ne; ++e)
        for (int k = 0; k < npe; ++k) adj_ptr[conn[npe * e + k] + 1]++;
    for (int64_t n = 0; n < nn; ++n) adj_ptr[n + 1] += adj_ptr[n];
    std::vector<int32_t> adj(adj_ptr[nn]);
    {
        std::vector<int64_t> fill(adj_ptr.begin(), adj_ptr.end() - 1);
        for (int64_t p = 0; p < ne; ++p) {
            const int32_t e = order[p];
            for (int k = 0; k < npe; ++k) adj[fill[conn[npe * e + k]]++] = e;
        }
    }
    // home tile of every element
    std::vector<int32_t> home(ne);
    for (int64_t p = 0; p < ne; ++p) home[order[p]] = tile_of[p];

    // owned nodes per tile (ascending node id), edges per tile
    std::vector<int64_t> own_ptr(nt + 1, 0), edg_ptr(nt + 1, 0);
    for (int64_t n = 0; n < nn; ++n) own_ptr[owner[n] + 1]++;
    for (int64_t g = 0; g < ned; ++g) {
        const int32_t ti = owner[edges[2 * g]], tj = owner[edges[2 * g + 1]];
        edg_ptr[ti + 1]++;
        if (tj != ti) edg_ptr[tj + 1]++;
    }
    for (int32_t t = 0; t < nt; ++t) { own_ptr[t + 1] += own_ptr[t]; edg_ptr[t + 1] += edg_ptr[t]; }
    std::vector<int32_t> own(nn), tedge(edg_ptr[nt]);
    {
        std::vector<int64_t> f(own_ptr.begin(), own_ptr.end() - 1);
        for (int64_t n = 0; n < nn; ++n) own[f[owner[n]]++] = (int32_t)n;
        std::vector<int64_t> fe(edg_ptr.begin(), edg_ptr.end() - 1);
        for (int64_t g = 0; g < ned; ++g) {
            const int32_t ti = owner[edges[2 * g]], tj = owner[edges[2 * g + 1]];
            tedge[fe[ti]++] = (int32_t)g;
            if (tj != ti) tedge[fe[tj]++] = (int32_t)g;
        }
    }

    P = HostPlan();
    P.ne = ne; P.nn = nn; P.ned = ned; P.tile_elems = T; P.npe = npe;
    P.tiles.resize(nt);
    P.elem_pack.reserve(ne + ne / 4);
    P.elem_gid.reserve(ne + ne / 4);
    P.node_src.reserve(2 * (nn + nn / 3));

    // Local node order inside a tile.  By node id -- unless the caller stores its parameter rows in ANOTHER order than it
    // numbers its nodes (x_src not increasing over the free nodes: a model that keeps a badly numbered mesh's rows along the
    // locality curve, hidenn_fem_amd/models.py reorder="auto"): then by ROW, so that the 64 lanes of a gather instruction
    // read neighbouring rows (a few 128-byte lines per wave instead of one line per lane).  Fixed rows go last.
    bool by_row = false;
    if (x_src) {
        int32_t last = -1;
        for (int64_t n = 0; n < nn && !by_row; ++n)
            if (x_src[n] >= 0) { by_row = x_src[n] < last; last = x_src[n]; }
    }
    auto row_key = [&](int32_t n) -> int64_t { return x_src[n] >= 0 ? (int64_t)x_src[n] : ((int64_t)1 << 32) + (int64_t)(~x_src[n]); };
    auto by_row_less = [&](int32_t a, int32_t b) { return row_key(a) < row_key(b); };
    std::vector<int32_t> stamp_e(ne, -1), stamp_n(nn, -1), lid(nn, 0);
    std::vector<int32_t> first_use_stamp(elem_order == 4 ? nn : 0, -1), first_use(elem_order == 4 ? nn : 0, 0);
    std::vector<int32_t> telems, halo, own_order_buf;
    for (int32_t t = 0; t < nt; ++t) {
        TileDesc &d = P.tiles[t];
        d = TileDesc();
        telems.clear();
        halo.clear();
        // elements: home first (sorted order), then halo in discovery order
        if (t < nt_main) {
            const int64_t p0 = bounds[t], p1 = bounds[t + 1];
            for (int64_t p = p0; p < p1; ++p) { stamp_e[order[p]] = t; telems.push_back(order[p]); }
        }
        const int64_t o0 = own_ptr[t], o1 = own_ptr[t + 1];
        for (int64_t i = o0; i < o1; ++i) {
            const int32_t n = own[i];
            for (int64_t a = adj_ptr[n]; a < adj_ptr[n + 1]; ++a) {
                const int32_t e = adj[a];
                if (stamp_e[e] != t) { stamp_e[e] = t; telems.push_back(e); }
            }
        }
        // local nodes: owned first
        const bool chunked = elem_order == 4 && npe == 3;
        std::vector<int32_t> &own_order = own_order_buf;       // this tile's owned nodes in local-id order
        own_order.assign(own.begin() + o0, own.begin() + o1);
        if (by_row) std::sort(own_order.begin(), own_order.end(), by_row_less);
        int32_t nloc = 0;
        for (int32_t n : own_order) { stamp_n[n] = t; lid[n] = nloc++; }
        d.n_owned = nloc;
        for (int32_t e : telems)
            for (int k = 0; k < npe; ++k) {
                const int32_t n = (int32_t)conn[npe * (int64_t)e + k];
                if (stamp_n[n] != t) { stamp_n[n] = t; halo.push_back(n); }
            }
        for (int64_t i = edg_ptr[t]; i < edg_ptr[t + 1]; ++i)
            for (int k = 0; k < 2; ++k) {
                const int32_t n = (int32_t)edges[2 * (int64_t)tedge[i] + k];
                if (stamp_n[n] != t) { stamp_n[n] = t; halo.push_back(n); }
            }
        if (by_row) std::sort(halo.begin(), halo.end(), by_row_less);
        else std::sort(halo.begin(), halo.end());
        if (d.n_owned + (int32_t)halo.size() > kMaxLocal) return 1;
        if (shards > 1)
            for (int32_t n : halo)
                if (shard_of[owner[n]] != shard_of[t]) { bnd[t] = 1; bnd[owner[n]] = 1; }
        int32_t chunk_rec[4] = {0, 0, 0, 0};
        if (chunked) {
            // ---- strips: sort the tile's elements along the longer side of their bounding box (centroids; without
            //      coordinates: list order = curve order of the home elements, then halo) and cut into kChunks equal runs
            const int n = (int)telems.size();
            std::vector<int32_t> strip_of(n, 0);
            {
                std::vector<std::pair<double, int32_t>> key(n);
                if (xy) {
                    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
                    std::vector<double> cx(n), cy(n);
                    for (int i = 0; i < n; ++i) {
                        double c[2] = {0, 0};
                        for (int k = 0; k < npe; ++k) {
                            const int64_t nd = conn[npe * (int64_t)telems[i] + k];
                            c[0] += xy[2 * nd]; c[1] += xy[2 * nd + 1];
                        }
                        cx[i] = c[0]; cy[i] = c[1];
                        for (int a = 0; a < 2; ++a) { lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[a]); }
                    }
                    const bool along_x = (hi[0] - lo[0]) >= (hi[1] - lo[1]);
                    for (int i = 0; i < n; ++i) key[i] = {along_x ? cx[i] : cy[i], i};
                } else {
                    for (int i = 0; i < n; ++i) key[i] = {(double)i, i};
                }
                std::stable_sort(key.begin(), key.end());
                for (int r = 0; r < n; ++r) strip_of[key[r].second] = (int32_t)(((int64_t)r * kChunks) / std::max(n, 1));
            }
            // ---- first strip that touches each local node; local ids = (owned | halo) x (first strip, node id)
            for (int i = 0; i < n; ++i)
                for (int k = 0; k < npe; ++k) {
                    const int32_t nd = (int32_t)conn[npe * (int64_t)telems[i] + k];
                    if (first_use_stamp[nd] != t) { first_use_stamp[nd] = t; first_use[nd] = strip_of[i]; }
                    else first_use[nd] = std::min(first_use[nd], strip_of[i]);
                }
            auto fu = [&](int32_t nd) { return first_use_stamp[nd] == t ? first_use[nd] : kChunks - 1; };   // edge-only nodes: last
            auto by_use = [&](int32_t a, int32_t b) { const int32_t fa = fu(a), fb = fu(b); return fa != fb ? fa < fb : a < b; };
            std::sort(own_order.begin(), own_order.end(), by_use);
            std::sort(halo.begin(), halo.end(), by_use);
            nloc = 0;
            for (int32_t nd : own_order) lid[nd] = nloc++;
            int32_t cnt_o[kChunks] = {0}, cnt_h[kChunks] = {0};
            for (int32_t nd : own_order) cnt_o[fu(nd)]++;
            for (int32_t nd : halo) cnt_h[fu(nd)]++;
            for (int32_t nd : halo) lid[nd] = nloc++;
            // ---- bank-aware groups inside every strip; every strip ends on a 16-slot boundary
            std::vector<int32_t> packed, part;
            int32_t ends[kChunks];
            for (int c = 0; c < kChunks; ++c) {
                part.clear();
                for (int i = 0; i < n; ++i)
                    if (strip_of[i] == c) part.push_back(telems[i]);
                pack_bank_groups(part, conn, npe, lid, d.n_owned, true, true, packed);
                ends[c] = (int32_t)packed.size();
                const int32_t len = ends[c] - (c ? ends[c - 1] : 0);
                if (chunk_cap > 0 && len > chunk_cap) return 1;              // a strip must fit one pass of the workgroup
                P.max_chunk_elems = std::max(P.max_chunk_elems, len);
            }
            telems.swap(packed);
            const int32_t po0 = (cnt_o[0] + 63) / 64, po1 = (cnt_o[0] + cnt_o[1] + 63) / 64;
            const int32_t ph0 = (cnt_h[0] + 63) / 64, ph1 = (cnt_h[0] + cnt_h[1] + 63) / 64;
            chunk_rec[0] = ends[0]; chunk_rec[1] = ends[1];
            chunk_rec[2] = po0 | (po1 << 8) | (ph0 << 16) | (ph1 << 24);
        } else {
            for (int32_t n : halo) lid[n] = nloc++;
            if (!(elem_order >= 5 && npe == 3)) order_tile_elements(telems, conn, npe, lid, d.n_owned, elem_order);
        }
        nloc = d.n_owned + (int32_t)halo.size();
        if (elem_order == 4) P.tile_chunks.insert(P.tile_chunks.end(), chunk_rec, chunk_rec + 4);

        d.elem_off = (int32_t)P.elem_pack.size();
        d.node_off = (int32_t)(P.node_src.size() / 2);
        d.n_node = nloc;
        d.edge_off = (int32_t)P.edge_pack.size();
        d.n_edge = (int32_t)(edg_ptr[t + 1] - edg_ptr[t]);
        if (elem_order >= 5 && npe == 3) {
            // ---- paired slots.  B is the successor of A when A's corner 0 is B's corner 0 and A's corner 2 is B's corner 1
            //      (the next element of the fan around n).  succ/pred are partial injective maps, so the candidate graph is
            //      a set of paths and cycles; consecutive elements are paired greedily along each.
            const int n = (int)telems.size();                   // telems: plain element ids here (order_tile_elements skipped)
            std::unordered_map<uint64_t, int32_t> by_edge0;    // (c0, c1) -> index in telems
            by_edge0.reserve((size_t)n * 2);
            for (int i = 0; i < n; ++i) {
                const int64_t e = telems[i];
                by_edge0[((uint64_t)conn[3 * e] << 32) | (uint32_t)conn[3 * e + 1]] = i;
            }
            std::vector<int32_t> succ(n, -1), pred(n, -1);
            for (int i = 0; i < n; ++i) {
                const int64_t e = telems[i];
                auto it = by_edge0.find(((uint64_t)conn[3 * e] << 32) | (uint32_t)conn[3 * e + 2]);   // B with (c0, c1) = (A.c0, A.c2)
                if (it != by_edge0.end() && it->second != i && pred[it->second] < 0) { succ[i] = it->second; pred[it->second] = i; }
            }
            std::vector<char> used(n, 0);
            std::vector<std::array<int32_t, 2>> slots;         // (A index, B index or -1)
            slots.reserve(n);
            for (int pass = 0; pass < 2; ++pass)               // pass 0: paths from their heads; pass 1: what is left (cycles)
                for (int i = 0; i < n; ++i) {
                    if (used[i] || (pass == 0 && pred[i] >= 0)) continue;
                    int a = i;
                    while (a >= 0 && !used[a]) {
                        used[a] = 1;
                        const int b = succ[a];
                        if (b >= 0 && !used[b]) { used[b] = 1; slots.push_back({a, b}); a = succ[b]; }
                        else { slots.push_back({a, -1}); a = -1; }
                    }
                }
            // pairs first (curve order kept), then singles: waves are uniform in `hasB` except one
            std::stable_sort(slots.begin(), slots.end(), [](const std::array<int32_t, 2> &x, const std::array<int32_t, 2> &y) { return (x[1] >= 0) > (y[1] >= 0); });
            std::vector<std::array<int32_t, 4>> items(slots.size());
            for (size_t s_ = 0; s_ < slots.size(); ++s_) {
                const int64_t ea = telems[slots[s_][0]];
                items[s_] = {lid[conn[3 * ea]], lid[conn[3 * ea + 1]], lid[conn[3 * ea + 2]],
                             slots[s_][1] >= 0 ? lid[conn[3 * (int64_t)telems[slots[s_][1]] + 2]] : -1};
            }
            std::vector<int32_t> order_s;
            std::vector<char> chain(slots.size(), 0);           // slot's b, c rows are carried into the next slot of its column
            size_t n_pair = 0;
            while (n_pair < slots.size() && slots[n_pair][1] >= 0) ++n_pair;
            d.pad = pair_block;                                 // column stride of the slot array (slot j of thread t at j*stride + t)
            if (elem_order == 6 && pair_block != 256) { set_error("plan: the strip order (lab) needs 256-thread tiles"); return -1; }
            if (elem_order == 6) {
                // ---- strips: pair Q follows pair P in a strip when Q.n = P.b and Q.d = P.c (the next split quad along a
                //      row): the thread that walks the strip keeps P's rows of b and c in registers and adds them to Q's
                //      rows of n and d -- a strip of L pairs flushes 2L + 2 node rows instead of 4L.  Strips are cut to at
                //      most H slots (the rows of a column) and bin-packed into <= 256 columns.
                const int S = (int)slots.size();
                const int H = std::min(kMaxRows, std::max(1, (S + 255) / 256));   // rows per column: as few as 256 threads allow
                if (S > 256 * kMaxRows) return 1;
                auto gnode = [&](int si, int k) -> int64_t {    // global node id of slot position k (n, b, c, d)
                    return k < 3 ? conn[3 * (int64_t)telems[slots[si][0]] + k] : conn[3 * (int64_t)telems[slots[si][1]] + 2];
                };
                std::unordered_map<uint64_t, int32_t> by_nd;
                by_nd.reserve(n_pair * 2);
                for (size_t si = 0; si < n_pair; ++si) by_nd[((uint64_t)gnode((int)si, 0) << 32) | (uint32_t)gnode((int)si, 3)] = (int32_t)si;
                std::vector<int32_t> nxt(n_pair, -1), prv(n_pair, -1);
                for (size_t si = 0; si < n_pair; ++si) {
                    auto it = by_nd.find(((uint64_t)gnode((int)si, 1) << 32) | (uint32_t)gnode((int)si, 2));
                    if (it != by_nd.end() && it->second != (int32_t)si && prv[it->second] < 0) { nxt[si] = it->second; prv[it->second] = (int32_t)si; }
                }
                std::vector<std::vector<int32_t>> strips, columns;
                {
                    const int L = H;
                    strips.clear();
                    std::vector<char> seen(n_pair, 0);
                    for (int pass = 0; pass < 2; ++pass)
                        for (size_t si = 0; si < n_pair; ++si) {
                            if (seen[si] || (pass == 0 && prv[si] >= 0)) continue;
                            int32_t a = (int32_t)si;
                            while (a >= 0 && !seen[a]) {
                                strips.emplace_back();
                                while (a >= 0 && !seen[a] && (int)strips.back().size() < L) { seen[a] = 1; strips.back().push_back(a); a = nxt[a]; }
                            }
                        }
                    std::stable_sort(strips.begin(), strips.end(), [](const std::vector<int32_t> &x, const std::vector<int32_t> &y) { return x.size() > y.size(); });
                    for (size_t si = n_pair; si < slots.size(); ++si) strips.push_back({(int32_t)si});   // singles last
                    // best-fit decreasing into exactly ceil(S / H) columns of H rows (fewest wave-rows); a strip that fits no
                    // single column is cut where the roomiest column ends (the cut slot just flushes its b and c)
                    const int ncol = (S + H - 1) / H;
                    columns.assign(ncol, {});
                    std::vector<std::vector<int32_t>> with_room(H + 1);      // column indices by free rows
                    for (int ci = ncol - 1; ci >= 0; --ci) with_room[H].push_back(ci);
                    for (size_t k = 0; k < strips.size(); ++k) {
                        std::vector<int32_t> st = strips[k];
                        while (!st.empty()) {
                            const int len = (int)st.size();
                            int r = len;
                            while (r <= H && with_room[r].empty()) ++r;
                            int take = len;
                            if (r > H) {                                     // no column holds it whole: fill the roomiest
                                r = len - 1;
                                while (r > 0 && with_room[r].empty()) --r;
                                take = r;
                            }
                            if (r <= 0) return -1;                           // cannot happen: ncol * H >= S
                            const int32_t ci = with_room[r].back();
                            with_room[r].pop_back();
                            columns[ci].insert(columns[ci].end(), st.begin(), st.begin() + take);
                            for (int q = 0; q + 1 < take; ++q) chain[st[q]] = 1;
                            if (r - take > 0) with_room[r - take].push_back(ci);
                            st.erase(st.begin(), st.begin() + take);
                        }
                    }
                    while (!columns.empty() && columns.back().empty()) columns.pop_back();
                }
                if (columns.size() > 256) return 1;
                // shape order (chain pattern as a number, long strips first), then bank-aware groups of 16 columns
                auto shape = [&](const std::vector<int32_t> &c) { uint32_t v = 0; for (int32_t si : c) v = v * 4 + (chain[si] ? 3u : (slots[si][1] >= 0 ? 2u : 1u)); for (size_t q = c.size(); q < (size_t)kMaxRows; ++q) v *= 4; return v; };
                std::stable_sort(columns.begin(), columns.end(), [&](const std::vector<int32_t> &x, const std::vector<int32_t> &y) {
                    const uint32_t sx = shape(x), sy = shape(y);
                    return sx != sy ? sx > sy : gnode(x[0], 0) < gnode(y[0], 0);
                });
                std::vector<std::array<int32_t, 4 * kMaxRows>> citems(columns.size());
                int rows_used = 0;
                for (size_t ci = 0; ci < columns.size(); ++ci) {
                    citems[ci].fill(-1);
                    rows_used = std::max(rows_used, (int)columns[ci].size());
                    for (size_t r = 0; r < columns[ci].size(); ++r) {
                        const int32_t si = columns[ci][r];
                        citems[ci][4 * r + 0] = items[si][0];
                        citems[ci][4 * r + 3] = items[si][3];
                        if (!chain[si]) { citems[ci][4 * r + 1] = items[si][1]; citems[ci][4 * r + 2] = items[si][2]; }
                    }
                }
                std::vector<int32_t> corder;
                pack_column_groups(citems, d.n_owned, corder);
                const int stride = (int)((corder.size() + 15) / 16 * 16);
                d.pad = stride;
                order_s.assign((size_t)rows_used * stride, -1);
                for (size_t t_ = 0; t_ < corder.size(); ++t_)
                    for (size_t r = 0; r < columns[corder[t_]].size(); ++r) order_s[r * stride + t_] = columns[corder[t_]][r];
                while (!order_s.empty() && order_s.back() < 0) order_s.pop_back();
                P.max_rows = std::max(P.max_rows, rows_used);
            } else {
            // one packing for pairs and singles (a single in a wave of pairs costs nothing extra: the wave runs B anyway);
            // the pairs stay in front, so the short last row holds the singles
            if (g_halfwave_pack) {
                pack_slot_halfwaves(items, d.n_owned, order_s);
            } else {
                std::vector<std::array<int32_t, 4>> part(items.begin(), items.begin() + n_pair);
                std::vector<int32_t> o;
                pack_slot_groups(part, d.n_owned, o);
                while (!o.empty() && (o.size() & 15)) o.push_back(-1);
                order_s = o;
                part.assign(items.begin() + n_pair, items.end());
                o.clear();
                pack_slot_groups(part, d.n_owned, o);
                for (int32_t v : o) order_s.push_back(v < 0 ? -1 : v + (int32_t)n_pair);
            }
            P.max_rows = std::max(P.max_rows, ((int)order_s.size() + pair_block - 1) / pair_block);
            }
            d.n_elem = (int32_t)order_s.size();
            for (int32_t si : order_s) {
                if (si < 0) {
                    P.elem_pack.push_back(kSkipBit);
                    P.elem_pack_hi.push_back(0u);
                    P.elem_gid.push_back(-1);
                    P.elem_gid_b.push_back(-1);
                    continue;
                }
                const int32_t ea = telems[slots[si][0]], eb = slots[si][1] >= 0 ? telems[slots[si][1]] : -1;
                const std::array<int32_t, 4> &L = items[si];
                P.elem_pack.push_back((uint32_t)L[0] | ((uint32_t)L[1] << kLocalBits) | ((uint32_t)L[2] << (2 * kLocalBits)) |
                                      (home[ea] == t ? kHomeBit : 0u));
                P.elem_pack_hi.push_back(eb >= 0 ? ((uint32_t)L[3] | (1u << 10) | (home[eb] == t ? (1u << 11) : 0u) | (chain[si] ? (1u << 12) : 0u)) : 0u);
                P.elem_gid.push_back(ea);
                P.elem_gid_b.push_back(eb);
                P.n_pairs += eb >= 0;
                P.n_chained += chain[si];
            }
            P.paired = true;
        } else {
        d.n_elem = (int32_t)telems.size();
        for (int32_t e : telems) {
            if (e < 0) {                                   // padding of a bank-conflict-free group
                P.elem_pack.push_back(kSkipBit);
                if (npe == 4) P.elem_pack_hi.push_back(0u);
                P.elem_gid.push_back(-1);
                continue;
            }
            const uint32_t l0 = lid[conn[npe * (int64_t)e]], l1 = lid[conn[npe * (int64_t)e + 1]],
                           l2 = lid[conn[npe * (int64_t)e + 2]];
            P.elem_pack.push_back(l0 | (l1 << kLocalBits) | (l2 << (2 * kLocalBits)) |
                                  (home[e] == t ? kHomeBit : 0u));
            if (npe == 4) P.elem_pack_hi.push_back((uint32_t)lid[conn[npe * (int64_t)e + 3]]);
            P.elem_gid.push_back(e);
        }
        }
        auto push_node = [&](int32_t n) {
            P.node_src.push_back(x_src ? x_src[n] : n);
            P.node_src.push_back(u_src ? u_src[n] : n);
        };
        for (int32_t n : own_order) { push_node(n); P.owned_gid_by_slot.push_back(n); }
        for (int32_t n : halo) { push_node(n); P.owned_gid_by_slot.push_back(n); }
        for (int64_t i = edg_ptr[t]; i < edg_ptr[t + 1]; ++i) {
            const int32_t g = tedge[i];
            const int64_t ni = edges[2 * (int64_t)g], nj = edges[2 * (int64_t)g + 1];
            P.edge_pack.push_back((uint32_t)lid[ni] | ((uint32_t)lid[nj] << kLocalBits) |
                                  (owner[ni] == t ? kHomeBit : 0u));
            P.edge_gid.push_back(g);
        }
        P.max_nodes = std::max(P.max_nodes, d.n_node);
        P.max_owned = std::max(P.max_owned, d.n_owned);
        P.max_elems = std::max(P.max_elems, d.n_elem);
        P.max_edges = std::max(P.max_edges, d.n_edge);
    }
    // ---- sharded plans: inside every rank's tile range the boundary tiles come first, so that a rank launches them as one
    //      contiguous sub-range, starts the interface exchange, and runs the interior sub-range under it.  Ownership was
    //      fixed above; the tile index only decides where a tile's records live, which workgroup walks it and where its
    //      partial energy goes (the loss is summed in the new tile order).
    P.shards = shards;
    P.pair_block = pair_block;
    P.shard_desc.clear();
    {
        std::vector<TileDesc> sorted_tiles;
        std::vector<int32_t> sorted_chunks;          // per-tile records indexed by tile (chunked order) move with their tile
        const bool has_chunks = P.tile_chunks.size() == (size_t)4 * nt;
        sorted_tiles.reserve(nt);
        auto take = [&](int32_t t) {
            sorted_tiles.push_back(P.tiles[t]);
            if (has_chunks) sorted_chunks.insert(sorted_chunks.end(), P.tile_chunks.begin() + 4 * (size_t)t, P.tile_chunks.begin() + 4 * (size_t)t + 4);
        };
        for (int32_t r = 0; r < shards; ++r) {
            const int32_t lo = (int32_t)(((int64_t)nt * r) / shards), hi = (int32_t)(((int64_t)nt * (r + 1)) / shards);
            for (int32_t t = lo; t < hi; ++t)
                if (bnd[t]) take(t);
            const int32_t mid = (int32_t)sorted_tiles.size();
            for (int32_t t = lo; t < hi; ++t)
                if (!bnd[t]) take(t);
            P.shard_desc.insert(P.shard_desc.end(), {lo, mid, hi, 0});
        }
        if (shards > 1) {
            P.tiles.swap(sorted_tiles);
            if (has_chunks) P.tile_chunks.swap(sorted_chunks);
        }
    }
    // owned node ids in (final tile order, local order): every node exactly once -- the TILE-MAJOR node order a caller can store
    // its parameter rows in (hfem_plan_export 11; hidenn_fem_amd/models.py reorder="tile")
    {
        std::vector<int32_t> og;
        og.reserve(nn);
        for (const TileDesc &d : P.tiles)
            for (int32_t l = 0; l < d.n_owned; ++l) og.push_back(P.owned_gid_by_slot[(size_t)d.node_off + l]);
        P.owned_gid.swap(og);
    }
    // ---- uniform node stride: tile t's row-map records start at t * node_stride, so a kernel can load them from its
    //      tile index alone, in parallel with the descriptor (one dependent memory round trip less: desc -> maps -> rows
    //      becomes {desc, maps} -> rows).  Padding repeats the tile's last record (a valid row pair: loads through it are
    //      harmless); kNodeTailPad more records follow the last tile so that lanes past a tile's stride stay inside the array.
    {
        const int32_t stride = std::max(16, (P.max_nodes + 15) / 16 * 16);   // >= 16: kernels clamp unguarded loads to stride - 1
        if ((int64_t)nt * stride + kNodeTailPad > (int64_t)std::numeric_limits<int32_t>::max()) {
            set_error("plan: tile arrays exceed int32 offsets");
            return -1;
        }
        std::vector<int32_t> ns((size_t)2 * ((size_t)nt * stride + kNodeTailPad));
        int32_t lastx = 0, lastu = 0;                       // running "last valid record" (tiles without nodes repeat it)
        for (int32_t t = 0; t < nt; ++t) {
            TileDesc &d = P.tiles[t];
            const int32_t *srcp = P.node_src.data() + 2 * (size_t)d.node_off;
            int32_t *dst = ns.data() + 2 * (size_t)t * stride;
            std::memcpy(dst, srcp, sizeof(int32_t) * 2 * (size_t)d.n_node);
            if (d.n_node > 0) { lastx = srcp[2 * (d.n_node - 1)]; lastu = srcp[2 * (d.n_node - 1) + 1]; }
            else if (t == 0 && nt > 1 && P.tiles[1].n_node > 0) { lastx = P.node_src[2 * (size_t)P.tiles[1].node_off]; lastu = P.node_src[2 * (size_t)P.tiles[1].node_off + 1]; }
            for (int32_t l = d.n_node; l < stride; ++l) { dst[2 * l] = lastx; dst[2 * l + 1] = lastu; }
            d.node_off = t * stride;
        }
        for (size_t l = (size_t)nt * stride; l < (size_t)nt * stride + kNodeTailPad; ++l) { ns[2 * l] = lastx; ns[2 * l + 1] = lastu; }
        P.node_records = (int64_t)P.node_src.size() / 2;
        P.node_src.swap(ns);
        P.node_stride = stride;
    }
    // ---- uniform slot stride, same reason: tile t's element records start at t * elem_stride; padding = skip records,
    //      kElemTailPad more after the last tile (unguarded loads of up to 6 x 256 lanes from a tile's start)
    {
        // paired plans: one column stride for the whole plan (the widest tile's), so that a thread's row-j record sits at
        // j * col_stride + t whatever the tile
        int32_t cs = 0;
        if (P.paired) {
            for (int32_t t = 0; t < nt; ++t) cs = std::max(cs, P.tiles[t].pad);
            P.max_elems = 0;
            for (int32_t t = 0; t < nt; ++t) {
                const TileDesc &d = P.tiles[t];
                const int32_t ne_new = d.n_elem > 0 ? ((d.n_elem - 1) / d.pad) * cs + (d.n_elem - 1) % d.pad + 1 : 0;
                P.max_elems = std::max(P.max_elems, ne_new);
            }
            P.col_stride = cs;
        }
        const int64_t stride = std::max(16, (P.max_elems + 15) / 16 * 16);   // >= 16 even for plans without elements (orphan-node tiles only)
        const size_t total = (size_t)nt * stride + kElemTailPad;
        if (total > (size_t)std::numeric_limits<int32_t>::max()) { set_error("plan: tile arrays exceed int32 offsets"); return -1; }
        const bool hi = !P.elem_pack_hi.empty(), gb = !P.elem_gid_b.empty();
        std::vector<uint32_t> ep(total, kSkipBit), eh(hi ? total : 0, 0u);
        std::vector<int32_t> eg(total, -1), egb(gb ? total : 0, -1);
        int64_t real = 0;
        for (int32_t t = 0; t < nt; ++t) {
            TileDesc &d = P.tiles[t];
            const size_t so = (size_t)d.elem_off, dn = (size_t)t * stride, n = (size_t)d.n_elem;
            int32_t n_new = 0;
            for (size_t i = 0; i < n; ++i) {
                const size_t k = cs > 0 ? (i / d.pad) * cs + i % d.pad : i;      // row-major re-stride
                ep[dn + k] = P.elem_pack[so + i];
                eg[dn + k] = P.elem_gid[so + i];
                if (hi) eh[dn + k] = P.elem_pack_hi[so + i];
                if (gb) egb[dn + k] = P.elem_gid_b[so + i];
                n_new = (int32_t)k + 1;
            }
            real += d.n_elem;
            d.elem_off = (int32_t)dn;
            d.n_elem = n_new;
            if (cs > 0) d.pad = cs;
        }
        P.elem_pack.swap(ep); P.elem_gid.swap(eg); P.elem_pack_hi.swap(eh); P.elem_gid_b.swap(egb);
        P.elem_stride = (int32_t)stride;
        P.elem_records = real;
    }
    if (P.elem_pack.size() > (size_t)std::numeric_limits<int32_t>::max() ||
        P.node_src.size() / 2 > (size_t)std::numeric_limits<int32_t>::max()) {
        set_error("plan: tile arrays exceed int32 offsets");
        return -1;
    }
    return 0;
}

}  // namespace

void set_plan_curve(int c) { set_locality_curve(c); }
void set_plan_snap(int percent) { set_tile_snap(percent); }
void set_plan_read_pack(int v) { set_halfwave_pack(v); }

int build_host_plan(const int64_t *conn, int npe, int64_t ne, int64_t nn, const double *coords,
                    const int32_t *x_src, const int32_t *u_src, const int64_t *edges,
                    int64_t ned, int32_t tile_elems, int32_t node_cap, int elem_order, int32_t chunk_cap,
                    HostPlan &out, int32_t pair_block, int32_t shards) {
    if (pair_block != 256 && pair_block != 512) { set_error("plan: pair_block must be 256 or 512"); return -1; }
    if (shards < 1 || shards > 4096) { set_error("plan: shards must be in [1, 4096]"); return -1; }
    if (npe != 3 && npe != 4) { set_error("plan: nodes per element must be 3 (TRI3) or 4 (QUAD4)"); return -1; }
    if (ne < 0 || nn < 0 || ned < 0 || nn > std::numeric_limits<int32_t>::max() ||
        ne > std::numeric_limits<int32_t>::max() || ned > std::numeric_limits<int32_t>::max()) {
        set_error("plan: sizes must be in [0, 2^31)");
        return -1;
    }
    if ((ne > 0 && !conn) || (ned > 0 && !edges)) { set_error("plan: null connectivity/edges"); return -1; }
    for (int64_t i = 0; i < npe * ne; ++i)
        if (conn[i] < 0 || conn[i] >= nn) { set_error("plan: connectivity index out of range"); return -1; }
    for (int64_t i = 0; i < 2 * ned; ++i)
        if (edges[i] < 0 || edges[i] >= nn) { set_error("plan: edge index out of range"); return -1; }
    if (tile_elems <= 0) tile_elems = 1024;
    if (tile_elems > 4096) tile_elems = 4096;

    std::vector<int32_t> order;
    morton_order(conn, npe, ne, nn, coords, order);
    for (int32_t T = tile_elems; T >= 16; T = (T * 2) / 3) {
        const int rc = try_build(conn, npe, ne, nn, coords, x_src, u_src, edges, ned, T, node_cap, order, elem_order, chunk_cap,
                                 pair_block, shards, out);
        if (rc == 0 && npe == 4 && out.max_elems > kMaxQuadSlots) continue;   // QUAD4 kernel: <= 4 slots x 256 threads
        if (rc == 0 && out.paired && out.max_rows > (pair_block == 512 ? 2 : kMaxRows)) continue;   // pair kernel: <= 6 slots per thread (2 at 512 threads)
        if (rc == 0) {
            out.conn32.assign(conn, conn + npe * ne);
            out.edges32.assign(edges, edges + 2 * ned);
            out.x_src_g.resize(nn);
            out.u_src_g.resize(nn);
            for (int64_t n = 0; n < nn; ++n) {
                out.x_src_g[n] = x_src ? x_src[n] : (int32_t)n;
                out.u_src_g[n] = u_src ? u_src[n] : (int32_t)n;
            }
        }
        if (rc <= 0) return rc;
    }
    set_error("plan: could not fit a tile into 1024 local nodes / element slots (node valence too high?)");
    return -1;
}

// ---- plan blobs ----------------------------------------------------------------------------------------------------
// A HostPlan as ONE relocatable byte string (hfem_plan_serialize / hfem_plan_deserialize): a planner run costs ~1 s per
// 10^6 elements, and the ranks of a multi-GPU job (or repeated runs on one mesh) all need the same plan -- one builds,
// the others load.  Layout: "HFEMPLAN", format version, the scalar fields, every array as {count, bytes padded to 8},
// the caller's trailer (launch options of the plan), and an FNV-1a checksum of everything before it.  Native endianness
// (a blob is a cache entry of one machine, not an interchange format).
namespace {
constexpr uint32_t kBlobVersion = 1;
struct BlobScalars {
    int64_t ne, nn, ned, elem_records, node_records, n_pairs, n_chained;
    int32_t tile_elems, npe, node_stride, elem_stride, col_stride, max_nodes, max_owned, max_elems, max_edges, max_rows,
        max_chunk_elems, shards, pair_block, paired;
};
uint64_t fnv1a(const unsigned char *p, size_t n) {
    // 8 bytes per step (a word-wise variant of FNV-1a): a corruption check, not a cryptographic hash
    uint64_t h = 1469598103934665603ull;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        h = (h ^ w) * 1099511628211ull;
    }
    for (; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}
template <typename T>
void put_vec(std::vector<unsigned char> &b, const std::vector<T> &v) {
    const uint64_t n = v.size();
    const size_t nb = (size_t)n * sizeof(T), at = b.size();
    b.resize(at + 8 + ((nb + 7) & ~(size_t)7), 0);
    std::memcpy(b.data() + at, &n, 8);
    if (nb) std::memcpy(b.data() + at + 8, v.data(), nb);
}
template <typename T>
bool get_vec(const unsigned char *p, size_t n, size_t &at, std::vector<T> &v) {
    if (at + 8 > n) return false;
    uint64_t cnt;
    std::memcpy(&cnt, p + at, 8);
    if (cnt > (n - at - 8) / sizeof(T)) return false;
    const size_t nb = (size_t)cnt * sizeof(T);
    v.resize((size_t)cnt);
    if (nb) std::memcpy(v.data(), p + at + 8, nb);
    at += 8 + ((nb + 7) & ~(size_t)7);
    return at <= n;
}
}  // namespace

void serialize_host_plan(const HostPlan &h, const void *trailer, size_t trailer_bytes, std::vector<unsigned char> &b) {
    b.clear();
    b.insert(b.end(), {'H', 'F', 'E', 'M', 'P', 'L', 'A', 'N'});
    const uint32_t head[2] = {kBlobVersion, (uint32_t)trailer_bytes};
    b.insert(b.end(), (const unsigned char *)head, (const unsigned char *)head + 8);
    BlobScalars sc{};
    sc.ne = h.ne; sc.nn = h.nn; sc.ned = h.ned; sc.elem_records = h.elem_records; sc.node_records = h.node_records;
    sc.n_pairs = h.n_pairs; sc.n_chained = h.n_chained; sc.tile_elems = h.tile_elems; sc.npe = h.npe;
    sc.node_stride = h.node_stride; sc.elem_stride = h.elem_stride; sc.col_stride = h.col_stride; sc.max_nodes = h.max_nodes;
    sc.max_owned = h.max_owned; sc.max_elems = h.max_elems; sc.max_edges = h.max_edges; sc.max_rows = h.max_rows;
    sc.max_chunk_elems = h.max_chunk_elems; sc.shards = h.shards; sc.pair_block = h.pair_block; sc.paired = h.paired ? 1 : 0;
    b.insert(b.end(), (const unsigned char *)&sc, (const unsigned char *)&sc + sizeof(sc));
    b.resize((b.size() + 7) & ~(size_t)7, 0);
    std::vector<int32_t> td((const int32_t *)h.tiles.data(), (const int32_t *)h.tiles.data() + 8 * h.tiles.size());
    put_vec(b, td);
    put_vec(b, h.elem_pack); put_vec(b, h.elem_pack_hi); put_vec(b, h.elem_gid); put_vec(b, h.elem_gid_b);
    put_vec(b, h.node_src); put_vec(b, h.edge_pack); put_vec(b, h.edge_gid); put_vec(b, h.tile_chunks);
    put_vec(b, h.conn32); put_vec(b, h.x_src_g); put_vec(b, h.u_src_g); put_vec(b, h.edges32);
    put_vec(b, h.shard_desc); put_vec(b, h.owned_gid);
    const size_t at = b.size();
    b.resize(at + ((trailer_bytes + 7) & ~(size_t)7), 0);
    if (trailer_bytes) std::memcpy(b.data() + at, trailer, trailer_bytes);
    const uint64_t sum = fnv1a(b.data(), b.size());
    b.insert(b.end(), (const unsigned char *)&sum, (const unsigned char *)&sum + 8);
}

int deserialize_host_plan(const void *blob, size_t n, HostPlan &h, void *trailer, size_t trailer_bytes) {
    const unsigned char *p = (const unsigned char *)blob;
    if (!p || n < 24 + sizeof(BlobScalars) || std::memcmp(p, "HFEMPLAN", 8) != 0) { set_error("plan blob: not a plan blob"); return -1; }
    uint32_t head[2];
    std::memcpy(head, p + 8, 8);
    if (head[0] != kBlobVersion) { set_error("plan blob: written by another format version"); return -1; }
    if (head[1] != trailer_bytes) { set_error("plan blob: written by another library version (trailer size)"); return -1; }
    uint64_t sum;
    std::memcpy(&sum, p + n - 8, 8);
    if (sum != fnv1a(p, n - 8)) { set_error("plan blob: checksum mismatch (truncated or corrupted)"); return -1; }
    size_t at = 16;
    BlobScalars sc;
    std::memcpy(&sc, p + at, sizeof(sc));
    at = (at + sizeof(sc) + 7) & ~(size_t)7;
    h = HostPlan();
    h.ne = sc.ne; h.nn = sc.nn; h.ned = sc.ned; h.elem_records = sc.elem_records; h.node_records = sc.node_records;
    h.n_pairs = sc.n_pairs; h.n_chained = sc.n_chained; h.tile_elems = sc.tile_elems; h.npe = sc.npe;
    h.node_stride = sc.node_stride; h.elem_stride = sc.elem_stride; h.col_stride = sc.col_stride; h.max_nodes = sc.max_nodes;
    h.max_owned = sc.max_owned; h.max_elems = sc.max_elems; h.max_edges = sc.max_edges; h.max_rows = sc.max_rows;
    h.max_chunk_elems = sc.max_chunk_elems; h.shards = sc.shards; h.pair_block = sc.pair_block; h.paired = sc.paired != 0;
    std::vector<int32_t> td;
    bool ok = get_vec(p, n - 8, at, td) && td.size() % 8 == 0;
    ok = ok && get_vec(p, n - 8, at, h.elem_pack) && get_vec(p, n - 8, at, h.elem_pack_hi) && get_vec(p, n - 8, at, h.elem_gid) &&
         get_vec(p, n - 8, at, h.elem_gid_b) && get_vec(p, n - 8, at, h.node_src) && get_vec(p, n - 8, at, h.edge_pack) &&
         get_vec(p, n - 8, at, h.edge_gid) && get_vec(p, n - 8, at, h.tile_chunks) && get_vec(p, n - 8, at, h.conn32) &&
         get_vec(p, n - 8, at, h.x_src_g) && get_vec(p, n - 8, at, h.u_src_g) && get_vec(p, n - 8, at, h.edges32) &&
         get_vec(p, n - 8, at, h.shard_desc) && get_vec(p, n - 8, at, h.owned_gid);
    ok = ok && at + ((trailer_bytes + 7) & ~(size_t)7) == n - 8;
    if (!ok) { set_error("plan blob: malformed"); return -1; }
    h.tiles.resize(td.size() / 8);
    if (!td.empty()) std::memcpy((void *)h.tiles.data(), td.data(), td.size() * 4);
    if (trailer_bytes) std::memcpy(trailer, p + at, trailer_bytes);
    // structural checks a kernel relies on: the blob passed its checksum, but it may come from a build with other limits
    const size_t nt = h.tiles.size();
    if ((h.npe != 3 && h.npe != 4) || h.node_stride < 0 || h.elem_stride < 0 || h.max_nodes > kMaxLocal ||
        h.node_src.size() < 2 * (nt * (size_t)h.node_stride) || h.elem_pack.size() < nt * (size_t)h.elem_stride ||
        ((h.npe == 4 || h.paired) && h.elem_pack_hi.size() != h.elem_pack.size()) || (int64_t)h.conn32.size() != h.npe * h.ne ||
        (int64_t)h.x_src_g.size() != h.nn || (int64_t)h.u_src_g.size() != h.nn || (int64_t)h.edges32.size() != 2 * h.ned ||
        h.shard_desc.size() != 4 * (size_t)std::max(1, h.shards)) {
        set_error("plan blob: inconsistent array sizes");
        return -1;
    }
    for (const TileDesc &d : h.tiles)
        if (d.n_node < 0 || d.n_node > h.max_nodes || d.n_owned < 0 || d.n_owned > d.n_node || d.n_elem < 0 || d.n_elem > h.max_elems ||
            d.node_off < 0 || (size_t)d.node_off + d.n_node > h.node_src.size() / 2 || d.elem_off < 0 ||
            (size_t)d.elem_off + d.n_elem > h.elem_pack.size() || d.n_edge < 0 || d.edge_off < 0 ||
            (size_t)d.edge_off + d.n_edge > h.edge_pack.size()) {
            set_error("plan blob: tile descriptor out of range");
            return -1;
        }
    return 0;
}

}  // namespace hfem

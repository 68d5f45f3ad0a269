// host_logic.hpp — the HOST-side decisions of a verify call that touch no device: how a mixed batch is bucketed by
// shape, and how the per-query stages are cut into groups that fit the workspace budget.  Plain C++ (no HIP): included by
// verify_api.inc, and compiled on its own by g++ with AddressSanitizer + UBSan for tests/host_logic_asan.cpp, which drives
// it from recorded and adversarial shape arrays — the host code of the mixed-batch path had never seen a sanitizer
// before round 4 (the r3a crash pointed at exactly this code or at the test's own threads).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "layout.hpp"

namespace rsv::host {

// workspace carving: 256-byte aligned sub-allocations of one buffer (base == nullptr: size probe)
struct Carve {
    char* base;
    size_t off = 0;
    template <class T> T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += sizeof(T) * count;
        return p;
    }
};

// One launch geometry: every proof of a bucket has n_queries = G.  maxM / maxInner / minLevel bound its trees.
struct Bucket { uint32_t G, maxM, maxInner, minLevel, count; size_t first; };

// The parser's two shape words per proof (k_parse.hpp): word0 = n_queries | M << 8 | n_inner << 16 | min level << 24 (0: the
// proof was rejected, no lane anywhere), word1 = the tree geometry.  Mixed batch: one bucket per n_queries; inside a bucket
// the slots are ordered by shape class, so that the lanes of a wavefront (and nearly every workgroup) walk trees of one
// geometry.  ids: slot -> proof.  cls / scratch: caller-owned (reused call after call).
inline void bucket_by_shape(const uint32_t* shape, uint32_t N, std::vector<Bucket>& buckets, std::vector<uint32_t>& ids,
                            std::vector<uint32_t>& cls) {
    struct Class { uint32_t w0, w1, count; size_t start; };
    std::vector<Class> classes;
    std::unordered_map<uint64_t, uint32_t> class_of;  // hashed lookup: an adversarial batch may carry thousands of distinct shapes
    buckets.clear();
    cls.resize(N);
    uint32_t last_w0 = 0, last_w1 = 0, last_c = 0;
    for (uint32_t p = 0; p < N; p++) {
        const uint32_t w0 = shape[2 * (size_t)p], w1 = shape[2 * (size_t)p + 1];
        if (!w0) { cls[p] = 0xFFFFFFFFu; continue; }
        uint32_t ci;
        if (!classes.empty() && w0 == last_w0 && w1 == last_w1) ci = last_c;
        else {
            auto it = class_of.find(((uint64_t)w0 << 32) | w1);
            if (it != class_of.end()) ci = it->second;
            else {
                ci = (uint32_t)classes.size();
                class_of.emplace(((uint64_t)w0 << 32) | w1, ci);
                classes.push_back({w0, w1, 0, 0});
            }
            last_w0 = w0; last_w1 = w1; last_c = ci;
        }
        cls[p] = ci;
        classes[ci].count++;
    }
    // class order: by n_queries, then by the shape words
    std::vector<uint32_t> order(classes.size());
    for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        const Class &a = classes[x], &b = classes[y];
        if ((a.w0 & 0xFFu) != (b.w0 & 0xFFu)) return (a.w0 & 0xFFu) < (b.w0 & 0xFFu);
        if (a.w0 != b.w0) return a.w0 < b.w0;
        return a.w1 < b.w1;
    });
    size_t at = 0;
    for (uint32_t ci : order) {
        Class& c2 = classes[ci];
        c2.start = at;
        at += c2.count;
        const uint32_t q = c2.w0 & 0xFFu;
        if (buckets.empty() || buckets.back().G != q) buckets.push_back({q, 0, 0, 255, 0, c2.start});
        Bucket& b = buckets.back();
        b.maxM = std::max(b.maxM, (c2.w0 >> 8) & 0xFFu);
        b.maxInner = std::max(b.maxInner, (c2.w0 >> 16) & 0xFFu);
        b.minLevel = std::min(b.minLevel, c2.w0 >> 24);
        b.count += c2.count;
    }
    ids.resize(at);
    std::vector<size_t> fill(classes.size());
    for (size_t i = 0; i < classes.size(); i++) fill[i] = classes[i].start;
    for (uint32_t p = 0; p < N; p++)
        if (cls[p] != 0xFFFFFFFFu) ids[fill[cls[p]]++] = p;
}

// One (bucket, slot range) of a group: cn slots of bucket bi from slot c0 on, launched with G lanes per proof, the dense
// top-of-tree cap from level Lc down.  Lt: the level at which the tree kernels hand their nodes over to the cap kernels
// (0: the whole cap inside the tree kernels); Lt2 <= Lt: the level from which k_cap_top (one lane per tree) walks to the
// root; the levels between Lt and Lt2 belong to k_cap_mid (one lane per subtree).
struct Entry { size_t bi; size_t c0; uint32_t cn, G, Lc, Lt, Lt2; size_t bytes; };

struct GroupPolicy {
    size_t budget;         // bytes of per-query workspace a group may use
    size_t max_fused;      // entries per launch (Fused<>)
    bool tree_cap_off;     // RSV_OPT_TREE_CAP = 2
    bool flow;             // the pass writes PoseidonFlow records
    bool flow_cap_off;     // RSV_OPT_FLOW_CAP = 2
    bool cap_top;          // the last levels of the cap in k_cap_top
    int cap_mid = 0;       // RSV_OPT_CAP_MID: 0 by the fill of the in-kernel cap levels, 1 hand over at the cap level wherever the cap kernels can take it, 2 never
};

// lanes per proof of a bucket's launches: tiny query counts are padded so that a workgroup holds at most 64 proofs
inline uint32_t padded_lanes(const Bucket& b) { return b.G < 4 ? 4u : b.G; }

// Cap levels of a bucket.  Lc: 2^Lc <= G, strictly below every tree's lowest leaf / data level (0: no cap).  Inside the
// tree kernels a cap level l keeps (proofs per workgroup) x 2^l lanes busy, each part-filled wave at the price of a full
// wave-level permutation: 16-query proofs (16 per workgroup) fill whole waves at levels 3 and 2 and leave only the last
// two levels to k_cap_top (Lt = Lt2 = 2); three 80-query proofs put 96, 48 and 24 lanes into 2 + 1 + 1 waves (256 slots
// for 168 nodes), nine 27-query proofs 72 and 36 lanes into 2 + 1 waves, 25 ten-query proofs 100 lanes into 2.  Such a
// bucket hands its level-Lc nodes over (Lt = Lc) and the cap kernels, whose lanes are dealt over the whole launch, take
// it from there: k_cap_mid a subtree per lane down to level Lt2, k_cap_top the rest (k_merkle.hpp).
// hand-over at the cap level: k_cap_top holds 2^Lt2 <= 8 nodes in registers, k_cap_mid walks Lt - Lt2 = 2 or 3 levels
inline void handover_levels(uint32_t Lc, uint32_t& Lt, uint32_t& Lt2) {
    Lt = Lc;
    Lt2 = Lc <= 3 ? Lc : (Lc >= 6 ? 3u : 2u);
}
inline void cap_levels(const Bucket& b, const GroupPolicy& pol, uint32_t block, uint32_t& Lc, uint32_t& Lt, uint32_t& Lt2) {
    const uint32_t G = padded_lanes(b);
    Lc = 0;
    while ((2u << Lc) <= G) Lc++;
    if (b.minLevel < 2) Lc = 0;
    else if (Lc > b.minLevel - 1) Lc = b.minLevel - 1;
    if (Lc > 6) Lc = 6;
    if (Lc < 2) Lc = 0;
    if (pol.tree_cap_off) Lc = 0;
    if (pol.flow && pol.flow_cap_off) Lc = 0;  // every lane walks (and records) its whole path
    Lt = Lt2 = (pol.cap_top && !pol.flow && Lc >= 3) ? (Lc >= 6 ? 3u : 2u) : 0u;  // the cap kernels write no records
    if (!Lt || pol.cap_mid == 2) return;
    // wave slots of the in-kernel cap levels against the nodes they hold
    const uint32_t pb = std::min<uint32_t>(block / G, 64u);
    uint32_t slots = 0, nodes = 0;
    for (uint32_t l = Lt; l < Lc; l++) { slots += ((pb << l) + 63u) / 64u * 64u; nodes += pb << l; }
    if (pol.cap_mid == 1 || slots * 100u > nodes * 115u) handover_levels(Lc, Lt, Lt2);
}

// per-query workspace of cn slots of a bucket (what verify_impl carves for one Entry)
inline size_t entry_bytes(const Bucket& b, uint32_t G, size_t cn, uint32_t Lt, uint32_t Lt2) {
    Carve probe{nullptr};
    probe.take<PlanHdr>(cn);
    probe.take<uint32_t>(cn * (b.maxM + 1) * G);
    probe.take<uint32_t>(cn * 2 * G);
    probe.take<uint32_t>(cn * (3 + b.maxInner) * G * 8);
    for (uint32_t L : {Lt, Lt2 != Lt ? Lt2 : 0u})
        if (L) {
            probe.take<uint32_t>((cn * 4 << L) * 8);
            probe.take<unsigned long long>(cn * 4);
            probe.take<uint32_t>((cn * (1 + b.maxInner) << L) * 8);
            probe.take<unsigned long long>(cn * (1 + b.maxInner));
        }
    return ((probe.off + 255) & ~(size_t)255) + 256;
}

// The batch is cut into GROUPS of at most max_fused entries whose workspaces fit the budget together; a group is ONE launch
// per stage.  A uniform batch is one entry (several groups only when it exceeds the budget), a mixed batch normally one
// group with an entry per n_queries.  Returns the workspace bytes the largest group needs.
inline size_t plan_groups(const std::vector<Bucket>& buckets, const GroupPolicy& pol, std::vector<std::vector<Entry>>& groups) {
    groups.clear();
    size_t need = 0;
    std::vector<Entry> cur;
    size_t used = 0;
    auto flush = [&]() { if (!cur.empty()) { groups.push_back(cur); need = std::max(need, used); cur.clear(); used = 0; } };
    // widest buckets first: their workgroups are the longest-running, and a launch's tail should be short ones
    for (size_t bk = buckets.size(); bk-- > 0;) {
        const size_t bi = bk;
        const Bucket& b = buckets[bi];
        const uint32_t G = padded_lanes(b);
        uint32_t Lc, Lt, Lt2;
        cap_levels(b, pol, 256u, Lc, Lt, Lt2);
        const size_t per_slot = entry_bytes(b, G, 1024, Lt, Lt2) / 1024 + 1;
        for (size_t c0 = 0; c0 < b.count;) {
            size_t room = pol.budget > used ? (pol.budget - used) / per_slot : 0;
            size_t cn = std::min<size_t>(b.count - c0, room);
            if (cn < std::min<size_t>(b.count - c0, 1024)) {  // not worth a sliver: start a new group
                if (!cur.empty()) { flush(); continue; }
                cn = std::min<size_t>(b.count - c0, 1024);       // a budget below 1 024 slots: take them anyway
            }
            Entry en{bi, c0, (uint32_t)cn, G, Lc, Lt, Lt2, entry_bytes(b, G, cn, Lt, Lt2)};
            cur.push_back(en);
            used += en.bytes;
            c0 += cn;
            if (cur.size() == pol.max_fused) flush();
        }
    }
    flush();
    return need;
}

// Which FRI tree each grid row (blockIdx.y) of a SMALL k_pair_merkle launch hashes.  A launch whose workgroups are all
// resident at once is not balanced by the dispatcher: workgroup i lands on compute unit i mod n_cu (round robin over the
// XCDs, then inside them), so with `blocks` workgroups per grid row the rows pile up on the same compute units — 1 024
// proofs of the standard shape are 64 x 9 workgroups: the first 64 compute units get rows 0, 4 and 8 (the first-layer tree
// and two more: 67 path steps per SIMD), the others two rows (40-44), and the kernel lasts as long as the fullest.  The
// trees differ in depth (first layer M + columns, inner layer i: M - 1 - i), so the host deals them out: compute units
// that receive more rows get the shallowest trees, the rest longest-first onto the least loaded (LPT).  Identity when
// the launch is larger than the machine holds (the dispatcher then balances by itself, longest trees first).
// live: grid rows that have work (a single-configuration launch is sized for the deepest tree the parser admits, MAX_LOG: its
// rows beyond the batch's real FRI layers exit at once and weigh nothing).
inline void pair_layer_order(uint32_t blocks, uint32_t layers, uint32_t live, uint32_t M, uint32_t n_cu, uint32_t resident_per_cu, uint8_t* y_of) {
    for (uint32_t k = 0; k < 32; k++) y_of[k] = (uint8_t)k;
    if (live > layers) live = layers;
    if (layers < 3 || live < 3 || layers > 32 || blocks == 0 || n_cu == 0 || (size_t)blocks * live > (size_t)n_cu * resident_per_cu || blocks >= n_cu) return;
    // path steps of tree y (relative weights are what matters)
    std::vector<uint32_t> len(layers);
    for (uint32_t y = 0; y < layers; y++) len[y] = y >= live ? 0u : (y == 0 ? M + 8 : (M > y ? M - y : 1) + 4);
    // rows per compute unit, and the compute units each grid position covers
    std::vector<uint32_t> rows_of_cu(n_cu, 0), load(n_cu, 0);
    auto first_cu = [&](uint32_t pos) { return (uint32_t)(((size_t)pos * blocks) % n_cu); };
    for (uint32_t pos = 0; pos < layers; pos++)
        for (uint32_t x = 0; x < blocks; x++) rows_of_cu[(first_cu(pos) + x) % n_cu]++;
    std::vector<uint8_t> pos_used(layers, 0), tree_used(layers, 0);
    std::vector<uint32_t> by_len(layers);
    for (uint32_t y = 0; y < layers; y++) by_len[y] = y;
    std::sort(by_len.begin(), by_len.end(), [&](uint32_t a, uint32_t b) { return len[a] != len[b] ? len[a] > len[b] : a < b; });  // longest first
    auto assign = [&](uint32_t pos, uint32_t tree) {
        y_of[pos] = (uint8_t)tree; pos_used[pos] = 1; tree_used[tree] = 1;
        for (uint32_t x = 0; x < blocks; x++) load[(first_cu(pos) + x) % n_cu] += len[tree];
    };
    auto max_rows = [&](uint32_t pos) { uint32_t m = 0; for (uint32_t x = 0; x < blocks; x++) m = std::max(m, rows_of_cu[(first_cu(pos) + x) % n_cu]); return m; };
    uint32_t min_rows = 0xFFFFFFFFu;
    for (uint32_t c = 0; c < n_cu; c++) if (rows_of_cu[c]) min_rows = std::min(min_rows, rows_of_cu[c]);
    // phase A: positions over compute units with more rows than the others take the shallowest trees, one per surplus row
    std::vector<uint32_t> surplus_taken(n_cu, 0);
    for (uint32_t pos = layers; pos-- > 0;) {
        const uint32_t c0 = first_cu(pos);
        if (max_rows(pos) > min_rows && surplus_taken[c0] + min_rows < rows_of_cu[c0]) {
            uint32_t tree = 0xFFFFFFFFu;
            for (uint32_t k = layers; k-- > 0;) if (!tree_used[by_len[k]]) { tree = by_len[k]; break; }  // shallowest left
            assign(pos, tree);
            for (uint32_t x = 0; x < blocks; x++) surplus_taken[(c0 + x) % n_cu]++;
        }
    }
    // phase B: the rest longest-first onto the free position whose compute units carry least
    for (uint32_t k = 0; k < layers; k++) {
        const uint32_t tree = by_len[k];
        if (tree_used[tree]) continue;
        uint32_t best = 0xFFFFFFFFu, best_load = 0xFFFFFFFFu;
        for (uint32_t pos = 0; pos < layers; pos++) {
            if (pos_used[pos]) continue;
            uint32_t m = 0;
            for (uint32_t x = 0; x < blocks; x++) m = std::max(m, load[(first_cu(pos) + x) % n_cu]);
            if (m < best_load) { best_load = m; best = pos; }
        }
        assign(best, tree);
    }
}

}  // namespace rsv::host

// host_logic_asan.cpp — TEST INFRASTRUCTURE: the product's host-only logic under AddressSanitizer + UBSan, in the CPU
// container (no HIP, no device, g++).  Built and driven by tests/test_host_logic_asan.py.
//
//   host_logic_asan program <hints file> <out file>   circuit_program.hpp: parse_template + build_program (the C++ mirror of
//                                                     the reference's gadgets, ~1 400 lines) on hints recorded on the CPU, then
//                                                     check_program on the result and on mutated copies of it; the program is
//                                                     written out for comparison with the Python restatement's
//   host_logic_asan buckets <seed> <rounds>           host_logic.hpp: bucket_by_shape + plan_groups on seeded shape arrays —
//                                                     recorded fixture shapes, garbage, thousands of distinct shapes, rejected
//                                                     proofs, 1 .. 128 queries — with the invariants the launcher relies on
//
// The hints file: 16 header words (magic, proof bytes, nq, M, n_inner, flow_count, n_pi, copies, 0...), then the proof
// (padded to words), trace_sib, trace_pos, trace_cols, fri_sib, fri_cols, flow [flow_count][32], swap (bytes, padded),
// public inputs [n_pi][5], set_walks (bytes, padded).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../recursive-stwo_amd/csrc/host_logic.hpp"
#include "../recursive-stwo_amd/csrc/circuit_program.hpp"

#define CHECK(c)                                                              \
    do {                                                                      \
        if (!(c)) {                                                           \
            fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

using namespace rsv;

static std::vector<uint32_t> read_words(const char* path) {
    FILE* f = fopen(path, "rb");
    CHECK(f);
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    CHECK(n > 0 && n % 4 == 0);
    std::vector<uint32_t> w((size_t)n / 4);
    CHECK(fread(w.data(), 4, w.size(), f) == w.size());
    fclose(f);
    return w;
}

static int run_program(const char* in_path, const char* out_path) {
    const std::vector<uint32_t> file = read_words(in_path);
    CHECK(file.size() > 16 && file[0] == 0x52535648u);
    const size_t len = file[1], nq = file[2], M = file[3], n_inner = file[4], flow_count = file[5], n_pi = file[6], copies = file[7];
    const size_t nt = 1 + n_inner;
    size_t at = 16;
    auto take = [&](size_t words) { const uint32_t* p = file.data() + at; at += words; CHECK(at <= file.size()); return p; };
    const uint32_t* proof_w = take((len + 3) / 4);
    const uint32_t* trace_sib = take(4 * nq * M * 8);
    const uint32_t* trace_pos = take(4 * nq);
    const uint32_t* trace_cols = take(4 * nq * 64);
    const uint32_t* fri_sib = take(nt * nq * M * 8);
    const uint32_t* fri_cols = take(nt * nq * 24);
    const uint32_t* flow = take(flow_count * 32);
    const uint8_t* swap = reinterpret_cast<const uint8_t*>(take((flow_count + 3) / 4));
    const uint32_t* pi = take(n_pi * 5);
    const uint8_t* walks = reinterpret_cast<const uint8_t*>(take((copies + 3) / 4));
    CHECK(at == file.size());
    // exact-size heap copies, so that the sanitizer sees every read past an array's end
    std::vector<uint32_t> v_tsib(trace_sib, trace_sib + 4 * nq * M * 8), v_tpos(trace_pos, trace_pos + 4 * nq),
        v_tcols(trace_cols, trace_cols + 4 * nq * 64), v_fsib(fri_sib, fri_sib + nt * nq * M * 8), v_fcols(fri_cols, fri_cols + nt * nq * 24),
        v_flow(flow, flow + flow_count * 32), v_proof(proof_w, proof_w + (len + 3) / 4);
    std::vector<uint8_t> v_swap(swap, swap + flow_count), v_walks(walks, walks + copies);

    circuit::Template d{};
    const uint8_t* proof = reinterpret_cast<const uint8_t*>(v_proof.data());
    CHECK(circuit::parse_template(proof, len, d) == RSV_OK);
    CHECK(d.nq == nq && d.M == M && d.n_inner == n_inner);
    // malformed templates are refused, never read out of bounds: every truncation, a misaligned buffer, absurd header words
    for (size_t cut : {(size_t)0, (size_t)4, (size_t)64, len / 2, len - 4, len - 1}) {
        std::vector<uint8_t> part(proof, proof + cut);
        circuit::Template t{};
        CHECK(circuit::parse_template(part.data(), cut, t) != RSV_OK || cut == len);
    }
    {
        std::vector<uint8_t> shifted(len + 8);
        memcpy(shifted.data() + 1, proof, len);
        circuit::Template t{};
        CHECK(circuit::parse_template(shifted.data() + 1, len, t) == RSV_E_SIZE);
        for (uint32_t word : {0u, 1u, 10u, 11u, 12u, 13u}) {
            std::vector<uint32_t> bad(v_proof);
            bad[word] = 0xFFFFFFFFu;
            circuit::Template t2{};
            (void)circuit::parse_template(reinterpret_cast<const uint8_t*>(bad.data()), len, t2);  // any status, no bad access
        }
    }
    d.trace_sib = v_tsib.data(); d.trace_pos = v_tpos.data(); d.trace_cols = v_tcols.data();
    d.fri_sib = v_fsib.data(); d.fri_cols = v_fcols.data();
    std::vector<std::pair<uint32_t, circuit::Q4>> inputs;
    for (size_t i = 0; i < n_pi; i++) inputs.push_back({pi[5 * i], circuit::Q4{pi[5 * i + 1], pi[5 * i + 2], pi[5 * i + 3], pi[5 * i + 4]}});
    circuit::BuiltProgram bp;
    CHECK(circuit::build_program(d, v_flow.data(), v_swap.data(), (uint32_t)flow_count, inputs, (uint32_t)copies, v_walks.data(), bp) == RSV_OK);
    // hints that do not belong to the template make the gadgets' own checks fail: an error status, never a crash
    {
        std::vector<uint32_t> bad_flow(v_flow);
        bad_flow[bad_flow.size() / 2] ^= 1;
        circuit::BuiltProgram b2;
        CHECK(circuit::build_program(d, bad_flow.data(), v_swap.data(), (uint32_t)flow_count, inputs, (uint32_t)copies, v_walks.data(), b2) == RSV_E_RANGE);
        circuit::BuiltProgram b3;  // a flow that is too short for the circuit
        CHECK(circuit::build_program(d, v_flow.data(), v_swap.data(), (uint32_t)flow_count - 1, inputs, (uint32_t)copies, v_walks.data(), b3) == RSV_E_RANGE);
    }
    const size_t n_levels = bp.level_offsets.size() - 1;
    rsv_witness_shape shape{d.lp, d.lq, d.pow_bits, d.blowup, d.log_last, d.nq, d.n_inner, (uint32_t)flow_count, (uint32_t)copies};
    CHECK(circuit::check_program(bp.instr.data(), bp.n_vars, bp.level_offsets.data(), n_levels, (uint32_t)bp.n_vars, shape) == RSV_OK);
    // mutated programs: every field of seeded instructions pushed out of range, the level table broken
    std::mt19937 rng(7);
    size_t refused = 0, tried = 0;
    for (int k = 0; k < 400; k++) {
        std::vector<uint32_t> bad(bp.instr);
        const size_t i = rng() % bp.n_vars;
        const int field = (int)(rng() % 8);
        const uint32_t vals[] = {0xFFFFFFFFu, (uint32_t)bp.n_vars, (uint32_t)bp.n_vars + 1, 0x7FFFFFFFu, 64u, 1u << 20};
        bad[i * 8 + field] = vals[rng() % 6];
        const int rc = circuit::check_program(bad.data(), bp.n_vars, bp.level_offsets.data(), n_levels, (uint32_t)bp.n_vars, shape);
        tried++;
        refused += rc != RSV_OK;
        // an opcode or a destination out of range is ALWAYS refused (an operand or an immediate may be unused by the op)
        if (field <= 1 && bad[i * 8 + field] >= bp.n_vars) CHECK(rc != RSV_OK);
    }
    CHECK(refused >= tried / 8);
    {
        std::vector<uint32_t> lv(bp.level_offsets);
        lv[n_levels / 2] = lv[n_levels / 2 + 1] + 1;  // not monotone
        CHECK(circuit::check_program(bp.instr.data(), bp.n_vars, lv.data(), n_levels, (uint32_t)bp.n_vars, shape) != RSV_OK);
        lv = bp.level_offsets;
        lv[n_levels] += 1;                             // does not cover the program
        CHECK(circuit::check_program(bp.instr.data(), bp.n_vars, lv.data(), n_levels, (uint32_t)bp.n_vars, shape) != RSV_OK);
        rsv_witness_shape s2 = shape;
        s2.n_queries = 3;
        CHECK(circuit::check_program(bp.instr.data(), bp.n_vars, bp.level_offsets.data(), n_levels, (uint32_t)bp.n_vars, s2) != RSV_OK);
    }
    FILE* f = fopen(out_path, "wb");
    CHECK(f);
    const uint32_t hdr[8] = {0x52535650u, (uint32_t)bp.n_vars, (uint32_t)n_levels, (uint32_t)bp.flow_wires.size(), (uint32_t)bp.gates.size(),
                             (uint32_t)bp.witness_ops.size(), 0, 0};
    fwrite(hdr, 4, 8, f);
    fwrite(bp.instr.data(), 4, bp.instr.size(), f);
    fwrite(bp.level_offsets.data(), 4, bp.level_offsets.size(), f);
    fwrite(bp.flow_wires.data(), 4, bp.flow_wires.size(), f);
    fwrite(bp.gates.data(), 4, bp.gates.size(), f);
    fwrite(bp.witness_ops.data(), 4, bp.witness_ops.size(), f);
    fclose(f);
    printf("program: %zu variables, %zu levels, %zu rows, %zu of %zu mutated programs refused\n", bp.n_vars, n_levels, bp.gates.size() / 6,
           refused, tried);
    return 0;
}

// shape word 0 of a proof as k_parse writes it
static uint32_t w0_of(uint32_t nq, uint32_t M, uint32_t n_inner, uint32_t min_level) { return nq | (M << 8) | (n_inner << 16) | (min_level << 24); }

static void check_buckets(const std::vector<uint32_t>& shape, uint32_t N, size_t budget, size_t max_fused, bool cap_top) {
    std::vector<host::Bucket> buckets;
    std::vector<uint32_t> ids, cls;
    host::bucket_by_shape(shape.data(), N, buckets, ids, cls);
    // ids: every live proof exactly once; buckets: a partition of the slots by n_queries, ascending, bounds of their members
    size_t live = 0;
    for (uint32_t p = 0; p < N; p++) live += shape[2 * (size_t)p] != 0;
    CHECK(ids.size() == live && cls.size() == N);
    std::vector<uint8_t> seen(N, 0);
    for (uint32_t p : ids) { CHECK(p < N && !seen[p] && shape[2 * (size_t)p] != 0); seen[p] = 1; }
    size_t at = 0;
    uint32_t prev_g = 0;
    for (const host::Bucket& b : buckets) {
        CHECK(b.first == at && b.count > 0 && b.G > prev_g);
        prev_g = b.G;
        for (size_t s = b.first; s < b.first + b.count; s++) {
            const uint32_t w0 = shape[2 * (size_t)ids[s]];
            CHECK((w0 & 0xFFu) == b.G && ((w0 >> 8) & 0xFFu) <= b.maxM && ((w0 >> 16) & 0xFFu) <= b.maxInner && (w0 >> 24) >= b.minLevel);
            if (s > b.first) {  // inside a bucket: slots ordered by shape class
                const uint32_t q0 = shape[2 * (size_t)ids[s - 1]], q1 = shape[2 * (size_t)ids[s - 1] + 1];
                CHECK(q0 < w0 || (q0 == w0 && q1 <= shape[2 * (size_t)ids[s] + 1]));
            }
        }
        at += b.count;
    }
    CHECK(at == live);
    // groups: every slot of every bucket in exactly one entry; a group within the budget unless it is a single minimum entry
    std::vector<std::vector<host::Entry>> groups;
    const host::GroupPolicy pol{budget, max_fused, false, false, false, cap_top, (int)((N + budget) % 3)};  // RSV_OPT_CAP_MID: every setting over the runs
    const size_t need = host::plan_groups(buckets, pol, groups);
    std::vector<std::vector<std::pair<size_t, size_t>>> ranges(buckets.size());
    size_t worst = 0;
    for (const auto& g : groups) {
        CHECK(!g.empty() && g.size() <= max_fused);
        size_t used = 0;
        for (const host::Entry& e : g) {
            CHECK(e.bi < buckets.size() && e.cn > 0 && e.c0 + e.cn <= buckets[e.bi].count);
            CHECK(e.G >= 4 && e.G >= buckets[e.bi].G && e.Lc <= 6 && e.Lt <= e.Lc && e.Lt2 <= 3 && e.Lt2 <= e.Lt && e.Lt - e.Lt2 <= 3 && (e.Lt - e.Lt2 != 1) && (e.Lt == 0 || (e.Lc >= 3 && e.Lt2 >= 2)));
            CHECK(e.bytes == host::entry_bytes(buckets[e.bi], e.G, e.cn, e.Lt, e.Lt2));
            ranges[e.bi].push_back({e.c0, e.cn});
            used += e.bytes;
        }
        CHECK(used <= budget || g.size() == 1 || g.back().cn <= 1024);
        worst = std::max(worst, used);
    }
    for (size_t bi = 0; bi < buckets.size(); bi++) {  // every slot of every bucket in exactly one entry
        std::sort(ranges[bi].begin(), ranges[bi].end());
        size_t at = 0;
        for (auto& r : ranges[bi]) { CHECK(r.first == at); at += r.second; }
        CHECK(at == buckets[bi].count);
    }
    CHECK(need == worst);
}

static int run_buckets(uint32_t seed, int rounds) {
    std::mt19937 rng(seed);
    // the shapes of the reference's fixtures (n_queries, M, n_inner, min level) as the parser reports them
    const uint32_t fixtures[][4] = {{16, 21, 10, 8}, {16, 22, 8, 14}, {16, 23, 9, 14}, {80, 21, 12, 10}, {27, 21, 10, 12}, {11, 25, 10, 16},
                                    {10, 26, 10, 17}, {8, 27, 10, 17}, {8, 28, 11, 17}};
    for (int r = 0; r < rounds; r++) {
        const uint32_t N = r == 0 ? 1 : (r == 1 ? 53248 : 1 + rng() % 20000);
        std::vector<uint32_t> shape(2 * (size_t)N);
        const int mode = r % 5;
        for (uint32_t p = 0; p < N; p++) {
            uint32_t w0, w1;
            if (mode == 0 || mode == 1) {  // the recursion chain's mix, some proofs rejected by the parser
                const uint32_t* f = fixtures[(mode == 1 ? p / 4096 : rng()) % 9];
                w0 = w0_of(f[0], f[1], f[2], f[3]);
                w1 = f[1] * 1000 + f[2];
                if (rng() % 17 == 0) w0 = 0;
            } else if (mode == 2) {        // adversarial: (almost) every proof its own shape
                w0 = w0_of(1 + rng() % 128, 1 + rng() % 30, rng() % 29, rng() % 30);
                w1 = rng();
            } else if (mode == 3) {        // everything rejected, or one survivor
                w0 = (p == N / 2 && (r & 8)) ? w0_of(128, 30, 28, 2) : 0;
                w1 = 0;
            } else {                       // tiny query counts (padded to 4 lanes), minimum levels around the cap's threshold
                w0 = w0_of(1 + rng() % 6, 5 + rng() % 6, rng() % 4, rng() % 5);
                w1 = rng() % 3;
            }
            shape[2 * (size_t)p] = w0;
            shape[2 * (size_t)p + 1] = w1;
        }
        const size_t budgets[] = {(size_t)1 << 20, (size_t)512 << 20, (size_t)8192 << 20};
        check_buckets(shape, N, budgets[r % 3], r % 2 ? 16 : 3, (r & 4) != 0);
    }
    printf("buckets: %d rounds, invariants hold\n", rounds);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 4 && std::string(argv[1]) == "program") return run_program(argv[2], argv[3]);
    if (argc == 4 && std::string(argv[1]) == "buckets") return run_buckets((uint32_t)atoi(argv[2]), atoi(argv[3]));
    fprintf(stderr, "usage: host_logic_asan program <hints> <out> | buckets <seed> <rounds>\n");
    return 2;
}

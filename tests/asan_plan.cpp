// TEST INFRASTRUCTURE: AddressSanitizer + UBSan harness AND property check for radian_amd/csrc/plan.hip -- the host code that turns a batch of
// reads into the tile descriptors every forward kernel indexes HBM with.  Random models (1..7 blocks, any dilations), random
// read sets (one sample ... several windows), random chunk / step.  For every layer's tile list of the chunk-mode and the global-mode plan:
//   * a sub-tile lies inside its segment; what a layer WRITES for a segment lies inside the activation tensor of P.total_rows rows, and no two
//     sub-tiles of a layer write the same row; the layer's row count is the sum of its segments;
//   * what a sub-tile READS exists: samples src_row .. src_row + in_len inside the batch's signal, rows taken from the stream
//     (alt_row + t for t >= alt_in / alt_res) inside the tensor;
//   * per decoded sequence, the rows the decoder will read (off1 / off2 / split / valid) lie inside the tensor.
// A violated property is what would be an out-of-bounds access on the GPU.
#include "../radian_amd/csrc/plan.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <random>

void rd_set_error(const char* fmt, ...) { (void)fmt; }
extern "C" hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = malloc(n); return hipSuccess; }
extern "C" hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
extern "C" const char* hipGetErrorString(hipError_t) { return "stub"; }

#define CHECK(cond, ...)                                                          \
    do {                                                                          \
        if (!(cond)) {                                                            \
            printf("property violated: %s  (", #cond);                           \
            printf(__VA_ARGS__);                                                  \
            printf(")\n");                                                        \
            return 1;                                                             \
        }                                                                         \
    } while (0)

static int check_plan(const rdi::ReadsPlan& P, int64_t n_samples, bool chunk_mode, int it)
{
    using namespace rdi;
    for (int li = 0; li < P.n_layers; li++) {
        if (!P.per_layer && li > 0) break;
        const std::vector<TileDesc>& v = P.tiles[li];
        std::vector<std::pair<int64_t, int64_t>> wr;
        int64_t rows = 0;
        for (const TileDesc& d : v) {
            if (d.seg_len == 0) {            // plan_pad_tiles' filler: a sub-tile without rows
                CHECK(d.t0 == 0 && d.in_len == 0 && d.alt_in == INT32_MAX && d.alt_res == INT32_MAX, "it %d layer %d filler", it, li);
                continue;
            }
            CHECK(d.seg_len >= 1 && d.t0 >= 0 && d.t0 < d.seg_len && d.t0 % 32 == 0, "it %d layer %d t0 %d len %d", it, li, d.t0, d.seg_len);
            CHECK(d.in_len >= d.seg_len, "it %d layer %d in_len %d < seg_len %d", it, li, d.in_len, d.seg_len);
            CHECK(d.seg_row >= 0 && d.seg_row + d.seg_len <= P.total_rows, "it %d layer %d rows %lld+%d of %lld", it, li, (long long)d.seg_row, d.seg_len, (long long)P.total_rows);
            CHECK(d.src_row >= 0 && d.src_row + d.in_len <= n_samples, "it %d layer %d samples %lld+%d of %lld", it, li, (long long)d.src_row, d.in_len, (long long)n_samples);
            if (d.alt_in != INT32_MAX || d.alt_res != INT32_MAX)
                CHECK(d.alt_row >= 0 && d.alt_row + d.seg_len <= P.total_rows && d.alt_in >= 0 && d.alt_res >= 0, "it %d layer %d alt %lld", it, li, (long long)d.alt_row);
            const int hi = std::min(d.t0 + 32, d.seg_len);
            wr.push_back({d.seg_row + d.t0, d.seg_row + hi});
            if (d.t0 == 0) rows += d.seg_len;
        }
        CHECK(rows == P.rows[li], "it %d layer %d rows %lld != %lld", it, li, (long long)rows, (long long)P.rows[li]);
        std::sort(wr.begin(), wr.end());
        for (size_t i = 1; i < wr.size(); i++) CHECK(wr[i].first >= wr[i - 1].second, "it %d layer %d: two sub-tiles write row %lld", it, li, (long long)wr[i].first);
    }
    if (chunk_mode) {
        CHECK(P.off1.size() == (size_t)P.n_windows && P.off2.size() == P.off1.size() && P.split.size() == P.off1.size() && P.valid.size() == P.off1.size(), "it %d sequence arrays", it);
        for (size_t w = 0; w < P.off1.size(); w++) {
            CHECK(P.split[w] >= 0 && P.split[w] <= P.valid[w] && P.valid[w] >= 0, "it %d window %zu split %d valid %d", it, w, P.split[w], P.valid[w]);
            CHECK(P.off1[w] >= 0 && P.off1[w] + P.split[w] <= P.total_rows, "it %d window %zu head rows", it, w);
            CHECK(P.off2[w] >= 0 && P.off2[w] + P.valid[w] <= P.total_rows, "it %d window %zu stream rows %lld+%d of %lld", it, w, (long long)P.off2[w], P.valid[w], (long long)P.total_rows);
        }
    }
    CHECK((int)P.read_win_off.size() >= 1 && P.read_win_off.back() == P.n_windows, "it %d window offsets", it);
    return 0;
}

// rd_plan_trie_runs (common.h, round 6): a launch's sequences cut into runs whose beam-search workspace fits the context's budget.
// Properties: the runs tile [k_begin, k_end) in order; node offsets restart at 0 in every run and are the running sum of 1 + W * len inside
// it; a run's nodes are that sum; a run of more than one sequence fits the budget (a single sequence may exceed it: it cannot be split).
static int check_trie_runs(std::mt19937_64& rng, int it)
{
    rd_ctx ctx;
    const int W = (int[]){1, 6, 10, 25, 64, 65, 100, 128, 129, 1024}[rng() % 10];
    ctx.trie_budget = (int64_t)1 << (14 + rng() % 22);
    const int n = (int)(rng() % 300), k_begin = n ? (int)(rng() % (n + 1)) : 0;
    std::vector<int32_t> len(n);
    for (int& x : len) x = (int)(rng() % (rng() % 5 ? 1100 : 60000));
    std::vector<int64_t> off(n, -7);
    std::vector<TrieRun> runs;
    rd_plan_trie_runs(&ctx, W, k_begin, n, [&](int k) { return (int64_t)len[k]; }, off.data(), runs);
    const int64_t per_node = W > RD_LANE_MAX_W ? 24 : 20, per_seq = W > RD_LANE_MAX_W ? (int64_t)rd_wide_scratch_bytes(W) : 0;
    int next = k_begin;
    for (const TrieRun& r : runs) {
        CHECK(r.k0 == next && r.k1 > r.k0 && r.k1 <= n, "it %d: run [%d, %d) after %d of %d", it, r.k0, r.k1, next, n);
        int64_t nodes = 0;
        for (int k = r.k0; k < r.k1; k++) {
            CHECK(off[k] == nodes, "it %d: node offset %lld of sequence %d, expected %lld", it, (long long)off[k], k, (long long)nodes);
            nodes += 1 + (int64_t)W * len[k];
        }
        CHECK(nodes == r.nodes, "it %d: run nodes %lld, expected %lld", it, (long long)r.nodes, (long long)nodes);
        const int64_t bytes = nodes * per_node + (int64_t)(r.k1 - r.k0) * per_seq;
        CHECK(r.k1 - r.k0 == 1 || bytes <= ctx.trie_budget, "it %d: run of %d sequences needs %lld bytes, budget %lld", it, r.k1 - r.k0, (long long)bytes, (long long)ctx.trie_budget);
        // (greedy: the next sequence would not have fitted)
        if (r.k1 < n) {
            const int64_t more = (1 + (int64_t)W * len[r.k1]) * per_node + per_seq;
            CHECK(bytes + more > ctx.trie_budget, "it %d: run [%d, %d) ends early", it, r.k0, r.k1);
        }
        next = r.k1;
    }
    CHECK(next == n || (runs.empty() && k_begin == n), "it %d: runs end at %d of %d", it, next, n);
    for (int k = 0; k < k_begin; k++) CHECK(off[k] == -7, "it %d: offset %d outside the range was written", it, k);
    return 0;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937_64 rng(17);
    for (int it = 0; it < 4 * iters; it++)
        if (check_trie_runs(rng, it)) return 1;
    long tiles = 0;
    for (int it = 0; it < iters; it++) {
        Model m;
        m.nblocks = 1 + (int)(rng() % 7);
        int halo = 0;
        for (int b = 0; b < m.nblocks; b++) {
            m.dil[b] = 1 << (rng() % 7);
            if (rng() % 5 == 0) m.dil[b] = 1 + (int)(rng() % 70);
            halo += 4 * m.dil[b];
        }
        const int chunk = (it % 3 == 0) ? 1024 : 8 + (int)(rng() % 1500);
        const int step = (rng() % 6 == 0) ? chunk : 1 + (int)(rng() % chunk);          // the entry points require 1 <= step <= chunk
        const int n_reads = 1 + (int)(rng() % 7);
        std::vector<int64_t> off(n_reads + 1, 0);
        for (int r = 0; r < n_reads; r++) {
            int64_t N = 1 + (int64_t)(rng() % (rng() % 4 ? 3 * chunk : 12 * chunk));
            if (rng() % 9 == 0) N = chunk + (int64_t)(rng() % 3) * step;          // exact multiples
            off[r + 1] = off[r] + N;
        }
        {
            rdi::ReadsPlan P;
            if (rdi::plan_reads_chunk(m, off.data(), n_reads, chunk, step, halo, P) != 0) { printf("plan_reads_chunk failed at %d\n", it); return 1; }
            if (check_plan(P, off[n_reads], true, it)) return 1;
            const size_t total = rdi::plan_pad_tiles(P);
            for (int li = 0; li < P.n_layers; li++)
                if (P.tiles[li].size() % 8) { printf("layer %d: %zu descriptors after padding\n", li, P.tiles[li].size()); return 1; }
            if (check_plan(P, off[n_reads], true, it) == 0) tiles += (long)total; else return 1;   // (padding entries: checked as harmless below)
        }
        {
            rdi::ReadsPlan P;
            bool streamed = false;
            if (rdi::plan_reads_global(m, off.data(), n_reads, chunk, step, halo, P, &streamed) != 0) { printf("plan_reads_global failed at %d\n", it); return 1; }
            if (check_plan(P, off[n_reads], false, it)) return 1;
        }
    }
    printf("%d geometries, %ld tile descriptors, every property holds, no sanitizer report\n", iters, tiles);
    return 0;
}

// stitch.hip -- host code: the chunk-mode fragment stitch, radian/sequence_assembly.py:19-48 (simple_assembly, add_count)
// + radian/basecall.py:122-123 (argmax of the vote matrix), for a batch of reads, on the host's cores.
//
// The reference lays fragment i against fragment i-1 at the FIRST LONGEST matching block that Python's
// difflib.SequenceMatcher(None, a, b).get_matching_blocks() reports, counts one vote per base and column, and takes the
// per-column argmax.  Everything here restates that, difflib included (CPython 3.10 Lib/difflib.py: __chain_b with the
// autojunk "popular element" rule for len(b) >= 200, find_longest_match, get_matching_blocks with its LIFO work list, sort
// and adjacent-block collapse), because the pure-Python difflib is what limits the chunk-mode driver once fragments are
// real (~200 bases per window): 4.5 ms per read per interpreter against ~40 us of GPU time.  Bit-exactness is pinned by the
// reference's golden cases and by randomised comparison with difflib itself (tests/test_host_cpu.py).
//
// No GPU is touched; the file is part of libradian_hip.so so that the host side stays one ctypes binding.
#include "common.h"
#include "../../include/radian_hip.h"

#include <algorithm>
#include <atomic>
#include <thread>

namespace {

struct Block {
    int a, b, size;
};

// difflib.SequenceMatcher(None, a, b) for sequences over {0, 1, 2, 3}
struct Matcher {
    const uint8_t* a;
    const uint8_t* b;
    int la, lb;
    std::vector<int> b2j[4];         // indices of each element in b, increasing; empty when the element is "popular"
    std::vector<int> j2len, newj2len, touched_old, touched_new;

    void chain_b()
    {
        for (int e = 0; e < 4; e++) b2j[e].clear();
        for (int j = 0; j < lb; j++) b2j[b[j] & 3].push_back(j);
        if (lb >= 200) {             // autojunk: elements that occur more than 1 + len(b) // 100 times are dropped from b2j
            const int ntest = lb / 100 + 1;
            for (int e = 0; e < 4; e++)
                if ((int)b2j[e].size() > ntest) b2j[e].clear();
        }
        j2len.assign(lb + 1, 0);     // j2len[j + 1] = length of the match ending at a[i-1], b[j]  (index 0 = "j - 1 = -1")
        newj2len.assign(lb + 1, 0);
    }

    Block find_longest_match(int alo, int ahi, int blo, int bhi)
    {
        int besti = alo, bestj = blo, bestsize = 0;
        touched_old.clear();
        for (int i = alo; i < ahi; i++) {
            touched_new.clear();
            for (int j : b2j[a[i] & 3]) {
                if (j < blo) continue;
                if (j >= bhi) break;
                const int k = j2len[j] + 1;          // j2len.get(j - 1, 0) + 1   (entry j holds index j - 1)
                newj2len[j + 1] = k;
                touched_new.push_back(j + 1);
                if (k > bestsize) {
                    besti = i - k + 1;
                    bestj = j - k + 1;
                    bestsize = k;
                }
            }
            for (int t : touched_old) j2len[t] = 0;  // j2len = newj2len (a dict in Python: only this row's entries exist)
            j2len.swap(newj2len);
            touched_old.swap(touched_new);
        }
        for (int t : touched_old) j2len[t] = 0;
        // extend by matching non-junk elements on both sides (there is no junk; popular elements count here)
        while (besti > alo && bestj > blo && a[besti - 1] == b[bestj - 1]) {
            besti--;
            bestj--;
            bestsize++;
        }
        while (besti + bestsize < ahi && bestj + bestsize < bhi && a[besti + bestsize] == b[bestj + bestsize]) bestsize++;
        return {besti, bestj, bestsize};
    }

    // -> (a - b) of the first longest block of get_matching_blocks()
    int displacement()
    {
        chain_b();
        struct Q {
            int alo, ahi, blo, bhi;
        };
        std::vector<Q> queue;
        std::vector<Block> blocks;
        queue.push_back({0, la, 0, lb});
        while (!queue.empty()) {
            const Q q = queue.back();
            queue.pop_back();
            const Block m = find_longest_match(q.alo, q.ahi, q.blo, q.bhi);
            if (m.size) {
                blocks.push_back(m);
                if (q.alo < m.a && q.blo < m.b) queue.push_back({q.alo, m.a, q.blo, m.b});
                if (m.a + m.size < q.ahi && m.b + m.size < q.bhi) queue.push_back({m.a + m.size, q.ahi, m.b + m.size, q.bhi});
            }
        }
        std::sort(blocks.begin(), blocks.end(), [](const Block& x, const Block& y) {
            if (x.a != y.a) return x.a < y.a;
            if (x.b != y.b) return x.b < y.b;
            return x.size < y.size;
        });
        // collapse adjacent blocks, append the (la, lb, 0) sentinel, take the first block of maximal size
        int best_a = 0, best_b = 0, best_size = -1;
        auto consider = [&](int i, int j, int k) {
            if (k > best_size) {
                best_a = i;
                best_b = j;
                best_size = k;
            }
        };
        int i1 = 0, j1 = 0, k1 = 0;
        for (const Block& m : blocks) {
            if (i1 + k1 == m.a && j1 + k1 == m.b) k1 += m.size;
            else {
                if (k1) consider(i1, j1, k1);
                i1 = m.a;
                j1 = m.b;
                k1 = m.size;
            }
        }
        if (k1) consider(i1, j1, k1);
        consider(la, lb, 0);
        return best_a - best_b;
    }
};

// one read: fragments (label arrays) -> consensus labels; returns the length, or -1 where the reference raises IndexError
// (its vote matrix grows by 1000 columns at most once per fragment, sequence_assembly.py:29-33,47)
int stitch_read(const uint8_t* const* frag, const int* flen, int m, uint8_t* out, Matcher& M, std::vector<int>& votes, std::vector<int>& starts)
{
    starts.assign(m, 0);
    int at = 0;
    for (int i = 1; i < m; i++) {
        M.a = frag[i - 1];
        M.la = flen[i - 1];
        M.b = frag[i];
        M.lb = flen[i];
        at += M.displacement();
        starts[i] = at;
    }
    int cap = 1000;
    for (int i = 0; i < m; i++) {
        const int st = starts[i], n = flen[i];
        if (i && st + n > cap) cap += 1000;
        const int kept = n + std::min(st, 0);                 // characters left of column 0 are dropped
        const int last = std::max(st, 0) + kept - 1;
        if (kept > 0 && last >= cap) return -1;
    }
    int width = 0;
    for (int i = 1; i < m; i++) width = std::max(width, starts[i] + flen[i]);
    if (width <= 0) return 0;                                 // (a single fragment gives an empty consensus, as in the reference)
    votes.assign((size_t)4 * width, 0);
    for (int i = 0; i < m; i++) {
        const int st = starts[i];
        const uint8_t* f = frag[i];
        for (int p = std::max(0, -st); p < flen[i]; p++) {
            const int col = st + p;
            if (col >= width) break;
            votes[(size_t)(f[p] & 3) * width + col]++;
        }
    }
    for (int c = 0; c < width; c++) {                         // np.argmax: the first maximum
        int best = 0, bv = votes[c];
        for (int e = 1; e < 4; e++)
            if (votes[(size_t)e * width + c] > bv) {
                bv = votes[(size_t)e * width + c];
                best = e;
            }
        out[c] = (uint8_t)best;
    }
    return width;
}

}  // namespace

extern "C" int rd_stitch_chunk(const uint8_t* labels, const int32_t* label_len, int chunk_len, const int32_t* read_win_off, int n_reads,
                               uint8_t* seq_out, const int64_t* seq_off, int32_t* seq_len, int n_threads)
{
    RD_REQUIRE(n_reads >= 0 && chunk_len >= 1, "rd_stitch_chunk: bad shape");
    if (n_reads == 0) return RD_OK;
    RD_REQUIRE(labels && label_len && read_win_off && seq_out && seq_off && seq_len, "rd_stitch_chunk: null argument");
    for (int r = 0; r < n_reads; r++) RD_REQUIRE(read_win_off[r + 1] >= read_win_off[r], "rd_stitch_chunk: window offsets must not decrease");
    const int nw = read_win_off[n_reads];
    for (int w = 0; w < nw; w++) RD_REQUIRE(label_len[w] >= 0 && label_len[w] <= chunk_len, "rd_stitch_chunk: label_len[%d] = %d out of range", w, label_len[w]);
    std::atomic<int> next(0);
    auto work = [&]() {
        Matcher M;
        std::vector<int> votes, starts, flen;
        std::vector<const uint8_t*> frag;
        for (;;) {
            const int r0 = next.fetch_add(16);
            if (r0 >= n_reads) break;
            for (int r = r0; r < std::min(n_reads, r0 + 16); r++) {
                const int w0 = read_win_off[r], m = read_win_off[r + 1] - w0;
                frag.resize(m);
                flen.resize(m);
                for (int i = 0; i < m; i++) {
                    frag[i] = labels + (size_t)(w0 + i) * chunk_len;
                    flen[i] = label_len[w0 + i];
                }
                seq_len[r] = stitch_read(frag.data(), flen.data(), m, seq_out + seq_off[r], M, votes, starts);
            }
        }
    };
    int nt = n_threads < 1 ? 1 : n_threads;
    if (nt > (n_reads + 15) / 16) nt = (n_reads + 15) / 16;
    if (nt <= 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < nt; i++) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    return RD_OK;
}

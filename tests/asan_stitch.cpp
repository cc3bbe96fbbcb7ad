// TEST INFRASTRUCTURE: AddressSanitizer + UBSan harness for radian_amd/csrc/stitch.hip (host code; sanitizers run on the CPU build only):
// random batches of reads with random fragments (empty ones, repeats of the previous fragment, shifted copies, fragments of 200+ labels
// for difflib's autojunk rule), every buffer heap-allocated at its exact size, one and several host threads.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
extern "C" int rd_stitch_chunk(const uint8_t* labels, const int32_t* label_len, int chunk_len, const int32_t* read_win_off, int n_reads,
                               uint8_t* seq_out, const int64_t* seq_off, int32_t* seq_len, int n_threads);
void rd_set_error(const char* fmt, ...) { (void)fmt; }
int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 3000;
    std::mt19937_64 rng(11);
    long reads_done = 0, failed_reads = 0;
    for (int it = 0; it < iters; it++) {
        const int chunk = (it % 4 == 0) ? 1024 : 16 + (int)(rng() % 400);
        const int n_reads = (int)(rng() % 6);
        std::vector<int32_t> off(n_reads + 1, 0);
        for (int r = 0; r < n_reads; r++) off[r + 1] = off[r] + (int)(rng() % 9);
        const int nw = off[n_reads];
        uint8_t* labels = (uint8_t*)malloc((size_t)nw * chunk ? (size_t)nw * chunk : 1);
        int32_t* len = (int32_t*)malloc(nw ? nw * sizeof(int32_t) : 1);
        std::vector<uint8_t> prev;
        int64_t cap = 0;
        std::vector<int64_t> soff(n_reads + 1, 0);
        for (int r = 0; r < n_reads; r++) {
            prev.clear();
            for (int w = off[r]; w < off[r + 1]; w++) {
                int L = 0;
                std::vector<uint8_t> f;
                const int kind = (int)(rng() % 8);
                if (kind == 0) L = 0;
                else if (kind == 1 && !prev.empty()) f = prev;                                     // the same fragment again
                else if (kind == 2 && prev.size() > 4) {                                            // a shifted copy with a new tail
                    const size_t sh = 1 + rng() % (prev.size() / 2);
                    f.assign(prev.begin() + sh, prev.end());
                    const int extra = (int)(rng() % 40);
                    for (int i = 0; i < extra; i++) f.push_back((uint8_t)(rng() % 4));
                } else {
                    L = (int)(rng() % (kind == 3 ? chunk + 1 : (chunk < 60 ? chunk + 1 : 60)));
                    const int alph = 1 + (int)(rng() % 4);                                         // few distinct labels: many equal runs
                    for (int i = 0; i < L; i++) f.push_back((uint8_t)(rng() % alph));
                }
                if ((int)f.size() > chunk) f.resize(chunk);
                L = (int)f.size();
                len[w] = L;
                if (L) memcpy(labels + (size_t)w * chunk, f.data(), L);
                cap += L;
                prev = f;
            }
            soff[r + 1] = cap;
        }
        uint8_t* out = (uint8_t*)malloc(cap ? cap : 1);
        int32_t* slen = (int32_t*)malloc(n_reads ? n_reads * sizeof(int32_t) : 1);
        const int rc = rd_stitch_chunk(labels, len, chunk, off.data(), n_reads, out, soff.data(), slen, 1 + (int)(rng() % 3));
        if (rc != 0) {
            printf("rd_stitch_chunk returned %d\n", rc);
            return 1;
        }
        for (int r = 0; r < n_reads; r++) {
            reads_done++;
            if (slen[r] < 0) failed_reads++;                  // the reference's IndexError case
            else if (slen[r] > soff[r + 1] - soff[r]) {
                printf("read %d: %d labels in a buffer of %ld\n", r, slen[r], (long)(soff[r + 1] - soff[r]));
                return 1;
            }
        }
        free(labels); free(len); free(out); free(slen);
    }
    printf("%ld reads stitched (%ld of them the reference's IndexError case), no sanitizer report\n", reads_done, failed_reads);
    return 0;
}

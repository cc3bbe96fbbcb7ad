// TEST INFRASTRUCTURE: AddressSanitizer + UBSan harness for radian_amd/csrc/fast5.hip (host code; sanitizers run on the CPU build only).
// usage: asan_fast5 <iterations> <file.fast5>...   Every file is parsed from an exact-size heap copy (a read past either end is an ASan report),
// every read copied out; then `iterations` mutated / truncated copies per file go through the same calls.  Whatever the reader answers
// (RD_OK, RD_ERR_FORMAT ...) is fine -- it must not touch memory outside the image, outside its output block, or loop for ever.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
struct rd_fast5;
extern "C" int rd_fast5_open_mem(const void* buf, size_t n, rd_fast5** out);
extern "C" void rd_fast5_close(rd_fast5* f);
extern "C" int rd_fast5_count(const rd_fast5* f, int64_t* n_reads);
extern "C" int rd_fast5_lengths(rd_fast5* f, int64_t lo, int64_t hi, int64_t* n_samples);
extern "C" int rd_fast5_read_batch(rd_fast5* f, int64_t lo, int64_t hi, int16_t* samples, int64_t cap, int64_t* offsets, char* ids, int id_stride);
void rd_set_error(const char* fmt, ...) { (void)fmt; }

static long g_ok_reads = 0, g_opened = 0, g_refused = 0;

static void drive(const unsigned char* img, size_t n, uint64_t* checksum)
{
    unsigned char* buf = (unsigned char*)malloc(n ? n : 1);   // exact size
    memcpy(buf, img, n);
    rd_fast5* f = nullptr;
    if (rd_fast5_open_mem(buf, n, &f) == 0) {
        g_opened++;
        int64_t cnt = 0;
        rd_fast5_count(f, &cnt);
        for (int64_t lo = 0; lo < cnt; lo += 7) {
            const int64_t hi = lo + 7 < cnt ? lo + 7 : cnt;
            std::vector<int64_t> lens((size_t)(hi - lo));
            if (rd_fast5_lengths(f, lo, hi, lens.data()) != 0) continue;
            int64_t tot = 0;
            bool sane = true;
            for (int64_t v : lens) {
                if (v < 0 || v > (1 << 24)) sane = false;     // a mutated length field: do not allocate terabytes
                tot += sane ? v : 0;
            }
            if (!sane) continue;
            int16_t* smp = (int16_t*)malloc(tot ? (size_t)tot * 2 : 1);    // exact size again
            std::vector<int64_t> off((size_t)(hi - lo + 1));
            std::vector<char> ids((size_t)(hi - lo) * 48);
            if (rd_fast5_read_batch(f, lo, hi, smp, tot, off.data(), ids.data(), 48) == 0) {
                g_ok_reads += hi - lo;
                if (checksum)
                    for (int64_t i = 0; i < tot; i++) *checksum = *checksum * 1099511628211ull + (uint16_t)smp[i];
            }
            free(smp);
        }
        rd_fast5_close(f);
    } else {
        g_refused++;
    }
    free(buf);
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const int iters = atoi(argv[1]);
    std::mt19937_64 rng(11);
    for (int a = 2; a < argc; a++) {
        FILE* fp = fopen(argv[a], "rb");
        if (!fp) return 3;
        std::vector<unsigned char> img;
        unsigned char tmp[65536];
        size_t k;
        while ((k = fread(tmp, 1, sizeof tmp, fp)) > 0) img.insert(img.end(), tmp, tmp + k);
        fclose(fp);
        uint64_t sum = 1469598103934665603ull;
        const long before = g_ok_reads;
        drive(img.data(), img.size(), &sum);
        printf("file %s: %ld reads, checksum %016llx\n", argv[a], g_ok_reads - before, (unsigned long long)sum);
        for (int it = 0; it < iters; it++) {
            std::vector<unsigned char> m = img;
            const int kind = (int)(rng() % 8);
            if (kind == 0) {
                m.resize((size_t)(rng() % (m.size() + 1)));                      // truncated file
            } else {
                const int muts = 1 + (int)(rng() % 6);
                for (int j = 0; j < muts; j++) {
                    // half of the mutations go into the first 16 KiB (superblock, root group, the first read's headers), where structure is dense
                    const size_t span = (rng() & 1) && m.size() > 16384 ? 16384 : m.size();
                    const size_t pos = (size_t)(rng() % span);
                    switch (rng() % 4) {
                        case 0: m[pos] = (unsigned char)rng(); break;
                        case 1: m[pos] ^= (unsigned char)(1u << (rng() % 8)); break;
                        case 2: m[pos] = 0xff; if (pos + 1 < m.size()) m[pos + 1] = 0xff; break;   // (undefined-address patterns, huge sizes)
                        default: {                                                                   // an 8-byte field replaced by a small or a wild number
                            uint64_t v = (rng() & 1) ? rng() % (2 * m.size() + 16) : rng();
                            for (int b = 0; b < 8 && pos + b < m.size(); b++) m[pos + b] = (unsigned char)(v >> (8 * b));
                        }
                    }
                }
            }
            drive(m.data(), m.size(), nullptr);
        }
    }
    printf("%ld opened, %ld refused, %ld reads copied: no sanitizer report\n", g_opened, g_refused, g_ok_reads);
    return 0;
}

// TEST INFRASTRUCTURE: AddressSanitizer + UBSan harness for radian_amd/csrc/lmjson.hip (host code; sanitizers run on the CPU build only):
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <random>
#include <string>
#include <vector>
#include <cmath>
extern "C" int rd_lm_json_probe(const char* buf, size_t n, int* k_out);
extern "C" int rd_lm_json_fill(const char* buf, size_t n, int k, double* table, int64_t* n_entries, int64_t* n_contexts);
void rd_set_error(const char* fmt, ...) { (void)fmt; }
int main(int argc, char** argv)
{
    std::mt19937_64 rng(7);
    const char* alphabet = "{}[]\":,ACGTacgu0123456789.eE+- \n\t\\xNaInf";
    long ok = 0, total = 0;
    const int iters = argc > 1 ? atoi(argv[1]) : 400000;
    for (int it = 0; it < iters; it++) {
        std::string s;
        const int k = 1 + rng() % 4;
        if (it % 3) {   // a valid model, then mutated
            s = "{";
            const int n = 1 + rng() % 12;
            for (int i = 0; i < n; i++) {
                if (i) s += (rng() % 5 ? ", " : ",");
                s += "\"";
                for (int j = 0; j < k; j++) s += "ACGT"[rng() % 4];
                s += "\": [";
                for (int j = 0; j < 4; j++) {
                    char b[64];
                    snprintf(b, sizeof b, rng() % 3 ? "%.17g" : "%g", (double)(rng() % 1000) / (1 + rng() % 1000) * (rng() % 7 ? 1 : 1e-300));
                    s += b;
                    if (j < 3) s += ", ";
                }
                s += "]";
            }
            s += "}";
            const int muts = rng() % 4;
            for (int m = 0; m < muts && !s.empty(); m++) {
                const size_t pos = rng() % s.size();
                switch (rng() % 3) {
                    case 0: s[pos] = alphabet[rng() % strlen(alphabet)]; break;
                    case 1: s.erase(pos, 1 + rng() % 3); break;
                    default: s.insert(pos, 1, alphabet[rng() % strlen(alphabet)]); break;
                }
            }
            if (rng() % 10 == 0) s.resize(rng() % (s.size() + 1));     // truncated file
        } else {
            const int n = rng() % 60;
            for (int i = 0; i < n; i++) s += alphabet[rng() % strlen(alphabet)];
        }
        char* buf = (char*)malloc(s.size() ? s.size() : 1);            // exact size: a read past the end is an ASan error
        memcpy(buf, s.data(), s.size());
        int kk = 0;
        total++;
        if (rd_lm_json_probe(buf, s.size(), &kk) == 0) {
            std::vector<double> table((size_t)4 << (2 * kk), NAN);
            int64_t ne = 0, nc = 0;
            if (rd_lm_json_fill(buf, s.size(), kk, table.data(), &ne, &nc) == 0) ok++;
        }
        free(buf);
    }
    printf("%ld of %ld texts accepted, no sanitizer report\n", ok, total);
    return 0;
}

// lmjson.hip -- host code: the RNA model file, radian/basecall.py:48-57 (json.load + re-keying), read straight into the dense table
// rd_load_lm takes.  The reference's default model has 4^11 = 4 194 304 contexts in ~420 MB of JSON; the standard library's parser
// builds 4.2 M Python strings and 21 M floats for it (21 s, 2 GB) before the table can be filled.  This is a one-pass scanner for the
// one shape such a file has -- an object of "ACGT..." keys, each with an array of four numbers --
//
//     { "AAAAAAAAAAA": [0.25, 0.25, 0.25, 0.25], "AAAAAAAAAAC": [ ... ], ... }
//
// and ANYTHING else (escapes in a key, another alphabet, nested values, NaN / Infinity literals, a fifth number, trailing text) makes
// it return RD_ERR_FORMAT without a verdict: the caller then runs the standard parser, whose errors are the reference's.  Numbers are
// converted by std::from_chars (locale-independent, correctly rounded: the double Python's float() gives for the same text); a repeated
// key keeps its last value (json.load's dict does the same); contexts the file does not hold stay NaN (sparse model: rd_load_lm).
//
// No GPU is touched; the file is part of libradian_hip.so so that the host side stays one ctypes binding.
#include "common.h"
#include "../../include/radian_hip.h"

#include <charconv>
#include <cmath>

namespace {

inline const char* skip_ws(const char* p, const char* e)
{
    while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++;
    return p;
}

inline int base_code(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

// JSON number grammar (RFC 8259): -? (0 | [1-9][0-9]*) (. [0-9]+)? ([eE] [+-]? [0-9]+)?   -> end of the token, or nullptr
inline const char* number_end(const char* p, const char* e)
{
    if (p < e && *p == '-') p++;
    if (p >= e) return nullptr;
    if (*p == '0') p++;
    else if (*p >= '1' && *p <= '9') {
        while (p < e && *p >= '0' && *p <= '9') p++;
    } else return nullptr;
    if (p < e && *p == '.') {
        p++;
        if (p >= e || *p < '0' || *p > '9') return nullptr;
        while (p < e && *p >= '0' && *p <= '9') p++;
    }
    if (p < e && (*p == 'e' || *p == 'E')) {
        p++;
        if (p < e && (*p == '+' || *p == '-')) p++;
        if (p >= e || *p < '0' || *p > '9') return nullptr;
        while (p < e && *p >= '0' && *p <= '9') p++;
    }
    return p;
}

}  // namespace

// Context length of the first key of the object in buf (1..13), or RD_ERR_FORMAT.
extern "C" int rd_lm_json_probe(const char* buf, size_t n, int* k_out)
{
    RD_REQUIRE(buf && k_out, "rd_lm_json_probe: null argument");
    const char* e = buf + n;
    const char* p = skip_ws(buf, e);
    if (p >= e || *p != '{') return RD_ERR_FORMAT;
    p = skip_ws(p + 1, e);
    if (p >= e || *p != '"') return RD_ERR_FORMAT;
    p++;
    int k = 0;
    while (p < e && base_code(*p) >= 0) {
        p++;
        k++;
    }
    if (p >= e || *p != '"' || k < 1 || k > 13) return RD_ERR_FORMAT;
    *k_out = k;
    return RD_OK;
}

// Fill table[4^k][4] (the caller has set every value to NaN) from the object in buf.  *n_entries = key/value pairs read,
// *n_contexts = distinct contexts among them.  RD_ERR_FORMAT: not the expected shape (the table is then in an undefined state).
extern "C" int rd_lm_json_fill(const char* buf, size_t n, int k, double* table, int64_t* n_entries, int64_t* n_contexts)
{
    RD_REQUIRE(buf && table && n_entries && n_contexts && k >= 1 && k <= 13, "rd_lm_json_fill: bad argument");
    const char* e = buf + n;
    const char* p = skip_ws(buf, e);
    if (p >= e || *p != '{') return RD_ERR_FORMAT;
    p = skip_ws(p + 1, e);
    int64_t entries = 0, distinct = 0;
    if (p < e && *p == '}') {
        p = skip_ws(p + 1, e);
        if (p != e) return RD_ERR_FORMAT;
        *n_entries = 0;
        *n_contexts = 0;
        return RD_OK;
    }
    for (;;) {
        if (p >= e || *p != '"' || e - p < k + 2) return RD_ERR_FORMAT;
        p++;
        size_t row = 0;
        for (int i = 0; i < k; i++) {
            const int c = base_code(p[i]);
            if (c < 0) return RD_ERR_FORMAT;
            row = (row << 2) | (size_t)c;
        }
        p += k;
        if (*p != '"') return RD_ERR_FORMAT;          // a longer or shorter key: mixed lengths are the standard parser's to report
        p = skip_ws(p + 1, e);
        if (p >= e || *p != ':') return RD_ERR_FORMAT;
        p = skip_ws(p + 1, e);
        if (p >= e || *p != '[') return RD_ERR_FORMAT;
        p = skip_ws(p + 1, e);
        double v[4];
        for (int i = 0; i < 4; i++) {
            const char* q = number_end(p, e);
            if (!q) return RD_ERR_FORMAT;
            const auto r = std::from_chars(p, q, v[i]);
            if (r.ec != std::errc() || r.ptr != q || std::isnan(v[i])) return RD_ERR_FORMAT;   // (out of range -> inf in Python: leave it to it)
            p = skip_ws(q, e);
            if (p >= e || *p != (i < 3 ? ',' : ']')) return RD_ERR_FORMAT;
            p = skip_ws(p + 1, e);
        }
        double* t = table + row * 4;
        if (std::isnan(t[0])) distinct++;
        t[0] = v[0];
        t[1] = v[1];
        t[2] = v[2];
        t[3] = v[3];
        entries++;
        if (p >= e) return RD_ERR_FORMAT;
        if (*p == ',') {
            p = skip_ws(p + 1, e);
            continue;
        }
        if (*p != '}') return RD_ERR_FORMAT;
        p = skip_ws(p + 1, e);
        if (p != e) return RD_ERR_FORMAT;
        break;
    }
    *n_entries = entries;
    *n_contexts = distinct;
    return RD_OK;
}

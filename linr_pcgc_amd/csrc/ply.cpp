// Host side of the frame input (datautils/custom_dataset.py:9-14: read_ply_o3d = open3d's C++ reader, then :263-269 keeps the
// rounded x, y, z): the body of an ASCII PLY - the format the 8iVFB / MVUB / Owlii sequences ship in - as one pass over the text.
// A loot frame is ~786 k lines of "x y z r g b": numpy's text reader needs ~0.7 s for it, 32 frames per GOP against ~0.6 s of GPU
// work for the whole GOP; this parser runs at memory-copy-like speed for the digits-only numbers such files hold and falls back
// to strtod for anything else (exponents, inf / nan).  No allocation, no locale, never reads past `len`.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "../../include/linr_hip.h"

namespace {
inline bool is_space(unsigned char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }

// one number starting at p (p < end, not a space); returns the position behind it, or nullptr on a malformed token
inline const char* parse_number(const char* p, const char* end, double* out) {
    const char* s = p;
    bool neg = false;
    if (*p == '-' || *p == '+') { neg = *p == '-'; ++p; }
    uint64_t mant = 0;
    int digits = 0, frac_digits = 0;
    while (p < end && (unsigned)(*p - '0') < 10u && digits < 15) { mant = mant * 10 + (unsigned)(*p - '0'); ++p; ++digits; }
    if (p < end && *p == '.' && digits < 15) {
        ++p;
        while (p < end && (unsigned)(*p - '0') < 10u && digits < 15) { mant = mant * 10 + (unsigned)(*p - '0'); ++p; ++digits; ++frac_digits; }
    }
    if (digits > 0 && (p == end || is_space((unsigned char)*p))) {          // the fast path: [sign] digits [. digits], <= 15 digits (exact in a double)
        static const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
        const double v = frac_digits ? (double)mant / p10[frac_digits] : (double)mant;          // both exact in a double => one correctly rounded division
        *out = neg ? -v : v;
        return p;
    }
    // anything else (exponent, more digits, inf, nan): a bounded copy through strtod
    const char* q = s;
    while (q < end && !is_space((unsigned char)*q)) ++q;
    char tmp[64];
    const size_t n = (size_t)(q - s);
    if (n == 0 || n >= sizeof(tmp)) return nullptr;
    memcpy(tmp, s, n);
    tmp[n] = 0;
    char* stop = nullptr;
    *out = strtod(tmp, &stop);
    return (stop == tmp + n) ? q : nullptr;
}
}  // namespace

extern "C" int linr_ply_parse_ascii(const char* text_h, size_t len, int64_t n_rows, int32_t n_cols, int32_t cx, int32_t cy,
                                    int32_t cz, int64_t* xyz_h, int64_t* rows_parsed_h) {
    if (rows_parsed_h) *rows_parsed_h = 0;
    if ((!text_h && len) || !xyz_h || n_rows < 0 || n_cols < 3 || n_cols > 64) return LINR_EINVAL;
    const int32_t c3[3] = {cx, cy, cz};
    for (int j = 0; j < 3; ++j)
        if (c3[j] < 0 || c3[j] >= n_cols) return LINR_EINVAL;
    const char* p = text_h;
    const char* const end = text_h + len;
    for (int64_t r = 0; r < n_rows; ++r) {
        double v[3] = {0, 0, 0};
        for (int32_t c = 0; c < n_cols; ++c) {
            while (p < end && *p != '\n' && is_space((unsigned char)*p)) ++p;          // blanks inside the line
            if (c == 0)
                while (p < end && is_space((unsigned char)*p)) ++p;                      // and empty lines between vertices
            if (p >= end || *p == '\n') return LINR_EINVAL;                              // the line (or the text) ended early
            double x;
            p = parse_number(p, end, &x);
            if (!p) return LINR_EINVAL;
            for (int j = 0; j < 3; ++j)
                if (c == c3[j]) v[j] = x;
        }
        while (p < end && *p != '\n') {                                                   // nothing but blanks up to the end of the line
            if (!is_space((unsigned char)*p)) return LINR_EINVAL;
            ++p;
        }
        for (int j = 0; j < 3; ++j) {
            if (!(std::fabs(v[j]) < 9.0e15)) return LINR_EINVAL;                          // nan / inf / beyond the int64-exact range
            xyz_h[3 * r + j] = (int64_t)std::nearbyint(v[j]);                             // round half to even, like numpy.rint
        }
        if (rows_parsed_h) *rows_parsed_h = r + 1;
    }
    return 0;
}

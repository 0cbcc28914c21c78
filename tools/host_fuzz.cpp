// ASan / UBSan fuzz of the host-side entry points (range coder, PLY parser): random and mutated inputs, bounds respected.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../include/linr_hip.h"
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    std::mt19937_64 rng(12345);
    auto u = [&](uint64_t n) { return (uint64_t)(rng() % n); };
    long checked = 0;
    for (int it = 0; it < iters; ++it) {
        // --- binary coder round trip + decode of corrupted / truncated / random streams ---
        const int64_t n = u(3000);
        std::vector<float> p(n);
        std::vector<uint8_t> s(n), out(2 * n + 64), dec(n + 1);
        const int mode = (int)u(4);
        for (int64_t i = 0; i < n; ++i) {
            float v = (float)(rng() >> 11) / (float)(1ull << 53);
            if (mode == 1) v = v < 0.5f ? 1e-7f * v : 1.0f - 1e-7f * v;
            if (mode == 2) v = (u(10) == 0) ? (u(2) ? 0.0f : 1.0f) : v;
            if (mode == 3 && u(50) == 0) v = std::nanf("");
            p[i] = v;
            s[i] = (uint8_t)(u(2));
        }
        std::vector<uint8_t> out_exact;
        const int64_t len = linr_ac_encode_binary(p.data(), s.data(), n, out.data(), (int64_t)out.size());
        if (len < 0) { printf("encode failed %lld\n", (long long)len); return 1; }
        out_exact.assign(out.begin(), out.begin() + len);          // exact-size buffer: reads past the end are ASan errors
        if (linr_ac_decode_binary(p.data(), n, out_exact.data(), len, dec.data()) != 0) { printf("decode rc\n"); return 1; }
        if (mode != 3 && n > 0 && memcmp(dec.data(), s.data(), n) != 0) { printf("round trip mismatch at iter %d (n %lld)\n", it, (long long)n); return 1; }
        // too small output buffer: must report the needed length / an error without writing past cap
        if (len > 2) {
            std::vector<uint8_t> small(len / 2);
            const int64_t r2 = linr_ac_encode_binary(p.data(), s.data(), n, small.data(), (int64_t)small.size());
            if (r2 >= 0 && r2 <= (int64_t)small.size()) { printf("short buffer accepted\n"); return 1; }
        }
        // truncated and garbage streams
        std::vector<uint8_t> trunc(out_exact.begin(), out_exact.begin() + (len ? u(len) : 0));
        linr_ac_decode_binary(p.data(), n, trunc.data(), (int64_t)trunc.size(), dec.data());
        std::vector<uint8_t> junk(u(200));
        for (auto& b : junk) b = (uint8_t)rng();
        linr_ac_decode_binary(p.data(), n, junk.data(), (int64_t)junk.size(), dec.data());
        // --- cdf16 coder with a shared random CDF ---
        const int lp = 2 + (int)u(300);
        std::vector<uint16_t> cdf(lp);
        { std::vector<uint32_t> cuts(lp - 1); for (auto& c : cuts) c = (uint32_t)u(65536); cuts[0] = 0; std::sort(cuts.begin(), cuts.end());
          for (int i = 0; i < lp - 1; ++i) cdf[i] = (uint16_t)cuts[i]; cdf[lp - 1] = 0; }
        const int64_t m = u(500);
        std::vector<int16_t> sym(m), sdec(m + 1);
        for (auto& v : sym) v = (int16_t)u(lp - 1);
        std::vector<uint8_t> o2(4 * m + 64);
        const int64_t l2 = linr_ac_encode_cdf16(cdf.data(), lp, 1, sym.data(), m, o2.data(), (int64_t)o2.size());
        if (l2 >= 0) {
            std::vector<uint8_t> ex(o2.begin(), o2.begin() + l2);
            linr_ac_decode_cdf16(cdf.data(), lp, 1, m, ex.data(), l2, sdec.data());
            std::vector<uint8_t> j2(u(100)); for (auto& b : j2) b = (uint8_t)rng();
            linr_ac_decode_cdf16(cdf.data(), lp, 1, m, j2.data(), (int64_t)j2.size(), sdec.data());
        }
        // --- PLY parser: valid bodies, mutated bodies, garbage ---
        const int rows = (int)u(40), cols = 3 + (int)u(4);
        std::string body;
        for (int r = 0; r < rows; ++r) {
            for (int c = 0; c < cols; ++c) {
                char buf[64];
                switch (u(5)) { case 0: snprintf(buf, 64, "%d", (int)u(2000) - 1000); break; case 1: snprintf(buf, 64, "%.3f", (double)u(100000) / 77.0); break;
                                case 2: snprintf(buf, 64, "%e", (double)u(100000) / 3.0); break; case 3: snprintf(buf, 64, "%lld", (long long)u(1ull << 50)); break;
                                default: snprintf(buf, 64, "%d.", (int)u(100)); }
                body += buf; body += (c + 1 < cols) ? (u(7) == 0 ? "\t" : " ") : (u(5) == 0 ? "\r\n" : "\n");
            }
            if (u(9) == 0) body += "\n";
        }
        std::vector<int64_t> xyz(3 * (rows + 1));
        int64_t donerows = 0;
        {   std::vector<char> exact(body.begin(), body.end());
            linr_ply_parse_ascii(exact.data(), exact.size(), rows, cols, 0, 1 % cols, 2 % cols, xyz.data(), &donerows); }
        std::string mut = body;
        for (int k = 0; k < 3 && !mut.empty(); ++k) { const size_t pos = u(mut.size()); const char pool[] = "0123456789.-+eE \n\txnaif"; mut[pos] = pool[u(sizeof(pool) - 1)]; }
        {   std::vector<char> exact(mut.begin(), mut.begin() + (mut.size() ? u(mut.size() + 1) : 0));
            linr_ply_parse_ascii(exact.data(), exact.size(), rows, cols, 0, 1, 2, xyz.data(), &donerows); }
        ++checked;
    }
    printf("fuzz ok: %ld iterations\n", checked);
    return 0;
}

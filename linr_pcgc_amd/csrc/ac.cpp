// Arithmetic-coder feed (host side).  Produces / consumes streams in torchac 0.9.3's format, following its published coder
// (byte-compatibility is by construction; the reference's one known-answer vector is consistent with it in two ways,
// include/linr_hip.h; cross-device model.bin interop depends on the last bit of expf), as the
// reference uses it: BinaryArithmeticCoding (models/module_utils.py:8-40, cdf = [0, 1-p, 1]) and the Laplace model stream
// (model_compression/model_size_est.py:470-482,545-563).  32-bit range coder with carry-less pending-bit handling
// over 16-bit cumulative frequencies; bits are packed MSB first, the tail is zero padded.
//
// The binary path never materialises a CDF row: the only entry that matters is
//   c1 = uint16(rint((1 - p) * 65534) + 1)         (torchac: round(cdf * (2^16 - (Lp-1))) + arange(Lp), Lp = 3)
// symbol 0 owns [0, c1), symbol 1 owns [c1, 2^16).  8 stages x scales are independent streams, coded in parallel.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include "../../include/linr_hip.h"

namespace {

struct BitSink {
    uint8_t* out; int64_t cap; int64_t len = 0; uint32_t acc = 0; int nacc = 0;
    BitSink(uint8_t* o, int64_t c) : out(o), cap(c) {}
    inline void put(uint32_t bit) {
        acc = (acc << 1) | bit;
        if (++nacc == 8) { if (len < cap) out[len] = (uint8_t)acc; ++len; acc = 0; nacc = 0; }
    }
    inline void put_with_pending(uint32_t bit, uint64_t& pending) {
        put(bit);
        for (; pending > 0; --pending) put(bit ^ 1u);
    }
    inline void finish() { while (nacc != 0) put(0); }
};

struct RangeEncoder {
    uint32_t low = 0, high = 0xFFFFFFFFu; uint64_t pending = 0; BitSink sink;
    RangeEncoder(uint8_t* o, int64_t c) : sink(o, c) {}
    inline void encode(uint32_t c_low, uint32_t c_high) {
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        high = (low - 1) + (uint32_t)((span * c_high) >> 16);
        low = low + (uint32_t)((span * c_low) >> 16);
        for (;;) {
            if (high < 0x80000000u) {
                sink.put_with_pending(0, pending);
            } else if (low >= 0x80000000u) {
                sink.put_with_pending(1, pending);
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                ++pending;
                low = (low << 1) & 0x7FFFFFFFu;
                high = (high << 1) | 0x80000001u;
                continue;
            } else {
                break;
            }
            low <<= 1;
            high = (high << 1) | 1u;
        }
    }
    inline int64_t finish() {
        ++pending;
        sink.put_with_pending(low < 0x40000000u ? 0u : 1u, pending);
        sink.finish();
        return sink.len;
    }
};

struct BitSource {
    const uint8_t* in; int64_t len; int64_t pos = 0; uint64_t buf = 0; int nbuf = 0;    // buf: next bits, MSB-aligned
    BitSource(const uint8_t* i, int64_t l) : in(i), len(l) {}
    // the next n (1..32) bits of the stream, MSB first; zeros past the end
    inline uint32_t bits(int n) {
        while (nbuf <= 56) {
            const uint64_t byte = pos < len ? in[pos] : 0u;
            ++pos;
            buf |= byte << (56 - nbuf);
            nbuf += 8;
        }
        const uint32_t v = (uint32_t)(buf >> (64 - n));
        buf <<= n;
        nbuf -= n;
        return v;
    }
};

struct RangeDecoder {
    uint32_t low = 0, high = 0xFFFFFFFFu, value = 0; BitSource src;
    RangeDecoder(const uint8_t* i, int64_t l) : src(i, l) { value = src.bits(32); }
    inline uint32_t target() const {
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        return (uint32_t)(uint16_t)((((uint64_t)value - (uint64_t)low + 1) * 0x10000u - 1) / span);
    }
    inline void consume(uint32_t c_low, uint32_t c_high) {
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        high = (low - 1) + (uint32_t)((span * c_high) >> 16);
        low = low + (uint32_t)((span * c_low) >> 16);
        for (;;) {
            const uint32_t diff = low ^ high;
            if (diff < 0x80000000u) {
                // torchac shifts one bit at a time while the top bits of low and high agree; all of those shifts at once
                int n = diff ? __builtin_clz(diff) : 31;
                if (n > 31) n = 31;
                low <<= n;
                high = (high << n) | ((1u << n) - 1u);
                value = (value << n) | src.bits(n);
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                low = (low << 1) & 0x7FFFFFFFu;
                high = (high << 1) | 0x80000001u;
                value -= 0x40000000u;
                value = (value << 1) | src.bits(1);
            } else {
                break;
            }
        }
    }
};

inline uint32_t binary_c1(float p) {
    // float32 arithmetic exactly as torch: (1 - p) * 65534, round half to even, int16 wrap, + 1
    const long scaled = lrintf((1.0f - p) * 65534.0f);      // current rounding mode = to nearest even, like torch.round
    return (uint32_t)((scaled + 1) & 0xFFFF);
}

}  // namespace

// ---- binary fast path ---------------------------------------------------------------------------------------------------
// Same coder, specialised: with cdf = [0, c1, 2^16] the interval update is  t = (span * c1) >> 16;  symbol 1: low += t
// (high = low - 1 + span is unchanged), symbol 0: high = low + t - 1.  Renormalisation shifts all leading bits on which
// low and high agree at once (torchac emits them one by one; the first one is followed by the pending complement bits),
// c1 is computed for a block of symbols at a time (vectorisable, off the serial dependency chain), and the symbol
// selects are branch-free (the symbol is the one thing the branch predictor cannot know).
namespace {

struct BitSinkFast {
    uint8_t* out; int64_t cap; int64_t len = 0; uint64_t acc = 0; int nacc = 0;       // nacc < 8 between calls
    BitSinkFast(uint8_t* o, int64_t c) : out(o), cap(c) {}
    inline void put_bits(uint32_t v, int n) {             // n in 0..32, v < 2^n
        acc = (acc << n) | v;
        nacc += n;
        while (nacc >= 8) {
            nacc -= 8;
            if (len < cap) out[len] = (uint8_t)(acc >> nacc);
            ++len;
        }
    }
    inline void put_run(uint32_t bit, uint64_t count) {
        const uint32_t word = bit ? 0xFFFFFFFFu : 0u;
        while (count >= 32) { put_bits(word, 32); count -= 32; }
        if (count) put_bits(word >> (32 - (int)count), (int)count);
    }
    inline void finish() { if (nacc) put_bits(0, 8 - nacc); }
};

constexpr int AC_CHUNK = 256;

inline void c1_block(const float* p, int n, uint32_t* c1) {
    int i = 0;
#if defined(__SSE2__)
    // binary_c1 four at a time: cvtps2dq rounds to nearest even under the default MXCSR, like lrintf; |scaled| <= 65534 fits
    const __m128 one = _mm_set1_ps(1.0f), scale = _mm_set1_ps(65534.0f);
    const __m128i inc = _mm_set1_epi32(1), m16 = _mm_set1_epi32(0xFFFF);
    for (; i + 4 <= n; i += 4) {
        const __m128i v = _mm_cvtps_epi32(_mm_mul_ps(_mm_sub_ps(one, _mm_loadu_ps(p + i)), scale));
        _mm_storeu_si128(reinterpret_cast<__m128i*>(c1 + i), _mm_and_si128(_mm_add_epi32(v, inc), m16));
    }
#endif
    for (; i < n; ++i) c1[i] = binary_c1(p[i]);
}

}  // namespace

extern "C" int64_t linr_ac_encode_binary(const float* prob_h, const uint8_t* sym_h, int64_t n, uint8_t* out_h, int64_t cap) {
    if (n < 0 || cap < 0 || (n > 0 && (!prob_h || !sym_h)) || (cap > 0 && !out_h)) return LINR_EINVAL;
    BitSinkFast sink(out_h, cap);
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint64_t pending = 0;
    uint32_t c1[AC_CHUNK];
    for (int64_t base = 0; base < n; base += AC_CHUNK) {
        const int m = (int)(n - base < AC_CHUNK ? n - base : AC_CHUNK);
        c1_block(prob_h + base, m, c1);
        for (int i = 0; i < m; ++i) {
            const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
            const uint32_t t = (uint32_t)((span * c1[i]) >> 16);
            const uint32_t one = sym_h[base + i] ? 0xFFFFFFFFu : 0u;
            const uint32_t lt = low + t;
            high = (high & one) | ((lt - 1u) & ~one);
            low = (lt & one) | (low & ~one);
            for (;;) {
                const uint32_t diff = low ^ high;
                if (diff < 0x80000000u) {
                    int k = diff ? __builtin_clz(diff) : 31;            // leading bits shared by low and high (>= 1)
                    if (k > 31) k = 31;
                    const uint32_t first = low >> 31;
                    if (pending) {
                        sink.put_bits(first, 1);
                        sink.put_run(first ^ 1u, pending);
                        pending = 0;
                        if (k > 1) sink.put_bits((low << 1) >> (33 - k), k - 1);
                    } else {
                        sink.put_bits(low >> (32 - k), k);
                    }
                    low <<= k;
                    high = (high << k) | ((1u << k) - 1u);
                } else if (low >= 0x40000000u && high < 0xC0000000u) {
                    ++pending;
                    low = (low << 1) & 0x7FFFFFFFu;
                    high = (high << 1) | 0x80000001u;
                } else {
                    break;
                }
            }
        }
    }
    ++pending;
    const uint32_t last = low < 0x40000000u ? 0u : 1u;
    sink.put_bits(last, 1);
    sink.put_run(last ^ 1u, pending);
    sink.finish();
    return sink.len <= cap ? sink.len : (int64_t)LINR_ENOSPC;
}

namespace {

// Bit window of the binary decoder: 64 bits, MSB-aligned, refilled eight bytes at a time while eight whole bytes remain
// (byte by byte in the tail; past the end the stream reads as zeros, as torchac's does).
struct BitWindow {
    const uint8_t* in; int64_t len; int64_t pos = 0; uint64_t buf = 0; int nbuf = 0;
    BitWindow(const uint8_t* i, int64_t l) : in(i), len(l) {}
    inline void refill() {                                   // afterwards nbuf >= 56
        if (__builtin_expect(pos + 8 <= len, 1)) {
            uint64_t w;
            std::memcpy(&w, in + pos, 8);
            buf |= __builtin_bswap64(w) >> nbuf;             // bits of a partly taken byte are OR-ed again, unchanged
            pos += (63 - nbuf) >> 3;
            nbuf |= 56;
        } else {
            while (nbuf <= 56) {
                const uint64_t byte = pos < len ? in[pos] : 0u;
                ++pos;
                buf |= byte << (56 - nbuf);
                nbuf += 8;
            }
        }
    }
    inline uint32_t take(int n) {                            // n in 0..32 (0 gives 0); needs nbuf >= n
        const uint32_t v = (uint32_t)((buf >> 1) >> (63 - n));
        buf <<= n;
        nbuf -= n;
        return v;
    }
};

}  // namespace

extern "C" int linr_ac_decode_binary(const float* prob_h, int64_t n, const uint8_t* in_h, int64_t in_len, uint8_t* sym_h) {
    if (n < 0 || in_len < 0 || (n > 0 && (!prob_h || !sym_h)) || (in_len > 0 && !in_h)) return LINR_EINVAL;
    BitWindow src(in_h, in_len);
    src.refill();
    uint32_t low = 0, high = 0xFFFFFFFFu, value = src.take(32);
    uint32_t c1[AC_CHUNK];
    for (int64_t base = 0; base < n; base += AC_CHUNK) {
        const int m = (int)(n - base < AC_CHUNK ? n - base : AC_CHUNK);
        c1_block(prob_h + base, m, c1);
        for (int i = 0; i < m; ++i) {
            // torchac's binary search over [0, c1, *] returns symbol 1 iff target >= c1, i.e. iff
            //   ((value - low + 1) * 2^16 - 1) / span >= c1  <=>  (value - low + 1) * 2^16 > c1 * span
            //   <=>  value - low + 1 > floor(c1 * span / 2^16)  <=>  value - low >= (c1 * span) >> 16 = t
            const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
            const uint32_t t = (uint32_t)((span * c1[i]) >> 16);
            const uint32_t s = (value - low >= t) ? 1u : 0u;
            sym_h[base + i] = (uint8_t)s;
            if (base + i == n - 1) break;                   // torchac does not renormalise after the last symbol
            const uint32_t one = 0u - s;
            const uint32_t lt = low + t;
            high = (high & one) | ((lt - 1u) & ~one);
            low = (lt & one) | (low & ~one);
            // Renormalisation.  torchac shifts one bit at a time: first while the top bits of low and high agree, then while
            // the interval straddles the middle with low = 01.., high = 10.. (each such step also takes 2^30 off value).  Both
            // runs are done at once: k agreeing bits, then j straddle steps (the top bits differ after the first run and after
            // every straddle step, so the first run cannot resume); j steps leave value = (value << j | bits) - 2^31 mod 2^32.
            // Most symbols of a trained model need neither, which the one branch below predicts.
            while (__builtin_expect((((low ^ high) ^ 0x80000000u) | ((low & ~high) << 1)) & 0x80000000u, 0)) {
                if (src.nbuf < 32) src.refill();
                const uint32_t diff = low ^ high;
                int k = __builtin_clz(diff | 1u);           // 0 when the top bits differ already
                low <<= k;
                high = (high << k) | ((1u << k) - 1u);
                value = (value << k) | src.take(k);
                if (src.nbuf < 32) src.refill();
                const int j = __builtin_clz(~((low & ~high) << 1));          // leading positions below the top with low 1, high 0
                low = (low << j) & 0x7FFFFFFFu;
                high = (high << j) | 0x80000000u | ((1u << j) - 1u);
                value = ((value << j) | src.take(j)) ^ (j ? 0x80000000u : 0u);
            }
        }
    }
    return 0;
}

extern "C" int64_t linr_ac_encode_cdf16(const uint16_t* cdf_h, int32_t lp, int32_t cdf_shared, const int16_t* sym_h,
                                        int64_t n, uint8_t* out_h, int64_t cap) {
    if (n < 0 || cap < 0 || lp < 2 || (n > 0 && (!cdf_h || !sym_h)) || (cap > 0 && !out_h)) return LINR_EINVAL;
    RangeEncoder enc(out_h, cap);
    const int max_symbol = lp - 2;
    for (int64_t i = 0; i < n; ++i) {
        const uint16_t* row = cdf_shared ? cdf_h : cdf_h + i * (int64_t)lp;
        const int s = sym_h[i];
        if (s < 0 || s > max_symbol) return LINR_EINVAL;
        enc.encode(row[s], s == max_symbol ? 0x10000u : (uint32_t)row[s + 1]);
    }
    const int64_t len = enc.finish();
    return len <= cap ? len : (int64_t)LINR_ENOSPC;
}

extern "C" int linr_ac_decode_cdf16(const uint16_t* cdf_h, int32_t lp, int32_t cdf_shared, int64_t n, const uint8_t* in_h,
                                    int64_t in_len, int16_t* sym_h) {
    if (n < 0 || in_len < 0 || lp < 2 || (n > 0 && (!cdf_h || !sym_h)) || (in_len > 0 && !in_h)) return LINR_EINVAL;
    RangeDecoder dec(in_h, in_len);
    const int max_symbol = lp - 2;
    for (int64_t i = 0; i < n; ++i) {
        const uint16_t* row = cdf_shared ? cdf_h : cdf_h + i * (int64_t)lp;
        const uint16_t t = (uint16_t)dec.target();
        int left = 0, right = max_symbol + 1;        // torchac's search (keeps its behaviour on non-monotone rows)
        while (left + 1 < right) {
            const int m = (left + right) / 2;
            const uint16_t v = row[m];
            if (v < t) left = m; else if (v > t) right = m; else { left = m; break; }
        }
        sym_h[i] = (int16_t)left;
        if (i == n - 1) break;
        dec.consume(row[left], left == max_symbol ? 0x10000u : (uint32_t)row[left + 1]);
    }
    return 0;
}

extern "C" int linr_ac_encode_binary_batch(const float* const* prob_h, const uint8_t* const* sym_h, const int64_t* n,
                                           int32_t n_streams, uint8_t* const* out_h, const int64_t* cap, int64_t* out_len,
                                           int32_t n_threads) {
    if (n_streams < 0 || (n_streams > 0 && (!prob_h || !sym_h || !n || !out_h || !cap || !out_len))) return LINR_EINVAL;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_streams) n_threads = n_streams;
    std::atomic<int> next(0);
    std::atomic<int> err(0);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_streams) return;
            const int64_t r = linr_ac_encode_binary(prob_h[i], sym_h[i], n[i], out_h[i], cap[i]);
            out_len[i] = r;
            if (r < 0) err.store((int)r);
        }
    };
    if (n_threads <= 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(work);
        for (auto& t : pool) t.join();
    }
    return err.load();
}

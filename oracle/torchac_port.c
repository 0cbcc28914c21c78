/* Oracle (test infrastructure): plain-C restatement of the range coder of torchac 0.9.3.
 *
 * torchac is a third-party, un-vendored dependency of the reference (enviroment.yaml:32); its call sites are
 * models/module_utils.py:26-40 (binary occupancy streams) and model_compression/model_size_est.py:470-482,
 * 545-563 (model stream).  This file restates its published algorithm (32-bit low/high range coder over
 * 16-bit CDFs, pending-bit carry, MSB-first packing) and is pinned by the model stream implied by
 * loot/gop_32_62/70/result.json (35,319 bytes with the CPU's pdf, 35,320 with a last-bit-different CUDA pdf: both consistent
 * with the artefact, tests/test_oracle_golden.py computes both).
 *
 * cdf: [n_sym][lp] uint16 (already converted: see oracle/ac.py), sym: [n_sym] int16 in [0, lp-2].
 */
#include <stdint.h>
#include <stddef.h>

typedef struct { uint8_t *out; size_t cap, len; uint8_t cache; int count; } bitw_t;

static void put_bit(bitw_t *w, int bit) {
    w->cache = (uint8_t)((w->cache << 1) | (bit & 1));
    if (++w->count == 8) {
        if (w->len < w->cap) w->out[w->len] = w->cache;
        w->len++;
        w->count = 0;
        w->cache = 0;
    }
}

static void put_bit_and_pending(bitw_t *w, int bit, uint64_t *pending) {
    put_bit(w, bit);
    while (*pending > 0) { put_bit(w, !bit); (*pending)--; }
}

/* returns the number of bytes the stream needs (> cap means truncated) */
size_t oracle_ac_encode(const uint16_t *cdf, const int16_t *sym, size_t n_sym, int lp, uint8_t *out, size_t cap) {
    bitw_t w = { out, cap, 0, 0, 0 };
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint64_t pending = 0;
    const int max_symbol = lp - 2;
    for (size_t i = 0; i < n_sym; ++i) {
        const int s = sym[i];
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        const uint16_t *row = cdf + i * (size_t)lp;
        const uint32_t c_low = row[s];
        const uint32_t c_high = (s == max_symbol) ? 0x10000u : row[s + 1];
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> 16);
        for (;;) {
            if (high < 0x80000000u) {
                put_bit_and_pending(&w, 0, &pending);
                low <<= 1; high <<= 1; high |= 1;
            } else if (low >= 0x80000000u) {
                put_bit_and_pending(&w, 1, &pending);
                low <<= 1; high <<= 1; high |= 1;
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                pending++;
                low <<= 1; low &= 0x7FFFFFFFu;
                high <<= 1; high |= 0x80000001u;
            } else {
                break;
            }
        }
    }
    pending += 1;
    put_bit_and_pending(&w, low < 0x40000000u ? 0 : 1, &pending);
    while (w.count != 0) put_bit(&w, 0);
    return w.len;
}

typedef struct { const uint8_t *in; size_t len, pos; uint8_t cache; int cached; } bitr_t;

static void get_bit(bitr_t *r, uint32_t *value) {
    if (r->cached == 0) {
        if (r->pos == r->len) { *value <<= 1; return; }
        r->cache = r->in[r->pos++];
        r->cached = 8;
    }
    *value <<= 1;
    *value |= (uint32_t)((r->cache >> (r->cached - 1)) & 1);
    r->cached--;
}

static int bin_search(const uint16_t *row, uint16_t target, int max_sym) {
    int left = 0, right = max_sym + 1;
    while (left + 1 < right) {
        const int m = (left + right) / 2;
        const uint16_t v = row[m];
        if (v < target) left = m;
        else if (v > target) right = m;
        else return m;
    }
    return left;
}

void oracle_ac_decode(const uint16_t *cdf, size_t n_sym, int lp, const uint8_t *in, size_t in_len, int16_t *sym_out) {
    bitr_t r = { in, in_len, 0, 0, 0 };
    uint32_t low = 0, high = 0xFFFFFFFFu, value = 0;
    const int max_symbol = lp - 2;
    for (int i = 0; i < 32; ++i) get_bit(&r, &value);
    for (size_t i = 0; i < n_sym; ++i) {
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        const uint16_t count = (uint16_t)((((uint64_t)value - (uint64_t)low + 1) * 0x10000u - 1) / span);
        const uint16_t *row = cdf + i * (size_t)lp;
        const int s = bin_search(row, count, max_symbol);
        sym_out[i] = (int16_t)s;
        if (i == n_sym - 1) break;
        const uint32_t c_low = row[s];
        const uint32_t c_high = (s == max_symbol) ? 0x10000u : row[s + 1];
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> 16);
        for (;;) {
            if (low >= 0x80000000u || high < 0x80000000u) {
                low <<= 1; high <<= 1; high |= 1;
                get_bit(&r, &value);
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                low <<= 1; low &= 0x7FFFFFFFu;
                high <<= 1; high |= 0x80000001u;
                value -= 0x40000000u;
                get_bit(&r, &value);
            } else {
                break;
            }
        }
    }
}

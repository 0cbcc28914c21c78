// Parameter layout of LINR_PCGC_Model in parameters() order (models/model_core.py:31-35, models/upsample.py:43-76),
// shared by the fp32 executor (net.hip) and the bf16 / uint8-weight inference executor (net_bf16.hip).
#pragma once
#include <stdint.h>
#define MAX_SCALES 16
#define MAX_BL 4              // block_layers of block_in (main.py:521 default 1; the outter blocks always have 1, upsample.py:72-76)
struct IncP {                 // one InceptionResNet layer (models/resnet.py:7-60)
    int64_t c00_w, c00_b;     // conv0_0  conv3 8->4
    int64_t c01_w, c01_b;     // conv0_1  conv3 4->4
    int64_t c10_w, c10_b;     // conv1_0  1x1 8->4   kernel [8][4]
    int64_t c11_w, c11_b;     // conv1_1  conv3 4->4
    int64_t c12_w, c12_b;     // conv1_2  1x1 4->4   kernel [4][4]
};
struct BlockP {
    int cin;
    int nl;                   // Inception layers of the ResNetBlock (.2.layers.0 .. nl-1)
    int64_t a_w, a_b;         // .0   conv3 cin->8
    IncP inc[MAX_BL];
    int64_t b_w, b_b;         // .3   conv3 8->8
};

struct Layout {
    int S;
    int BL;                                        // block_layers of block_in
    int64_t emb;                                   // [S][8]
    int64_t m0_w[MAX_SCALES], m0_b[MAX_SCALES];    // Linear(15,16): weight [16][15]
    int64_t m2_w[MAX_SCALES], m2_b[MAX_SCALES];    // Linear(16,8):  weight [8][16]
    BlockP block_in;
    int64_t h0_w[8], h0_b[8], h2_w[8], h2_b[8];    // inner_mlps.k.0: Linear(8,24), Linear(24,1)
    int64_t pr_w[8], pr_b[8];                      // prune_blocks.k.0.conv: conv3 8->8
    BlockP outter[7];
    int64_t total;
};

static inline int64_t take(int64_t& cur, int64_t n) { int64_t o = cur; cur += n; return o; }

static inline void layout_block(BlockP& b, int cin, int nl, int64_t& cur) {
    b.cin = cin;
    b.nl = nl;
    b.a_w = take(cur, 27 * cin * 8);  b.a_b = take(cur, 8);
    for (int l = 0; l < nl; ++l) {
        IncP& q = b.inc[l];
        q.c00_w = take(cur, 27 * 8 * 4);  q.c00_b = take(cur, 4);
        q.c01_w = take(cur, 27 * 4 * 4);  q.c01_b = take(cur, 4);
        q.c10_w = take(cur, 8 * 4);       q.c10_b = take(cur, 4);
        q.c11_w = take(cur, 27 * 4 * 4);  q.c11_b = take(cur, 4);
        q.c12_w = take(cur, 4 * 4);       q.c12_b = take(cur, 4);
    }
    b.b_w = take(cur, 27 * 8 * 8);    b.b_b = take(cur, 8);
}

static inline bool make_layout(Layout& L, int S, int BL = 1) {
    if (S < 1 || S > MAX_SCALES || BL < 1 || BL > MAX_BL) return false;
    L.S = S;
    L.BL = BL;
    int64_t cur = 0;
    L.emb = take(cur, (int64_t)S * 8);
    for (int s = 0; s < S; ++s) {
        L.m0_w[s] = take(cur, 16 * 15); L.m0_b[s] = take(cur, 16);
        L.m2_w[s] = take(cur, 8 * 16);  L.m2_b[s] = take(cur, 8);
    }
    layout_block(L.block_in, 8, BL, cur);
    for (int k = 0; k < 8; ++k) {
        L.h0_w[k] = take(cur, 24 * 8); L.h0_b[k] = take(cur, 24);
        L.h2_w[k] = take(cur, 24);     L.h2_b[k] = take(cur, 1);
    }
    for (int k = 0; k < 8; ++k) { L.pr_w[k] = take(cur, 27 * 8 * 8); L.pr_b[k] = take(cur, 8); }
    for (int k = 0; k < 7; ++k) layout_block(L.outter[k], k + 1, 1, cur);
    L.total = cur;
    return true;
}


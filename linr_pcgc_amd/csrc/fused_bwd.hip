// Backward-data AND weight gradient of a 3x3x3 convolution from ONE gather of the output gradient (csrc/net.hip is the caller).
//
// Every conv-type kernel of the training step sits on the L1 gather floor (~17 us per row pass), and the
// backward of a convolution out = conv(in; W) used to make two such passes: backward-data gathers the output gradient
//     gin[i] = sum_k  g[nbr(i, 26 - k)] W[k]^T
// and the weight gradient gathered the input, gW[k] = sum_j in[nbr(j, k)]^T g[j].  Substituting j = nbr(i, 26 - k) turns the
// second into
//     gW[k] = sum_i  in[i]^T g[nbr(i, 26 - k)]
// - the SAME gathered rows, multiplied with the row's own input instead of the weights.  The kernel below is the
// backward-data convolution of csrc/fused.hip (lane = output row, weights register-resident, v_mfma_f32_4x4x1 with the
// weight 4-vector broadcast, taps in LINR_TAP order: bit-identical input gradients) that additionally parks every gathered
// row in a wave-private LDS image [tap][row][8 floats], 8 taps (one "chunk") at a time, and multiplies the chunk with the
// rows' own inputs from a (tap, channel quad) view of the image:
//   * the 64 lanes are four quarters of 16 (tap of the chunk, quad) pairs; quarter Q handles rows 16 Q .. 16 Q + 15 of the
//     wave's 64-row tile.  v_mfma_f32_4x4x1 with CBSZ = 2 broadcasts the A operand of block ABID *inside each quarter*, so the
//     four quarters multiply four different rows per instruction: B = the lane's gathered component g[nbr(row)][4 q + c]
//     (one ds_read_b128 per row), A = in[row][4 h .. 4 h + 3] held by the quarter's block (row % 4), D = gW[tap][4 h + i][4 q + c];
//   * a tap's slot in the image is 64 x 32 + 32 bytes, which makes the transposed ds_read_b128 of every 16-lane group
//     conflict-free (16-byte slot index = lane-in-quarter + 2 row (mod 16)); the writes are 2 KB contiguous per tap;
//   * the image is double-buffered and wave-private (LDS operations of one wave execute in order: no barrier in the row
//     loop): while the taps of chunk c are gathered, multiplied with the weights and written, the rows of chunk c - 1 are
//     multiplied with the inputs - two rows per tap, so the matrix cores always have independent work beside the gathers.
//     The last chunk of a tile is finished during the first taps of the wave's next tile.
// 880 MFMAs per 64-row tile (432 backward-data + 448 weight-gradient, of which 432 useful) keep ~400 registers alive: one wave per
// SIMD, one 256-thread block per CU, 130 KB of LDS; the block is persistent over `tiles_per_wave` tiles per wave, folds its 16
// (wave, quarter) partial sums in fixed order and writes ONE slab row - the reduction contract of every weight-gradient
// kernel (common.h: LinrWgradDst).  The same skeleton serves the two other convolution pairs of an Inception layer (KIND below).
//
// conv_bwd_wgrad_single_k below is this schedule in ONE instruction stream per SIMD; the launchers use its wave-specialised form
// conv_bwd_wgrad_k (csrc/fused_bwd_split.h: producer / consumer wave pairs, bit-identical results, 8 % faster) and keep the
// single-stream kernel selectable (LINR_FUSED_SPLIT=0) as the reference the parity test compares against.
#include "common.h"
#include "conv_common.h"
#include <stdlib.h>

#define FB_WAVES 4
#ifndef FB_LAB
#define FB_LAB 0                         // kernel-floor experiments (tools/fused_lab.sh): 1 no gathers, 2 no weight-gradient MFMAs,
#endif                                   // 4 no backward-data MFMAs, 8 no LDS traffic
#define FB_HP 1040                       // bytes per (tap, channel quad) plane: 64 rows x 16 B + 16 B  (65 x 16 B = 1 mod 16)
#define FB_TP (2 * FB_HP)                // bytes per tap slot of the 32-byte images (130 x 16 B = 2 mod 16)
#define FB_BUF (8 * FB_TP)               // one chunk: 8 taps x 32 B (KIND 0, 1) or 16 taps x 16 B (KIND 2)
#define FB_WAVE_BYTES (2 * FB_BUF)       // double-buffered

// The three convolution pairs of a block's backward pass that the kernel serves:
//   KIND 0  conv 8->8 (prune convolutions, tail convolutions, block_in's first convolution): g [n][8] gathered whole
//   KIND 1  the two 4->4 convolutions of an Inception layer: gathers [gI[:, 0:4] | gM[:, 0:4]] from two matrices, own rows H
//   KIND 2  conv0_0 8->4: gathers gH[:, 0:4] (16 bytes per tap; chunks of 16 taps), own rows A
struct FbArgs {
    const float* g;        // gathered output gradient, zero row at index -1; ld 8
    const float* g1;       // KIND 1: the second gathered matrix gM, ld 4, zero row at index -1
    const float* xin;      // own rows [n][8]: the convolution's input (KIND 0), H (KIND 1, also the ReLU mask), A (KIND 2, also the mask)
    const float* W;        // KIND 0: [27][8][8]; KIND 1: W01 [27][4][4]; KIND 2: W00 [27][8][4]      (ME layout [k][cin][cout])
    const float* W1;       // KIND 1: W11 [27][4][4]
    const float* res;      // KIND 2: gI [n][8], added to the input gradient
    float* out;            // [n][8] input gradient
    unsigned flags;        // KIND 2: LINR_RELU_MASK (mask by xin > 0), LINR_ACCUM (+= old out)
    int tiles_per_wave;
    int nb_slab;           // slab rows the reduction will read: rows >= gridDim.x get zeros for this kernel's parameters
};
struct FbDst2 { int64_t w_off1, b_off1; };       // KIND 1: slab offsets of the second convolution

template <int KIND> struct FbT {
    static constexpr int CT = KIND == 2 ? 16 : 8;               // taps per chunk
    static constexpr int NCH = (27 + CT - 1) / CT;              // chunks per tile
    static constexpr int XN = KIND == 2 ? 1 : 2;                // 16-byte pieces gathered per tap
    static constexpr int WGM = KIND == 1 ? 4 : 8;               // weight-gradient MFMAs per row
    static constexpr int ntaps(int ch) { return 27 - ch * CT < CT ? 27 - ch * CT : CT; }
    // row sets of a chunk: KIND 0 multiplies its 3-tap last chunk in eighths (CBSZ = 1), everything else in quarters
    static constexpr int sets(int ch) { return (KIND == 0 && ch == NCH - 1) ? 8 : 4; }
    static constexpr int prows(int ch) { return 64 / sets(ch); }
    static constexpr int WELEMS = KIND == 0 ? 1728 : KIND == 1 ? 432 : 864;      // kernel elements (per convolution)
    static constexpr int BELEMS = KIND == 0 ? 8 : 4;
};

// EPI 3 (KIND 0 only): also gM = (gin[4:8] @ W12^T) * (M > 0)  (PwArgs as in cconv_mfma_k)
template <int KIND, int EPI>
__global__ __launch_bounds__(FB_WAVES * 64, 1) void conv_bwd_wgrad_single_k(FbArgs a, const int32_t* __restrict__ lo,
                                                                    const uint32_t* __restrict__ mask, int64_t ld, int64_t n,
                                                                    PwArgs pw, LinrWgradDst d, FbDst2 d2, Grp gp) {
    using T = FbT<KIND>;
    constexpr int CT = T::CT, NCH = T::NCH, XN = T::XN, WGM = T::WGM;
    __shared__ float4 smem[FB_WAVES * FB_WAVE_BYTES / 16];
    __shared__ float sbias[FB_WAVES][8];
    {   // group offsets: in = g, e5 = g1, res = xin, w = W, e6 = W1, act = res, out; e0..e2 = pointwise epilogue (KIND 0 / 2) or
        // e0 / e1 = slab offsets of the second convolution (KIND 1), e1 / e2 = of conv1_0 (KIND 2), e5 / e6 = of conv1_2 (EPI 3);
        // e3 / e4 = slab offsets of kernel / bias
        const int gi = blockIdx.y;
        a.g += gp.in[gi]; a.xin += gp.res[gi]; a.W += gp.w[gi]; a.out += gp.out[gi];
        if constexpr (KIND == 1) { a.g1 += gp.e5[gi]; a.W1 += gp.e6[gi]; d2.w_off1 += gp.e0[gi]; d2.b_off1 += gp.e1[gi]; }
        if constexpr (KIND == 2) { if (a.res) a.res += gp.act[gi]; pw.w += gp.e0[gi]; d2.w_off1 += gp.e1[gi]; d2.b_off1 += gp.e2[gi]; }
        if constexpr (EPI == 3) { pw.w += gp.e0[gi]; pw.aux += gp.e1[gi]; pw.aux_out += gp.e2[gi]; d2.w_off1 += gp.e5[gi]; d2.b_off1 += gp.e6[gi]; }
        d.w_off += gp.e3[gi]; d.b_off += gp.e4[gi];
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // backward-data weights, register-resident, as A-operand images of v_mfma_f32_4x4x1 (block = lane >> 2, j = lane & 3):
    //   KIND 0 (cconv_mfma_k<8, 8, BWD>):  wv[g][i] block b = W(k = 8 g + b / 2, gathered channel i, produced channels 4 (b % 2) + j)
    //   KIND 1 (cconv_dual44_k<BWD>):      wv[g][i] block b = (b even ? W01 : W11)(k = 8 g + b / 2, gathered i, produced j)
    //   KIND 2 (cconv_mfma_k<4, 8, BWD>):  wv[g][i] block b = W00(k = 8 g + b / 2, gathered i, produced 4 (b % 2) + j)
    constexpr int WI = KIND == 0 ? 8 : 4;
    float wv[4][WI];
    {
        const int blk = lane >> 2, j = lane & 3;
        const int kl = blk >> 1, co = 4 * (blk & 1) + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k = g * 8 + kl;
#pragma unroll
            for (int i = 0; i < WI; ++i) {
                float w = 0.0f;
                if (k < 27) {
                    if constexpr (KIND == 0) w = a.W[(k * 8 + co) * 8 + i];
                    if constexpr (KIND == 1) w = ((blk & 1) ? a.W1 : a.W)[(k * 4 + j) * 4 + i];
                    if constexpr (KIND == 2) w = a.W[(k * 8 + co) * 4 + i];
                }
                wv[g][i] = w;
            }
        }
    }
    // weight-gradient roles of the lane.  Quarters (CBSZ = 2; KIND 1: pairs of blocks inside a quarter, CBSZ = 1): quarter Q
    // multiplies rows 16 Q .. 16 Q + 15 of the wave's tile; the lane reads plane (slot, quad) of the image:
    //   KIND 0: 16 lanes = 8 slots x 2 quads (gathered channels 4 q ..), A = own row's channels 4 h .. (both halves) from block r % 4
    //   KIND 1: 16 lanes = 2 convolutions x 8 slots, A = H[row][4 conv ..] from block r % 2 of the convolution's 8 lanes
    //   KIND 2: 16 lanes = 16 slots (one plane per tap), A as KIND 0
    const int Q = lane >> 4;
    const int wq = KIND == 0 ? (lane & 1) : KIND == 1 ? ((lane >> 3) & 1) : 0;
    const int wslot = KIND == 0 ? ((lane & 15) >> 1) : KIND == 1 ? (lane & 7) : (lane & 15);
    const int ablk = KIND == 1 ? ((lane >> 2) & 1) : ((lane >> 2) & 3), ai = lane & 3;
    char* img = reinterpret_cast<char*>(smem) + wave * FB_WAVE_BYTES;
    char* imgW = img + lane * 16;
    const char* imgR = img + (KIND == 2 ? wslot * FB_HP : wslot * FB_TP + wq * FB_HP) + (16 * Q) * 16;
    // KIND 0: the last chunk holds 3 taps only (27 = 3 x 8 + 3): six (slot, quad) pairs.  It runs with CBSZ = 1 - EIGHT row sets of
    // 8 lanes, A broadcast inside each pair of blocks - so its 16 rows of a quarter become 8 rows of an eighth: 64 MFMAs, not 128.
    const int E8 = lane >> 3, wq3 = lane & 1, wslot3 = (lane & 7) >> 1, ablk3 = (lane >> 2) & 1;
    const char* imgR3 = img + FB_BUF + wslot3 * FB_TP + wq3 * FB_HP + (8 * E8) * 16;
    f32x4 wacc[NCH][8];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int c = 0; c < 8; ++c) wacc[ch][c] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.0f;
    // own-row A operands: xa = this tile, layout of the quarters; xlp = the PREVIOUS tile in the layout of the last chunk (whose rows
    // are multiplied during the first chunk of the next tile); xl = this tile in that layout (KIND 0: eighths; else = xa)
    float xa[8], xl[8], xlp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xa[j] = 0.0f; xl[j] = 0.0f; xlp[j] = 0.0f; }
    // the first tile multiplies "the last chunk of the previous tile" with xlp = 0: that buffer must hold finite numbers
    for (int o = lane * 16; o < FB_BUF; o += 64 * 16)
        *reinterpret_cast<float4*>(img + ((NCH - 1) & 1) * FB_BUF + o) = make_float4(0.f, 0.f, 0.f, 0.f);

    const int64_t T64 = (n + 63) >> 6;
    const int64_t tb0 = (int64_t)blockIdx.x * (FB_WAVES * a.tiles_per_wave);
    const int64_t tb1 = (tb0 + FB_WAVES * a.tiles_per_wave < T64) ? tb0 + FB_WAVES * a.tiles_per_wave : T64;
    const char* pad = reinterpret_cast<const char*>(a.g - 8);
    const char* pad1 = KIND == 1 ? reinterpret_cast<const char*>(a.g1 - 4) : nullptr;
    // With one wave per SIMD nothing hides a latency but the wave's own instruction stream, so the row loop is software-pipelined
    // ACROSS tiles: the gathers run PF taps ahead (a ring of RING = PF + 1 rows; 27 % RING == 0 keeps the ring slots compile-time
    // constants from tile to tile), the last PF taps of a tile already gather the first taps of the wave's next tile, whose index
    // words were loaded at step 1 and decoded at step 10 and whose own-row inputs were loaded at step 3; the transposed image
    // reads of a step are issued one step ahead.
    constexpr int PF = 8, RING = 9;
    static_assert(27 % RING == 0 && PF + 1 == RING, "ring slots must not depend on the tile");
    // Weight-gradient MFMA number m of the rows [r0, ...) of chunk pc (whose image the caller has read into b[]): row r0 + m / WGM.
    // B = the lane's transposed read of its (slot, quad) plane, A = own-row inputs XA broadcast inside the row set.
    auto wg_mfma = [&](auto pcc, auto r0c, auto mc, const float4* b, const float (&XA)[8]) {
        constexpr int pc = decltype(pcc)::value, r0 = decltype(r0c)::value, m = decltype(mc)::value;
        constexpr int j = m / WGM, c = m % 4, r = r0 + j;
        const float B = c == 0 ? b[j].x : c == 1 ? b[j].y : c == 2 ? b[j].z : b[j].w;
        if constexpr (KIND == 1) {          // accumulator c: gW(conv)[tap][own channel i][gathered channel c]
            wacc[pc][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(XA[r >> 1], B, wacc[pc][c], 1, r & 1, 0);
        } else {                            // accumulator (c, h): gW[tap][own channel 4 h + i][gathered channel 4 q + c]
            constexpr int h = (m % 8) / 4;
            if constexpr (T::sets(pc) == 8)
                wacc[pc][c * 2 + h] = __builtin_amdgcn_mfma_f32_4x4x1f32(XA[2 * (r >> 1) + h], B, wacc[pc][c * 2 + h], 1, r & 1, 0);
            else
                wacc[pc][c * 2 + h] = __builtin_amdgcn_mfma_f32_4x4x1f32(XA[2 * (r >> 2) + h], B, wacc[pc][c * 2 + h], 2, r & 3, 0);
        }
    };
    // index words of a row (9 column bases + the 27-bit mask of the compressed map) and their decode into the 27 byte offsets of
    // the mirrored taps (decode_offsets<true> split in two so that the loads' latency lies behind a few taps of MFMAs); offsets
    // are in units of 32-byte rows (KIND 1 halves them for its 16-byte matrix)
    const char* lob = reinterpret_cast<const char*>(lo);
    const uint32_t ld4 = (uint32_t)ld << 2;
    auto idx_load = [&](int64_t row, int32_t (&raw)[10]) {
        const uint32_t rb = (uint32_t)row << 2;
        raw[9] = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(mask) + rb);
#pragma unroll
        for (int q = 0; q < 9; ++q) raw[q] = *reinterpret_cast<const int32_t*>(lob + (rb + (uint32_t)q * ld4));
    };
    auto idx_decode = [&](const int32_t (&raw)[10], uint32_t (&off)[27]) {
        const uint32_t m = (uint32_t)raw[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const uint32_t L = ((uint32_t)raw[q] + 1u) << 5;
            const int m0 = __builtin_amdgcn_sbfe(m, 3 * q, 1), m1 = __builtin_amdgcn_sbfe(m, 3 * q + 1, 1),
                      m2 = __builtin_amdgcn_sbfe(m, 3 * q + 2, 1);
            const uint32_t t1 = L + (uint32_t)__mul24(m0, -32);
            const uint32_t t2 = t1 + (uint32_t)__mul24(m1, -32);
            off[26 - q] = L & (uint32_t)m0; off[26 - (q + 9)] = t1 & (uint32_t)m1; off[26 - (q + 18)] = t2 & (uint32_t)m2;
        }
    };
    auto gather = [&](uint32_t o, f32x4 (&xr)[XN]) {
        xr[0] = *reinterpret_cast<const f32x4*>(pad + o);
        if constexpr (KIND == 0) xr[1] = *reinterpret_cast<const f32x4*>(pad + o + 16);
        if constexpr (KIND == 1) xr[1] = *reinterpret_cast<const f32x4*>(pad1 + (o >> 1));
    };
    auto xa_load = [&](int64_t row0, float (&XA)[8]) {
        if constexpr (KIND == 1) {
#pragma unroll
            for (int rq = 0; rq < 8; ++rq) {
                const int64_t r = row0 + 16 * Q + 2 * rq + ablk;
                XA[rq] = (r < n) ? a.xin[r * 8 + 4 * wq + ai] : 0.0f;
            }
        } else {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int64_t r = row0 + 16 * Q + 4 * rq + ablk;
#pragma unroll
                for (int h = 0; h < 2; ++h) XA[2 * rq + h] = (r < n) ? a.xin[r * 8 + 4 * h + ai] : 0.0f;
            }
        }
    };
    auto xl_load = [&](int64_t row0, float (&XL)[8]) {         // KIND 0: the eighths' layout
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int64_t r = row0 + 8 * E8 + 2 * rq + ablk3;
#pragma unroll
            for (int h = 0; h < 2; ++h) XL[2 * rq + h] = (r < n) ? a.xin[r * 8 + 4 * h + ai] : 0.0f;
        }
    };
    float w12[16];                               // EPI 3: the 1x1 kernel of the epilogue, read once
#pragma unroll
    for (int j = 0; j < 16; ++j) w12[j] = 0.0f;
    if constexpr (EPI == 3) {
#pragma unroll
        for (int j = 0; j < 16; ++j) w12[j] = pw.w[j];
    }
    // EPI 3: conv1_2 (1x1, 4 -> 4) rides along: its weight gradient sum_rows M[row]^T gI[row][4:8] is a product of two OWN-row
    // quantities the epilogue holds anyway - 16 + 4 per-lane accumulators (plain VALU, 20 FMAs per tile), one wave reduction per block
    float g12[20];
#pragma unroll
    for (int j = 0; j < 20; ++j) g12[j] = 0.0f;
    float w10[32];                               // KIND 2: conv1_0's 1x1 kernel [8][4] of the epilogue, read once
#pragma unroll
    for (int j = 0; j < 32; ++j) w10[j] = 0.0f;
    if constexpr (KIND == 2) {
#pragma unroll
        for (int j = 0; j < 32; ++j) w10[j] = pw.w[j];
    }
    int64_t tile = tb0 + wave;
    if (tile < tb1) {                            // wave-uniform
        uint32_t off[27], offn[27];
        int32_t raw[10];
        float xan[8], xln[8];
        f32x4 x[RING][XN];                       // the ring of gathered rows: 16-byte register tuples
        float4 bq[2][6];
        {
            const int64_t r = (tile << 6) + lane;
            idx_load(r < n ? r : n - 1, raw);
            idx_decode(raw, off);
            xa_load(tile << 6, xa);
            if constexpr (KIND == 0) xl_load(tile << 6, xl);
#pragma unroll
            for (int u = 0; u < PF; ++u) gather(off[LINR_TAP(u)], x[u]);
            // step 0 multiplies the first rows of "the last chunk of the previous tile" (the zeroed buffer; xlp = 0): EVERY register
            // it will feed to the matrix cores is loaded - 0 x (whatever the register held) is only 0 while that is finite
            constexpr int nr0 = T::prows(NCH - 1) / T::ntaps(0);
#pragma unroll
            for (int j = 0; j < nr0; ++j)
                bq[0][j] = *reinterpret_cast<const float4*>((KIND == 0 ? imgR3 : imgR + ((NCH - 1) & 1) * FB_BUF) + j * 16);
        }
        for (; tile < tb1; tile += FB_WAVES) {
            const int64_t row_raw = (tile << 6) + lane;
            const bool live = row_raw < n;
            const int64_t row = live ? row_raw : n - 1;            // every lane stays in the MFMAs (they ignore EXEC)
            const int64_t ntile = (tile + FB_WAVES < tb1) ? tile + FB_WAVES : tile;      // the last tile "prefetches" itself
            const int64_t nrow_raw = (ntile << 6) + lane;
            const int64_t nrow = nrow_raw < n ? nrow_raw : n - 1;
            f32x4 acc[2] = {(f32x4){0.0f, 0.0f, 0.0f, 0.0f}, (f32x4){0.0f, 0.0f, 0.0f, 0.0f}};
            // own-row operands of the epilogue, requested early (step 5): EPI 3: M; KIND 1: H; KIND 2: gI, gH[4:8], A
            float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0, e2 = e0, e3 = e0, e4 = e0;
            __builtin_amdgcn_sched_barrier(0);
            static_for<27>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                constexpr int k = LINR_TAP(kk);
                constexpr int g = k / 8, ab = (k % 8) * 2;
                constexpr int ch = kk / CT, slot = kk % CT;
                constexpr int pc = (ch + NCH - 1) % NCH;           // the chunk whose rows are multiplied beside this tap
                constexpr int nst = T::ntaps(ch);                  // taps of this chunk: the rows of chunk pc spread over them
                constexpr int r0 = (slot * T::prows(pc)) / nst, nr = ((slot + 1) * T::prows(pc)) / nst - r0;
                // the same for the next step (step 0 of the next tile behind step 26)
                constexpr int kn = (kk + 1) % 27, chn = kn / CT, slotn = kn % CT, pcn = (chn + NCH - 1) % NCH, nstn = T::ntaps(chn);
                constexpr int r0n = (slotn * T::prows(pcn)) / nstn, nrn = ((slotn + 1) * T::prows(pcn)) / nstn - r0n;
                if constexpr (!(FB_LAB & 1)) gather(kk + PF < 27 ? off[LINR_TAP((kk + PF) % 27)] : offn[LINR_TAP((kk + PF) % 27)], x[(kk + PF) % RING]);
                if constexpr (kk == 1) idx_load(nrow, raw);
                if constexpr (kk == 3) xa_load(ntile << 6, xan);
                if constexpr (KIND == 0 && kk == 4) xl_load(ntile << 6, xln);
                if constexpr (kk == 10) idx_decode(raw, offn);
                if constexpr (kk == 5) {
                    if constexpr (EPI == 3) e0 = *reinterpret_cast<const float4*>(pw.aux + row * 4);
                    if constexpr (KIND == 1) {
                        e0 = *reinterpret_cast<const float4*>(a.xin + row * 8);
                        e1 = *reinterpret_cast<const float4*>(a.xin + row * 8 + 4);
                    }
                    if constexpr (KIND == 2) {
                        e0 = *reinterpret_cast<const float4*>(a.res + row * 8);
                        e1 = *reinterpret_cast<const float4*>(a.res + row * 8 + 4);
                        e2 = *reinterpret_cast<const float4*>(a.g + row * 8 + 4);
                        if (a.flags & LINR_RELU_MASK) {
                            e3 = *reinterpret_cast<const float4*>(a.xin + row * 8);
                            e4 = *reinterpret_cast<const float4*>(a.xin + row * 8 + 4);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4(&xk)[XN] = x[kk % RING];
                // Pin: the MFMAs below consume x[kk] only from here on.  Without it instruction selection slides the whole
                // backward-data MFMA chain PF taps up, right behind each load (sched_barrier orders the machine scheduler,
                // not the DAG), and every gather is waited for the moment it is issued.
                if constexpr (XN == 2) asm volatile("" : "+v"(xk[0]), "+v"(xk[1]));
                else asm volatile("" : "+v"(xk[0]));
                if constexpr (!(FB_LAB & 8)) {
                    if constexpr (KIND == 2) {
                        *reinterpret_cast<f32x4*>(imgW + (ch & 1) * FB_BUF + slot * FB_HP) = xk[0];
                    } else {
                        *reinterpret_cast<f32x4*>(imgW + (ch & 1) * FB_BUF + slot * FB_TP) = xk[0];
                        *reinterpret_cast<f32x4*>(imgW + (ch & 1) * FB_BUF + slot * FB_TP + FB_HP) = xk[XN - 1];
                    }
                }
                // next step's transposed reads: behind this tap's write (a new chunk reads the buffer just completed) - except at
                // step 26, whose successor (step 0) shares the register buffer
                auto read_next = [&]() {
                    if constexpr (FB_LAB & 8) return;
#pragma unroll
                    for (int j = 0; j < nrn; ++j)
                        bq[kn & 1][j] = (KIND == 0 && pcn == NCH - 1) ? *reinterpret_cast<const float4*>(imgR3 + (r0n + j) * 16)
                                                                      : *reinterpret_cast<const float4*>(imgR + (pcn & 1) * FB_BUF + (r0n + j) * 16);
                };
                if constexpr (kk != 26) read_next();
                // backward-data MFMAs (two dependent chains) with the weight-gradient MFMAs of chunk pc (independent
                // accumulators) between the pairs, in source order; the machine scheduler is left free to regroup them
                // (pinning every pair with a scheduling barrier measured 1.3 % of the step slower)
                constexpr int NBW = KIND == 0 ? 8 : 4;             // backward-data MFMA pairs of a tap
                static_for<NBW>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (!(FB_LAB & 4)) {
                        if constexpr (KIND == 0) {
                            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[i / 4][i % 4], acc[0], 4, ab, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[i / 4][i % 4], acc[1], 4, ab + 1, 0);
                        } else if constexpr (KIND == 1) {
                            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[0][i], acc[0], 4, ab, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[1][i], acc[1], 4, ab + 1, 0);
                        } else {
                            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[0][i], acc[0], 4, ab, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[0][i], acc[1], 4, ab + 1, 0);
                        }
                    }
                    if constexpr (!(FB_LAB & 2)) {
                        constexpr int tot = nr * WGM, lo_m = (i * tot) / NBW, hi_m = ((i + 1) * tot) / NBW;
                        static_for<hi_m - lo_m>([&](auto tc) {
                            constexpr int m = lo_m + decltype(tc)::value;
                            if constexpr (ch == 0) wg_mfma(std::integral_constant<int, pc>{}, std::integral_constant<int, r0>{},
                                                           std::integral_constant<int, m>{}, bq[kk & 1], xlp);
                            else wg_mfma(std::integral_constant<int, pc>{}, std::integral_constant<int, r0>{},
                                         std::integral_constant<int, m>{}, bq[kk & 1], xa);
                        });
                    }
                });
                if constexpr (KIND == 2 && kk == 26 && !(FB_LAB & 8)) {
                    // conv1_0 (1x1, 8 -> 4) rides along: its weight gradient sum_rows A[row]^T gH[row][4:8] is the product the idle
                    // lanes of the last chunk (11 taps on 16 slots) would compute if slot 11 held the rows' OWN gH[:, 4:8] - which
                    // the epilogue has in registers anyway (e2).  One LDS write per tile instead of a pointwise weight-gradient launch.
                    *reinterpret_cast<float4*>(imgW + ((NCH - 1) & 1) * FB_BUF + 11 * FB_HP) = e2;
                    if (live) { bsum[4] += e2.x; bsum[5] += e2.y; bsum[6] += e2.z; bsum[7] += e2.w; }
                }
                if constexpr (kk == 26) read_next();
                if constexpr (k == 13) {                           // the centre tap is the row's own gradient: bias gradient
                    if (live) {
#pragma unroll
                        for (int j = 0; j < 4 * XN; ++j) bsum[j] += xk[j / 4][j % 4];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int j = 0; j < 27; ++j) off[j] = offn[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (KIND == 0) { xlp[j] = xl[j]; xl[j] = xln[j]; } else xlp[j] = xa[j];
                xa[j] = xan[j];
            }
            if (live) {
                float o[8];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[4 * h + j] = acc[h][j];
                if constexpr (EPI == 3) {          // gM = (gin[4:8] @ W12^T) * (M > 0)   (W12 [4][4])
                    const float mv[4] = {e0.x, e0.y, e0.z, e0.w};
                    float gm[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t = 0.0f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) t = fmaf(o[4 + q], w12[i * 4 + q], t);
                        gm[i] = mv[i] > 0.0f ? t : 0.0f;
                    }
                    *reinterpret_cast<float4*>(pw.aux_out + row * 4) = make_float4(gm[0], gm[1], gm[2], gm[3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) g12[i * 4 + q] = fmaf(mv[i], o[4 + q], g12[i * 4 + q]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) g12[16 + q] += o[4 + q];
                }
                if constexpr (KIND == 1) {         // gH = [bwd(gI[:, 0:4]; W01) | bwd(gM; W11)] * (H > 0)
                    const float hv[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = hv[j] > 0.0f ? o[j] : 0.0f;
                }
                float* op = a.out + row * 8;
                if constexpr (KIND == 2) {         // cconv_mfma_k's EPI 4: + gI, + old (ACCUM), + gH[4:8] @ W10^T, * (A > 0)
                    const float rv[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += rv[j];
                    if (a.flags & LINR_ACCUM) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] += op[j];
                    }
                    const float gq[4] = {e2.x, e2.y, e2.z, e2.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float t = 0.0f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) t = fmaf(gq[q], w10[i * 4 + q], t);
                        o[i] += t;
                    }
                    if (a.flags & LINR_RELU_MASK) {
                        const float av[8] = {e3.x, e3.y, e3.z, e3.w, e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = av[j] > 0.0f ? o[j] : 0.0f;
                    }
                }
                *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
        {   // the last chunk of the wave's last tile
            constexpr int pr = T::prows(NCH - 1);
            float4 b[pr];
#pragma unroll
            for (int j = 0; j < pr; ++j)
                b[j] = *reinterpret_cast<const float4*>((KIND == 0 ? imgR3 : imgR + ((NCH - 1) & 1) * FB_BUF) + j * 16);
            static_for<pr * WGM>([&](auto mc) {
                wg_mfma(std::integral_constant<int, NCH - 1>{}, std::integral_constant<int, 0>{}, mc, b, xlp);
            });
        }
    }
    __syncthreads();
    // ---- fold: the (wave, row set) partials of every element in fixed order, one slab row per block ---------------------------
    float* sacc = reinterpret_cast<float*>(smem);                 // [wave][lane][33]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = bsum[j];
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) t += __shfl_xor(t, dd, 64);
        if (lane == 0) sbias[wave][j] = t;
    }
    float* dst = d.base + (int64_t)blockIdx.x * d.block_stride;
    const int tid = threadIdx.x;
    if constexpr (EPI == 3) {          // conv1_2's gradients: fixed shuffle tree inside the wave, the four waves in order
        __shared__ float s12[FB_WAVES][20];
#pragma unroll
        for (int j = 0; j < 20; ++j) {
            float t = g12[j];
#pragma unroll
            for (int dd = 32; dd > 0; dd >>= 1) t += __shfl_xor(t, dd, 64);
            if (lane == 0) s12[wave][j] = t;
        }
        __syncthreads();
        if (tid < 20) {
            const float t = ((s12[0][tid] + s12[1][tid]) + s12[2][tid]) + s12[3][tid];
            dst[(tid < 16 ? d2.w_off1 + tid : d2.b_off1 + (tid - 16))] = t;
        }
    }
    static_for<NCH>([&](auto chc) {
        constexpr int ch = decltype(chc)::value;
        constexpr int ntaps = T::ntaps(ch);
        constexpr int sets = T::sets(ch), width = 64 / sets;      // row sets per wave and the lanes of one
        float* mine = sacc + (wave * 64 + lane) * 33;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) mine[c * 4 + i] = wacc[ch][c][i];
        __syncthreads();
        constexpr int per_tap = KIND == 0 ? 64 : 32;              // outputs per tap (KIND 1: both convolutions)
        constexpr int nslots = ntaps + ((KIND == 2 && ch == NCH - 1) ? 1 : 0);        // KIND 2: slot 11 of the last chunk = conv1_0
        for (int e = tid; e < nslots * per_tap; e += FB_WAVES * 64) {
            const int slot = e / per_tap, r = e % per_tap;
            int kp, idx, dofs;
            const int kk = CT * ch + slot;
            const int k = kk / 9 + 3 * ((kk / 3) % 3) + 9 * (kk % 3);
            if constexpr (KIND == 0) {          // r = ci * 8 + co; lane (2 slot + co / 4), accumulator (co % 4, ci / 4), register ci % 4
                const int ci = r >> 3, co = r & 7;
                kp = 2 * slot + (co >> 2); idx = ((co & 3) * 2 + (ci >> 2)) * 4 + (ci & 3);
                dofs = (int)d.w_off + k * 64 + r;
            } else if constexpr (KIND == 1) {   // r = conv * 16 + ci * 4 + co; lane (8 conv + slot), accumulator co, register ci
                const int cv = r >> 4, ci = (r >> 2) & 3, co = r & 3;
                kp = 8 * cv + slot; idx = co * 4 + ci;
                dofs = (int)(cv ? d2.w_off1 : d.w_off) + k * 16 + (r & 15);
            } else {                            // r = ci * 4 + co; lane slot, accumulator (co, ci / 4), register ci % 4
                const int ci = r >> 2, co = r & 3;
                kp = slot; idx = (co * 2 + (ci >> 2)) * 4 + (ci & 3);
                dofs = slot < ntaps ? (int)d.w_off + k * 32 + r : (int)d2.w_off1 + r;          // conv1_0.kernel [8][4]
            }
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < FB_WAVES; ++w)
#pragma unroll
                for (int qq = 0; qq < sets; ++qq) t += sacc[(w * 64 + width * qq + kp) * 33 + idx];
            dst[dofs] = t;
        }
        __syncthreads();
    });
    if (tid < (KIND == 2 ? 8 : 4 * XN)) {
        const float t = ((sbias[0][tid] + sbias[1][tid]) + sbias[2][tid]) + sbias[3][tid];
        if constexpr (KIND == 0) dst[d.b_off + tid] = t;
        else dst[(tid < 4 ? d.b_off : d2.b_off1) + (tid & 3)] = t;          // KIND 1: second convolution; KIND 2: conv1_0
    }
    for (int64_t r = (int64_t)blockIdx.x + gridDim.x; r < a.nb_slab; r += gridDim.x) {
        float* z = d.base + r * d.block_stride;
        for (int e = tid; e < T::WELEMS; e += FB_WAVES * 64) {
            z[d.w_off + e] = 0.0f;
            if constexpr (KIND == 1) z[d2.w_off1 + e] = 0.0f;
        }
        if (tid < T::BELEMS) {
            z[d.b_off + tid] = 0.0f;
            if constexpr (KIND != 0) z[d2.b_off1 + tid] = 0.0f;
        }
        if constexpr (KIND == 2) { if (tid < 32) z[d2.w_off1 + tid] = 0.0f; }
        if constexpr (EPI == 3) { if (tid < 20) z[(tid < 16 ? d2.w_off1 + tid : d2.b_off1 + (tid - 16))] = 0.0f; }
    }
}

#include "fused_bwd_split.h"

// LINR_FUSED_SPLIT=0 selects the single-stream kernels (conv_bwd_wgrad_single_k), anything else the wave-specialised ones
static bool fb_split() {
    const char* e = getenv("LINR_FUSED_SPLIT");
    return !(e && e[0] == '0');
}

// Grid of the fused kernels.  One block per CU is resident (registers, LDS) and a block's prologue (weights, index decode,
// first gathers) and epilogue (fold, slab row) run with idle matrix cores, so the launch is sized as ONE round of
// long-lived blocks: about CUs / groups blocks per group (never more than the slab's nb rows), m tiles per wave.
static int fb_cus() {
    static const int v = [] {
        const char* e = getenv("LINR_FUSED_CUS");
        if (e && atoi(e) > 0) return atoi(e);
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    return v;
}
static inline void fb_grid(int64_t n, int nb, int ngroups, int& tiles_per_wave, int& blocks) {
    const int64_t t64 = (n + 63) >> 6;
    int64_t target = fb_cus() / (ngroups < 1 ? 1 : ngroups);
    if (target < 1) target = 1;
    if (target > nb) target = nb;
    int64_t m = (t64 + FB_WAVES * target - 1) / (FB_WAVES * target);
    if (m < 1) m = 1;
    tiles_per_wave = (int)m;
    blocks = (int)((t64 + FB_WAVES * m - 1) / (FB_WAVES * m));
    if (blocks < 1) blocks = 1;
}

// slab rows a fused launch over `ngroups` groups writes (= its blocks per group)
int linr_fused_bwd_rows(int64_t n, int nb, int ngroups) {
    int tpw = 1, blocks = 1;
    fb_grid(n, nb, ngroups, tpw, blocks);
    return blocks;
}

// g: output gradient (gathered), xin: the convolution's input, W: its kernel; out: input gradient; slab partials into d.
// rows_written == nullptr: rows 0 .. nb - 1 of the slab are all written (the rows beyond the grid's blocks get zeros);
// otherwise only the grid's rows are written and *rows_written tells the caller how many (its reduction must stop there).
// pw != nullptr selects the gM epilogue, which also produces conv1_2's kernel / bias gradient (M^T gin[:, 4:8]; slab offsets
// w12_off / b12_off).
int linr_conv88_bwd_wgrad_launch(const float* g, const float* xin, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                                 const float* W, float* out, const PwArgs* pw, LinrWgradDst d, int nb, hipStream_t s,
                                 const Grp* gp, int ngroups, int* rows_written, int64_t w12_off, int64_t b12_off) {
    if (rows_written) *rows_written = 0;
    if (n == 0) return 0;
    const Grp g0 = gp ? *gp : Grp();
    FbArgs a = {g, nullptr, xin, W, nullptr, nullptr, out, 0u, 1, rows_written ? 0 : nb};
    int blocks = 1;
    fb_grid(n, nb, ngroups, a.tiles_per_wave, blocks);
    if (rows_written) *rows_written = blocks;
    const dim3 grid(blocks, ngroups);
    if (fb_split()) {
        if (pw) conv_bwd_wgrad_k<0, 3><<<grid, FS_THREADS, 0, s>>>(a, lo, mask, ld, n, *pw, d, FbDst2{w12_off, b12_off}, g0);
        else conv_bwd_wgrad_k<0, 0><<<grid, FS_THREADS, 0, s>>>(a, lo, mask, ld, n, PwArgs(), d, FbDst2{0, 0}, g0);
    } else if (pw) conv_bwd_wgrad_single_k<0, 3><<<grid, FB_WAVES * 64, 0, s>>>(a, lo, mask, ld, n, *pw, d, FbDst2{w12_off, b12_off}, g0);
    else conv_bwd_wgrad_single_k<0, 0><<<grid, FB_WAVES * 64, 0, s>>>(a, lo, mask, ld, n, PwArgs(), d, FbDst2{0, 0}, g0);
    return linr_launch_rc();
}

// both 4->4 convolutions of an Inception layer: gH = [bwd(gI[:, 0:4]; W01) | bwd(gM; W11)] * (H > 0) and the two kernel / bias
// gradients from one gather of [gI[:, 0:4] | gM]
int linr_dual44_bwd_wgrad_launch(const float* gI, const float* gM, const float* H, const int32_t* lo, const uint32_t* mask,
                                 int64_t ld, int64_t n, const float* w01, const float* w11, float* gH, float* big,
                                 int64_t block_stride, int64_t w_off0, int64_t b_off0, int64_t w_off1, int64_t b_off1, int nb,
                                 hipStream_t s, const Grp* gp, int ngroups, int* rows_written) {
    if (rows_written) *rows_written = 0;
    if (n == 0) return 0;
    const Grp g0 = gp ? *gp : Grp();
    FbArgs a = {gI, gM, H, w01, w11, nullptr, gH, 0u, 1, rows_written ? 0 : nb};
    int blocks = 1;
    fb_grid(n, nb, ngroups, a.tiles_per_wave, blocks);
    if (rows_written) *rows_written = blocks;
    LinrWgradDst d = {big, block_stride, w_off0, b_off0, 4};
    if (fb_split()) conv_bwd_wgrad_k<1, 0><<<dim3(blocks, ngroups), FS_THREADS, 0, s>>>(a, lo, mask, ld, n, PwArgs(), d, FbDst2{w_off1, b_off1}, g0);
    else conv_bwd_wgrad_single_k<1, 0><<<dim3(blocks, ngroups), FB_WAVES * 64, 0, s>>>(a, lo, mask, ld, n, PwArgs(), d, FbDst2{w_off1, b_off1}, g0);
    return linr_launch_rc();
}

// conv0_0 (8->4) of an Inception layer: gA = (bwd(gH[:, 0:4]; W00) + gI (+ old gA: LINR_ACCUM) + gH[:, 4:8] @ W10^T) (* (A > 0):
// LINR_RELU_MASK) and the kernel / bias gradient of conv0_0 from one gather of gH[:, 0:4]; the kernel / bias gradient of the 1x1
// conv1_0 (A^T gH[:, 4:8]) comes out of the same launch (slab offsets w10_off / b10_off)
int linr_conv84_bwd_wgrad_launch(const float* gH, const float* A, const float* gI, const int32_t* lo, const uint32_t* mask,
                                 int64_t ld, int64_t n, const float* w00, const float* w10, float* gA, unsigned flags,
                                 LinrWgradDst d, int64_t w10_off, int64_t b10_off, int nb, hipStream_t s, const Grp* gp, int ngroups,
                                 int* rows_written) {
    if (rows_written) *rows_written = 0;
    if (n == 0) return 0;
    const Grp g0 = gp ? *gp : Grp();
    FbArgs a = {gH, nullptr, A, w00, nullptr, gI, gA, flags & (LINR_RELU_MASK | LINR_ACCUM), 1, rows_written ? 0 : nb};
    int blocks = 1;
    fb_grid(n, nb, ngroups, a.tiles_per_wave, blocks);
    if (rows_written) *rows_written = blocks;
    PwArgs pw = {w10, nullptr, nullptr, nullptr};
    if (fb_split()) conv_bwd_wgrad_k<2, 0><<<dim3(blocks, ngroups), FS_THREADS, 0, s>>>(a, lo, mask, ld, n, pw, d, FbDst2{w10_off, b10_off}, g0);
    else conv_bwd_wgrad_single_k<2, 0><<<dim3(blocks, ngroups), FB_WAVES * 64, 0, s>>>(a, lo, mask, ld, n, pw, d, FbDst2{w10_off, b10_off}, g0);
    return linr_launch_rc();
}

extern "C" int linr_spconv_bwd_fused(const float* gout, const float* in, const int32_t* lo, const uint32_t* mask, int64_t ld,
                                     int64_t n, const float* W, float* gin, float* slab, int32_t nblocks, void* stream) {
    if (n < 0 || ld < n || nblocks < 1) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!gout || !in || !lo || !mask || !W || !gin || !slab) return LINR_EINVAL;
    if (!linr_aligned16(gout) || !linr_aligned16(gin)) return LINR_EALIGN;
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull || ld >= ((int64_t)1 << 26)) return LINR_EINVAL;
    LinrWgradDst d = {slab, 1736, 0, 1728, 8};
    return linr_conv88_bwd_wgrad_launch(gout, in, lo, mask, ld, n, W, gin, nullptr, d, nblocks, (hipStream_t)stream, nullptr, 1, nullptr);
}

// The Inception layer's backward with its two conv pairs fused (what the executor launches): gM by the caller (tail conv
// epilogue or linr_linear), then [gH + dW01, db01, dW11, db11], then [gX + dW00, db00, dW10, db10].  slab: [nblocks][1776] =
// [W00 864 | b00 4 | W01 432 | b01 4 | W11 432 | b11 4 | W10 32 | b10 4]
extern "C" int linr_inception_bwd_fused(const float* gI, const float* gM, const float* x, const float* H, const int32_t* lo,
                                        const uint32_t* mask, int64_t ld, int64_t n, const linr_inception_params* q, float* gH,
                                        float* gX, uint32_t flags, float* slab, int32_t nblocks, void* stream) {
    if (n < 0 || ld < n || nblocks < 1) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!gI || !gM || !x || !H || !lo || !mask || !q || !q->w00 || !q->w01 || !q->w10 || !q->w11 || !gH || !gX || !slab) return LINR_EINVAL;
    if (flags & ~(LINR_RELU_MASK | LINR_ACCUM)) return LINR_EINVAL;
    if (!linr_aligned16(gI) || !linr_aligned16(gM) || !linr_aligned16(gH) || !linr_aligned16(gX) || !linr_aligned16(H) ||
        !linr_aligned16(x)) return LINR_EALIGN;
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull || ld >= ((int64_t)1 << 26)) return LINR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int rc = linr_dual44_bwd_wgrad_launch(gI, gM, H, lo, mask, ld, n, q->w01, q->w11, gH, slab, 1776, 868, 1300, 1304, 1736, nblocks, s,
                                          nullptr, 1, nullptr);
    if (rc) return rc;
    LinrWgradDst d = {slab, 1776, 0, 864, 8};
    return linr_conv84_bwd_wgrad_launch(gH, x, gI, lo, mask, ld, n, q->w00, q->w10, gX, flags, d, 1740, 1772, nblocks, s, nullptr, 1, nullptr);
}

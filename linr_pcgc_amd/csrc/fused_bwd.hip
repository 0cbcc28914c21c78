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
// kernel (common.h: LinrWgradDst).
#include "common.h"
#include "conv_common.h"
#include <stdlib.h>

#define FB_WAVES 4
#ifndef FB_LAB
#define FB_LAB 16                        // kernel-floor experiments (tools/fused_lab.sh): 1 no gathers, 2 no weight-gradient MFMAs,
#endif                                   // 4 no backward-data MFMAs, 8 no LDS traffic, 16 no per-pair scheduling barrier
#define FB_HP 1040                       // bytes per (tap, channel quad) plane: 64 rows x 16 B + 16 B  (65 x 16 B = 1 mod 16)
#define FB_TP (2 * FB_HP)                // bytes per tap slot (130 x 16 B = 2 mod 16)
#define FB_BUF (8 * FB_TP)               // one chunk: 8 taps
#define FB_WAVE_BYTES (2 * FB_BUF)       // double-buffered

struct FbArgs {
    const float* g;        // [n][8] output gradient, zero row at index -1 (gathered)
    const float* xin;      // [n][8] the convolution's input (own rows)
    const float* W;        // [27][8][8] kernel (ME layout [k][cin][cout])
    float* out;            // [n][8] input gradient
    int tiles_per_wave;
    int nb_slab;           // slab rows the reduction will read: rows >= gridDim.x get zeros for this kernel's parameters
};

// EPI 0: plain; EPI 3: also gM = (gin[4:8] @ W12^T) * (M > 0)  (PwArgs as in cconv_mfma_k)
template <int EPI>
__global__ __launch_bounds__(FB_WAVES * 64, 1) void conv88_bwd_wgrad_k(FbArgs a, const int32_t* __restrict__ lo,
                                                                      const uint32_t* __restrict__ mask, int64_t ld, int64_t n,
                                                                      PwArgs pw, LinrWgradDst d, Grp gp) {
    __shared__ float4 smem[FB_WAVES * FB_WAVE_BYTES / 16];
    __shared__ float sbias[FB_WAVES][8];
    {   // group offsets: in = g, res = xin, w = W, out; e0..e2 = pointwise epilogue; e3 / e4 = slab offsets of kernel / bias
        const int gi = blockIdx.y;
        a.g += gp.in[gi]; a.xin += gp.res[gi]; a.W += gp.w[gi]; a.out += gp.out[gi];
        if constexpr (EPI == 3) { pw.w += gp.e0[gi]; pw.aux += gp.e1[gi]; pw.aux_out += gp.e2[gi]; }
        d.w_off += gp.e3[gi]; d.b_off += gp.e4[gi];
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // backward-data weights, register-resident (cconv_mfma_k<8, 8, BWD>): block b of wv[g][i] holds W(k = 8 g + b / 2,
    // gathered channel i, produced channels 4 (b % 2) .. + 3)
    float wv[4][8];
    {
        const int blk = lane >> 2, j = lane & 3;
        const int kl = blk >> 1, co = 4 * (blk & 1) + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k = g * 8 + kl;
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[g][i] = (k < 27) ? a.W[(k * 8 + co) * 8 + i] : 0.0f;
        }
    }
    // weight-gradient roles of the lane: quarter Q, (slot of the chunk, quad) = kap; A-operand holder: block ablk, component ai
    const int Q = lane >> 4, kap = lane & 15, wq = kap & 1, wslot = kap >> 1;
    const int ablk = (lane >> 2) & 3, ai = lane & 3;
    char* img = reinterpret_cast<char*>(smem) + wave * FB_WAVE_BYTES;
    char* imgW = img + lane * 16;
    const char* imgR = img + wslot * FB_TP + wq * FB_HP + (16 * Q) * 16;
    // the last chunk holds 3 taps only (27 = 3 x 8 + 3): six (slot, quad) pairs.  It runs with CBSZ = 1 - EIGHT row sets of 8 lanes,
    // A broadcast inside each pair of blocks - so its 16 rows of a quarter become 8 rows of an eighth: 64 MFMAs instead of 128.
    const int E8 = lane >> 3, wq3 = lane & 1, wslot3 = (lane & 7) >> 1, ablk3 = (lane >> 2) & 1;
    const char* imgR3 = img + FB_BUF + wslot3 * FB_TP + wq3 * FB_HP + (8 * E8) * 16;
    f32x4 wacc[4][4][2];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) wacc[ch][c][h] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.0f;
    float xa[8], xb[8], xbp[8];                  // own-row A operands: quarters (chunks 0-2), eighths (chunk 3; xbp: of the previous tile)
#pragma unroll
    for (int j = 0; j < 8; ++j) { xa[j] = 0.0f; xb[j] = 0.0f; xbp[j] = 0.0f; }
    // the first tile multiplies "the last chunk of the previous tile" with xap = 0: that buffer must hold finite numbers
    for (int o = lane * 16; o < FB_BUF; o += 64 * 16) *reinterpret_cast<float4*>(img + FB_BUF + o) = make_float4(0.f, 0.f, 0.f, 0.f);

    const int64_t T64 = (n + 63) >> 6;
    const int64_t tb0 = (int64_t)blockIdx.x * (FB_WAVES * a.tiles_per_wave);
    const int64_t tb1 = (tb0 + FB_WAVES * a.tiles_per_wave < T64) ? tb0 + FB_WAVES * a.tiles_per_wave : T64;
    const char* pad = reinterpret_cast<const char*>(a.g - 8);
    // With one wave per SIMD nothing hides a latency but the wave's own instruction stream, so the row loop is software-pipelined
    // ACROSS tiles: the gathers run PF taps ahead (a ring of RING = PF + 1 rows; 27 % RING == 0 keeps the ring slots compile-time
    // constants from tile to tile), the last PF taps of a tile already gather the first taps of the wave's next tile, whose index
    // words were loaded at step 1 and decoded at step 10 and whose own-row inputs were loaded at step 3; the transposed image
    // reads of a step are issued one step ahead.
    constexpr int PF = 8, RING = 9;
    static_assert(27 % RING == 0 && PF + 1 == RING, "ring slots must not depend on the tile");
    // Weight-gradient MFMA number m of the rows [r0, r0 + NR) of chunk pc (whose image the caller has read into b[]): row
    // r0 + m / 8, input half h = (m % 8) / 4, gathered component c = m % 4.  B = the lane's transposed read of its (slot, quad),
    // A = own-row inputs XA, broadcast inside the quarter from block (row % 4).
    auto wg_mfma = [&](auto pcc, auto r0c, auto mc, const float4* b, const float (&XA)[8]) {
        constexpr int pc = decltype(pcc)::value, r0 = decltype(r0c)::value, m = decltype(mc)::value;
        constexpr int j = m / 8, h = (m % 8) / 4, c = m % 4, r = r0 + j;
        const float B = c == 0 ? b[j].x : c == 1 ? b[j].y : c == 2 ? b[j].z : b[j].w;
        if constexpr (pc == 3)      // eighths: XA = xb layout, block (row % 2) of the lane's pair of blocks
            wacc[pc][c][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(XA[2 * (r >> 1) + h], B, wacc[pc][c][h], 1, r & 1, 0);
        else
            wacc[pc][c][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(XA[2 * (r >> 2) + h], B, wacc[pc][c][h], 2, r & 3, 0);
    };
    // index words of a row (9 column bases + the 27-bit mask of the compressed map) and their decode into the 27 byte offsets of
    // the mirrored taps (decode_offsets<true> split in two so that the loads' latency lies behind a few taps of MFMAs)
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
    auto xa_load = [&](int64_t row0, float (&XA)[8]) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int64_t r = row0 + 16 * Q + 4 * rq + ablk;
#pragma unroll
            for (int h = 0; h < 2; ++h) XA[2 * rq + h] = (r < n) ? a.xin[r * 8 + 4 * h + ai] : 0.0f;
        }
    };
    float w12[16];                               // EPI 3: the 1x1 kernel of the epilogue, read once
#pragma unroll
    for (int j = 0; j < 16; ++j) w12[j] = 0.0f;
    if constexpr (EPI == 3) {
#pragma unroll
        for (int j = 0; j < 16; ++j) w12[j] = pw.w[j];
    }
    auto xb_load = [&](int64_t row0, float (&XB)[8]) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int64_t r = row0 + 8 * E8 + 2 * rq + ablk3;
#pragma unroll
            for (int h = 0; h < 2; ++h) XB[2 * rq + h] = (r < n) ? a.xin[r * 8 + 4 * h + ai] : 0.0f;
        }
    };
    int64_t tile = tb0 + wave;
    if (tile < tb1) {                            // wave-uniform
        uint32_t off[27], offn[27];
        int32_t raw[10];
        float xan[8], xbn[8];
        f32x4 x[RING][2];                        // the ring of gathered rows: two 16-byte register tuples per row
        float4 bq[2][6];
        {
            const int64_t r = (tile << 6) + lane;
            idx_load(r < n ? r : n - 1, raw);
            idx_decode(raw, off);
            xa_load(tile << 6, xa);
            xb_load(tile << 6, xb);
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                x[u][0] = *reinterpret_cast<const f32x4*>(pad + off[LINR_TAP(u)]);
                x[u][1] = *reinterpret_cast<const f32x4*>(pad + off[LINR_TAP(u)] + 16);
            }
            bq[0][0] = *reinterpret_cast<const float4*>(imgR3);             // step 0: row 0 of "the last chunk of the previous tile" (zeros)
        }
        for (; tile < tb1; tile += FB_WAVES) {
            const int64_t row_raw = (tile << 6) + lane;
            const bool live = row_raw < n;
            const int64_t row = live ? row_raw : n - 1;            // every lane stays in the MFMAs (they ignore EXEC)
            const int64_t ntile = (tile + FB_WAVES < tb1) ? tile + FB_WAVES : tile;      // the last tile "prefetches" itself
            const int64_t nrow_raw = (ntile << 6) + lane;
            const int64_t nrow = nrow_raw < n ? nrow_raw : n - 1;
            f32x4 acc[2] = {(f32x4){0.0f, 0.0f, 0.0f, 0.0f}, (f32x4){0.0f, 0.0f, 0.0f, 0.0f}};
            float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);        // EPI 3: the row's M, requested early (step 5)
            __builtin_amdgcn_sched_barrier(0);
            static_for<27>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                constexpr int k = LINR_TAP(kk);
                constexpr int g = k / 8, ab = (k % 8) * 2;
                constexpr int ch = kk / 8, slot = kk % 8;
                constexpr int pc = (ch + 3) % 4;                   // the chunk whose rows are multiplied beside this tap
                constexpr int nst = ch < 3 ? 8 : 3;                // taps of this chunk: the 16 rows of chunk pc spread over them
                constexpr int prows = pc == 3 ? 8 : 16;            // rows per lane of chunk pc (eighths / quarters)
                constexpr int r0 = (slot * prows) / nst, nr = ((slot + 1) * prows) / nst - r0;
                // the same for the next step (step 0 of the next tile behind step 26)
                constexpr int kn = (kk + 1) % 27, chn = kn / 8, slotn = kn % 8, pcn = (chn + 3) % 4, nstn = chn < 3 ? 8 : 3;
                constexpr int prowsn = pcn == 3 ? 8 : 16;
                constexpr int r0n = (slotn * prowsn) / nstn, nrn = ((slotn + 1) * prowsn) / nstn - r0n;
                if constexpr (!(FB_LAB & 1)) {
                {
                    const uint32_t o = kk + PF < 27 ? off[LINR_TAP((kk + PF) % 27)] : offn[LINR_TAP((kk + PF) % 27)];
                    x[(kk + PF) % RING][0] = *reinterpret_cast<const f32x4*>(pad + o);
                    x[(kk + PF) % RING][1] = *reinterpret_cast<const f32x4*>(pad + o + 16);
                }
                }
                if constexpr (kk == 1) idx_load(nrow, raw);
                if constexpr (kk == 3) xa_load(ntile << 6, xan);
                if constexpr (kk == 4) xb_load(ntile << 6, xbn);
                if constexpr (kk == 10) idx_decode(raw, offn);
                if constexpr (EPI == 3 && kk == 5) m4 = *reinterpret_cast<const float4*>(pw.aux + row * 4);
                __builtin_amdgcn_sched_barrier(0);
                f32x4(&xk)[2] = x[kk % RING];
                // Pin: the MFMAs below consume x[kk] only from here on.  Without it instruction selection slides the whole
                // backward-data MFMA chain PF taps up, right behind each load (sched_barrier orders the machine scheduler,
                // not the DAG), and every gather is waited for the moment it is issued.
                asm volatile("" : "+v"(xk[0]), "+v"(xk[1]));
                if constexpr (!(FB_LAB & 8)) {
                *reinterpret_cast<f32x4*>(imgW + (ch & 1) * FB_BUF + slot * FB_TP) = xk[0];
                *reinterpret_cast<f32x4*>(imgW + (ch & 1) * FB_BUF + slot * FB_TP + FB_HP) = xk[1];
                }
                // next step's transposed reads: behind this tap's write (a new chunk reads the buffer just completed) - except at
                // step 26, whose successor (step 0) shares the register buffer
                auto read_next = [&]() {
                    if constexpr (FB_LAB & 8) return;
#pragma unroll
                    for (int j = 0; j < nrn; ++j)
                        bq[kn & 1][j] = pcn == 3 ? *reinterpret_cast<const float4*>(imgR3 + (r0n + j) * 16)
                                                 : *reinterpret_cast<const float4*>(imgR + (pcn & 1) * FB_BUF + (r0n + j) * 16);
                };
                if constexpr (kk != 26) read_next();
                // The backward-data MFMAs are two dependent chains (acc[0], acc[1]); issued back to back the second link of a
                // chain stalls on the first.  The weight-gradient MFMAs of chunk pc (eight independent accumulators) go between
                // the pairs, nr per pair, and a scheduling barrier per group keeps the machine scheduler from sorting them apart.
                static_for<8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (!(FB_LAB & 4)) {
                    acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[i / 4][i % 4], acc[0], 4, ab, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[i / 4][i % 4], acc[1], 4, ab + 1, 0);
                    }
                    if constexpr (!(FB_LAB & 2))
                    static_for<nr>([&](auto tc) {
                        constexpr int m = i * nr + decltype(tc)::value;
                        if constexpr (ch == 0) wg_mfma(std::integral_constant<int, pc>{}, std::integral_constant<int, r0>{},
                                                       std::integral_constant<int, m>{}, bq[kk & 1], xbp);
                        else wg_mfma(std::integral_constant<int, pc>{}, std::integral_constant<int, r0>{},
                                     std::integral_constant<int, m>{}, bq[kk & 1], xa);
                    });
                    if constexpr (!(FB_LAB & 16)) __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (kk == 26) read_next();
                if constexpr (k == 13) {                           // the centre tap is the row's own gradient: bias gradient
                    if (live) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) bsum[j] += xk[j / 4][j % 4];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int j = 0; j < 27; ++j) off[j] = offn[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) { xa[j] = xan[j]; xbp[j] = xb[j]; xb[j] = xbn[j]; }
            if (live) {
                float o[8];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[4 * h + j] = acc[h][j];
                if constexpr (EPI == 3) {          // gM = (gin[4:8] @ W12^T) * (M > 0)   (W12 [4][4])
                    const float mv[4] = {m4.x, m4.y, m4.z, m4.w};
                    float gm[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t = 0.0f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) t = fmaf(o[4 + q], w12[i * 4 + q], t);
                        gm[i] = mv[i] > 0.0f ? t : 0.0f;
                    }
                    *reinterpret_cast<float4*>(pw.aux_out + row * 4) = make_float4(gm[0], gm[1], gm[2], gm[3]);
                }
                float* op = a.out + row * 8;
                *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
        {   // the last chunk of the wave's last tile
            float4 b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = *reinterpret_cast<const float4*>(imgR3 + j * 16);
            static_for<64>([&](auto mc) {
                wg_mfma(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}, mc, b, xbp);
            });
        }
    }
    __syncthreads();
    // ---- fold: 16 (wave, quarter) partials per element in fixed order, one slab row per block --------------------------------
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
    static_for<4>([&](auto chc) {
        constexpr int ch = decltype(chc)::value;
        constexpr int ntaps = ch < 3 ? 8 : 3;
        float* mine = sacc + (wave * 64 + lane) * 33;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) mine[(c * 2 + h) * 4 + i] = wacc[ch][c][h][i];
        __syncthreads();
        for (int e = tid; e < ntaps * 64; e += FB_WAVES * 64) {
            const int slot = e >> 6, ci = (e >> 3) & 7, co = e & 7;
            const int kp = 2 * slot + (co >> 2), idx = ((co & 3) * 2 + (ci >> 2)) * 4 + (ci & 3);
            constexpr int sets = ch == 3 ? 8 : 4, width = 64 / sets;         // row sets per wave (eighths / quarters) and their lanes
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < FB_WAVES; ++w)
#pragma unroll
                for (int qq = 0; qq < sets; ++qq) t += sacc[(w * 64 + width * qq + kp) * 33 + idx];
            const int kk = 8 * ch + slot;
            const int k = kk / 9 + 3 * ((kk / 3) % 3) + 9 * (kk % 3);
            dst[d.w_off + k * 64 + ci * 8 + co] = t;
        }
        __syncthreads();
    });
    if (tid < 8) dst[d.b_off + tid] = ((sbias[0][tid] + sbias[1][tid]) + sbias[2][tid]) + sbias[3][tid];
    for (int64_t r = (int64_t)blockIdx.x + gridDim.x; r < a.nb_slab; r += gridDim.x) {
        float* z = d.base + r * d.block_stride;
        for (int e = tid; e < 1728; e += FB_WAVES * 64) z[d.w_off + e] = 0.0f;
        if (tid < 8) z[d.b_off + tid] = 0.0f;
    }
}

// Grid of the fused kernels.  One block per CU is resident (registers, LDS) and a block's prologue (weights, index decode,
// first gathers) and epilogue (16-way fold, slab row) run with idle matrix cores, so the launch is sized as ONE round of
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

// g: output gradient (gathered), xin: the convolution's input, W: its kernel; out: input gradient; slab partials into d
// (rows 0 .. nb - 1 of the slab are all written: the blocks beyond the grid's get zeros).  pw != nullptr selects the gM epilogue.
int linr_conv88_bwd_wgrad_launch(const float* g, const float* xin, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                                 const float* W, float* out, const PwArgs* pw, LinrWgradDst d, int nb, hipStream_t s,
                                 const Grp* gp, int ngroups) {
    if (n == 0) return 0;
    const Grp g0 = gp ? *gp : Grp();
    FbArgs a = {g, xin, W, out, 1, nb};
    int blocks = 1;
    fb_grid(n, nb, ngroups, a.tiles_per_wave, blocks);
    const dim3 grid(blocks, ngroups);
    if (pw) conv88_bwd_wgrad_k<3><<<grid, FB_WAVES * 64, 0, s>>>(a, lo, mask, ld, n, *pw, d, g0);
    else conv88_bwd_wgrad_k<0><<<grid, FB_WAVES * 64, 0, s>>>(a, lo, mask, ld, n, PwArgs(), d, g0);
    return linr_launch_rc();
}

extern "C" int linr_spconv_bwd_fused(const float* gout, const float* in, const int32_t* lo, const uint32_t* mask, int64_t ld,
                                     int64_t n, const float* W, float* gin, float* slab, int32_t nblocks, void* stream) {
    if (n < 0 || ld < n || nblocks < 1) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!gout || !in || !lo || !mask || !W || !gin || !slab) return LINR_EINVAL;
    if (!linr_aligned16(gout) || !linr_aligned16(gin)) return LINR_EALIGN;
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull) return LINR_EINVAL;
    LinrWgradDst d = {slab, 1736, 0, 1728, 8};
    return linr_conv88_bwd_wgrad_launch(gout, in, lo, mask, ld, n, W, gin, nullptr, d, nblocks, (hipStream_t)stream, nullptr, 1);
}

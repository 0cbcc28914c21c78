// bf16 training executor (BASELINE config[4]: "bf16 SparseConv"): the overfit step of main.py:305-321 with bf16 feature and
// gradient rows, fp32 master weights and fp32 accumulation - beside the fp32 executor of csrc/net.hip, never instead of it.
//
// Replaces, like csrc/net.hip: LINR_PCGC_Model.logic_core / forward (models/model_core.py:38-81), CNP.forward
// (models/upsample.py:163-217), make_block (:88-97), InceptionResNet.forward (models/resnet.py:12-60) and the autograd graph that
// main.py:315-316 differentiates; the optimiser step (main.py:231-237,319) runs on the fp32 master parameters.
//
// Numerics - ONE rule: every matrix that is written to memory is rounded to bf16 (RNE) and every consumer, also one fused into the
// producing kernel, sees the STORED value.  Stored forward: x_low, A, H, M, I, O / x_glob, the prune convolutions' outputs C;
// stored backward: gC, gO, the fan-in sum of x_glob's gradient, gI, gM, gH, gA, gx_low.  The 3x3x3 kernels are rounded to bf16
// inside the kernels from the fp32 master; biases, the 1x1 convolutions, the scale-context MLP, the head MLP, sigmoid and the
// bits are fp32 arithmetic on fp32 parameters; every product is accumulated in fp32; weight gradients are fp32 sums of products
// of stored (bf16) rows; probabilities and bits are fp32 / fp64.  oracle/network_bf16.py (train_forward_backward) emulates
// exactly these roundings with autograd.
//
// Layout: feature / gradient matrices are bf16 [1 + rows][8] (16-byte rows, an all-zero row in front: absent neighbours gather
// it) - ONE dwordx4 gather per tap where the fp32 executor needs two.
//
// Kernels:
//   forward    bconv_k<MODE, 1, true> (csrc/bf16_common.h): lane = output row, v_mfma_f32_4x4x4_16b_bf16 with the weight block
//              broadcast (CBSZ = 4): 64 rows x 4 cout x 4 cin per instruction, a quarter of the fp32 path's matrix instructions
//   backward   bbwd_k<KIND>: backward-data AND weight gradient of a convolution from ONE gather of the output gradient (the
//              re-indexing of csrc/fused_bwd.hip: gW[k] = sum_i in[i]^T g[nbr(i, 26 - k)]).  Backward-data is the forward kernel
//              at the mirrored taps.  For the weight gradient the ROWS are the K dimension of the matrix instruction: the gathered
//              rows are parked in a wave-private LDS image [tap slot][row][16 B] as they arrive (one ds_write_b128 per tap) and read
//              back TRANSPOSED by ds_read_b64_tr_b16 - a lane receives one channel of four consecutive rows - as are the rows' own
//              inputs.  Convolutions 8->8 (KIND 0, and KIND 2 with the second convolution 8->4 of the same input) use
//              v_mfma_f32_16x16x32_bf16: M = the 8 own channels (padded to 16), N = two tap slots x 8 columns, K = 32 rows - 28
//              weight-gradient instructions per 64-row tile beside the 108 (4x4x4) of backward-data; the 4->4 pair of KIND 1 stays on
//              v_mfma_f32_4x4x4_16b_bf16 (16 independent 4 x 4 blocks, four rows per instruction).  ~250 registers and 10 KB of LDS
//              per wave, two waves per SIMD, 14 gathers in flight per wave and the next tile's first gathers issued before the
//              epilogue of the current one.
//   first conv bocc7m_k / bocc_wgrad7_k: the seven 1->8 .. 7->8 convolutions that read the occupancy codes share one gather; the forward is
//              ONE matrix product on v_mfma_f32_16x16x32_bf16 (columns = (group, cout) as M, rows as N, the gathered 16-byte rows ARE the B
//              operands; 22 us against 31 for the 4x4x4 form bocc7_k, kept behind LINR_BOCC7_MFMA16=0).
//   per step   tpack_k rounds every 3x3x3 kernel once into the operand images the kernels above load (wimg).
#include "bf16_common.h"
#include "head_bwd.h"
#include "sce.h"
#include "net_shared.h"
#include <stdlib.h>
#include <vector>

#define TRY(e) do { int rc_ = (e); if (rc_) return rc_; } while (0)

// profiling / poison classes of this executor (include/linr_hip.h: linr_prof_*)
enum { TK_BWD88 = 17, TK_BWD_DUAL = 18, TK_BWD_C00 = 19, TK_FWD = 20, TK_HEAD_BWD = 21, TK_FIRST_WGRAD = 22, TK_MISC = 23 };

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// v_mfma_f32_16x16x32_bf16: D[m][n] += sum_k A[m][k] B[k][n]; lane l holds A[l & 15][8 (l >> 4) + j], B[8 (l >> 4) + j][l & 15] in element j
// of its two 64-bit halves (lo = j 0..3, hi = j 4..7) and D[4 (l >> 4) + reg][l & 15]
__device__ __forceinline__ f32x4 mfma16(s16x4 alo, s16x4 ahi, s16x4 blo, s16x4 bhi, f32x4 c) {
    const s16x8 a = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7), b = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// transposed LDS read (ds_read_b64_tr_b16): per 16-lane group, lane 4 q + p supplies the address of an 8-byte piece (q, p); lane
// 4 p + j of the group receives element j of the pieces (0, p) .. (3, p) - four rows of one 16-bit column
__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}

// ---- fused backward: backward-data + weight gradient from one gather ----------------------------------------------------------------
#define BB_WAVES 4
#define BB_SLOT 1088                    // bytes of one image slot: 64 rows x 16 B + 64 (consecutive slots start 16 banks apart)
#define TB_OUT_F32 2u                   // input gradient stored as fp32 [n][8] (rounded to bf16 first): gx_low for the scale context
#define TB_MAXG 8
//   KIND 0  conv 8->8 (prune convolutions, tail convolutions, first convolutions): g [n][8] gathered whole, own rows = its input
//   KIND 1  the two 4->4 convolutions of an Inception layer: gathers G2 = [gI[:, 0:4] | gM], own rows H (also the ReLU mask)
//   KIND 2  conv0_0 8->4: gathers gH[:, 0:4] (8 bytes per tap), own rows A (also the mask); conv1_0 (1x1) rides along
//   EPI 3 (KIND 0, tail convolution): also gM = (gI[:, 4:8] @ W12^T) * (M > 0), G2 = [gI[:, 0:4] | gM]; conv1_2's gradients ride along
struct BbArgs {
    const bf16_t* g;  const bf16_t* xin;  const float* P;
    bf16_t* out;  float* out_f32;
    const bf16_t* m;  bf16_t* g2;  const bf16_t* res;
    const int32_t* lo;  const uint32_t* mask;  int64_t ld, n;
    int tiles_per_wave;
    float* big;  int64_t block_stride;
    unsigned flags;
    int64_t g_g[TB_MAXG], g_x[TB_MAXG], g_out[TB_MAXG], g_m[TB_MAXG], g_g2[TB_MAXG], g_res[TB_MAXG];     // element offsets per group
    int64_t w[TB_MAXG], b[TB_MAXG];            // parameter (= slab) offsets: the kernel / bias (KIND 1: conv0_1)
    int64_t w1[TB_MAXG], b1[TB_MAXG];          // KIND 1: conv1_1
    int64_t wp[TB_MAXG], bp[TB_MAXG];          // EPI 3: conv1_2;  KIND 2: conv1_0
    int cin[TB_MAXG];                          // KIND 0: input channels of the kernel [27][cin][8]
};

template <int KIND> struct BbT {
    static constexpr int CT = KIND == 0 ? 4 : 8;                // taps per chunk = taps per weight-gradient instruction
    static constexpr int SL = KIND == 1 ? 8 : 4;                // image slots per chunk
    static constexpr int NCH = (27 + CT - 1) / CT;              // chunks per tile: 7 / 4 / 4
    static constexpr int WAVE_BYTES = BB_SLOT * (1 + 2 * SL);   // own rows + two chunk buffers
    static constexpr int NW = KIND == 0 ? 108 : 54;             // backward-data weight blocks
    // weight gradient: KIND 0 / 2 on v_mfma_f32_16x16x32_bf16 (M = the 8 own channels, padded to 16; N = two image slots x 8 columns; K = 32
    // rows): two accumulators per chunk; KIND 1 (two 4 x 4 products per tap, a quarter of such a tile) on v_mfma_f32_4x4x4_16b_bf16: one
    static constexpr bool BIG = KIND != 1;
    static constexpr int NACC = BIG ? 2 * NCH : NCH;
};

template <int KIND, int EPI>
__global__ __launch_bounds__(BB_WAVES * 64, 2) void bbwd_k(BbArgs a) {
    using T = BbT<KIND>;
    constexpr int CT = T::CT, SL = T::SL, NCH = T::NCH, NG = (T::NW + 15) / 16, NACC = T::NACC;
    constexpr bool BIG = T::BIG;
    __shared__ uint4 smem[BB_WAVES * T::WAVE_BYTES / 16];
    __shared__ float sbias[BB_WAVES][8];
    __shared__ float s12[BB_WAVES][20];
    const int gi = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bf16_t* gbase = a.g + a.g_g[gi];
    const bf16_t* xin = a.xin + a.g_x[gi];
    const float* P = a.P;
    // ---- backward-data weight blocks: block (lane >> 2) of wv[r] is combo c = 16 r + block; lane i = lane & 3 holds A[i][0..3] ------
    //   KIND 0: c = 4 k + 2 h + q -> W[k][4 h + i][4 q + kk]     (produced = input channel 4 h + i, gathered = output channel 4 q + kk)
    //   KIND 1: c = 2 k + t       -> (t ? W11 : W01)[k][i][kk]
    //   KIND 2: c = 2 k + h       -> W00[k][4 h + i][kk]
    s16x4 wv[NG];
    {
        const int blk = lane >> 2, i = lane & 3;
        float raw[NG][4];
        bool ok[NG];
#pragma unroll
        for (int r = 0; r < NG; ++r) {
            const int c = 16 * r + blk;
            const int k0 = KIND == 0 ? c / 4 : c / 2, k = k0 < 27 ? k0 : 26;
            ok[r] = k0 < 27;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                int64_t idx;
                if constexpr (KIND == 0) idx = a.w[gi] + ((int64_t)k * 8 + 4 * ((c >> 1) & 1) + i) * 8 + 4 * (c & 1) + kk;
                else if constexpr (KIND == 1) idx = ((c & 1) ? a.w1[gi] : a.w[gi]) + ((int64_t)k * 4 + i) * 4 + kk;
                else idx = a.w[gi] + ((int64_t)k * 8 + 4 * (c & 1) + i) * 4 + kk;
                raw[r][kk] = P[idx];
            }
        }
#pragma unroll
        for (int r = 0; r < NG; ++r) {
            s16x4 v;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) v[kk] = ok[r] ? (short)f2bf(raw[r][kk]) : (short)0;
            wv[r] = v;
        }
    }
    // ---- the lane's roles in the transposed reads: group tg = lane >> 4, piece row tq = (lane >> 2) & 3, piece column tp = lane & 3 ---
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    char* img = reinterpret_cast<char*>(smem) + wave * T::WAVE_BYTES;
    char* ximgW = img + lane * 16;
    const char* ximgR = img + tq * 16 + 8 * (tp & 1);                       // own rows: channel quad tp & 1
    char* bufW = img + BB_SLOT + lane * 16;
    // gathered rows: KIND 0: slot tg (tap 4 c + tg), channel quad tp >> 1; KIND 1: slot 2 tg + (tp >> 1), quad tp & 1 (the
    // convolution); KIND 2: slot tg holds the tap PAIR 2 tg, 2 tg + 1 as the two halves of a 16-byte row: half tp >> 1
    const char* bufR = img + BB_SLOT + (KIND == 1 ? (2 * tg + (tp >> 1)) * BB_SLOT + 8 * (tp & 1) : tg * BB_SLOT + 8 * (tp >> 1)) + tq * 16;
    // 16x16x32 form (KIND 0 / 2): group tg of 16 lanes = rows 8 tg .. 8 tg + 7 of a 32-row K step (two transposed reads of four rows);
    // A columns = own channels (tp & 1: quads 0 / 1; the lanes m >= 8 of the tile are padding), B columns = slot pair: piece tp of
    // [slot 2 u: 8 columns | slot 2 u + 1: 8 columns]
    const char* ximgR16 = img + (8 * tg + tq) * 16 + 8 * (tp & 1);
    const char* bufR16 = img + BB_SLOT + (tp >> 1) * BB_SLOT + (8 * tg + tq) * 16 + 8 * (tp & 1);
    f32x4 wacc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c) wacc[c] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.0f;
    float g12[20];                                                            // EPI 3: conv1_2's kernel [4][4] and bias gradients
#pragma unroll
    for (int j = 0; j < 20; ++j) g12[j] = 0.0f;
    float wpw[32];                                                            // EPI 3: W12 [4][4];  KIND 2: W10 [8][4]
#pragma unroll
    for (int j = 0; j < 32; ++j) wpw[j] = 0.0f;
    if constexpr (EPI == 3) {
#pragma unroll
        for (int j = 0; j < 16; ++j) wpw[j] = P[a.wp[gi] + j];
    }
    if constexpr (KIND == 2) {
#pragma unroll
        for (int j = 0; j < 32; ++j) wpw[j] = P[a.wp[gi] + j];
    }
    const int64_t n = a.n;
    const int64_t T64 = (n + 63) >> 6;
    const int64_t tb0 = (int64_t)blockIdx.x * (BB_WAVES * a.tiles_per_wave);
    const int64_t tb1 = (tb0 + BB_WAVES * a.tiles_per_wave < T64) ? tb0 + BB_WAVES * a.tiles_per_wave : T64;
    const char* pad = reinterpret_cast<const char*>(gbase - 8);
    constexpr int PF = 14;                                                   // gathers in flight per wave: two waves per SIMD need them deep
    typedef typename std::conditional<KIND == 2, uint2, uint4>::type XR;      // (5 / 8 / 11 / 14 / 17 / 20: 0.892 / 0.890 / 0.867 / 0.852 / 0.852 / 0.860 ms per step)
    // The row loop is software-pipelined ACROSS tiles (two to three waves share a SIMD, but every wave still walks through the same
    // phases): the index words, the own row and the epilogue's own-row operands of the wave's NEXT tile are requested at step 1 of the
    // current one; the 16 transposed reads of a chunk are issued behind its last tap and its 16 weight-gradient instructions are
    // spread over the taps of the NEXT chunk, beside that chunk's gathers and backward-data instructions (independent accumulators).
    struct Own { uint4 xr, res, gh; uint2 m; };
    auto idx_load = [&](int64_t row, uint32_t (&raw)[10]) { load_words16(a.lo, a.mask, a.ld, row, raw); };
    auto idx_decode = [&](const uint32_t (&raw)[10], uint32_t (&off)[27]) {      // mirrored taps: off[k] pairs with W[k] (backward-data
        uint32_t fo[27];                                                         // gathers nbr(row, 26 - k)); absent -> 0 (the pad row)
        decode_words16(raw, fo);
#pragma unroll
        for (int k = 0; k < 27; ++k) off[k] = fo[26 - k];
    };
    auto own_load = [&](int64_t tile, Own& o) {
        const int64_t row_raw = (tile << 6) + lane;
        const bool live = row_raw < n;
        const int64_t row = live ? row_raw : n - 1;
        // own row (dead lanes: zeros, so that nothing of them reaches a weight gradient)
        const uint32_t rb16 = (uint32_t)row << 4;                              // rows < 2^27: byte offsets fit 32 bits (saddr-form accesses)
        o.xr = make_uint4(0u, 0u, 0u, 0u);
        if (live) o.xr = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(xin) + rb16);
        o.res = make_uint4(0u, 0u, 0u, 0u); o.gh = o.res; o.m = make_uint2(0u, 0u);
        if constexpr (EPI == 3) o.m = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.m + a.g_m[gi]) + (rb16 >> 1));
        if constexpr (KIND == 2) {
            o.res = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.res + a.g_res[gi]) + rb16);
            o.gh = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(gbase) + rb16);
        }
    };
    int64_t tile = tb0 + wave;
    if (tile < tb1) {                                                        // wave-uniform
        uint32_t off[27], raw[10];
        Own ow, own_n;
        XR x[PF + 1];                                                        // the ring of gathered rows; a tile's first PF gathers are issued
        {                                                                    // before the PREVIOUS tile's epilogue (here: before the loop)
            const int64_t r = (tile << 6) + lane;
            idx_load(r < n ? r : n - 1, raw);
            idx_decode(raw, off);
            own_load(tile, ow);
#pragma unroll
            for (int u = 0; u < PF; ++u) x[u] = *reinterpret_cast<const XR*>(pad + off[LINR_TAP(u)]);
        }
        for (; tile < tb1; tile += BB_WAVES) {
            const int64_t row_raw = (tile << 6) + lane;
            const bool live = row_raw < n;
            const int64_t row = live ? row_raw : n - 1;                      // every lane stays in the matrix instructions (they ignore EXEC)
            const int64_t ntile = tile + BB_WAVES < tb1 ? tile + BB_WAVES : tile;          // the last tile "prefetches" itself
            *reinterpret_cast<uint4*>(ximgW) = ow.xr;
            s16x4 av[16], bv[16];
            if constexpr (BIG) {                                              // av[2 ks + h]: rows 32 ks + 8 tg + 4 h .. + 3 of the own channels
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = tr_read(ximgR16 + (32 * (j >> 1) + 4 * (j & 1)) * 16);
            } else {
#pragma unroll
                for (int rq = 0; rq < 16; ++rq) av[rq] = tr_read(ximgR + rq * 64);
            }
            f32x4 acc[2] = {(f32x4){0.0f, 0.0f, 0.0f, 0.0f}, (f32x4){0.0f, 0.0f, 0.0f, 0.0f}};
            sfor<27>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;                      // step; k = LINR_TAP(kk) the tap it handles
                constexpr int k = LINR_TAP(kk);
                constexpr int ch = kk / CT, s = kk % CT;
                constexpr int nst = 27 - ch * CT < CT ? 27 - ch * CT : CT;   // taps of this chunk
                if constexpr (kk + PF < 27) x[(kk + PF) % (PF + 1)] = *reinterpret_cast<const XR*>(pad + off[LINR_TAP(kk + PF)]);
                if constexpr (kk == 1) {
                    const int64_t r = (ntile << 6) + lane;
                    idx_load(r < n ? r : n - 1, raw);
                    own_load(ntile, own_n);
                }
                const XR xk = x[kk % (PF + 1)];
                {
                    if constexpr (KIND == 0) {
                        const s16x4 q0 = __builtin_bit_cast(s16x4, make_uint2(xk.x, xk.y));
                        const s16x4 q1 = __builtin_bit_cast(s16x4, make_uint2(xk.z, xk.w));
                        constexpr int c0 = 4 * k;
                        acc[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[c0 / 16], q0, acc[0], 4, c0 % 16, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[(c0 + 2) / 16], q0, acc[1], 4, (c0 + 2) % 16, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[(c0 + 1) / 16], q1, acc[0], 4, (c0 + 1) % 16, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[(c0 + 3) / 16], q1, acc[1], 4, (c0 + 3) % 16, 0);
                    } else if constexpr (KIND == 1) {
                        const s16x4 q0 = __builtin_bit_cast(s16x4, make_uint2(xk.x, xk.y));
                        const s16x4 q1 = __builtin_bit_cast(s16x4, make_uint2(xk.z, xk.w));
                        constexpr int c0 = 2 * k;
                        acc[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[c0 / 16], q0, acc[0], 4, c0 % 16, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[(c0 + 1) / 16], q1, acc[1], 4, (c0 + 1) % 16, 0);
                    } else {
                        const s16x4 q0 = __builtin_bit_cast(s16x4, xk);
                        constexpr int c0 = 2 * k;
                        acc[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[c0 / 16], q0, acc[0], 4, c0 % 16, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[(c0 + 1) / 16], q0, acc[1], 4, (c0 + 1) % 16, 0);
                    }
                }
                if constexpr (KIND == 2 && kk == 24) {
                    // conv1_0 (1x1, 8 -> 4) rides along: its weight gradient sum_rows A[row]^T gH[row][4:8] is what the idle pseudo-tap 27
                    // of the last chunk computes when its half-slot holds the rows' OWN gH[:, 4:8] (written here, behind the reads of
                    // chunk 1, which shares the buffer)
                    uint2 own = make_uint2(0u, 0u);
                    if (live) own = make_uint2(ow.gh.z, ow.gh.w);
                    *reinterpret_cast<uint2*>(bufW + (((NCH - 1) & 1) * SL + 1) * BB_SLOT + 8) = own;
                }
                // park the gathered row in the chunk's image
                if constexpr (KIND == 2) *reinterpret_cast<uint2*>(bufW + ((ch & 1) * SL + (s >> 1)) * BB_SLOT + 8 * (s & 1)) = xk;
                else *reinterpret_cast<uint4*>(bufW + ((ch & 1) * SL + s) * BB_SLOT) = xk;
                if constexpr (k == 13) {                                      // the centre tap is the row's own gradient: bias gradient
                    if (live) {
                        if constexpr (KIND == 2) {
                            bsum[0] += bf2f((bf16_t)(xk.x & 0xffff)); bsum[1] += bf2f((bf16_t)(xk.x >> 16));
                            bsum[2] += bf2f((bf16_t)(xk.y & 0xffff)); bsum[3] += bf2f((bf16_t)(xk.y >> 16));
                        } else {
                            float t[8];
                            unpack_row(xk, t);
#pragma unroll
                            for (int j = 0; j < 8; ++j) bsum[j] += t[j];
                        }
                    }
                }
                if constexpr (ch >= 1) {                                      // this step's share of the previous chunk's weight gradient
                    if constexpr (BIG) {                                      // 4 instructions: (slot pair u, K step ks) = mi >> 1, mi & 1
                        constexpr int m0 = (s * 4) / nst, m1 = ((s + 1) * 4) / nst;
                        sfor<m1 - m0>([&](auto mc) {
                            constexpr int mi = m0 + decltype(mc)::value, u = mi >> 1, ks = mi & 1;
                            wacc[2 * (ch - 1) + u] = mfma16(av[2 * ks], av[2 * ks + 1], bv[4 * u + 2 * ks], bv[4 * u + 2 * ks + 1], wacc[2 * (ch - 1) + u]);
                        });
                    } else {
                        constexpr int r0 = (s * 16) / nst, r1 = ((s + 1) * 16) / nst;
                        sfor<r1 - r0>([&](auto rc) {
                            constexpr int rq = r0 + decltype(rc)::value;
                            wacc[ch - 1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av[rq], bv[rq], wacc[ch - 1], 0, 0, 0);
                        });
                    }
                }
                if constexpr (s == nst - 1) {                                 // the chunk is complete: its transposed reads (8 / 16)
                    if constexpr (BIG) {                                      // bv[4 u + 2 ks + h]
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            bv[j] = tr_read(bufR16 + ((ch & 1) * SL + 2 * (j >> 2)) * BB_SLOT + (32 * ((j >> 1) & 1) + 4 * (j & 1)) * 16);
                    } else {
#pragma unroll
                        for (int rq = 0; rq < 16; ++rq) bv[rq] = tr_read(bufR + (ch & 1) * SL * BB_SLOT + rq * 64);
                    }
                }
            });
            if constexpr (BIG) {                                              // the last chunk of the tile
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    wacc[2 * (NCH - 1) + (mi >> 1)] = mfma16(av[2 * (mi & 1)], av[2 * (mi & 1) + 1], bv[4 * (mi >> 1) + 2 * (mi & 1)],
                                                             bv[4 * (mi >> 1) + 2 * (mi & 1) + 1], wacc[2 * (NCH - 1) + (mi >> 1)]);
            } else {
#pragma unroll
                for (int rq = 0; rq < 16; ++rq)
                    wacc[NCH - 1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av[rq], bv[rq], wacc[NCH - 1], 0, 0, 0);
            }
            // the next tile's offsets (its index words came in at step 1) and its first gathers, in flight during this tile's epilogue
            idx_decode(raw, off);
            if (tile + BB_WAVES < tb1) {
#pragma unroll
                for (int u = 0; u < PF; ++u) x[u] = *reinterpret_cast<const XR*>(pad + off[LINR_TAP(u)]);
            }
            // ---- epilogue: the row's input gradient, rounded to bf16; everything derived from it uses the rounded value ---------------
            float o[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = acc[0][j]; o[4 + j] = acc[1][j]; }
            if constexpr (KIND == 1) {                // gH = [bwd(gI[:, 0:4]; W01) | bwd(gM; W11)] * (H > 0)
                float hv[8];
                unpack_row(ow.xr, hv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = hv[j] > 0.0f ? o[j] : 0.0f;
            }
            if constexpr (KIND == 2) {                // gA = (bwd(gH[:, 0:4]; W00) + gI + gH[:, 4:8] @ W10^T) * (A > 0)
                float rv[8], gh[8], av8[8];
                unpack_row(ow.res, rv);
                unpack_row(ow.gh, gh);
                unpack_row(ow.xr, av8);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += rv[j];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float t = 0.0f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) t = fmaf(gh[4 + q], wpw[i * 4 + q], t);
                    o[i] += t;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = av8[j] > 0.0f ? o[j] : 0.0f;
                if (live) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) bsum[4 + q] += gh[4 + q];
                }
            }
            const uint4 po = pack_row(o);                                     // the ONE rounding of the row (v_cvt_pk_bf16_f32)
            {
                if (live) {
                    const uint32_t rb16 = (uint32_t)row << 4;
                    if (EPI == 3 || (a.flags & TB_OUT_F32)) unpack_row(po, o);   // what follows uses the STORED values
                    if (a.flags & TB_OUT_F32) {
                        float* op = a.out_f32 + row * 8;
                        *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
                        *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
                    } else {
                        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.out + a.g_out[gi]) + rb16) = po;
                    }
                    if constexpr (EPI == 3) {             // gM = (gI[:, 4:8] @ W12^T) * (M > 0);  G2 = [gI[:, 0:4] | gM]
                        const float mv[4] = {bf2f((bf16_t)(ow.m.x & 0xffff)), bf2f((bf16_t)(ow.m.x >> 16)), bf2f((bf16_t)(ow.m.y & 0xffff)),
                                             bf2f((bf16_t)(ow.m.y >> 16))};
                        float g2[8];
#pragma unroll
                        for (int j = 0; j < 4; ++j) g2[j] = o[j];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float t = 0.0f;
#pragma unroll
                            for (int q = 0; q < 4; ++q) t = fmaf(o[4 + q], wpw[i * 4 + q], t);
                            g2[4 + i] = mv[i] > 0.0f ? t : 0.0f;
                        }
                        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.g2 + a.g_g2[gi]) + rb16) = pack_row(g2);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int q = 0; q < 4; ++q) g12[i * 4 + q] = fmaf(mv[i], o[4 + q], g12[i * 4 + q]);
#pragma unroll
                        for (int q = 0; q < 4; ++q) g12[16 + q] += o[4 + q];
                    }
                }
            }
            ow = own_n;
        }
    }
    __syncthreads();
    // ---- fold: the four waves' partial sums of every element in wave order, one slab row per block ------------------------------------
    // [wave][NACC * 4][LW lanes]: the 16x16x32 tiles keep their useful rows m < 8 in the lanes 0..31 (the other half is padding)
    constexpr int LW = BIG ? 32 : 64;
    float* sacc = reinterpret_cast<float*>(smem);
    if (lane < LW) {
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) sacc[((wave * NACC + c) * 4 + i) * LW + lane] = wacc[c][i];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = bsum[j];
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) t += __shfl_xor(t, dd, 64);
        if (lane == 0) sbias[wave][j] = t;
    }
    if constexpr (EPI == 3) {
#pragma unroll
        for (int j = 0; j < 20; ++j) {
            float t = g12[j];
#pragma unroll
            for (int dd = 32; dd > 0; dd >>= 1) t += __shfl_xor(t, dd, 64);
            if (lane == 0) s12[wave][j] = t;
        }
    }
    __syncthreads();
    float* dst = a.big + (int64_t)blockIdx.x * a.block_stride;
    const int tid = threadIdx.x;
    const int cinv = KIND == 0 ? a.cin[gi] : 8;
    constexpr int per_tap = KIND == 0 ? 64 : 32;                             // kernel elements per tap (KIND 1: both convolutions)
    constexpr int nsteps = KIND == 2 ? 28 : 27;                              // KIND 2: pseudo-step 27 = conv1_0
    for (int e = tid; e < nsteps * per_tap; e += BB_WAVES * 64) {
        const int kk = e / per_tap, r = e % per_tap;                         // step kk handled tap k (common.h: LINR_TAP)
        const int k = kk / 9 + 3 * ((kk / 3) % 3) + 9 * (kk % 3);
        int c = kk / CT, L, i;
        const int s = kk % CT;
        int64_t dofs;
        bool valid = true;
        if constexpr (KIND == 0) {          // r = ci * 8 + co: accumulator 2 c + s / 2, tile row ci, tile column 8 (s % 2) + co
            const int ci = r >> 3, co = r & 7;
            c = 2 * c + (s >> 1); L = 16 * (ci >> 2) + 8 * (s & 1) + co; i = ci & 3;
            valid = ci < cinv;
            dofs = a.w[gi] + ((int64_t)k * cinv + ci) * 8 + co;
        } else if constexpr (KIND == 1) {   // r = conv * 16 + ci * 4 + co: group s / 2, block conv | (s % 2) << 1, lane co, register ci
            const int cv = r >> 4, ci = (r >> 2) & 3, co = r & 3;
            L = 16 * (s >> 1) + 4 * (cv | ((s & 1) << 1)) + co; i = ci;
            dofs = (cv ? a.w1[gi] : a.w[gi]) + (int64_t)k * 16 + (r & 15);
        } else {                            // r = ci * 4 + co: slot s / 2 -> accumulator 2 c + slot / 2, column 8 (slot % 2) + 4 (s % 2) + co
            const int ci = r >> 2, co = r & 3, slot = s >> 1;
            c = 2 * c + (slot >> 1); L = 16 * (ci >> 2) + 8 * (slot & 1) + 4 * (s & 1) + co; i = ci & 3;
            dofs = kk < 27 ? a.w[gi] + (int64_t)k * 32 + r : a.wp[gi] + r;  // conv1_0.kernel [8][4]
        }
        if (!valid) continue;
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < BB_WAVES; ++w) t += sacc[((w * NACC + c) * 4 + i) * LW + L];
        dst[dofs] = t;
    }
    if (tid < 8) {
        const float t = ((sbias[0][tid] + sbias[1][tid]) + sbias[2][tid]) + sbias[3][tid];
        if constexpr (KIND == 0) dst[a.b[gi] + tid] = t;
        else if constexpr (KIND == 1) dst[(tid < 4 ? a.b[gi] : a.b1[gi]) + (tid & 3)] = t;
        else dst[(tid < 4 ? a.b[gi] : a.bp[gi]) + (tid & 3)] = t;
    }
    if constexpr (EPI == 3) {
        if (tid < 20) {
            const float t = ((s12[0][tid] + s12[1][tid]) + s12[2][tid]) + s12[3][tid];
            dst[tid < 16 ? a.wp[gi] + tid : a.bp[gi] + (tid - 16)] = t;
        }
    }
}

static int tb_cus() {
    static const int v = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    return v;
}
// one round of long-lived blocks: about 2 x CUs / groups blocks per group (two blocks are resident per CU), never more than nb
static void tb_grid(int64_t n, int nb, int ngroups, int& tiles_per_wave, int& blocks) {
    const int64_t t64 = (n + 63) >> 6;
    int64_t target = 2 * tb_cus() / (ngroups < 1 ? 1 : ngroups);
    if (target < 1) target = 1;
    if (target > nb) target = nb;
    int64_t m = (t64 + BB_WAVES * target - 1) / (BB_WAVES * target);
    if (m < 1) m = 1;
    tiles_per_wave = (int)m;
    blocks = (int)((t64 + BB_WAVES * m - 1) / (BB_WAVES * m));
    if (blocks < 1) blocks = 1;
}

// ---- weight-block images of the forward kernels, packed once per step ------------------------------------------------------------------
// Every forward convolution keeps its bf16 weight blocks (the A operands of v_mfma_f32_4x4x4_16b_bf16) in registers; building them from
// the fp32 master in front of every 256-row workgroup costs 4 scattered loads + conversions per register.  tpack_k builds every
// lane's registers ONCE per step (the parameters change with every Adam update): image m = 64 lanes x 8 bytes.
//   [0, 63)     9 convolutions cin->8 x 7 images: block_in.0, block_in.3, outter_blocks.0-6.3          (bconv_k MODE 0)
//   [63, 119)   prune convolutions 0-7 x 7                                                              (MODE 1)
//   [119, 151)  conv0_0 of block 0-7 x 4                                                                (MODE 2)
//   [151, 183)  [conv0_1 | conv1_1] of block 0-7 x 4                                                    (MODE 3)
//   [183, 217)  the shared first convolution of the 7 outter blocks x 34                                (bocc7_k)
//   [217, 273)  the same kernels as the A operands of v_mfma_f32_16x16x32_bf16: 28 images of 64 x 16 bytes (bocc7m_k)
#define TP_CONV0 0
#define TP_PRUNE 63
#define TP_C00 119
#define TP_DUAL 151
#define TP_OCC 183
#define TP_OCCM 217                     // [217, 273)  the same seven convolutions as ONE matrix: 28 operand images of 16 bytes per lane (bocc7m_k)
#define TP_IMAGES 273
struct TPack { int64_t conv0_w[9], pr_w[8], c00_w[8], c01_w[8], c11_w[8], occ_w[7]; };
__host__ __device__ constexpr int oj_g(int j) { return j < 8 ? j >> 1 : 4 + ((j - 8) >> 2); }
__host__ __device__ constexpr int oj_h(int j) { return j < 8 ? j & 1 : ((j - 8) >> 1) & 1; }
__host__ __device__ constexpr int oj_q(int j) { return j < 8 ? 0 : (j - 8) & 1; }
// bocc7m_k: slot s (0..26) of the matrix product's K dimension -> tap; column-major over the map's nine (x, y) columns, so that the
// four slots of a K step are neighbours in memory (the three z taps of a column are consecutive rows)
__host__ __device__ constexpr int bm_tap(int s) { return s / 3 + 9 * (s % 3); }

// (blocks behind the images clear the pad rows of the arena's matrices - zero_pads16_k's work: one launch less per step)
__global__ __launch_bounds__(64) void tpack_k(const float* __restrict__ P, TPack t, uint2* __restrict__ wimg, bf16_t* __restrict__ pad_base, BPads pl) {
    const int m = blockIdx.x, lane = threadIdx.x;
    if (m >= TP_IMAGES) {
        const int b = m - TP_IMAGES;
        if (b < pl.n && lane < pl.w[b]) pad_base[pl.off[b] + lane] = 0;
        return;
    }
    const int blk = lane >> 2, i = lane & 3;
    float w[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (m < TP_DUAL) {
        // block c = 16 r + blk of the layer's register r:  cin->8 / prune: c = 4 k + 2 h + q -> W[k][4 q + kk][4 h + i];
        // conv0_0: c = 2 k + q -> W00[k][4 q + kk][i]
        const bool c00 = m >= TP_C00;
        const int per = c00 ? 4 : 7, layer = c00 ? (m - TP_C00) / 4 : m / 7, r = c00 ? (m - TP_C00) % 4 : m % 7;
        const int c = 16 * r + blk;
        const int k = c00 ? c >> 1 : c >> 2;
        (void)per;
        if (k < 27) {
            const float* W = P + (c00 ? t.c00_w[layer] : (m < TP_PRUNE ? t.conv0_w[layer] : t.pr_w[layer - 9]));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (c00) w[kk] = W[(k * 8 + 4 * (c & 1) + kk) * 4 + i];
                else w[kk] = W[(k * 8 + 4 * (c & 1) + kk) * 8 + 4 * ((c >> 1) & 1) + i];
            }
        }
    } else if (m < TP_OCC) {            // c = 2 k + t -> (t ? W11 : W01)[k][kk][i]
        const int layer = (m - TP_DUAL) / 4, r = (m - TP_DUAL) % 4;
        const int c = 16 * r + blk, k = c >> 1;
        if (k < 27) {
            const float* W = P + ((c & 1) ? t.c11_w[layer] : t.c01_w[layer]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) w[kk] = W[(k * 4 + kk) * 4 + i];
        }
    } else if (m < TP_OCCM) {           // c = 20 k + j, j -> (group g, output quad h, input quad q): W_g[k][4 q + kk][4 h + i], cin = g + 1
        const int c = 16 * (m - TP_OCC) + blk, k = c / 20, j = c % 20;
        if (k < 27) {
            const int g = oj_g(j), h = oj_h(j), q = oj_q(j), cin = g + 1;
            const float* W = P + t.occ_w[g];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                if (4 * q + kk < cin) w[kk] = W[(k * cin + 4 * q + kk) * 8 + 4 * h + i];
        }
    } else {                             // bocc7m_k: image im = 4 ks + mt, half = input channels 4 half ..: lane (col = lane & 15, kb = lane >> 4)
        const int im = (m - TP_OCCM) >> 1, half = (m - TP_OCCM) & 1;
        const int ks = im >> 2, mt = im & 3, col = 16 * mt + (lane & 15), g = col >> 3, co = col & 7, slot = 4 * ks + (lane >> 4);
        if (slot < 27 && g < 7) {
            const int k = bm_tap(slot), cin = g + 1;
            const float* W = P + t.occ_w[g];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                if (4 * half + kk < cin) w[kk] = W[(k * cin + 4 * half + kk) * 8 + co];
        }
        wimg[(int64_t)TP_OCCM * 64 + (im * 64 + lane) * 2 + half] = make_uint2(pack2(w[0], w[1]), pack2(w[2], w[3]));
        return;
    }
    wimg[m * 64 + lane] = make_uint2(pack2(w[0], w[1]), pack2(w[2], w[3]));
}

// ---- the seven first convolutions of the outter blocks from ONE gather of the occupancy rows --------------------------------------
// A[b] = relu(conv3(occ[:, :b]; a_b) + bias) for b = 1..7 (models/upsample.py:206-214, make_block :88-97): the seven convolutions read
// the SAME rows, so forward and weight gradient gather them once instead of seven times (csrc/fused.hip: occ_conv7_k and
// csrc/occ_wgrad.hip: occ_wgrad7_k do the same for the fp32 executor).  Per tap the (group g, output quad h, input quad q) blocks
// that exist - input channels 4 q .. are only present when 4 q < g + 1: 2 blocks for the groups 0..3, 4 for the groups 4..6 = 20.
struct BoArgs {
    const bf16_t* occ;  bf16_t* out;  const float* P;  const uint2* wimg;
    const int32_t* lo;  const uint32_t* mask;  int64_t ld, n;
    int64_t b[7], g_out[7];
};

// forward: lane = output row, 540 weight blocks (27 taps x 20, images TP_OCC.. of tpack_k) in 68 registers, 14 accumulators; the arithmetic
// of bconv_k<0> per group (same blocks, same tap order, fp32 accumulation from the bias) - bit-identical to seven separate launches
__global__ __launch_bounds__(LINR_BLOCK) void bocc7_k(BoArgs a) {
    constexpr int NG = (27 * 20 + 15) / 16;
    const int lane = threadIdx.x & 63;
    s16x4 wv[NG];
    {
        const uint2* wp = a.wimg + (int64_t)TP_OCC * 64 + lane;
#pragma unroll
        for (int r = 0; r < NG; ++r) wv[r] = __builtin_bit_cast(s16x4, wp[r * 64]);
    }
    const int64_t row_raw = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    const bool live = row_raw < a.n;
    const int64_t row = live ? row_raw : a.n - 1;
    const char* pad = reinterpret_cast<const char*>(a.occ - 8);
    uint32_t off[27];
    decode_offsets16(a.lo, a.mask, a.ld, row, off);
    f32x4 acc[7][2];
#pragma unroll
    for (int g = 0; g < 7; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[g][h][j] = a.P[a.b[g] + 4 * h + j];
#ifndef BOCC_PF
#define BOCC_PF 4
#endif
    constexpr int PF = BOCC_PF;
    uint4 x[PF + 1];
#pragma unroll
    for (int u = 0; u < PF; ++u) x[u] = *reinterpret_cast<const uint4*>(pad + off[LINR_TAP(u)]);
    sfor<27>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        constexpr int k = LINR_TAP(kk);
        if constexpr (kk + PF < 27) x[(kk + PF) % (PF + 1)] = *reinterpret_cast<const uint4*>(pad + off[LINR_TAP(kk + PF)]);
        const uint4 r = x[kk % (PF + 1)];
        const s16x4 q0 = __builtin_bit_cast(s16x4, make_uint2(r.x, r.y));
        const s16x4 q1 = __builtin_bit_cast(s16x4, make_uint2(r.z, r.w));
        sfor<20>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int c0 = 20 * k + j, g = oj_g(j), h = oj_h(j);
            if constexpr (oj_q(j) == 0) acc[g][h] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[c0 / 16], q0, acc[g][h], 4, c0 % 16, 0);
            else acc[g][h] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[c0 / 16], q1, acc[g][h], 4, c0 % 16, 0);
        });
    });
    if (!live) return;
#pragma unroll
    for (int g = 0; g < 7; ++g) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = fmaxf(acc[g][0][j], 0.0f); o[4 + j] = fmaxf(acc[g][1][j], 0.0f); }
        *reinterpret_cast<uint4*>(a.out + a.g_out[g] + row * 8) = pack_row(o);
    }
}

// The same seven convolutions as ONE matrix product on v_mfma_f32_16x16x32_bf16 (round 6; bocc7_k stays behind LINR_BOCC7_MFMA16=0):
//   out^T [56 columns = (group, cout), padded to 64] x [64 rows] = W^T [64 x 224] . gathered^T [224 = 28 slots x 8 channels x 64 rows]
// with the COLUMNS as M and the rows as N: the A operand of lane (m = lane & 15, kb = lane >> 4) are the eight input channels of slot
// 4 ks + kb for column 16 mt + m (tpack_k's images TP_OCCM.., 16 bytes per lane, kept in LDS for the whole block), the B operand of lane
// (n = lane & 15, kb) IS the gathered 16-byte occupancy row of neighbour slot 4 ks + kb of row 16 nt + n - no transposition anywhere - and
// a lane's four accumulator registers are four consecutive output channels of one row: 8-byte stores.  112 matrix instructions of 16
// cycles per 64-row tile instead of 540 of 8.  The byte offsets are decoded lane = row as everywhere (decode_offsets16) and handed to the
// (row, slot) lanes through a wave-private LDS image [row][kb][ks] (row stride 36 words: conflict-free 16-byte accesses both ways).
// Accumulation order differs from bocc7_k's (32 products per instruction): results agree to fp32 summation order, not bit for bit.
#define BM_XSTR 36
__global__ __launch_bounds__(256, 2) void bocc7m_k(BoArgs a, int tile_quads) {
    __shared__ uint4 wl[28 * 64];
    __shared__ uint32_t xo[4][64 * BM_XSTR];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = lane & 15, kb = lane >> 4;
    {
        const uint4* wsrc = reinterpret_cast<const uint4*>(a.wimg + (int64_t)TP_OCCM * 64);
        uint4 t[7];                                               // (all seven loads in flight: a rolled loop waits for each)
#pragma unroll
        for (int i = 0; i < 7; ++i) t[i] = wsrc[i * 256 + threadIdx.x];
#pragma unroll
        for (int i = 0; i < 7; ++i) wl[i * 256 + threadIdx.x] = t[i];
    }
    // a lane's columns: 16 mt + 4 kb + r -> group 2 mt + (kb >> 1), output channels 4 (kb & 1) + r
    float bias[4][4];
    int64_t gout[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const bool hi = (kb >> 1) != 0;
        const bool real = !(mt == 3 && hi);                     // group 7 does not exist
        const int64_t bo = real ? (hi ? a.b[(2 * mt + 1) % 7] : a.b[2 * mt]) : 0;
        gout[mt] = hi ? a.g_out[(2 * mt + 1) % 7] : a.g_out[2 * mt];
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[mt][r] = real ? a.P[bo + 4 * (kb & 1) + r] : 0.0f;
    }
    __syncthreads();
    uint32_t* xw = xo[wave];
    const char* pad = reinterpret_cast<const char*>(a.occ - 8);
    // the map words of a wave's NEXT tile are loaded behind the gathers of the current one (the decode is a dependent load chain:
    // words -> offsets -> LDS -> gathers); all 28 gathers of a tile are in flight before its first matrix instruction
    uint32_t raw[10];
    auto words = [&](int64_t tq) {
        const int64_t row_raw = (tq * 4 + wave) * 64 + lane;
        const int64_t row = row_raw < a.n ? row_raw : a.n - 1;
        load_words16(a.lo, a.mask, a.ld, row, raw);
    };
    if ((int64_t)blockIdx.x < tile_quads) words(blockIdx.x);
    for (int64_t tq = blockIdx.x; tq < tile_quads; tq += gridDim.x) {
        const int64_t row0 = (tq * 4 + wave) * 64;
        {
            uint32_t off[27];
            decode_words16(raw, off);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t v[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) v[ks] = (ks < 7 && 4 * ks + q < 27) ? off[bm_tap((4 * ks + q) % 27)] : 0u;
                *reinterpret_cast<uint4*>(xw + lane * BM_XSTR + q * 8) = make_uint4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<uint4*>(xw + lane * BM_XSTR + q * 8 + 4) = make_uint4(v[4], v[5], v[6], v[7]);
            }
        }
        uint4 x[7][4];
        {
            uint32_t go[4][8];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const uint4 lo4 = *reinterpret_cast<const uint4*>(xw + (16 * nt + n16) * BM_XSTR + kb * 8);
                const uint4 hi4 = *reinterpret_cast<const uint4*>(xw + (16 * nt + n16) * BM_XSTR + kb * 8 + 4);
                go[nt][0] = lo4.x; go[nt][1] = lo4.y; go[nt][2] = lo4.z; go[nt][3] = lo4.w;
                go[nt][4] = hi4.x; go[nt][5] = hi4.y; go[nt][6] = hi4.z; go[nt][7] = hi4.w;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {                          // K step major (pinned): the first matrix instructions wait for the oldest gathers
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) x[ks][nt] = *reinterpret_cast<const uint4*>(pad + go[nt][ks]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        words(tq + gridDim.x);                                    // (unconditional - rows are clamped - so that the waits below count exactly)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){bias[mt][0], bias[mt][1], bias[mt][2], bias[mt][3]};
        sfor<7>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            uint4 w[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) w[mt] = wl[(ks * 4 + mt) * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[mt]),
                                                                          __builtin_bit_cast(bf16x8, x[ks][nt]), acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                    // K steps in order: each waits for its own four gathers only
        });
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int64_t row = row0 + 16 * nt + n16;
            if (row >= a.n) continue;                             // (also the waves of the last quad that have no tile)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (mt == 3 && (kb >> 1)) continue;               // group 7
                const f32x4 v = acc[mt][nt];
                *reinterpret_cast<uint2*>(a.out + gout[mt] + row * 8 + 4 * (kb & 1)) =
                    make_uint2(pack2(fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)), pack2(fmaxf(v[2], 0.0f), fmaxf(v[3], 0.0f)));
            }
        }
    }
}

// weight gradients gW_g[k][ci][co] = sum_rows occ[nbr(row, k)][ci] gA_g[row][co], gb_g = sum_rows gA_g[row]: v_mfma_f32_16x16x32_bf16 with the
// ROWS as its K dimension (32 per instruction; transposed LDS reads, like bbwd_k): M = 2 taps x 8 input channels of the gathered
// occupancy rows, N = 2 groups x 8 output channels of the rows' own gradients - every lane of every tile useful (but for the
// triangle ci <= group).  Wave w of a block owns the taps of steps 7 w .. 7 w + 6 as four tap pairs (wave 3: six taps and, in its
// seventh slot, an all-ones pseudo tap whose product is the bias gradient) and walks over ALL tiles of the block with 4 x 4
// accumulators (tap pair x group pair): no fold across waves, every wave writes its own part of the block's slab row.  32 matrix
// instructions and 32 transposed reads per 64-row tile and wave (the 4x4x4 form: 176 and 48).
#define OW_SLOTS 16                     // 7 tap slots + the 7 groups' own rows + 2 never written (the pair partner of group 6: its columns are unused)
#define OW_WAVES 8                      // two waves per tap set (alternating tiles): two waves per SIMD, one block per CU (139 KB of LDS)
struct OwArgs {
    const bf16_t* occ;  const bf16_t* g;
    const int32_t* lo;  const uint32_t* mask;  int64_t ld, n;
    int tiles_per_block;
    float* big;  int64_t block_stride;
    int64_t g_g[7], w[7], b[7];
};

__global__ __launch_bounds__(OW_WAVES * 64, 2) void bocc_wgrad7_k(OwArgs a) {
    extern __shared__ uint4 smem[];                                            // OW_WAVES * OW_SLOTS * BB_SLOT bytes
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3, half = wave8 >> 2;                             // tap set; which tiles of the block (even / odd)
    const int ntaps = wave == 3 ? 6 : 7;
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    char* img = reinterpret_cast<char*>(smem) + wave8 * (OW_SLOTS * BB_SLOT);
    char* imgW = img + lane * 16;
    // both operands: piece tp of [slot 2 j: 8 columns | slot 2 j + 1: 8 columns], rows 8 tg + tq (+ 32 ks + 4 h) - A: tap slots 0.., B: own rows 7..
    const char* aR = img + (tp >> 1) * BB_SLOT + (8 * tg + tq) * 16 + 8 * (tp & 1);
    const char* bR = aR + 7 * BB_SLOT;
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const int64_t n = a.n;
    const int64_t T64 = (n + 63) >> 6;
    const int64_t tb0 = (int64_t)blockIdx.x * a.tiles_per_block;
    const int64_t tb1 = tb0 + a.tiles_per_block < T64 ? tb0 + a.tiles_per_block : T64;
    const char* pad = reinterpret_cast<const char*>(a.occ - 8);
    if (wave == 3) {                                                           // the pseudo tap: ones (bf16 1.0 = 0x3f80) in slot 6, written once
        *reinterpret_cast<uint4*>(imgW + 6 * BB_SLOT) = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    }
    // Two waves per SIMD: the loads of the wave's next tile (index words, its 7 gathers, the 7 groups' own rows) are still issued before
    // the matrix instructions of the current one.
    uint4 x[7], own[7], xn[7], ownn[7];
    auto load_tile = [&](int64_t tile, auto nxt) {
        constexpr bool NXT = decltype(nxt)::value;
        const int64_t row_raw = (tile << 6) + lane;
        const bool live = row_raw < n;
        const int64_t row = live ? row_raw : n - 1;
        uint32_t off[27];
        decode_offsets16(a.lo, a.mask, a.ld, row, off);
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            // off[LINR_TAP(7 wave + t)]: the table is indexed with compile-time steps, the wave selects among the four constants
            const uint32_t o = wave == 0 ? off[LINR_TAP(t)] : wave == 1 ? off[LINR_TAP(7 + t)] : wave == 2 ? off[LINR_TAP(14 + t)] : off[LINR_TAP(t < 6 ? 21 + t : 26)];
            const uint4 v = *reinterpret_cast<const uint4*>(pad + o);
            if constexpr (NXT) xn[t] = v; else x[t] = v;
        }
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);                              // dead lanes contribute nothing
            if (live) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.g + a.g_g[g]) + ((uint32_t)row << 4));
            if constexpr (NXT) ownn[g] = v; else own[g] = v;
        }
    };
    if (tb0 + half < tb1) load_tile(tb0 + half, std::false_type{});
    for (int64_t tile = tb0 + half; tile < tb1; tile += 2) {
#pragma unroll
        for (int t = 0; t < 7; ++t)
            if (t < ntaps) *reinterpret_cast<uint4*>(imgW + t * BB_SLOT) = x[t];
#pragma unroll
        for (int g = 0; g < 7; ++g) *reinterpret_cast<uint4*>(imgW + (7 + g) * BB_SLOT) = own[g];
        if (tile + 2 < tb1) load_tile(tile + 2, std::true_type{});
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s16x4 av[4][2], bv[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    av[j][h] = tr_read(aR + 2 * j * BB_SLOT + (32 * ks + 4 * h) * 16);
                    bv[j][h] = tr_read(bR + 2 * j * BB_SLOT + (32 * ks + 4 * h) * 16);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = mfma16(av[i][0], av[i][1], bv[j][0], bv[j][1], d[i][j]);
        }
#pragma unroll
        for (int t = 0; t < 7; ++t) { x[t] = xn[t]; own[t] = ownn[t]; }
    }
    // ---- the two waves of a tap set: even tiles + odd tiles, through the second wave's own image (16 x 4 x 64 floats = 16 KB) ----------------
    __syncthreads();
    float* fold = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + (wave + 4) * (OW_SLOTS * BB_SLOT));
    if (half == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) fold[((i * 4 + j) * 4 + r) * 64 + lane] = d[i][j][r];
    }
    __syncthreads();
    if (half == 1) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) d[i][j][r] += fold[((i * 4 + j) * 4 + r) * 64 + lane];
    // ---- every tap set writes its own slab entries: tile (tap pair i, group pair j), row m = 4 (lane >> 4) + reg, column lane & 15 ------------
    float* dst = a.big + (int64_t)blockIdx.x * a.block_stride;
    const int nn = lane & 15, g_lo = nn >> 3, co = nn & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = 2 * j + g_lo;
            if (g > 6) continue;
            const int cin = g + 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 4 * (lane >> 4) + r, slot = 2 * i + (m >> 3), ci = m & 7;
                if (slot < ntaps) {
                    const int kk = 7 * wave + slot;
                    const int k = kk / 9 + 3 * ((kk / 3) % 3) + 9 * (kk % 3);
                    if (ci < cin) dst[a.w[g] + ((int64_t)k * cin + ci) * 8 + co] = d[i][j][r];
                } else if (wave == 3 && slot == 6 && ci == 0) {
                    dst[a.b[g] + co] = d[i][j][r];                            // the all-ones pseudo tap: every row holds the bias gradient
                }
            }
        }
}

// ---- backward of the occupancy heads (csrc/head_bwd.h, shared with the fp32 executor's head_bwd_k) -----------------------------------
// bf16 rows in (the STORED C_k) and out (gC); the MLP, its weight gradients and gz are fp32 arithmetic in this executor as well.
struct THeadArgs {
    const bf16_t* c;  const float* p;  const float* target;  int target_ld;
    const float* P;  float gscale;  bf16_t* gc;  int64_t n;
    float* big;  int64_t block_stride;
    int64_t g_c[8], g_p[8], g_t[8], g_gc[8];
    int64_t w1[8], b1[8], w2[8], b2[8];
    int active;
};

__global__ __launch_bounds__(HB_WAVES * 64, 2) void thead_bwd_k(THeadArgs A) {
    __shared__ float lds[HB_LDS_FLOATS];
    const int gi = blockIdx.y;
    const bf16_t* Cm = A.c + A.g_c[gi];
    bf16_t* GC = A.gc + A.g_gc[gi];
    HbParams h;
    h.p = A.p + A.g_p[gi]; h.target = A.target + A.g_t[gi]; h.target_ld = A.target_ld;
    h.w1 = A.P + A.w1[gi]; h.b1 = A.P + A.b1[gi]; h.w2 = A.P + A.w2[gi];
    h.gscale = A.gscale; h.n = A.n;
    h.dst = A.big + (int64_t)blockIdx.x * A.block_stride;
    h.off_w1 = A.w1[gi]; h.off_b1 = A.b1[gi]; h.off_w2 = A.w2[gi]; h.off_b2 = A.b2[gi];
    h.active = A.active;
    head_bwd_body<u32x4>(h,
        [&](int64_t row) { return *reinterpret_cast<const u32x4*>(Cm + row * 8); },
        [](const u32x4& r, float (&c)[8]) { unpack_row(make_uint4(r[0], r[1], r[2], r[3]), c); },
        [&](int64_t row, const float (&g)[8]) { *reinterpret_cast<uint4*>(GC + row * 8) = pack_row(g); }, lds);
}

// x_glob's gradient: the fan-in of the eight priors (models/upsample.py:206-214 under autograd), summed in fp32 in the order of the
// fp32 executor's sum8_k, stored once as bf16
struct TPtr8 { const bf16_t* p[8]; };
__global__ __launch_bounds__(LINR_BLOCK) void tsum8_k(TPtr8 src, int64_t n, bf16_t* __restrict__ dst) {
    const int64_t r = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (r >= n) return;
    float t[8], u[8];
    unpack_row(*reinterpret_cast<const uint4*>(src.p[7] + r * 8), t);
#pragma unroll
    for (int k = 6; k >= 0; --k) {
        unpack_row(*reinterpret_cast<const uint4*>(src.p[k] + r * 8), u);
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = t[j] + u[j];
    }
    *reinterpret_cast<uint4*>(dst + r * 8) = pack_row(t);
}

// ---- arena --------------------------------------------------------------------------------------------------------------------------
struct TArena {
    int64_t rows;
    char* base;
    int64_t cur;                     // bytes
    float *gX0, *PR, *GSUM, *BIG;
    uint2* WIMG;                     // [TP_IMAGES][64] weight-block images of the forward kernels (tpack_k)
    double* part;
    bf16_t *X0, *OCC, *A[8], *H[8], *I[8], *O[8], *C[8], *M[8];
    bf16_t *gC[8], *gO[8], *gXG, *gI[8], *G2[8], *gH[8], *gA[8];
    BPads pads;
    bf16_t* mats;
    int64_t n_params;
};

static void* tbytes(TArena& a, int64_t nbytes) {
    void* p = a.base ? a.base + a.cur : nullptr;
    a.cur += (nbytes + 63) & ~(int64_t)63;
    return p;
}
static bf16_t* tmat(TArena& a, int w = 8) {          // [1 + rows][w] with the zero row in front
    bf16_t* p = a.base ? reinterpret_cast<bf16_t*>(a.base + a.cur) : nullptr;
    if (a.pads.n >= BPADS_MAX) abort();                  // (99 matrices today)
    a.pads.off[a.pads.n] = (a.cur - (int64_t)(reinterpret_cast<char*>(a.mats) - a.base)) / 2;
    a.pads.w[a.pads.n++] = w;
    a.cur += (((a.rows + 1) * w * 2) + 63) & ~(int64_t)63;
    return p ? p + w : nullptr;
}

static void make_tarena(TArena& a, int64_t rows, char* base, int64_t n_params, bool own_occ) {
    a.rows = rows; a.base = base; a.cur = 0; a.pads.n = 0; a.n_params = n_params;
    a.gX0 = (float*)tbytes(a, rows * 8 * 4);
    a.PR = (float*)tbytes(a, rows * 8 * 4);
    a.part = (double*)tbytes(a, (int64_t)8 * linr_grid(rows, LINR_BLOCK) * 8);
    a.WIMG = (uint2*)tbytes(a, (int64_t)TP_IMAGES * 64 * 8);
    a.GSUM = (float*)tbytes(a, n_params * 4);
    a.BIG = (float*)tbytes(a, (int64_t)LINR_WG_BLOCKS * n_params * 4);
    a.mats = reinterpret_cast<bf16_t*>(base + a.cur);
    a.X0 = tmat(a);
    a.OCC = own_occ ? tmat(a) : nullptr;
    for (int b = 0; b < 8; ++b) { a.A[b] = tmat(a); a.H[b] = tmat(a); a.I[b] = tmat(a); a.O[b] = tmat(a); a.M[b] = tmat(a, 4); }
    for (int k = 0; k < 8; ++k) { a.C[k] = tmat(a); a.gC[k] = tmat(a); a.gO[k] = tmat(a); }
    a.gXG = tmat(a);
    for (int b = 0; b < 8; ++b) { a.gI[b] = tmat(a); a.G2[b] = tmat(a); a.gH[b] = tmat(a); a.gA[b] = tmat(a); }
}

extern "C" size_t linr_net_train_bf16_arena_bytes(int64_t rows, int32_t block_layers) {
    if (rows < 0 || block_layers > 1) return 0;                  // block_layers 1 only (every BASELINE config)
    Layout L;
    make_layout(L, MAX_SCALES, 1);
    TArena a;
    make_tarena(a, rows, nullptr, L.total, true);
    return (size_t)a.cur + 64;
}

// occupancy fp32 [rows][8] -> bf16 [1 + rows][8] with the zero row in front (out points at the pad row): a frame's occupancy does
// not change over the epochs, so a caller can convert it once and hand it to every step (occ_bf16 of the entries below)
extern "C" int linr_occ_to_bf16(const float* occ, int64_t rows, uint16_t* out_padded, void* stream) {
    if (rows < 0) return LINR_EINVAL;
    if (!out_padded) return LINR_EINVAL;
    if ((((uintptr_t)out_padded) & 15u) || (occ && !linr_aligned16(occ))) return LINR_EALIGN;
    if (rows > 0 && !occ) return LINR_EINVAL;                  // every argument check in front of the first launch
    hipStream_t s = (hipStream_t)stream;
    BPads pl;
    pl.n = 1; pl.off[0] = 0; pl.w[0] = 8;
    zero_pads16_k<<<1, 64, 0, s>>>(out_padded, pl);
    if (rows > 0) {
        occ_bf16_k<<<linr_grid(rows, LINR_BLOCK), LINR_BLOCK, 0, s>>>(occ, rows, out_padded + 8);
    }
    return linr_launch_rc();
}

// ---- executor -------------------------------------------------------------------------------------------------------------------------
struct TCtx {
    const linr_frame* f;
    Layout L;
    TArena A;
    const float* P;
    const bf16_t* OCC;               // bf16 occupancy rows (zero row in front): the caller's copy or the arena's
    hipStream_t s;
    int64_t R;
    int nb;
    std::vector<LinrShortRange> shortr;
    void note_short(int64_t b, int64_t e, int rows) { if (rows < nb) shortr.push_back({b, e, rows}); }
};

static int tcheck(const linr_frame* f, const float* params, void* arena, size_t arena_bytes, const uint16_t* occ_bf16, TCtx& c) {
    if (!f || !params || !arena) return LINR_EINVAL;
    if (f->rows < 0 || f->n_scales < 1 || f->n_scales > MAX_SCALES || !f->row_off_h || !f->scale_idx_h) return LINR_EINVAL;
    if (f->block_layers > 1) return LINR_EINVAL;                // block_layers 1 only
    if (!make_layout(c.L, f->model_scale_num, 1)) return LINR_EINVAL;
    if (f->row_off_h[0] != 0 || f->row_off_h[f->n_scales] != f->rows) return LINR_EINVAL;
    for (int s = 0; s < f->n_scales; ++s) {
        if (f->row_off_h[s + 1] < f->row_off_h[s]) return LINR_EINVAL;
        if (f->scale_idx_h[s] < 0 || f->scale_idx_h[s] >= f->model_scale_num) return LINR_EINVAL;
    }
    if (f->rows > 0 && (!f->nbr_lo || !f->nbr_mask || !f->offset_feat || !f->occ || f->nbr_ld < f->rows)) return LINR_EINVAL;   // compressed map only
    if (f->rows >= ((int64_t)1 << 27) - 1) return LINR_EINVAL;                    // 32-bit byte offsets of the 16-byte gathers
    if (arena_bytes < linr_net_train_bf16_arena_bytes(f->rows, 1)) return LINR_ENOSPC;
    if (((uintptr_t)arena) & 63u) return LINR_EALIGN;
    if (occ_bf16 && (((uintptr_t)occ_bf16) & 15u)) return LINR_EALIGN;
    c.f = f; c.P = params; c.R = f->rows;
    c.nb = linr_wg_blocks_for(f->rows);
    make_tarena(c.A, f->rows, (char*)arena, c.L.total, true);
    c.OCC = occ_bf16 ? reinterpret_cast<const bf16_t*>(occ_bf16) + 8 : c.A.OCC;
    return 0;
}

static BArgs tbase(const TCtx& c) {
    BArgs a = BArgs();
    a.lo = c.f->nbr_lo; a.mask = c.f->nbr_mask; a.ld = c.f->nbr_ld; a.n = c.R;
    a.codes = nullptr; a.minv = 0.0f; a.range = 0.0f; a.pf = c.P; a.wimg = c.A.WIMG;
    for (int g = 0; g < BMAXG; ++g) a.cin[g] = 8;
    return a;
}

template <int MODE>
static int tlaunch(const TCtx& c, const BArgs& a, int groups) {
    LinrProf ps(c.s, TK_FWD, groups);
    bconv_k<MODE, 2, true><<<dim3(linr_grid(c.R, LINR_BLOCK), groups), LINR_BLOCK, 0, c.s>>>(a);
    return linr_launch_rc();
}

// conv3 cin->8 (+ res) (ReLU) over `groups` layers; ptrs[g] are the groups' matrices (offsets are taken against ptrs[0])
static int tconv(const TCtx& c, const bf16_t* const* in, bf16_t* const* out, const bf16_t* const* res, int relu, int groups,
                 const int64_t* w, const int64_t* b, int layer0) {
    BArgs a = tbase(c);
    a.in = in[0]; a.out = out[0]; a.res = res ? res[0] : nullptr; a.relu = relu;
    for (int g = 0; g < groups; ++g) {
        a.g_in[g] = in[g] - in[0]; a.g_out[g] = out[g] - out[0]; a.g_res[g] = res ? res[g] - res[0] : 0;
        a.w[g] = w[g]; a.b[g] = b[g]; a.cin[g] = 8; a.wi[g] = TP_CONV0 + 7 * (layer0 + g);       // (layers of tpack_k's first range)
    }
    return tlaunch<0>(c, a, groups);
}

// the teacher-forced forward of all 8 stages with every activation the backward pass needs kept in the arena
static int tforward(TCtx& c, float* probs, double* bits_acc) {
    const TArena& a = c.A;
    const Layout& L = c.L;
    const linr_frame* f = c.f;
    const int64_t nblk = linr_grid(c.R, LINR_BLOCK);
    {
        LinrProf ps(c.s, TK_MISC, 1);
        if (c.OCC == a.OCC) occ_bf16_k<<<linr_grid(c.R, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(f->occ, c.R, a.OCC);
        TPack tp;
        tp.conv0_w[0] = L.block_in.a_w; tp.conv0_w[1] = L.block_in.b_w;
        for (int g = 0; g < 7; ++g) { tp.conv0_w[2 + g] = L.outter[g].b_w; tp.occ_w[g] = L.outter[g].a_w; }
        for (int k = 0; k < 8; ++k) {
            const IncP& q = (k == 0 ? L.block_in : L.outter[k - 1]).inc[0];
            tp.pr_w[k] = L.pr_w[k]; tp.c00_w[k] = q.c00_w; tp.c01_w[k] = q.c01_w; tp.c11_w[k] = q.c11_w;
        }
        tpack_k<<<TP_IMAGES + a.pads.n, 64, 0, c.s>>>(c.P, tp, a.WIMG, a.mats, a.pads);
        SceArgs sa;
        sa.n_scales = f->n_scales;
        for (int s = 0; s < f->n_scales; ++s) {
            const int si = f->scale_idx_h[s];
            sa.row_off[s] = f->row_off_h[s];
            sa.emb[s] = L.emb + si * 8; sa.w1[s] = L.m0_w[si]; sa.b1[s] = L.m0_b[si]; sa.w2[s] = L.m2_w[si]; sa.b2[s] = L.m2_b[si];
        }
        sa.row_off[f->n_scales] = f->rows;
        sa.blk_off[0] = 0;
        for (int s = 0; s < f->n_scales; ++s)
            sa.blk_off[s + 1] = sa.blk_off[s] + (int)linr_grid(sa.row_off[s + 1] - sa.row_off[s], LINR_BLOCK);
        PadList none;
        none.n = 0;
        sce_fwd_k<bf16_t><<<sa.blk_off[sa.n_scales], LINR_BLOCK, 0, c.s>>>(c.P, f->offset_feat, sa, c.R, nullptr, nullptr, a.X0, nullptr, none);
    }
    {   // A[0] = relu(conv3(x_low)) (block_in) and A[b] = relu(conv3(occ[:, :b])) (outter block b)
        const bf16_t* in0[1] = {a.X0};
        bf16_t* out0[1] = {a.A[0]};
        TRY(tconv(c, in0, out0, nullptr, 1, 1, &L.block_in.a_w, &L.block_in.a_b, 0));
        BoArgs o;
        o.occ = c.OCC; o.out = a.A[1]; o.P = c.P; o.wimg = a.WIMG; o.lo = f->nbr_lo; o.mask = f->nbr_mask; o.ld = f->nbr_ld; o.n = c.R;
        for (int g = 0; g < 7; ++g) { o.b[g] = L.outter[g].a_b; o.g_out[g] = a.A[g + 1] - a.A[1]; }
        LinrProf ps(c.s, TK_FWD, 7);
        const char* e16 = getenv("LINR_BOCC7_MFMA16");       // 0: bocc7_k (v_mfma_f32_4x4x4_16b_bf16); read per call: tests compare the two
        const int mfma16 = e16 ? atoi(e16) : 1;
        if (mfma16) {
            const int64_t tq = (linr_grid(c.R, 64) + 3) / 4;
            const int64_t blocks = tq < 2 * (int64_t)tb_cus() ? tq : 2 * (int64_t)tb_cus();
            bocc7m_k<<<(int)blocks, 256, 0, c.s>>>(o, (int)tq);
        } else {
            bocc7_k<<<linr_grid(c.R, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(o);
        }
        TRY(linr_launch_rc());
    }
    {   // the Inception layer of all eight blocks: H = [relu(conv0_0(A)) | relu(conv1_0(A))], then I, M
        BArgs h = tbase(c), i2 = tbase(c);
        h.in = a.A[0]; h.out = a.H[0];
        i2.in = a.H[0]; i2.out = a.I[0]; i2.res = a.A[0]; i2.m_out = a.M[0];
        for (int g = 0; g < 8; ++g) {
            const IncP& q = (g == 0 ? L.block_in : L.outter[g - 1]).inc[0];
            h.g_in[g] = a.A[g] - a.A[0]; h.g_out[g] = a.H[g] - a.H[0];
            h.w[g] = q.c00_w; h.b[g] = q.c00_b; h.w2[g] = q.c10_w; h.b2[g] = q.c10_b; h.wi[g] = TP_C00 + 4 * g;
            i2.g_in[g] = a.H[g] - a.H[0]; i2.g_out[g] = a.I[g] - a.I[0]; i2.g_res[g] = a.A[g] - a.A[0]; i2.g_m[g] = a.M[g] - a.M[0];
            i2.w[g] = q.c01_w; i2.b[g] = q.c01_b; i2.w2[g] = q.c11_w; i2.b2[g] = q.c11_b; i2.w3[g] = q.c12_w; i2.b3[g] = q.c12_b;
            i2.wi[g] = TP_DUAL + 4 * g;
        }
        TRY(tlaunch<2>(c, h, 8));
        TRY(tlaunch<3>(c, i2, 8));
    }
    {   // x_glob = O[0] = conv3(I[0]); prior_b = O[b] = conv3(I[b]) + x_glob
        const bf16_t* in0[1] = {a.I[0]};
        bf16_t* out0[1] = {a.O[0]};
        TRY(tconv(c, in0, out0, nullptr, 0, 1, &L.block_in.b_w, &L.block_in.b_b, 1));
        const bf16_t* in[7]; bf16_t* out[7]; const bf16_t* res[7]; int64_t w[7], b[7];
        for (int g = 0; g < 7; ++g) { in[g] = a.I[g + 1]; out[g] = a.O[g + 1]; res[g] = a.O[0]; w[g] = L.outter[g].b_w; b[g] = L.outter[g].b_b; }
        TRY(tconv(c, in, out, res, 0, 7, w, b, 2));
    }
    {   // the 8 heads: C_k = prune conv(prior_k) (stored), p_k, bits partials
        BArgs h = tbase(c);
        h.in = a.O[0]; h.out = a.C[0];
        h.target = f->occ; h.target_ld = 8;
        h.p_out = a.PR; h.partial = a.part;
        for (int k = 0; k < 8; ++k) {
            h.g_in[k] = a.O[k] - a.O[0]; h.g_out[k] = a.C[k] - a.C[0];
            h.w[k] = L.pr_w[k]; h.b[k] = L.pr_b[k]; h.wi[k] = TP_PRUNE + 7 * k;
            h.h_w1[k] = L.h0_w[k]; h.h_b1[k] = L.h0_b[k]; h.h_w2[k] = L.h2_w[k]; h.h_b2[k] = L.h2_b[k];
            h.t_col[k] = k; h.p_off[k] = (int64_t)k * c.R; h.part_off[k] = (int64_t)k * nblk;
        }
        TRY(tlaunch<1>(c, h, 8));
    }
    if (bits_acc) TRY(linr_bits_finish_launch(a.part, (int)(8 * nblk), bits_acc, c.s));
    if (probs) TRY(linr_hip_rc(hipMemcpyAsync(probs, a.PR, (size_t)c.R * 8 * sizeof(float), hipMemcpyDeviceToDevice, c.s)));
    return linr_launch_rc();
}

static BbArgs bb_base(const TCtx& c) {
    BbArgs a = BbArgs();
    a.P = c.P; a.lo = c.f->nbr_lo; a.mask = c.f->nbr_mask; a.ld = c.f->nbr_ld; a.n = c.R;
    a.big = c.A.BIG; a.block_stride = c.L.total;
    for (int g = 0; g < TB_MAXG; ++g) a.cin[g] = 8;
    return a;
}

template <int KIND, int EPI>
static int bb_launch(TCtx& c, BbArgs& a, int groups, int kind_prof, int* rows) {
    int blocks = 1;
    tb_grid(c.R, c.nb, groups, a.tiles_per_wave, blocks);
    *rows = blocks;
    LinrProf ps(c.s, kind_prof, groups);
    bbwd_k<KIND, EPI><<<dim3(blocks, groups), BB_WAVES * 64, 0, c.s>>>(a);
    return linr_launch_rc();
}

// backward of gscale * bits: leaves the parameter gradient in the arena's GSUM (flat, parameters() order)
static int tbackward(TCtx& c, float gscale) {
    const TArena& a = c.A;
    const Layout& L = c.L;
    const float gz_scale = gscale * 1.4426950408889634f;       // d(bits)/d(nats) = 1/ln 2
    c.shortr.clear();
    {   // heads: gC and the four head-parameter gradients
        THeadArgs h = THeadArgs();
        h.c = a.C[0]; h.p = a.PR; h.target = c.f->occ; h.target_ld = 8; h.P = c.P; h.gscale = gz_scale; h.gc = a.gC[0]; h.n = c.R;
        h.big = a.BIG; h.block_stride = L.total;
        for (int k = 0; k < 8; ++k) {
            h.g_c[k] = a.C[k] - a.C[0]; h.g_p[k] = (int64_t)k * c.R; h.g_t[k] = k; h.g_gc[k] = a.gC[k] - a.gC[0];
            h.w1[k] = L.h0_w[k]; h.b1[k] = L.h0_b[k]; h.w2[k] = L.h2_w[k]; h.b2[k] = L.h2_b[k];
        }
        h.active = hb_blocks(c.R, 8, tb_cus(), c.nb);
        LinrProf ps(c.s, TK_HEAD_BWD, 8);
        thead_bwd_k<<<dim3(h.active, 8), HB_WAVES * 64, 0, c.s>>>(h);
        TRY(linr_launch_rc());
        c.note_short(L.h0_w[0], L.h2_b[7] + 1, h.active);          // the heads' parameters are one contiguous range
    }
    int rows = 0;
    {   // C_k = conv3(prior_k; prune_k): gO[k] = bwd(gC[k]) and the kernel / bias gradients, one gather of gC
        BbArgs b = bb_base(c);
        b.g = a.gC[0]; b.xin = a.O[0]; b.out = a.gO[0];
        for (int k = 0; k < 8; ++k) {
            b.g_g[k] = a.gC[k] - a.gC[0]; b.g_x[k] = a.O[k] - a.O[0]; b.g_out[k] = a.gO[k] - a.gO[0];
            b.w[k] = L.pr_w[k]; b.b[k] = L.pr_b[k];
        }
        TRY((bb_launch<0, 0>(c, b, 8, TK_BWD88, &rows)));
        c.note_short(L.pr_w[0], L.pr_b[7] + 8, rows);
    }
    {   // prior_k = x_glob (+ outter block k): x_glob receives every gO
        TPtr8 src;
        for (int k = 0; k < 8; ++k) src.p[k] = a.gO[k];
        LinrProf ps(c.s, TK_MISC, 1);
        tsum8_k<<<linr_grid(c.R, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(src, c.R, a.gXG);
        TRY(linr_launch_rc());
    }
    {   // O = conv3(I; b): gI = bwd(gO), gM, G2 = [gI[:, 0:4] | gM], the kernel / bias gradients and conv1_2's
        BbArgs b = bb_base(c);
        const bf16_t* gO0 = a.gXG;                             // slot 0 = block_in, whose output gradient is the fan-in sum
        b.g = gO0; b.xin = a.I[0]; b.out = a.gI[0]; b.m = a.M[0]; b.g2 = a.G2[0];
        for (int g = 0; g < 8; ++g) {
            const BlockP& bp = g == 0 ? L.block_in : L.outter[g - 1];
            b.g_g[g] = (g == 0 ? a.gXG : a.gO[g]) - gO0; b.g_x[g] = a.I[g] - a.I[0]; b.g_out[g] = a.gI[g] - a.gI[0];
            b.g_m[g] = a.M[g] - a.M[0]; b.g_g2[g] = a.G2[g] - a.G2[0];
            b.w[g] = bp.b_w; b.b[g] = bp.b_b; b.wp[g] = bp.inc[0].c12_w; b.bp[g] = bp.inc[0].c12_b;
        }
        TRY((bb_launch<0, 3>(c, b, 8, TK_BWD88, &rows)));
        // everything of a block behind its first conv comes from fused launches over the same groups: one range per block
        for (int g = 0; g < 8; ++g) {
            const BlockP& bp = g == 0 ? L.block_in : L.outter[g - 1];
            c.note_short(bp.inc[0].c00_w, bp.b_b + 8, rows);
        }
    }
    {   // both 4->4 convs: gH = [bwd(gI[:, 0:4]; W01) | bwd(gM; W11)] * (H > 0) and their gradients, one gather of G2
        BbArgs b = bb_base(c);
        b.g = a.G2[0]; b.xin = a.H[0]; b.out = a.gH[0];
        for (int g = 0; g < 8; ++g) {
            const IncP& q = (g == 0 ? L.block_in : L.outter[g - 1]).inc[0];
            b.g_g[g] = a.G2[g] - a.G2[0]; b.g_x[g] = a.H[g] - a.H[0]; b.g_out[g] = a.gH[g] - a.gH[0];
            b.w[g] = q.c01_w; b.b[g] = q.c01_b; b.w1[g] = q.c11_w; b.b1[g] = q.c11_b;
        }
        int r2 = 0;
        TRY((bb_launch<1, 0>(c, b, 8, TK_BWD_DUAL, &r2)));
        if (r2 != rows) return LINR_EINVAL;
    }
    {   // conv0_0 (8->4): gA = (bwd(gH[:, 0:4]; W00) + gI + gH[:, 4:8] @ W10^T) * (A > 0), its gradients and conv1_0's
        BbArgs b = bb_base(c);
        b.g = a.gH[0]; b.xin = a.A[0]; b.out = a.gA[0]; b.res = a.gI[0];
        for (int g = 0; g < 8; ++g) {
            const IncP& q = (g == 0 ? L.block_in : L.outter[g - 1]).inc[0];
            b.g_g[g] = a.gH[g] - a.gH[0]; b.g_x[g] = a.A[g] - a.A[0]; b.g_out[g] = a.gA[g] - a.gA[0]; b.g_res[g] = a.gI[g] - a.gI[0];
            b.w[g] = q.c00_w; b.b[g] = q.c00_b; b.wp[g] = q.c10_w; b.bp[g] = q.c10_b;
        }
        int r2 = 0;
        TRY((bb_launch<2, 0>(c, b, 8, TK_BWD_C00, &r2)));
        if (r2 != rows) return LINR_EINVAL;
    }
    {   // A[b] = relu(conv3(occ[:, :b]; a)): kernel / bias gradients of the seven first convolutions from ONE gather of the occupancy
        OwArgs o;
        o.occ = c.OCC; o.g = a.gA[1]; o.lo = c.f->nbr_lo; o.mask = c.f->nbr_mask; o.ld = c.f->nbr_ld; o.n = c.R;
        o.big = a.BIG; o.block_stride = L.total;
        for (int g = 0; g < 7; ++g) { o.g_g[g] = a.gA[g + 1] - a.gA[1]; o.w[g] = L.outter[g].a_w; o.b[g] = L.outter[g].a_b; }
        const int64_t t64 = (c.R + 63) >> 6;
        int64_t target = tb_cus();                            // one 8-wave block per CU (two waves per SIMD)
        if (target > c.nb) target = c.nb;
        o.tiles_per_block = (int)((t64 + target - 1) / target);
        rows = (int)((t64 + o.tiles_per_block - 1) / o.tiles_per_block);
        LinrProf ps(c.s, TK_FIRST_WGRAD, 7);
        static const bool big_lds = hipFuncSetAttribute((const void*)bocc_wgrad7_k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        OW_WAVES * OW_SLOTS * BB_SLOT) == hipSuccess;
        if (!big_lds) return LINR_EINVAL;
        bocc_wgrad7_k<<<rows, OW_WAVES * 64, OW_WAVES * OW_SLOTS * BB_SLOT, c.s>>>(o);
        TRY(linr_launch_rc());
        for (int g = 0; g < 7; ++g) c.note_short(L.outter[g].a_w, L.outter[g].a_b + 8, rows);
    }
    {   // A[0] = relu(conv3(x_low; a)) of block_in: gx_low (fp32 for the scale context's backward) and its gradients
        BbArgs b = bb_base(c);
        b.g = a.gA[0]; b.xin = a.X0; b.out_f32 = a.gX0; b.flags = TB_OUT_F32;
        b.w[0] = L.block_in.a_w; b.b[0] = L.block_in.a_b;
        TRY((bb_launch<0, 0>(c, b, 1, TK_BWD88, &rows)));
        c.note_short(L.block_in.a_w, L.block_in.a_b + 8, rows);
    }
    return linr_bwd_tail_launch(c.f, L, c.P, a.gX0, nullptr, a.BIG, a.GSUM, c.nb, c.shortr.data(), (int)c.shortr.size(), c.s);
}

// ---- C-ABI ----------------------------------------------------------------------------------------------------------------------------
extern "C" int linr_net_forward_train_bf16(const linr_frame* f, const float* params, void* arena, size_t arena_bytes,
                                           const uint16_t* occ_bf16, float* probs, double* bits_acc, void* stream) {
    TCtx c;
    TRY(tcheck(f, params, arena, arena_bytes, occ_bf16, c));
    c.s = (hipStream_t)stream;
    if (c.R == 0) return 0;
    return tforward(c, probs, bits_acc);
}

extern "C" int linr_net_backward_bf16(const linr_frame* f, const float* params, void* arena, size_t arena_bytes,
                                      const uint16_t* occ_bf16, float gscale, float* grads, void* stream) {
    TCtx c;
    TRY(tcheck(f, params, arena, arena_bytes, occ_bf16, c));
    if (!grads) return LINR_EINVAL;
    c.s = (hipStream_t)stream;
    if (c.R == 0) return 0;
    TRY(tbackward(c, gscale));
    return linr_axpy(c.A.GSUM, c.L.total, grads, 1, stream);
}

extern "C" int linr_net_train_step_bf16(const linr_frame* f, float* params, void* arena, size_t arena_bytes,
                                        const uint16_t* occ_bf16, float gscale, float* exp_avg, float* exp_avg_sq, double lr,
                                        int64_t step, const int64_t* scale_steps_h, double beta1, double beta2, double eps,
                                        double weight_decay, double* bits_acc, void* stream) {
    if (!exp_avg || !exp_avg_sq || !bits_acc || step < 1) return LINR_EINVAL;
    TCtx c;
    TRY(tcheck(f, params, arena, arena_bytes, occ_bf16, c));
    if (scale_steps_h) {          // checked before anything is launched
        for (int s = 0; s < c.L.S; ++s)
            if (scale_steps_h[s] < 0) return LINR_EINVAL;
        for (int j = 0; j < f->n_scales; ++j)
            if (f->row_off_h[j + 1] > f->row_off_h[j] && scale_steps_h[f->scale_idx_h[j]] < 1) return LINR_EINVAL;
    }
    c.s = (hipStream_t)stream;
    if (c.R == 0) return 0;
    TRY(tforward(c, nullptr, bits_acc));
    TRY(tbackward(c, gscale));
    LinrProf ps(c.s, TK_MISC, 0);
    return linr_adam_step_launch(c.L, params, c.A.GSUM, exp_avg, exp_avg_sq, lr, step, scale_steps_h, beta1, beta2, eps, weight_decay, c.s);
}

extern "C" int linr_spconv_bwd_fused_bf16(const uint16_t* gout, const uint16_t* in, const int32_t* lo, const uint32_t* mask, int64_t ld,
                                          int64_t n, const float* W, uint16_t* gin, float* slab, int32_t nblocks, int32_t* rows_written,
                                          void* stream) {
    if (rows_written) *rows_written = 0;
    if (n < 0 || ld < n || nblocks < 1) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!gout || !in || !lo || !mask || !W || !gin || !slab || !rows_written) return LINR_EINVAL;
    if ((((uintptr_t)gout) & 15u) || (((uintptr_t)in) & 15u) || (((uintptr_t)gin) & 15u)) return LINR_EALIGN;
    if (n >= ((int64_t)1 << 27) - 1) return LINR_EINVAL;
    BbArgs a = BbArgs();
    a.g = gout; a.xin = in; a.P = W; a.out = gin; a.lo = lo; a.mask = mask; a.ld = ld; a.n = n;
    a.big = slab; a.block_stride = 1736;
    a.w[0] = 0; a.b[0] = 1728; a.cin[0] = 8;
    int blocks = 1;
    tb_grid(n, nblocks, 1, a.tiles_per_wave, blocks);
    *rows_written = blocks;
    bbwd_k<0, 0><<<dim3(blocks, 1), BB_WAVES * 64, 0, (hipStream_t)stream>>>(a);
    return linr_launch_rc();
}

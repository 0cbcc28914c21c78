// Backward of one occupancy head (models/upsample.py:137-161: conv -> PointwiseMLP([8, 24, 1]) -> sigmoid, under the BCE bits of
// models/model_core.py:72-81), shared by the fp32 executor (csrc/fused.hip: head_bwd_k, fp32 rows) and the bf16 training executor
// (csrc/train_bf16.hip: thead_bwd_k, bf16 rows in and out; the MLP itself is fp32 arithmetic in both).  Not part of the C-ABI.
//
// Per row r (lane = row, 64 rows per wave tile), with c = the prune convolution's output, p = the stored probability, t = the target:
//   gz   = gscale * (p - t)                                 (the BCE / sigmoid pair, clamped like torch's BCELoss backward)
//   hpre = b1 + W1 c                                        54 x v_mfma_f32_4x4x1 (weights broadcast: CBSZ/ABID, K = 1)
//   u[j] = [hpre[j] > 0] gz                                 (the hidden gradient without its w2[j] factor)
//   gC   = (W1 diag(w2))^T u                                48 x 4x4x1 (the w2 factor lives in the weight images)
//   gw2[j] += u[j] hpre[j],  gb1'[j] += u[j],  gb2 += gz    per-lane accumulators
//   gW1'[j][i] += u[j] c[i]                                 rows are the reduction: both operands are TRANSPOSED inside each quad of
//                                                           lanes by 4x4x1 products with one-hot B operands (exact: x * 1 + 0), after
//                                                           which a quad's four lanes hold four channels of ONE row and a 4x4x1
//                                                           instruction accumulates sixteen rows' 4x4 outer products (one per block)
//                                                           at full efficiency: 32 + 48 instructions per 64 rows, no LDS, where the
//                                                           16x16x4 form through an LDS tile spent 1024 matrix cycles at 42 %.
// The w2[j] factor of gW1 / gb1 is applied ONCE per block when the partials are written (gW1 = diag(w2) gW1').  Everything a wave
// accumulates stays in registers over all its tiles; blocks are long-lived (one round over the chip), a block folds its four waves
// through LDS in a fixed order and writes ONE slab row.  No atomics; the summation order depends only on (n, grid).
#pragma once
#include "common.h"
#include <utility>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class F, int... Ks>
__device__ __forceinline__ void hb_for_impl(F&& f, std::integer_sequence<int, Ks...>) { (f(std::integral_constant<int, Ks>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void hb_for(F&& f) { hb_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct HbRaw32 { f32x4 a, b; };          // a row of c as loaded: fp32 (two 16-byte halves) / bf16 (one)

#define HB_WAVES 4
#ifndef HB_LAB
#define HB_LAB 0                         // lab builds (tools/head_lab.sh): 1 no hpre MFMAs, 2 no gC MFMAs, 4 no transposes, 8 no X^T G, 16 no mask / per-lane sums, 32 next tile's loads not pinned, 64 gC stored one tile late
#endif
#ifndef HB_GC_CHAINS
#define HB_GC_CHAINS 2                   // independent accumulation chains per output quad of gC
#endif
#define HB_LSTR 68                       // lane stride of the fold buffer: 4 r + lane -> 64 distinct banks over 16 registers
#define HB_LDS_FLOATS (HB_WAVES * 49 * HB_LSTR + 4 * 49)

struct HbParams {                        // one head (group), resolved by the calling kernel
    const float* p;  const float* target;  int target_ld;
    const float* w1;  const float* b1;  const float* w2;      // [24][8], [24], [24]
    float gscale;                        // d loss / d nats
    int64_t n;
    float* dst;                          // this block's slab row
    int64_t off_w1, off_b1, off_w2, off_b2;
    int active;                          // blocks that own tiles; blocks beyond write a zero row
};

// how many long-lived blocks (of HB_WAVES waves) a launch over `groups` heads of n rows uses: one round of two blocks per CU
static inline int hb_blocks(int64_t n, int groups, int cus, int max_blocks) {
    const int64_t t64 = (n + 63) >> 6;
    int64_t b = (int64_t)2 * cus / (groups < 1 ? 1 : groups);
    if (b > max_blocks) b = max_blocks;
    const int64_t need = (t64 + HB_WAVES - 1) / HB_WAVES;
    if (b > need) b = need;
    return (int)(b < 1 ? 1 : b);
}

// Raw: a row of c as it is loaded (kept in the load's own destination registers until its tile comes up: no copies, no early wait);
// LoadC: (int64_t row) -> Raw (row < n guaranteed); Unpack: (const Raw&, float (&c)[8]); StoreG: (int64_t row, const float (&g)[8])
template <class Raw, class LoadC, class Unpack, class StoreG>
__device__ __forceinline__ void head_bwd_body(const HbParams& A, LoadC loadc, Unpack unpack, StoreG storeg, float* __restrict__ lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = lane >> 2, j4 = lane & 3;
    if ((int)blockIdx.x >= A.active) {   // a slab row nobody accumulates into: zeros (the reduction sums every row of the launch)
        for (int e = threadIdx.x; e < 241; e += HB_WAVES * 64) {
            const int64_t o = e < 192 ? A.off_w1 + e : (e < 216 ? A.off_b1 + (e - 192) : (e < 240 ? A.off_w2 + (e - 216) : A.off_b2));
            A.dst[o] = 0.0f;
        }
        return;
    }
    // A-operand images: block (lane >> 2) of register v is "combo" 16 v + block
    //   wA: combo = 6 i + hq -> W1[4 hq + j][i] (i < 8), combos 48..53 -> b1[4 (combo - 48) + j]
    //   wB: combo = 2 jj + q -> W1[jj][4 q + j] * w2[jj]  (jj < 24)
    float wA[4], wB[3], oh[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int cb = 16 * v + blk;
        wA[v] = cb < 48 ? A.w1[(4 * (cb % 6) + j4) * 8 + cb / 6] : (cb < 54 ? A.b1[4 * (cb - 48) + j4] : 0.0f);
        oh[v] = j4 == v ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const int cb = 16 * v + blk;
        wB[v] = A.w1[(cb / 2) * 8 + 4 * (cb % 2) + j4] * A.w2[cb / 2];
    }
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 acc[6][2];                     // acc[jq][iq][m] at lane 4 b + n: sum over the rows of block b of u[4 jq + m] c[4 iq + n]
#pragma unroll
    for (int a = 0; a < 6; ++a) { acc[a][0] = z4; acc[a][1] = z4; }
    float gw2[24], gb1[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) { gw2[j] = 0.0f; gb1[j] = 0.0f; }
    float gz_sum = 0.0f;

    const int64_t tiles = (A.n + 63) >> 6;
    const int64_t stride = (int64_t)A.active * HB_WAVES;
    int64_t t = (int64_t)blockIdx.x * HB_WAVES + wave;
    // The inputs of a wave's NEXT tile are loaded into the very registers the current tile has just been unpacked from, at the top of the
    // iteration, and a scheduling barrier keeps them there: left alone, hipcc sinks those loads behind two thirds of the tile's
    // arithmetic (register pressure) and the next iteration waits for them at its first instruction.  Worth 2-3 % (fp32 60.8 -> 59.0 us,
    // bf16 58.1 -> 56.7, same box): load latency is NOT what holds this kernel - two waves per SIMD (208-224 registers) run ~1 us of
    // dependent 4x4x1 chains and mask arithmetic per tile beside a row stream at 2.1 (bf16) / 3.5 (fp32) TB/s.  Rows beyond n (the last tile's tail, the prefetch behind a wave's last tile) read row n - 1 instead: finite data
    // whose gz is forced to 0.
    const int64_t last = A.n - 1;
    Raw cb;
    float pb, tb;
    {
        const int64_t row = t * 64 + lane;
        const int64_t rc = row < last ? row : last;
        cb = loadc(rc); pb = A.p[rc]; tb = A.target[rc * A.target_ld];
    }
    auto tile = [&](const int64_t tk, const float (&c)[8], const float pp, const float tt, float (&o)[8]) {
        const int64_t row = tk * 64 + lane;
        const bool live = row < A.n;
        // torch: grad = (p - t) / max((1 - p) p, 1e-12) [BCELoss] * (1 - p) p [sigmoid]: the quotient and the product cancel unless clamped
        const float q = (1.0f - pp) * pp;
        const float gz = (live ? A.gscale : 0.0f) * (pp - tt) * (q < 1e-12f ? q * 1e12f : 1.0f);          // (branch-free)
        gz_sum += gz;
        // hpre = b1 + W1 c  (6 output quads; bias through x = 1, then inputs ascending: the forward head's chains)
        f32x4 hp[6];
        if constexpr ((HB_LAB & 1) == 0) {
        hb_for<6>([&](auto hc) {
            constexpr int hq = decltype(hc)::value;
            hp[hq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[3], 1.0f, z4, 4, hq, 0);   // combo 48 + hq
        });
        hb_for<8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            hb_for<6>([&](auto hc) {
                constexpr int hq = decltype(hc)::value;
                constexpr int cb = 6 * i + hq;
                hp[hq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[cb / 16], c[i], hp[hq], 4, cb % 16, 0);
            });
        });
        } else {
#pragma unroll
            for (int hq = 0; hq < 6; ++hq) hp[hq] = (f32x4){c[hq], c[(hq + 1) & 7], c[(hq + 2) & 7], wA[hq & 3]};
        }
        // quad transposes: T[x][m] at lane 4 b + n = x[channel 4 q + n] of row 4 b + m   (D_b[m][n] += A_b[m] onehot_k[n], A = channel 4 q + k)
        f32x4 Tc[2];
        if constexpr ((HB_LAB & 4) == 0) {
        Tc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(c[0], oh[0], z4, 0, 0, 0);
        Tc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(c[4], oh[0], z4, 0, 0, 0);
        hb_for<3>([&](auto kc) {
            constexpr int k = decltype(kc)::value + 1;
            Tc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(c[k], oh[k], Tc[0], 0, 0, 0);
            Tc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(c[4 + k], oh[k], Tc[1], 0, 0, 0);
        });
        } else {
            Tc[0] = (f32x4){c[0], c[1], c[2], c[3]};
            Tc[1] = (f32x4){c[4], c[5], c[6], c[7]};
        }
        float u[24];
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            const float hv = hp[j >> 2][j & 3];
            if constexpr ((HB_LAB & 16) == 0) {
                u[j] = hv > 0.0f ? gz : 0.0f;
                gw2[j] = fmaf(u[j], hv, gw2[j]);
                gb1[j] += u[j];
            } else {
                u[j] = hv;
            }
        }
        // gC = (W1 diag(w2))^T u  (2 output quads, hidden units ascending)
        f32x4 gcq[2];
        if constexpr ((HB_LAB & 2) == 0) {
            // HB_GC_CHAINS independent chains per quad (chain ch takes the hidden units jj = ch mod HB_GC_CHAINS), folded in chain order
            f32x4 gch[2][HB_GC_CHAINS];
            hb_for<24>([&](auto jc) {
                constexpr int jj = decltype(jc)::value;
                constexpr int ch = jj % HB_GC_CHAINS;
                hb_for<2>([&](auto qc) {
                    constexpr int qq = decltype(qc)::value;
                    constexpr int cb = 2 * jj + qq;
                    if constexpr (jj < HB_GC_CHAINS) gch[qq][ch] = __builtin_amdgcn_mfma_f32_4x4x1f32(wB[cb / 16], u[jj], z4, 4, cb % 16, 0);
                    else gch[qq][ch] = __builtin_amdgcn_mfma_f32_4x4x1f32(wB[cb / 16], u[jj], gch[qq][ch], 4, cb % 16, 0);
                });
            });
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                gcq[qq] = gch[qq][0];
#pragma unroll
                for (int ch = 1; ch < HB_GC_CHAINS; ++ch) gcq[qq] += gch[qq][ch];
            }
        } else {
            gcq[0] = (f32x4){u[0], u[1], u[2], u[3]};
            gcq[1] = (f32x4){u[4], u[5], u[6], u[7]};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = gcq[i >> 2][i & 3];
        if constexpr ((HB_LAB & 64) == 0) {
            if (live) storeg(row, o);
        }
        f32x4 Tu[6];
        if constexpr ((HB_LAB & 4) == 0) {
        hb_for<6>([&](auto jc) {
            constexpr int jq = decltype(jc)::value;
            Tu[jq] = __builtin_amdgcn_mfma_f32_4x4x1f32(u[4 * jq], oh[0], z4, 0, 0, 0);
        });
        hb_for<3>([&](auto kc) {
            constexpr int k = decltype(kc)::value + 1;
            hb_for<6>([&](auto jc) {
                constexpr int jq = decltype(jc)::value;
                Tu[jq] = __builtin_amdgcn_mfma_f32_4x4x1f32(u[4 * jq + k], oh[k], Tu[jq], 0, 0, 0);
            });
        });
        } else {
#pragma unroll
            for (int jq = 0; jq < 6; ++jq) Tu[jq] = (f32x4){u[4 * jq], u[4 * jq + 1], u[4 * jq + 2], u[4 * jq + 3]};
        }
        // X^T G: one row per block and instruction
        if constexpr ((HB_LAB & 8) == 0) {
        hb_for<4>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            hb_for<6>([&](auto jc) {
                constexpr int jq = decltype(jc)::value;
                acc[jq][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(Tu[jq][m], Tc[0][m], acc[jq][0], 0, 0, 0);
                acc[jq][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(Tu[jq][m], Tc[1][m], acc[jq][1], 0, 0, 0);
            });
        });
        } else {
#pragma unroll
            for (int jq = 0; jq < 6; ++jq) { acc[jq][0] += Tu[jq]; acc[jq][1] += Tc[jq & 1]; }
        }
    };
    // (lab, HB_LAB & 64: the gC rows of a tile stored at the top of the NEXT iteration, behind that iteration's loads, so that no wait
    // for a tile's inputs includes a younger store - measured slower: fp32 65.2 against 59.0 us, bf16 57.3 against 56.7)
    float og[8];
    int64_t orow = A.n;                  // the row og belongs to (>= n: nothing pending)
    for (; t < tiles; t += stride) {
        float c[8];
        unpack(cb, c);
        const float pp = pb, tt = tb;
        {
            const int64_t row = (t + stride) * 64 + lane;
            const int64_t rc = row < last ? row : last;
            cb = loadc(rc); pb = A.p[rc]; tb = A.target[rc * A.target_ld];
        }
        if constexpr ((HB_LAB & 64) != 0) {
            if (orow < A.n) storeg(orow, og);
        }
        if constexpr ((HB_LAB & 32) == 0) __builtin_amdgcn_sched_barrier(0);
        tile(t, c, pp, tt, og);
        orow = t * 64 + lane;
    }
    if constexpr ((HB_LAB & 64) != 0) {
        if (orow < A.n) storeg(orow, og);
    }
    // ---- fold: blocks of lanes and waves in a fixed order --------------------------------------------------------------------------
    float* mine = lds + wave * 49 * HB_LSTR;
    float* part = lds + HB_WAVES * 49 * HB_LSTR;
    // (1) gW1': register r = 8 jq + 4 iq + m of every lane
#pragma unroll
    for (int jq = 0; jq < 6; ++jq)
#pragma unroll
        for (int iq = 0; iq < 2; ++iq)
#pragma unroll
            for (int m = 0; m < 4; ++m) mine[(8 * jq + 4 * iq + m) * HB_LSTR + lane] = acc[jq][iq][m];
    __syncthreads();
    if (threadIdx.x < 192) {
        const int r = threadIdx.x >> 2, nn = threadIdx.x & 3;                 // j = 4 jq + m, i = 4 iq + n
        float s = 0.0f;
        for (int w = 0; w < HB_WAVES; ++w) {
            const float* src = lds + (w * 49 + r) * HB_LSTR + nn;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
            for (int b = 0; b < 16; b += 4) { a0 += src[4 * b]; a1 += src[4 * b + 4]; a2 += src[4 * b + 8]; a3 += src[4 * b + 12]; }
            s += (a0 + a1) + (a2 + a3);
        }
        const int jq = r >> 3, iq = (r >> 2) & 1, m = r & 3;
        const int j = 4 * jq + m, i = 4 * iq + nn;
        A.dst[A.off_w1 + j * 8 + i] = s * A.w2[j];
    }
    __syncthreads();
    // (2) the per-lane sums: gw2 (24), gb1' (24), gb2
#pragma unroll
    for (int j = 0; j < 24; ++j) { mine[j * HB_LSTR + lane] = gw2[j]; mine[(24 + j) * HB_LSTR + lane] = gb1[j]; }
    mine[48 * HB_LSTR + lane] = gz_sum;
    __syncthreads();
    if (threadIdx.x < 4 * 49) {
        const int w = threadIdx.x / 49, r = threadIdx.x % 49;
        const float* src = lds + (w * 49 + r) * HB_LSTR;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
        for (int l = 0; l < 64; l += 4) { a0 += src[l]; a1 += src[l + 1]; a2 += src[l + 2]; a3 += src[l + 3]; }
        part[w * 49 + r] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (threadIdx.x < 49) {
        const int r = threadIdx.x;
        const float s = ((part[r] + part[49 + r]) + part[98 + r]) + part[147 + r];
        if (r < 24) A.dst[A.off_w2 + r] = s;
        else if (r < 48) A.dst[A.off_b1 + (r - 24)] = s * A.w2[r - 24];
        else A.dst[A.off_b2] = s;
    }
}

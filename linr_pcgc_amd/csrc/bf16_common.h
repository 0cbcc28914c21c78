// bf16 device helpers and the forward convolution kernel shared by the bf16 / uint8-weight inference executor (csrc/net_bf16.hip)
// and the bf16 training executor (csrc/train_bf16.hip).  Not part of the C-ABI.
#pragma once
#include "common.h"
#include "layout.h"
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

__device__ __forceinline__ bf16_t f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }     // RNE (v_cvt_pk_bf16_f32)
__device__ __forceinline__ float bf2f(bf16_t b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// torch: recon = q / sym_max * ten_range + min_n, each step ONE fp32 rounding: hipcc contracts a * b + c into an fma by
// default (-ffp-contract=fast-honor-pragmas), which changes the last bit where the sum cancels, so contraction is switched
// off here; the division is IEEE (correctly rounded is hipcc's default for fp32 divide)
__device__ __forceinline__ float dequant_code(uint8_t code, float range, float minv) {
#pragma clang fp contract(off)
    const float t = (float)code / 255.0f;
    const float u = t * range;
    return u + minv;
}
__device__ __forceinline__ float dequant(const uint8_t* __restrict__ codes, int64_t i, float range, float minv) {
    return dequant_code(codes[i], range, minv);
}

[[maybe_unused]] static __global__ __launch_bounds__(LINR_BLOCK) void dequant_all_k(const uint8_t* __restrict__ codes, int64_t n, float range, float minv,
                                                            float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i < n) out[i] = dequant(codes, i, range, minv);
}

__device__ __forceinline__ void unpack_row(const uint4 r, float (&x)[8]) {
    x[0] = bf2f((bf16_t)(r.x & 0xffff)); x[1] = bf2f((bf16_t)(r.x >> 16));
    x[2] = bf2f((bf16_t)(r.y & 0xffff)); x[3] = bf2f((bf16_t)(r.y >> 16));
    x[4] = bf2f((bf16_t)(r.z & 0xffff)); x[5] = bf2f((bf16_t)(r.z >> 16));
    x[6] = bf2f((bf16_t)(r.w & 0xffff)); x[7] = bf2f((bf16_t)(r.w >> 16));
}
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }
__device__ __forceinline__ uint4 pack_row(const float (&x)[8]) {
    return make_uint4(pack2(x[0], x[1]), pack2(x[2], x[3]), pack2(x[4], x[5]), pack2(x[6], x[7]));
}

// byte offsets (from the pad row) of the 27 neighbours of `row` for 16-byte rows; absent -> 0 (the pad row itself)
// (branch-free, the arithmetic of conv_common.h's decode_offsets: per column L = (lo + 1) << 4, a tap's offset is L advanced by one row per
// present tap below it and zeroed when absent - m_j = -bit_j (v_bfe_i32), o_j = t_j & m_j, t_{j+1} = t_j - 16 m_j: 9 VALU operations per column;
// the ten index words through 32-bit byte offsets from uniform bases (saddr-form loads) while 9 * ld * 4 < 2^32)
__device__ __forceinline__ void decode_words16(const uint32_t (&raw)[10], uint32_t (&off)[27]) {
    const uint32_t m = raw[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const uint32_t L = (raw[q] + 1u) << 4;
        const int m0 = __builtin_amdgcn_sbfe(m, 3 * q, 1), m1 = __builtin_amdgcn_sbfe(m, 3 * q + 1, 1), m2 = __builtin_amdgcn_sbfe(m, 3 * q + 2, 1);
        const uint32_t t1 = L + (uint32_t)__mul24(m0, -16);
        const uint32_t t2 = t1 + (uint32_t)__mul24(m1, -16);
        off[q] = L & (uint32_t)m0; off[q + 9] = t1 & (uint32_t)m1; off[q + 18] = t2 & (uint32_t)m2;
    }
}
__device__ __forceinline__ void load_words16(const int32_t* __restrict__ lo, const uint32_t* __restrict__ mask, int64_t ld, int64_t row,
                                             uint32_t (&raw)[10]) {
    const uint32_t rb = (uint32_t)row << 2;
    const char* lob = reinterpret_cast<const char*>(lo);
    const uint32_t ld4 = (uint32_t)ld << 2;
    const bool small = ld < ((int64_t)1 << 26);            // wave-uniform
    raw[9] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(mask) + rb);
#pragma unroll
    for (int q = 0; q < 9; ++q)
        raw[q] = small ? *reinterpret_cast<const uint32_t*>(lob + (rb + (uint32_t)q * ld4)) : (uint32_t)(lo + (int64_t)q * ld)[row];
}
__device__ __forceinline__ void decode_offsets16(const int32_t* __restrict__ lo, const uint32_t* __restrict__ mask, int64_t ld,
                                                 int64_t row, uint32_t (&off)[27]) {
    uint32_t raw[10];
    load_words16(lo, mask, ld, row, raw);
    decode_words16(raw, off);
}

#include <utility>
#include <type_traits>
template <class F, int... Ks>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Ks...>) { (f(std::integral_constant<int, Ks>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

#define BMAXG 8
// MODE 0: conv3 cin->8 (+bias) (+res) (ReLU)            -> out bf16 [n][8]
// MODE 1: prune conv 8->8 + head MLP + sigmoid (+ BCE)   -> p fp32 [n]                (upsample.py:137-161)
// MODE 2: conv0_0 (8->4) + conv1_0 (1x1 8->4), both ReLU -> H bf16 [n][8]             (resnet.py:55-57)
// MODE 3: conv0_1 on H[:,0:4], conv1_1 on H[:,4:8] + ReLU, conv1_2 (1x1) and the residual x -> I bf16 [n][8]
struct BArgs {
    const bf16_t* in; bf16_t* out; const bf16_t* res;
    const int32_t* lo; const uint32_t* mask; int64_t ld, n;
    const uint8_t* codes; float minv, range; const float* pf;      // uint8 codes (conv kernels) and the fp32 de-quantised copy
    int relu;
    int64_t g_in[BMAXG], g_out[BMAXG], g_res[BMAXG];               // element offsets per group (gridDim.y)
    int64_t w[BMAXG], b[BMAXG];                                    // main kernel / bias (MODE 3: conv0_1)
    int64_t w2[BMAXG], b2[BMAXG];                                  // MODE 2: conv1_0;  MODE 3: conv1_1
    int64_t w3[BMAXG], b3[BMAXG];                                  // MODE 3: conv1_2
    int cin[BMAXG];                                                // MODE 0: valid input channels (kernel is [27][cin][8])
    int64_t h_w1[BMAXG], h_b1[BMAXG], h_w2[BMAXG], h_b2[BMAXG];    // MODE 1: head MLP
    const float* target; int target_ld; int64_t t_col[BMAXG];      // occupancy column (fp32) or NULL
    float* p_out; int64_t p_off[BMAXG];
    double* partial; int64_t part_off[BMAXG];
    bf16_t* m_out; int64_t g_m[BMAXG];                             // TRAIN, MODE 3: M = relu(conv1_1) bf16 [n][4], kept for the backward pass
    const uint2* wimg; int wi[BMAXG];                              // SRC 2: pre-packed weight-block images [image][64 lanes], first image per group
};

// SRC 0: the model is the uint8 code vector a.codes (de-quantised here), a.pf its de-quantised fp32 copy;  SRC 1: a.pf = the fp32
// master parameters, the 3x3x3 kernels are rounded to bf16 here (training);  SRC 2: the same blocks, packed once per step by
// csrc/train_bf16.hip's tpack_k (the lane's registers are then NG coalesced 8-byte loads instead of 4 NG scattered loads + conversions
// in front of every 256-row workgroup).  TRAIN: everything the backward pass needs is stored
// (MODE 3: M; MODE 1: the prune conv's output C, rounded to bf16 like every stored activation, and the head reads the STORED value -
// the inference executor feeds the unrounded accumulators to the head).
template <int MODE, int SRC = 0, bool TRAIN = false>
__global__ __launch_bounds__(LINR_BLOCK) void bconv_k(BArgs a) {
    constexpr int CPT = (MODE == 0 || MODE == 1) ? 4 : 2;        // MFMAs (weight blocks) per tap
    constexpr int NG = (27 * CPT + 15) / 16;                      // register pairs holding them
    constexpr int NACC = (MODE == 2) ? 1 : 2;
    const int gi = blockIdx.y;
    const bf16_t* in = a.in + a.g_in[gi];
    const int lane = threadIdx.x & 63;
    // ---- weight blocks: block (lane >> 2) of wv[g] is combo c = 16 g + block; lane i = lane & 3 holds A[i][0..3] ----------
    //   MODE 0/1: c = 4 k + 2 h + q  -> W[k][4q + kk][4h + i]
    //   MODE 2  : c = 2 k + q        -> W00[k][4q + kk][i]
    //   MODE 3  : c = 2 k + t        -> t = 0: W01[k][kk][i], t = 1: W11[k][kk][i]
    s16x4 wv[NG];
    if constexpr (SRC == 2) {
        const uint2* wp = a.wimg + (int64_t)a.wi[gi] * 64 + lane;
#pragma unroll
        for (int g = 0; g < NG; ++g) wv[g] = __builtin_bit_cast(s16x4, wp[g * 64]);
    } else {
        // all of the lane's code bytes first (unconditional loads from clamped, always valid indices: in flight together), then the
        // de-quantisation - with the loads under `if (k < 27)` / `if (ci < cinv)` every byte was a load-and-wait of its own, up to 28
        // round trips in front of the first tap of every 256-row workgroup
        const int blk = lane >> 2, i = lane & 3;
        const int cinv = (MODE == 0) ? a.cin[gi] : 8;
        typename std::conditional<SRC == 0, uint8_t, float>::type raw[NG][4];
        bool ok[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int c = 16 * g + blk;
            const int k0 = c / CPT, k = k0 < 27 ? k0 : 26;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                int64_t idx;
                bool valid = k0 < 27;
                if constexpr (MODE == 0 || MODE == 1) {
                    const int h = (c >> 1) & 1, q = c & 1, ci0 = 4 * q + kk;
                    valid = valid && ci0 < cinv;
                    const int ci = ci0 < cinv ? ci0 : 0;
                    idx = a.w[gi] + ((int64_t)k * cinv + ci) * 8 + 4 * h + i;
                } else if constexpr (MODE == 2) {
                    const int q = c & 1;
                    idx = a.w[gi] + ((int64_t)k * 8 + 4 * q + kk) * 4 + i;
                } else {
                    const int t = c & 1;
                    idx = (t ? a.w2[gi] : a.w[gi]) + ((int64_t)k * 4 + kk) * 4 + i;
                }
                if constexpr (SRC == 0) raw[g][kk] = a.codes[idx];
                else raw[g][kk] = a.pf[idx];
                ok[g][kk] = valid;
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            s16x4 v;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float wf;
                if constexpr (SRC == 0) wf = dequant_code(raw[g][kk], a.range, a.minv);
                else wf = raw[g][kk];
                v[kk] = ok[g][kk] ? (short)f2bf(wf) : (short)0;
            }
            wv[g] = v;
        }
    }
    const int64_t row_raw = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    const bool live = row_raw < a.n;
    const int64_t row = live ? row_raw : a.n - 1;              // every lane stays in the MFMAs (they ignore EXEC)
    const char* pad = reinterpret_cast<const char*>(in - 8);
    uint32_t off[27];
    decode_offsets16(a.lo, a.mask, a.ld, row, off);
    f32x4 acc[NACC];
    {
        const float* b0 = a.pf + a.b[gi];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] = b0[j];
        if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[1][j] = b0[4 + j];
        }
        if constexpr (MODE == 3) {
            const float* b1 = a.pf + a.b2[gi];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[1][j] = b1[j];
        }
    }
#ifndef BCONV_PF
#define BCONV_PF 4
#endif
    constexpr int PF = BCONV_PF;
    uint4 x[PF + 1];
#pragma unroll
    for (int u = 0; u < PF; ++u) x[u] = *reinterpret_cast<const uint4*>(pad + off[LINR_TAP(u)]);
    __builtin_amdgcn_sched_barrier(0);
    sfor<27>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;              // step; k = the tap it handles (common.h: LINR_TAP, the fp32 family's order)
        constexpr int k = LINR_TAP(kk);
        if constexpr (kk + PF < 27) x[(kk + PF) % (PF + 1)] = *reinterpret_cast<const uint4*>(pad + off[LINR_TAP(kk + PF)]);
        __builtin_amdgcn_sched_barrier(0);
        const uint4 r = x[kk % (PF + 1)];
        const s16x4 q0 = __builtin_bit_cast(s16x4, make_uint2(r.x, r.y));
        const s16x4 q1 = __builtin_bit_cast(s16x4, make_uint2(r.z, r.w));
        if constexpr (MODE == 0 || MODE == 1) {
            constexpr int c0 = 4 * k;
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[c0 / 16], q0, acc[0], 4, c0 % 16, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[(c0 + 2) / 16], q0, acc[1], 4, (c0 + 2) % 16, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[(c0 + 1) / 16], q1, acc[0], 4, (c0 + 1) % 16, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[(c0 + 3) / 16], q1, acc[1], 4, (c0 + 3) % 16, 0);
        } else if constexpr (MODE == 2) {
            constexpr int c0 = 2 * k;
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[c0 / 16], q0, acc[0], 4, c0 % 16, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[(c0 + 1) / 16], q1, acc[0], 4, (c0 + 1) % 16, 0);
        } else {
            constexpr int c0 = 2 * k;
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[c0 / 16], q0, acc[0], 4, c0 % 16, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv[(c0 + 1) / 16], q1, acc[1], 4, (c0 + 1) % 16, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (MODE == 1) {
        // ---- occupancy head on the fp32 accumulators (the conv output row is never rounded to bf16) --------------------------
        const float* w1 = a.pf + a.h_w1[gi];
        const float* b1 = a.pf + a.h_b1[gi];
        const float* w2 = a.pf + a.h_w2[gi];
        float c[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = acc[0][j]; c[4 + j] = acc[1][j]; }
        if constexpr (TRAIN) {
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = bf2f(f2bf(c[j]));
        }
        // (every store of this kernel comes BEHIND the MLP: with a store to memory the compiler cannot tell apart from the parameters
        // in front of them, the 265 uniform weight loads below become per-lane vector loads - 54 extra 64-lane loads per tile, the cost
        // of two convolution passes - instead of scalar loads)
        // hidden layer on the matrix cores (round 6; before: 192 v_fma per lane with scalar weights, as long as the convolution itself):
        // v_mfma_f32_4x4x1 with the weight 4-vector broadcast (CBSZ = 4), K = 1 - every instruction IS one fmaf per output, in the
        // order of the loop it replaces (bias first, inputs ascending), so the probabilities keep their bits (stream format unchanged).
        //   wA: combo = 6 i + hq -> W1[4 hq + j][i] (i < 8), combos 48..53 -> b1[4 (combo - 48) + j]   (the fp32 executor's head images)
        float wA[4];
        {
            const int blk = lane >> 2, j4 = lane & 3;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int cb = 16 * v + blk;
                wA[v] = cb < 48 ? w1[(4 * (cb % 6) + j4) * 8 + cb / 6] : (cb < 54 ? b1[4 * (cb - 48) + j4] : 0.0f);
            }
        }
        f32x4 hp[6];
        sfor<6>([&](auto hc) {
            constexpr int hq = decltype(hc)::value;
            hp[hq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[3], 1.0f, (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, 4, hq, 0);       // combo 48 + hq
        });
        sfor<8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            sfor<6>([&](auto hc) {
                constexpr int hq = decltype(hc)::value;
                constexpr int cb = 6 * i + hq;
                hp[hq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[cb / 16], c[i], hp[hq], 4, cb % 16, 0);
            });
        });
        float z = a.pf[a.h_b2[gi]];
#pragma unroll
        for (int j = 0; j < 24; ++j) z = fmaf(fmaxf(hp[j >> 2][j & 3], 0.0f), w2[j], z);
        const float p = 1.0f / (1.0f + expf(-z));
        float t = 0.0f;
        if (a.partial != nullptr && live) t = a.target[a.t_col[gi] + row * a.target_ld];
        if constexpr (TRAIN) {
            if (live) *reinterpret_cast<uint4*>(a.out + a.g_out[gi] + row * 8) = pack_row(c);
        }
        if (live) a.p_out[a.p_off[gi] + row] = p;
        if (a.partial != nullptr) {
            __shared__ double sred[LINR_BLOCK / 64];
            double nats = 0.0;
            if (live) nats = (double)((t - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - t * fmaxf(logf(p), -100.0f));
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) nats += __shfl_xor(nats, d, 64);
            if (lane == 0) sred[threadIdx.x >> 6] = nats;
            __syncthreads();
            if (threadIdx.x == 0) {
                double tot = sred[0];
                for (int w = 1; w < LINR_BLOCK / 64; ++w) tot += sred[w];
                a.partial[a.part_off[gi] + blockIdx.x] = tot;
            }
        }
        return;
    } else {
        if (!live) return;
        float o[8];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = acc[0][j]; o[4 + j] = acc[1][j]; }
            if (a.res != nullptr) {
                float r[8];
                unpack_row(*reinterpret_cast<const uint4*>(a.res + a.g_res[gi] + row * 8), r);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += r[j];
            }
            if (a.relu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.0f);
            }
        } else if constexpr (MODE == 2) {
            float xs[8];
            unpack_row(*reinterpret_cast<const uint4*>(in + row * 8), xs);
            const float* w10 = a.pf + a.w2[gi];
            const float* b10 = a.pf + a.b2[gi];
            float h1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) h1[j] = b10[j];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) h1[j] = fmaf(xs[i], w10[i * 4 + j], h1[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = fmaxf(acc[0][j], 0.0f); o[4 + j] = fmaxf(h1[j], 0.0f); }
        } else {
            float xr[8];
            unpack_row(*reinterpret_cast<const uint4*>(a.res + a.g_res[gi] + row * 8), xr);
            const float* w12 = a.pf + a.w3[gi];
            const float* b12 = a.pf + a.b3[gi];
            // M = relu(conv1_1) is rounded to bf16 like every stored activation would be, so that a future split of this
            // kernel (M in memory) cannot change the bits
            float m[4], i1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = bf2f(f2bf(fmaxf(acc[1][j], 0.0f)));
#pragma unroll
            for (int j = 0; j < 4; ++j) i1[j] = b12[j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) i1[j] = fmaf(m[i], w12[i * 4 + j], i1[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = acc[0][j] + xr[j]; o[4 + j] = i1[j] + xr[4 + j]; }
            if constexpr (TRAIN)
                *reinterpret_cast<uint2*>(a.m_out + a.g_m[gi] + row * 4) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
        }
        *reinterpret_cast<uint4*>(a.out + a.g_out[gi] + row * 8) = pack_row(o);
    }
}


// occupancy fp32 [n][8] -> bf16 [n][8] (exact: 0 / 1)
[[maybe_unused]] static __global__ __launch_bounds__(LINR_BLOCK) void occ_bf16_k(const float* __restrict__ occ, int64_t n, bf16_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (r >= n) return;
    const float4 a = *reinterpret_cast<const float4*>(occ + r * 8);
    const float4 b = *reinterpret_cast<const float4*>(occ + r * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    *reinterpret_cast<uint4*>(out + r * 8) = pack_row(v);
}

// clears the pad rows (row -1) of the arena's bf16 matrices: `off` in bf16 elements from `base`, `w` elements each
#define BPADS_MAX 128
struct BPads { int64_t off[BPADS_MAX]; int w[BPADS_MAX]; int n; };
[[maybe_unused]] static __global__ void zero_pads16_k(bf16_t* __restrict__ base, BPads pl) {
    if ((int)blockIdx.x < pl.n && (int)threadIdx.x < pl.w[blockIdx.x]) base[pl.off[blockIdx.x] + threadIdx.x] = 0;
}

// Scale-context forward kernel (models/model_core.py:48-53) shared by the fp32 executor (csrc/net.hip) and the bf16 training
// executor (csrc/train_bf16.hip).  Not part of the C-ABI.
#pragma once
#include "common.h"
#include "layout.h"

struct PadList { int64_t off[200]; int w[200]; int n; };

#include <utility>
#include <type_traits>
template <class F, int... Ks>
__device__ __forceinline__ void sce_for_impl(F&& f, std::integer_sequence<int, Ks...>) { (f(std::integral_constant<int, Ks>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sce_for(F&& f) { sce_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Scale context of all scales in one launch (model_core.py:48-53): x0[r] = W2 relu(W1 [emb | offset_feat[r]] + b1) + b2 with
// the weights of r's scale: fmaf chains with the bias first and the inputs ascending, like linear_k<15,16> + linear_k<16,8> on
// [emb | offset_feat | 0].  hid (optional: the op-level entry) is the hidden layer; the executors pass NULL and sce_bwd_all_k recomputes it.
// A workgroup never straddles two scales (blk_off: first workgroup of every scale), so the scale - and with it every weight
// address - is uniform.  Round 2: the weights through the scalar cache into SGPRs instead of ~400 broadcast vector loads per row
// (46.8 -> ~17 us for the forward kernel at 337 k rows); round 6: as seven weight-image registers of broadcast 4x4x1 matrix
// instructions (see the kernel).
struct SceArgs {
    int64_t row_off[MAX_SCALES + 1];
    int64_t emb[MAX_SCALES], w1[MAX_SCALES], b1[MAX_SCALES], w2[MAX_SCALES], b2[MAX_SCALES];   // parameter offsets per scale
    int blk_off[MAX_SCALES + 1];
    int wg_off[MAX_SCALES + 1];          // sce_bwd_all_k: first workgroup of every scale (= its slab rows in front)
    int n_scales;
};

// scale of workgroup b and the row of this thread (-1: none)
__device__ __forceinline__ int64_t sce_row_of(const SceArgs& a, int b, int& s) {
    s = 0;
    for (int i = 1; i < a.n_scales; ++i) s += (b >= a.blk_off[i]) ? 1 : 0;
    const int64_t r = a.row_off[s] + (int64_t)(b - a.blk_off[s]) * LINR_BLOCK + threadIdx.x;
    return r < a.row_off[s + 1] ? r : -1;
}

// The blocks behind the last row block clear the arena's pad rows (PadList; one pad per 32 threads): the first kernel that reads
// a pad row comes after this one on the stream.
// XT = float: x0 fp32 [n][8];  XT = unsigned short: x0 rounded to bf16 [n][8] (the bf16 training executor; hid stays fp32)
template <class XT>
__global__ __launch_bounds__(LINR_BLOCK) void sce_fwd_k(const float* __restrict__ P, const float* __restrict__ off, SceArgs a,
                                                        int64_t n, float* __restrict__ mix, float* __restrict__ hid,
                                                        XT* __restrict__ x0, float* __restrict__ pad_base, PadList pl) {
    const int row_blocks = a.blk_off[a.n_scales];
    if ((int)blockIdx.x >= row_blocks) {
        if (pad_base == nullptr) return;
        const int b = ((int)blockIdx.x - row_blocks) * (LINR_BLOCK / 32) + (int)(threadIdx.x >> 5), t = threadIdx.x & 31;
        if (b < pl.n && t < pl.w[b]) pad_base[pl.off[b] + t] = 0.0f;
        return;
    }
    int s;
    const int64_t r_raw = sce_row_of(a, (int)blockIdx.x, s);
    const bool live = r_raw >= 0;
    const int64_t r = live ? r_raw : a.row_off[s + 1] - 1;     // every lane stays in the matrix instructions (they ignore EXEC)
    const float* emb = P + a.emb[s];
    const float* W1 = P + a.w1[s];
    const float* b1 = P + a.b1[s];
    const float* W2 = P + a.w2[s];
    const float* b2 = P + a.b2[s];
    // Both layers on v_mfma_f32_4x4x1 with the weight 4-vector broadcast (CBSZ = 4), K = 1: every instruction IS one fmaf per output in
    // the order of the loops it replaces (bias first, inputs ascending) - same bits - and the ~400 weights of a scale sit in SEVEN vector
    // registers per lane instead of coming through the scalar cache in ~25 load / wait rounds per wave (round 6: 17 -> see profiles/).
    //   wA: block (lane >> 2) of register v is combo 16 v + block; combo 4 i + hq -> W1[4 hq + j][i] (i < 15), 60 + hq -> b1[4 hq + j]
    //   wB: combo 2 i + q -> W2[4 q + j][i] (i < 16), 32 + q -> b2[4 q + j]
    float wA[4], wB[3];
    {
        const int lane = threadIdx.x & 63, blk = lane >> 2, j4 = lane & 3;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int c = 16 * v + blk;
            wA[v] = c < 60 ? W1[(4 * (c & 3) + j4) * 15 + (c >> 2)] : b1[4 * (c - 60) + j4];
        }
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int c = 16 * v + blk;
            wB[v] = c < 32 ? W2[(4 * (c & 1) + j4) * 16 + (c >> 1)] : (c < 34 ? b2[4 * (c - 32) + j4] : 0.0f);
        }
    }
    float x[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = emb[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[8 + i] = off[r * 7 + i];
    x[15] = 0.0f;
    if (mix && live) {                         // only the op-level entry wants the MLP input back (uniform)
        float4* mp = reinterpret_cast<float4*>(mix + r * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) mp[v] = make_float4(x[4 * v], x[4 * v + 1], x[4 * v + 2], x[4 * v + 3]);
    }
    typedef float sce_f32x4 __attribute__((ext_vector_type(4)));
    const sce_f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    sce_f32x4 hq4[4];
    sce_for<4>([&](auto hc) {
        constexpr int hq = decltype(hc)::value;
        hq4[hq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[3], 1.0f, z4, 4, 12 + hq, 0);          // combo 60 + hq
    });
    sce_for<15>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        sce_for<4>([&](auto hc) {
            constexpr int hq = decltype(hc)::value;
            constexpr int c = 4 * i + hq;
            hq4[hq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[c / 16], x[i], hq4[hq], 4, c % 16, 0);
        });
    });
    float h[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) h[o] = fmaxf(hq4[o >> 2][o & 3], 0.0f);
    if (live && hid != nullptr) {
        float4* hp = reinterpret_cast<float4*>(hid + r * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) hp[v] = make_float4(h[4 * v], h[4 * v + 1], h[4 * v + 2], h[4 * v + 3]);
    }
    sce_f32x4 yq[2];
    sce_for<2>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        yq[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(wB[2], 1.0f, z4, 4, q, 0);                   // combo 32 + q
    });
    sce_for<16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        sce_for<2>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int c = 2 * i + q;
            yq[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(wB[c / 16], h[i], yq[q], 4, c % 16, 0);
        });
    });
    if (!live) return;
    const float y[8] = {yq[0][0], yq[0][1], yq[0][2], yq[0][3], yq[1][0], yq[1][1], yq[1][2], yq[1][3]};
    if constexpr (sizeof(XT) == 4) {
        float4* yp = reinterpret_cast<float4*>(x0 + r * 8);
        yp[0] = make_float4(y[0], y[1], y[2], y[3]);
        yp[1] = make_float4(y[4], y[5], y[6], y[7]);
    } else {
        unsigned w[4];
#pragma unroll
        for (int v = 0; v < 4; ++v)
            w[v] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)y[2 * v]) |
                   ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)y[2 * v + 1]) << 16);
        *reinterpret_cast<uint4*>(x0 + r * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}


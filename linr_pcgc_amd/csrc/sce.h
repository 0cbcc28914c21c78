// Scale-context forward kernel (models/model_core.py:48-53) shared by the fp32 executor (csrc/net.hip) and the bf16 training
// executor (csrc/train_bf16.hip).  Not part of the C-ABI.
#pragma once
#include "common.h"
#include "layout.h"

struct PadList { int64_t off[200]; int w[200]; int n; };

// Scale context of all scales in one launch (model_core.py:48-53): x0[r] = W2 relu(W1 [emb | offset_feat[r]] + b1) + b2 with
// the weights of r's scale: fmaf chains with the bias first and the inputs ascending, like linear_k<15,16> + linear_k<16,8> on
// [emb | offset_feat | 0].  HID is kept for the backward pass (sce_bwd_all_k); the op-level entry also returns the MLP input.
// A workgroup never straddles two scales (blk_off: first workgroup of every scale), so the scale - and with it every weight
// address - is uniform: the weights come through the scalar cache into SGPRs (s_load + v_fmac with an SGPR operand) instead of
// ~400 broadcast vector loads per row (46.8 -> ~12 us for the forward kernel at 337 k rows, 18.9 -> ~9 for the backward one).
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
    const int64_t r = sce_row_of(a, (int)blockIdx.x, s);
    if (r < 0) return;
    const float* emb = P + a.emb[s];
    const float* W1 = P + a.w1[s];
    const float* b1 = P + a.b1[s];
    const float* W2 = P + a.w2[s];
    const float* b2 = P + a.b2[s];
    float x[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = emb[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[8 + i] = off[r * 7 + i];
    x[15] = 0.0f;
    if (mix) {                                 // only the op-level entry wants the MLP input back (uniform)
        float4* mp = reinterpret_cast<float4*>(mix + r * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) mp[v] = make_float4(x[4 * v], x[4 * v + 1], x[4 * v + 2], x[4 * v + 3]);
    }
    float h[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) h[o] = b1[o];
#pragma unroll
    for (int i = 0; i < 15; ++i)
#pragma unroll
        for (int o = 0; o < 16; ++o) h[o] = fmaf(x[i], W1[o * 15 + i], h[o]);
#pragma unroll
    for (int o = 0; o < 16; ++o) h[o] = fmaxf(h[o], 0.0f);
    float4* hp = reinterpret_cast<float4*>(hid + r * 16);
#pragma unroll
    for (int v = 0; v < 4; ++v) hp[v] = make_float4(h[4 * v], h[4 * v + 1], h[4 * v + 2], h[4 * v + 3]);
    float y[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) y[o] = b2[o];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int o = 0; o < 8; ++o) y[o] = fmaf(h[i], W2[o * 16 + i], y[o]);
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


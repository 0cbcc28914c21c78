// bf16 / uint8-weight inference executor (BASELINE config[4]: "bf16 SparseConv + int8 weight pack").
//
// The codec never runs the trained fp32 weights: encoder.py:101-103 codes the geometry with the DE-QUANTISED model
// (Model_Estimate.compress_model -> new_model), i.e. with w = q / 255 * (max - min) + min for the uint8 codes q of
// quant_uniform2 (model_compression/model_size_est.py:72-91) - exactly what the decoder rebuilds from model.bin.  This
// executor takes those codes as the model: 54,712 bytes instead of 219 KB, de-quantised inside the kernels.
//   * feature matrices are bf16 [1 + rows][8] (16 B per row: half the gather bytes of the fp32 path, one dwordx4 per tap),
//     accumulation is fp32;
//   * every 3x3x3 convolution runs on v_mfma_f32_4x4x4_16b_bf16 with CBSZ = 4: the A operand (a 4 cout x 4 cin weight block,
//     de-quantised from the uint8 codes and rounded to bf16 once per kernel, register-resident) is broadcast from block ABID
//     to all 16 blocks, B = four input channels of the lane's gathered row, D = four output channels of the lane's row:
//     64 rows x 4 cout x 4 cin per instruction - a quarter of the fp32 path's MFMAs;
//   * biases, the pointwise convolutions (conv1_0, conv1_2), the scale-context MLP and the head MLP stay fp32 VALU math on
//     the de-quantised fp32 parameters (one tiny prologue kernel writes them to the arena);
//   * inference only (encode / decode / codec): no activations are kept for a backward pass.
// Encoder (all 8 stages, grouped launches) and decoder (stage by stage) run the same kernels with the same per-row
// instruction sequence, so their probabilities are bit-identical and the 16-bit CDF quantisation cannot diverge.
#include "common.h"
#include "layout.h"
#include <math.h>

#define TRY(e) do { int rc_ = (e); if (rc_) return rc_; } while (0)

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

__global__ __launch_bounds__(LINR_BLOCK) void dequant_all_k(const uint8_t* __restrict__ codes, int64_t n, float range, float minv,
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
__device__ __forceinline__ void decode_offsets16(const int32_t* __restrict__ lo, const uint32_t* __restrict__ mask, int64_t ld,
                                                 int64_t row, uint32_t (&off)[27]) {
    const uint32_t m = mask[row];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const uint32_t base = (uint32_t)lo[(int64_t)q * ld + row] + 1u;
        const uint32_t b0 = (m >> (3 * q)) & 1u, b1 = (m >> (3 * q + 1)) & 1u, b2 = (m >> (3 * q + 2)) & 1u;
        off[q] = b0 ? base * 16u : 0u;
        off[q + 9] = b1 ? (base + b0) * 16u : 0u;
        off[q + 18] = b2 ? (base + b0 + b1) * 16u : 0u;
    }
}

#include <utility>
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
};

template <int MODE>
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
    {
        // all of the lane's code bytes first (unconditional loads from clamped, always valid indices: in flight together), then the
        // de-quantisation - with the loads under `if (k < 27)` / `if (ci < cinv)` every byte was a load-and-wait of its own, up to 28
        // round trips in front of the first tap of every 256-row workgroup
        const int blk = lane >> 2, i = lane & 3;
        const int cinv = (MODE == 0) ? a.cin[gi] : 8;
        uint8_t raw[NG][4];
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
                raw[g][kk] = a.codes[idx];
                ok[g][kk] = valid;
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            s16x4 v;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) v[kk] = ok[g][kk] ? (short)f2bf(dequant_code(raw[g][kk], a.range, a.minv)) : (short)0;
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
    constexpr int PF = 4;
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
        float z = a.pf[a.h_b2[gi]];
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            float hj = b1[j];
#pragma unroll
            for (int i = 0; i < 8; ++i) hj = fmaf(c[i], w1[j * 8 + i], hj);
            z = fmaf(fmaxf(hj, 0.0f), w2[j], z);
        }
        const float p = 1.0f / (1.0f + expf(-z));
        if (live) a.p_out[a.p_off[gi] + row] = p;
        if (a.partial != nullptr) {
            __shared__ double sred[LINR_BLOCK / 64];
            double nats = 0.0;
            if (live) {
                const float t = a.target[a.t_col[gi] + row * a.target_ld];
                nats = (double)((t - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - t * fmaxf(logf(p), -100.0f));
            }
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
        }
        *reinterpret_cast<uint4*>(a.out + a.g_out[gi] + row * 8) = pack_row(o);
    }
}

// ---- scale context (model_core.py:48-53) -> x0 bf16; occupancy fp32 -> bf16 copy with the pad row ----------------------------
struct BSce {
    int64_t row_off[MAX_SCALES + 1];
    int64_t emb[MAX_SCALES], w1[MAX_SCALES], b1[MAX_SCALES], w2[MAX_SCALES], b2[MAX_SCALES];
    int n_scales;
};
__global__ __launch_bounds__(LINR_BLOCK) void sce_bf16_k(const float* __restrict__ P, const float* __restrict__ off, BSce a, int64_t n,
                                                        bf16_t* __restrict__ x0) {
    const int64_t r = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (r >= n) return;
    int s = 0;
    for (int i = 1; i < a.n_scales; ++i) s += (r >= a.row_off[i]) ? 1 : 0;
    const float* emb = P + a.emb[s];
    const float* W1 = P + a.w1[s];
    const float* b1 = P + a.b1[s];
    const float* W2 = P + a.w2[s];
    const float* b2 = P + a.b2[s];
    float x[15];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = emb[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[8 + i] = off[r * 7 + i];
    float h[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) h[o] = b1[o];
#pragma unroll
    for (int i = 0; i < 15; ++i)
#pragma unroll
        for (int o = 0; o < 16; ++o) h[o] = fmaf(x[i], W1[o * 15 + i], h[o]);
    float y[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) y[o] = b2[o];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int o = 0; o < 8; ++o) y[o] = fmaf(fmaxf(h[i], 0.0f), W2[o * 16 + i], y[o]);
    *reinterpret_cast<uint4*>(x0 + r * 8) = pack_row(y);
}

__global__ __launch_bounds__(LINR_BLOCK) void occ_bf16_k(const float* __restrict__ occ, int64_t n, bf16_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (r >= n) return;
    const float4 a = *reinterpret_cast<const float4*>(occ + r * 8);
    const float4 b = *reinterpret_cast<const float4*>(occ + r * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    *reinterpret_cast<uint4*>(out + r * 8) = pack_row(v);
}

struct BPads { int64_t off[64]; int n; };
__global__ void zero_pads16_k(bf16_t* __restrict__ base, BPads pl) {
    if ((int)blockIdx.x < pl.n && threadIdx.x < 8) base[pl.off[blockIdx.x] + threadIdx.x] = 0;
}

// ---- arena --------------------------------------------------------------------------------------------------------------------
struct BArena {
    int64_t rows;
    char* base;
    int64_t cur;                     // bytes
    float* PF;                       // [n_params] de-quantised fp32 parameters
    double* part;                    // [8 * grid] block partials of the bits
    bf16_t *X0, *OCC, *A[8], *H[8], *I[8], *O[8], *Hx[MAX_BL - 1], *Ix[MAX_BL - 1];
    BPads pads;
    bf16_t* mats;                    // start of the bf16 matrices (pad offsets are relative to it)
};

static bf16_t* bmat(BArena& a) {
    bf16_t* p = a.base ? reinterpret_cast<bf16_t*>(a.base + a.cur) : nullptr;
    a.pads.off[a.pads.n++] = (a.cur - (int64_t)(reinterpret_cast<char*>(a.mats) - a.base)) / 2;
    a.cur += (a.rows + 1) * 16;
    return p ? p + 8 : nullptr;
}

static void make_barena(BArena& a, int64_t rows, char* base, int64_t n_params, int block_layers) {
    a.rows = rows; a.base = base; a.cur = 0; a.pads.n = 0;
    a.PF = reinterpret_cast<float*>(base);
    a.cur += ((n_params * 4 + 63) / 64) * 64;
    a.part = reinterpret_cast<double*>(base + a.cur);
    a.cur += (((int64_t)8 * linr_grid(rows, LINR_BLOCK) * 8 + 63) / 64) * 64;
    a.mats = reinterpret_cast<bf16_t*>(base + a.cur);
    a.X0 = bmat(a); a.OCC = bmat(a);
    for (int b = 0; b < 8; ++b) { a.A[b] = bmat(a); a.H[b] = bmat(a); a.I[b] = bmat(a); a.O[b] = bmat(a); }
    for (int l = 0; l + 1 < MAX_BL; ++l) {
        a.Hx[l] = a.Ix[l] = nullptr;
        if (l + 1 < block_layers) { a.Hx[l] = bmat(a); a.Ix[l] = bmat(a); }
    }
}

extern "C" size_t linr_net_bf16_arena_bytes(int64_t rows, int32_t block_layers) {
    if (rows < 0) return 0;
    if (block_layers < 1) block_layers = 1;
    if (block_layers > MAX_BL) return 0;
    Layout L;
    make_layout(L, MAX_SCALES, block_layers);
    BArena a;
    make_barena(a, rows, nullptr, L.total, block_layers);
    return (size_t)a.cur + 64;
}

// ---- executor ------------------------------------------------------------------------------------------------------------------
struct BCtx {
    const linr_frame* f;
    Layout L;
    BArena A;
    const uint8_t* codes;
    float minv, range;
    hipStream_t s;
    int64_t R;
};

static BArgs base_args(const BCtx& c) {
    BArgs a = BArgs();
    a.lo = c.f->nbr_lo; a.mask = c.f->nbr_mask; a.ld = c.f->nbr_ld; a.n = c.R;
    a.codes = c.codes; a.minv = c.minv; a.range = c.range; a.pf = c.A.PF;
    for (int g = 0; g < BMAXG; ++g) a.cin[g] = 8;
    return a;
}

template <int MODE>
static int blaunch(const BCtx& c, const BArgs& a, int groups) {
    linr_poison_hook(c.s, 14);
    bconv_k<MODE><<<dim3(linr_grid(c.R, LINR_BLOCK), groups), LINR_BLOCK, 0, c.s>>>(a);
    return linr_launch_rc();
}

// conv3 cin->8: groups share `in`/`out`/`res` base pointers through element offsets
static int bconv_plain(const BCtx& c, const bf16_t* in, bf16_t* out, const bf16_t* res, int relu, int groups, const int64_t* g_in,
                       const int64_t* g_out, const int64_t* g_res, const int64_t* w, const int64_t* b, const int* cin) {
    BArgs a = base_args(c);
    a.in = in; a.out = out; a.res = res; a.relu = relu;
    for (int g = 0; g < groups; ++g) {
        a.g_in[g] = g_in ? g_in[g] : 0; a.g_out[g] = g_out ? g_out[g] : 0; a.g_res[g] = g_res ? g_res[g] : 0;
        a.w[g] = w[g]; a.b[g] = b[g]; a.cin[g] = cin ? cin[g] : 8;
    }
    return blaunch<0>(c, a, groups);
}

// one Inception layer: X -> H -> I (two launches), per group
static int binception(const BCtx& c, const bf16_t* X, bf16_t* H, bf16_t* I, int groups, const int64_t* gX, const int64_t* gH,
                      const int64_t* gI, const IncP* const* q) {
    {
        BArgs a = base_args(c);
        a.in = X; a.out = H;
        for (int g = 0; g < groups; ++g) {
            a.g_in[g] = gX ? gX[g] : 0; a.g_out[g] = gH ? gH[g] : 0;
            a.w[g] = q[g]->c00_w; a.b[g] = q[g]->c00_b; a.w2[g] = q[g]->c10_w; a.b2[g] = q[g]->c10_b;
        }
        TRY(blaunch<2>(c, a, groups));
    }
    BArgs a = base_args(c);
    a.in = H; a.out = I; a.res = X;
    for (int g = 0; g < groups; ++g) {
        a.g_in[g] = gH ? gH[g] : 0; a.g_out[g] = gI ? gI[g] : 0; a.g_res[g] = gX ? gX[g] : 0;
        a.w[g] = q[g]->c01_w; a.b[g] = q[g]->c01_b; a.w2[g] = q[g]->c11_w; a.b2[g] = q[g]->c11_b;
        a.w3[g] = q[g]->c12_w; a.b3[g] = q[g]->c12_b;
    }
    return blaunch<3>(c, a, groups);
}

__global__ __launch_bounds__(LINR_BLOCK) void badd_rows_k(const bf16_t* __restrict__ src, int64_t n, bf16_t* __restrict__ dst) {
    const int64_t r = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (r >= n) return;
    float a[8], b[8];
    unpack_row(*reinterpret_cast<const uint4*>(src + r * 8), a);
    unpack_row(*reinterpret_cast<const uint4*>(dst + r * 8), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] += a[j];
    *reinterpret_cast<uint4*>(dst + r * 8) = pack_row(b);
}

// make_block for one block slot (upsample.py:88-97), single launch group
static int bblock(const BCtx& c, const BlockP& bp, const bf16_t* in, int slot, const bf16_t* res) {
    const BArena& a = c.A;
    const int cin = bp.cin;
    TRY(bconv_plain(c, in, a.A[slot], nullptr, 1, 1, nullptr, nullptr, nullptr, &bp.a_w, &bp.a_b, &cin));
    const bf16_t* X = a.A[slot];
    bf16_t* Il = nullptr;
    for (int l = 0; l < bp.nl; ++l) {
        bf16_t* H = l == 0 ? a.H[slot] : a.Hx[l - 1];
        bf16_t* I = l == 0 ? a.I[slot] : a.Ix[l - 1];
        const IncP* q = &bp.inc[l];
        TRY(binception(c, X, H, I, 1, nullptr, nullptr, nullptr, &q));
        X = I; Il = I;
    }
    if (bp.nl > 1) badd_rows_k<<<linr_grid(c.R, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(a.A[slot], c.R, Il);     // resnet.py:160-161
    return bconv_plain(c, Il, a.O[slot], res, 0, 1, nullptr, nullptr, nullptr, &bp.b_w, &bp.b_b, nullptr);
}

static int bheads(const BCtx& c, int k0, int k1, float* probs_stage_major, double* part) {
    const BArena& a = c.A;
    BArgs h = base_args(c);
    h.in = a.O[k0];
    h.target = part ? c.f->occ : nullptr; h.target_ld = 8;
    h.p_out = probs_stage_major; h.partial = part;
    const int64_t nblk = linr_grid(c.R, LINR_BLOCK);
    for (int k = k0; k < k1; ++k) {
        const int g = k - k0;
        h.g_in[g] = a.O[k] - a.O[k0];
        h.w[g] = c.L.pr_w[k]; h.b[g] = c.L.pr_b[k];
        h.h_w1[g] = c.L.h0_w[k]; h.h_b1[g] = c.L.h0_b[k]; h.h_w2[g] = c.L.h2_w[k]; h.h_b2[g] = c.L.h2_b[k];
        h.t_col[g] = k; h.p_off[g] = (int64_t)k * c.R; h.part_off[g] = (int64_t)k * nblk;
    }
    return blaunch<1>(c, h, k1 - k0);
}

extern "C" int linr_net_forward_bf16(const linr_frame* f, const uint8_t* codes, float min_param, float max_param, void* arena,
                                     size_t arena_bytes, int32_t stage_begin, int32_t stage_end, float* probs, double* bits_acc,
                                     void* stream) {
    if (!f || !codes || !arena || !probs) return LINR_EINVAL;
    if (f->rows < 0 || f->n_scales < 1 || f->n_scales > MAX_SCALES || !f->row_off_h || !f->scale_idx_h) return LINR_EINVAL;
    if (stage_begin < 0 || stage_end > 8 || stage_begin >= stage_end) return LINR_EINVAL;
    BCtx c;
    if (!make_layout(c.L, f->model_scale_num, f->block_layers < 1 ? 1 : f->block_layers)) return LINR_EINVAL;
    if (f->row_off_h[0] != 0 || f->row_off_h[f->n_scales] != f->rows) return LINR_EINVAL;
    for (int s = 0; s < f->n_scales; ++s) {
        if (f->row_off_h[s + 1] < f->row_off_h[s]) return LINR_EINVAL;
        if (f->scale_idx_h[s] < 0 || f->scale_idx_h[s] >= f->model_scale_num) return LINR_EINVAL;
    }
    if (f->rows == 0) return 0;
    if (!f->nbr_lo || !f->nbr_mask || !f->offset_feat || !f->occ || f->nbr_ld < f->rows) return LINR_EINVAL;   // compressed map only
    if (f->rows >= ((int64_t)1 << 27) - 1) return LINR_EINVAL;                    // 32-bit byte offsets of the 16-byte gathers
    if (arena_bytes < linr_net_bf16_arena_bytes(f->rows, c.L.BL)) return LINR_ENOSPC;
    if (((uintptr_t)arena) & 63u) return LINR_EALIGN;
    c.f = f; c.codes = codes; c.minv = min_param; c.range = max_param - min_param;      // fp32 subtraction, like ten_range
    c.s = (hipStream_t)stream; c.R = f->rows;
    make_barena(c.A, f->rows, (char*)arena, c.L.total, c.L.BL);
    const BArena& a = c.A;
    const Layout& L = c.L;
    occ_bf16_k<<<linr_grid(c.R, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(f->occ, c.R, a.OCC);
    if (stage_begin == 0) {
        dequant_all_k<<<linr_grid(L.total, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(codes, L.total, c.range, c.minv, a.PF);
        zero_pads16_k<<<a.pads.n, 64, 0, c.s>>>(a.mats, a.pads);
        BSce sa;
        sa.n_scales = f->n_scales;
        for (int s = 0; s < f->n_scales; ++s) {
            const int si = f->scale_idx_h[s];
            sa.row_off[s] = f->row_off_h[s];
            sa.emb[s] = L.emb + si * 8; sa.w1[s] = L.m0_w[si]; sa.b1[s] = L.m0_b[si]; sa.w2[s] = L.m2_w[si]; sa.b2[s] = L.m2_b[si];
        }
        sa.row_off[f->n_scales] = f->rows;
        linr_poison_hook(c.s, 14);
        sce_bf16_k<<<linr_grid(c.R, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(a.PF, f->offset_feat, sa, c.R, a.X0);
        TRY(bblock(c, L.block_in, a.X0, 0, nullptr));                                   // O[0] = x_glob
    }
    double* part = bits_acc ? a.part : nullptr;
    const int64_t nblk = linr_grid(c.R, LINR_BLOCK);
    if (stage_begin == 0 && stage_end == 8) {
        // teacher-forced: the 7 outter blocks and the 8 heads as grouped launches (same kernels, same per-row arithmetic as
        // the staged path below)
        int64_t gA[7], gH[7], gI[7], gO[7], zero7[7], aw[7], ab[7], bw[7], bb[7];
        int cin[7];
        const IncP* q[7];
        for (int g = 0; g < 7; ++g) {
            const BlockP& bp = L.outter[g];
            gA[g] = a.A[g + 1] - a.A[1]; gH[g] = a.H[g + 1] - a.H[1]; gI[g] = a.I[g + 1] - a.I[1]; gO[g] = a.O[g + 1] - a.O[1];
            zero7[g] = 0; aw[g] = bp.a_w; ab[g] = bp.a_b; bw[g] = bp.b_w; bb[g] = bp.b_b; cin[g] = g + 1; q[g] = &bp.inc[0];
        }
        TRY(bconv_plain(c, a.OCC, a.A[1], nullptr, 1, 7, zero7, gA, nullptr, aw, ab, cin));
        TRY(binception(c, a.A[1], a.H[1], a.I[1], 7, gA, gH, gI, q));
        TRY(bconv_plain(c, a.I[1], a.O[1], a.O[0], 0, 7, gI, gO, zero7, bw, bb, nullptr));
        TRY(bheads(c, 0, 8, probs, part));
    } else {
        for (int k = stage_begin; k < stage_end; ++k) {
            if (k > 0) TRY(bblock(c, L.outter[k - 1], a.OCC, k, a.O[0]));
            TRY(bheads(c, k, k + 1, probs, part));
        }
    }
    if (bits_acc)
        TRY(linr_bits_finish_launch(a.part + (int64_t)stage_begin * nblk, (int)((stage_end - stage_begin) * nblk), bits_acc, c.s));
    return linr_launch_rc();
}

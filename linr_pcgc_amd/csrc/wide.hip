// 3x3x3 convolutions wider than 8 channels (--hidden_channel_conv 16 / 32, main.py:520; models/upsample.py:38-76 `channels=`,
// models/resnet.py:12-51 `channels // 2`) on channel-BLOCKED activations: a C-wide feature matrix is C / 8 separate [rows + 1][8]
// matrices with the zero row in front (linr_pcgc_amd/wide_net.py), so every kernel of the 8-wide network still applies to a block.
//
// Round 3 ran a convolution Ci -> Co as (Ci / 8)(Co / 8) launches of the 8 -> 8 kernel, each gathering its input block again
// (profiles/r03_wide_speed.txt: 12.9 ms per step at width 16).  wconv_k gathers every input block of a row ONCE per tap and feeds all
// the output channels of up to two output blocks from it - the organisation of occ_conv7_k (csrc/fused.hip): lane = output row, the
// weights of a tap as an A-operand image in LDS (block b of VGPR v = combo 16 v + b = (gathered channel, output quad)),
// v_mfma_f32_4x4x1 with the weight 4-vector broadcast (CBSZ = 4), so a tap of a 16 -> 16 convolution is 4 gathers for 64 matrix
// instructions where the blocked form issued 8 gathers for them, and 32 -> 32 is 8 for 256 instead of 32 for 256.
// Per-output arithmetic: bias, then taps in LINR_TAP order, gathered channel ascending over ALL input blocks - one chain per output,
// the same in training, encoding and stage-by-stage decoding (wide_net.py runs the same launches in all three).
#include "common.h"
#include "conv_common.h"
#include <stdlib.h>

#ifndef LINR_CONV_BLOCK
#define LINR_CONV_BLOCK 256
#endif
#define WC_MAXB 4
struct WcArgs {
    const float* in[WC_MAXB];      // gathered blocks (fwd: the input, bwd: the output gradient), each with its zero row at [-8, 0)
    const float* res[2];           // per produced block or nullptr
    const float* act[2];           // LINR_RELU_MASK: mask by act > 0
    float* out[2];                 // produced blocks
    const float* W;                // [27][cin][cout] (ME layout)
    const float* bias;             // fwd: [cout] or nullptr
    int cin, cout;                 // of the convolution (fwd: gathered = cin, produced = cout; bwd: the other way round)
    int gvalid;                    // gathered channels that exist (first convolutions of the outter blocks: cin = k < 8)
    int pb0;                       // first produced block of this launch
    unsigned flags;
    // pointwise side path of a wide Inception layer fused into the epilogue (models/resnet.py:55-60; template parameter EPI):
    //   EPI 1 (fwd conv0_0 C -> h): out2 = relu(own input row @ W10 + b10)                         (conv1_0 reads the centre tap's row)
    //   EPI 2 (fwd conv1_1 h -> h): out2 = (this kernel's output row) @ W12 + b12 + aux            (conv1_2 + the residual's upper half)
    //   EPI 3 (bwd tail conv):      out2 = ((produced upper half) @ W12^T) * (aux > 0)             (backward of conv1_2 and of M's ReLU)
    //   EPI 4 (bwd conv0_0):        produced += (aux own row) @ W10^T, in front of the mask        (backward-data of conv1_0; aux = gH1)
    // pw_W: [cin_pw][cout_pw] (ME layout), pw_b [cout_pw] or nullptr; pw_on: this launch runs the side path (EPI 3: the launch that
    // produces the upper half, whose first local block is pw_hi0)
    const float* pw_W; const float* pw_b;
    const float* pw_aux[2];
    float* pw_out[2];
    int pw_on, pw_hi0;
};

template <int GB, int PB, bool BWD, int EPI = 0>
__global__ __launch_bounds__(LINR_CONV_BLOCK) void wconv_k(WcArgs a, const int32_t* __restrict__ lo, const uint32_t* __restrict__ mask,
                                                          int64_t ld, int64_t n) {
    constexpr int GIN = 8 * GB, NQ = 2 * PB, NV = GB * PB;          // NV = GIN * NQ / 16 A-operand registers per tap
    constexpr int PL = NV >= 4 ? NV / 4 : 1, PW = NV >= 4 ? 4 : NV;  // planes of PW registers: wl[k][plane][lane][PW]
    extern __shared__ float wl[];
    {   // the weight image: all of a thread's (scattered, L2-resident) loads in flight at once, then the LDS stores - as a rolled loop
        // every element waited for its own load: 27 round trips to the L2 in front of the first tile of a 16 -> 16 convolution
        constexpr int TOT = 27 * NV * 64, IT = (TOT + LINR_CONV_BLOCK - 1) / LINR_CONV_BLOCK;
        float wv[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int e = (int)threadIdx.x + LINR_CONV_BLOCK * i;
            const int w = e % PW, ln = (e / PW) % 64, pl = (e / (PW * 64)) % PL, k = e / (NV * 64);
            const int v = pl * PW + w, blk = ln >> 2, j = ln & 3;
            const int c = 16 * v + blk, gi = c / NQ, oq = c % NQ, po = 8 * a.pb0 + 4 * oq + j;
            float val = 0.0f;
            if (e < TOT && gi < a.gvalid) val = BWD ? a.W[((int64_t)k * a.cin + po) * a.cout + gi] : a.W[((int64_t)k * a.cin + gi) * a.cout + po];
            wv[i] = val;
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int e = (int)threadIdx.x + LINR_CONV_BLOCK * i;
            if (e < TOT) wl[e] = wv[i];
        }
    }
    // the side path's weights behind the image (read back with uniform addresses: LDS broadcasts)
    constexpr int PWI = EPI == 1 ? 8 * GB : EPI == 2 ? 8 * PB : EPI == 3 ? 4 * GB : EPI == 4 ? 16 * GB : 0;      // rows of pw_W (cin_pw)
    constexpr int PWO = EPI == 1 ? 8 * PB : EPI == 2 ? 8 * PB : EPI == 3 ? 4 * GB : EPI == 4 ? 8 * GB : 0;       // columns (cout_pw)
    float* wp = wl + 27 * NV * 64;
    // read through a pointer that is per-lane in form: with a uniform one the compiler keeps the weights in SGPRs and spills hundreds
    const float* wq = wp + __builtin_amdgcn_mbcnt_lo(0u, 0u);
    if constexpr (EPI != 0) {
        for (int e = threadIdx.x; e < PWI * PWO; e += LINR_CONV_BLOCK) wp[e] = a.pw_W[e];
        if (EPI <= 2) for (int e = threadIdx.x; e < PWO; e += LINR_CONV_BLOCK) wp[PWI * PWO + e] = a.pw_b ? a.pw_b[e] : 0.0f;
    }
    __syncthreads();
    // the produced channels' bias: once per workgroup (uniform loads), not once per tile
    float bz[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) bz[q][j] = (!BWD && a.bias) ? a.bias[8 * a.pb0 + 4 * q + j] : 0.0f;
    const int lane = threadIdx.x & 63;
    const char* pad[GB];
#pragma unroll
    for (int g = 0; g < GB; ++g) pad[g] = reinterpret_cast<const char*>(a.in[g] - 8);
    const float* wlane = wl + lane * PW;
    const int64_t tiles = (n + LINR_CONV_BLOCK - 1) / LINR_CONV_BLOCK;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row_raw = tile * LINR_CONV_BLOCK + threadIdx.x;
        const bool live = row_raw < n;
        const int64_t row = live ? row_raw : n - 1;          // every lane stays in the MFMAs (they ignore EXEC)
        uint32_t off[27];
        decode_offsets<BWD>(lo, mask, ld, row, 32u, off);
        f32x4 acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[q][j] = bz[q][j];
        constexpr int PF = GB >= 4 ? 1 : 2;
        f32x4 x[PF + 1][2 * GB];
        float wr[2][NV];
        auto gather = [&](int u, uint32_t o) {
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                x[u][2 * g] = *reinterpret_cast<const f32x4*>(pad[g] + o);
                x[u][2 * g + 1] = *reinterpret_cast<const f32x4*>(pad[g] + o + 16);
            }
        };
        auto wread = [&](int u, int k) {
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int w = 0; w < PW; ++w) wr[u][pl * PW + w] = wlane[(k * PL + pl) * 64 * PW + w];
        };
#pragma unroll
        for (int u = 0; u < PF; ++u) gather(u, off[LINR_TAP(u)]);
        wread(0, LINR_TAP(0));
        __builtin_amdgcn_sched_barrier(0);
        static_for<27>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;            // step; tap LINR_TAP(kk)
            if constexpr (kk + PF < 27) gather((kk + PF) % (PF + 1), off[LINR_TAP(kk + PF)]);
            if constexpr (kk + 1 < 27) wread((kk + 1) & 1, LINR_TAP(kk + 1));
            __builtin_amdgcn_sched_barrier(0);
            // gathered channel outermost: consecutive MFMAs write different accumulators, every output sees its channels ascending
            static_for<GIN>([&](auto gc) {
                constexpr int gi = decltype(gc)::value;
                static_for<NQ>([&](auto qc) {
                    constexpr int oq = decltype(qc)::value;
                    constexpr int c = gi * NQ + oq;
                    acc[oq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[kk & 1][c / 16], x[kk % (PF + 1)][gi / 4][gi % 4], acc[oq], 4, c % 16, 0);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        if (!live) continue;
        // epilogue order of the 8-wide kernels: + res, + old (LINR_ACCUM), [EPI 4: + the pointwise backward], * (act > 0)
        // (LINR_RELU_MASK), ReLU
        float o[PB][8];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) o[pb][4 * h + j] = acc[2 * pb + h][j];
            if (a.res[pb]) {
                const float* r = a.res[pb] + row * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[pb][j] += r[j];
            }
            if (a.flags & LINR_ACCUM) {
                const float* op = a.out[pb] + row * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[pb][j] += op[j];
            }
        }
        if constexpr (EPI == 4) {
            // + gH1[row] @ W10^T: per produced channel ci the chain of linr_linear_wide's backward-data pass (gradient channels
            // ascending from 0), added to what the convolution, the residual's share and the old content gave
            float gh[8 * GB];
#pragma unroll
            for (int b = 0; b < GB; ++b) {
                const float4 t0 = *reinterpret_cast<const float4*>(a.pw_aux[b] + row * 8), t1 = *reinterpret_cast<const float4*>(a.pw_aux[b] + row * 8 + 4);
                gh[8 * b] = t0.x; gh[8 * b + 1] = t0.y; gh[8 * b + 2] = t0.z; gh[8 * b + 3] = t0.w;
                gh[8 * b + 4] = t1.x; gh[8 * b + 5] = t1.y; gh[8 * b + 6] = t1.z; gh[8 * b + 7] = t1.w;
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float* wrow = wq + (8 * (a.pb0 + pb) + j) * PWO;
                    float t = 0.0f;
#pragma unroll
                    for (int co = 0; co < PWO; ++co) t = fmaf(gh[co], wrow[co], t);
                    o[pb][j] = t + o[pb][j];
                }
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            if (a.flags & LINR_RELU_MASK) {
                const float* m = a.act[pb] + row * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[pb][j] = m[j] > 0.0f ? o[pb][j] : 0.0f;
            }
            if (a.flags & LINR_RELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[pb][j] = fmaxf(o[pb][j], 0.0f);
            }
            float* op = a.out[pb] + row * 8;
            *reinterpret_cast<float4*>(op) = make_float4(o[pb][0], o[pb][1], o[pb][2], o[pb][3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(o[pb][4], o[pb][5], o[pb][6], o[pb][7]);
        }
        if constexpr (EPI == 1) {
            // conv1_0 on the row itself: bias, then the input channels ascending (linr_linear_wide's chain), ReLU
            float xin[8 * GB];
#pragma unroll
            for (int b = 0; b < GB; ++b) {
                const float4 t0 = *reinterpret_cast<const float4*>(a.in[b] + row * 8), t1 = *reinterpret_cast<const float4*>(a.in[b] + row * 8 + 4);
                xin[8 * b] = t0.x; xin[8 * b + 1] = t0.y; xin[8 * b + 2] = t0.z; xin[8 * b + 3] = t0.w;
                xin[8 * b + 4] = t1.x; xin[8 * b + 5] = t1.y; xin[8 * b + 6] = t1.z; xin[8 * b + 7] = t1.w;
            }
            float y[PWO];
#pragma unroll
            for (int j = 0; j < PWO; ++j) y[j] = wq[PWI * PWO + j];
#pragma unroll
            for (int i = 0; i < PWI; ++i)
#pragma unroll
                for (int j = 0; j < PWO; ++j) y[j] = fmaf(xin[i], wq[i * PWO + j], y[j]);
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                float* q = a.pw_out[b] + row * 8;
                *reinterpret_cast<float4*>(q) = make_float4(fmaxf(y[8 * b], 0.f), fmaxf(y[8 * b + 1], 0.f), fmaxf(y[8 * b + 2], 0.f), fmaxf(y[8 * b + 3], 0.f));
                *reinterpret_cast<float4*>(q + 4) = make_float4(fmaxf(y[8 * b + 4], 0.f), fmaxf(y[8 * b + 5], 0.f), fmaxf(y[8 * b + 6], 0.f), fmaxf(y[8 * b + 7], 0.f));
            }
        }
        if constexpr (EPI == 2) {
            // conv1_2 on this kernel's own output row M, + the residual's upper half
            float y[PWO];
#pragma unroll
            for (int j = 0; j < PWO; ++j) y[j] = wq[PWI * PWO + j];
#pragma unroll
            for (int i = 0; i < PWI; ++i)
#pragma unroll
                for (int j = 0; j < PWO; ++j) y[j] = fmaf(o[i / 8][i % 8], wq[i * PWO + j], y[j]);
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                const float* r = a.pw_aux[b] + row * 8;
                float* q = a.pw_out[b] + row * 8;
                *reinterpret_cast<float4*>(q) = make_float4(y[8 * b] + r[0], y[8 * b + 1] + r[1], y[8 * b + 2] + r[2], y[8 * b + 3] + r[3]);
                *reinterpret_cast<float4*>(q + 4) = make_float4(y[8 * b + 4] + r[4], y[8 * b + 5] + r[5], y[8 * b + 6] + r[6], y[8 * b + 7] + r[7]);
            }
        }
        if constexpr (EPI == 3) {
            // gM = (gI[upper half] @ W12^T) * (M > 0): per channel ci of M the gradient channels ascending (the backward-data chain)
            if (a.pw_on) {          // uniform
                constexpr int NHB = PWI / 8;                         // blocks of the upper half = of M
                constexpr int HI0 = GB == 2 ? 1 : 0;                 // its first local block: C = 16: the launch makes blocks 0, 1; C = 32: 2, 3
#pragma unroll
                for (int b = 0; b < NHB; ++b) {
                    const float* mrow = a.pw_aux[b] + row * 8;
                    float g[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float* wrow = wq + (8 * b + j) * PWO;
                        float t = 0.0f;
#pragma unroll
                        for (int co = 0; co < PWO; ++co) t = fmaf(o[HI0 + co / 8][co % 8], wrow[co], t);
                        g[j] = mrow[j] > 0.0f ? t : 0.0f;
                    }
                    float* q = a.pw_out[b] + row * 8;
                    *reinterpret_cast<float4*>(q) = make_float4(g[0], g[1], g[2], g[3]);
                    *reinterpret_cast<float4*>(q + 4) = make_float4(g[4], g[5], g[6], g[7]);
                }
            }
        }
    }
}

template <int GB, int PB, bool BWD, int EPI = 0>
static int wc_launch(const WcArgs& a, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, hipStream_t s) {
    constexpr int PWI = EPI == 1 ? 8 * GB : EPI == 2 ? 8 * PB : EPI == 3 ? 4 * GB : EPI == 4 ? 16 * GB : 0;
    constexpr int PWO = EPI == 1 ? 8 * PB : EPI == 2 ? 8 * PB : EPI == 3 ? 4 * GB : EPI == 4 ? 8 * GB : 0;
    constexpr size_t lds = ((size_t)27 * GB * PB * 64 + (size_t)PWI * PWO + PWO) * 4;
    static const int cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
        return v;
    }();
    // the weight image is built once per workgroup, which then loops over row tiles (two workgroups per CU as in occ_conv7_k)
    const int64_t tiles = linr_grid(n, LINR_CONV_BLOCK);
    const int64_t want = (int64_t)cus * 2;
    const int64_t per = (tiles + want - 1) / want;
    const int64_t grid = (tiles + per - 1) / per;
    wconv_k<GB, PB, BWD, EPI><<<(unsigned)grid, LINR_CONV_BLOCK, lds, s>>>(a, lo, mask, ld, n);
    return linr_launch_rc();
}

template <bool BWD>
static int wc_dispatch(int gb, int pb, int epi, const WcArgs& a, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, hipStream_t s) {
    if (epi == 0) {
        if (gb == 1 && pb == 1) return wc_launch<1, 1, BWD>(a, lo, mask, ld, n, s);
        if (gb == 1 && pb == 2) return wc_launch<1, 2, BWD>(a, lo, mask, ld, n, s);
        if (gb == 2 && pb == 1) return wc_launch<2, 1, BWD>(a, lo, mask, ld, n, s);
        if (gb == 2 && pb == 2) return wc_launch<2, 2, BWD>(a, lo, mask, ld, n, s);
        if (gb == 4 && pb == 1) return wc_launch<4, 1, BWD>(a, lo, mask, ld, n, s);
        if (gb == 4 && pb == 2) return wc_launch<4, 2, BWD>(a, lo, mask, ld, n, s);
        return LINR_EINVAL;
    }
    // the side paths of a wide Inception layer: C = 16 (h = 8) and C = 32 (h = 16)
    if constexpr (!BWD) {
        if (epi == 1 && gb == 2 && pb == 1) return wc_launch<2, 1, false, 1>(a, lo, mask, ld, n, s);      // conv0_0 16 -> 8 (+ conv1_0)
        if (epi == 1 && gb == 4 && pb == 2) return wc_launch<4, 2, false, 1>(a, lo, mask, ld, n, s);      // conv0_0 32 -> 16
        if (epi == 2 && gb == 1 && pb == 1) return wc_launch<1, 1, false, 2>(a, lo, mask, ld, n, s);      // conv1_1 8 -> 8 (+ conv1_2)
        if (epi == 2 && gb == 2 && pb == 2) return wc_launch<2, 2, false, 2>(a, lo, mask, ld, n, s);      // conv1_1 16 -> 16
    } else {
        if (epi == 3 && gb == 2 && pb == 2) return wc_launch<2, 2, true, 3>(a, lo, mask, ld, n, s);       // tail conv 16 <- 16 (+ gM)
        if (epi == 3 && gb == 4 && pb == 2) return wc_launch<4, 2, true, 3>(a, lo, mask, ld, n, s);       // tail conv 32 <- 32
        if (epi == 4 && gb == 1 && pb == 2) return wc_launch<1, 2, true, 4>(a, lo, mask, ld, n, s);       // conv0_0 16 <- 8 (+ conv1_0's)
        if (epi == 4 && gb == 2 && pb == 2) return wc_launch<2, 2, true, 4>(a, lo, mask, ld, n, s);       // conv0_0 32 <- 16
    }
    return LINR_EINVAL;
}

// in / out / res / act: HOST arrays of device pointers to the [rows][8] blocks (gathered blocks must have the zero row in front).
// fwd (bwd = 0): gathers cin channels in ceil(cin / 8) blocks (cin < 8: one block whose channels >= cin are ignored), produces cout / 8
// blocks.  bwd (bwd = 1): gathers the output gradient in cout / 8 blocks at the mirrored taps, produces the input gradient in cin / 8
// blocks (cin a multiple of 8).  flags: LINR_RELU, LINR_ACCUM, LINR_RELU_MASK as in linr_spconv_cmap.  Up to 32 channels either side.
static int spconv_wide_impl(int32_t bwd, const float* const* in_h, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                            const float* W, const float* bias, int32_t cin, int32_t cout, const float* const* res_h,
                            const float* const* act_h, float* const* out_h, uint32_t flags, const linr_wide_pw* pw, void* stream) {
    if (n < 0 || ld < n || cin < 1 || cin > 32 || cout < 8 || cout > 32 || cout % 8) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!in_h || !out_h || !lo || !mask || !W) return LINR_EINVAL;
    if (flags & ~(LINR_RELU | LINR_ACCUM | LINR_RELU_MASK)) return LINR_EINVAL;
    if ((flags & LINR_RELU_MASK) && !act_h) return LINR_EINVAL;
    if (bwd && cin % 8) return LINR_EINVAL;
    if (!bwd && cin > 8 && cin % 8) return LINR_EINVAL;
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull || ld >= ((int64_t)1 << 26)) return LINR_EINVAL;
    const int gch = bwd ? cout : cin, pch = bwd ? cin : cout;
    const int gb = (gch + 7) / 8, npb = pch / 8;
    if (gb == 3 || npb == 3) return LINR_EINVAL;          // widths are 8, 16 or 32
    WcArgs a;
    for (int g = 0; g < WC_MAXB; ++g) a.in[g] = g < gb ? in_h[g] : nullptr;
    for (int g = 0; g < gb; ++g)
        if (!a.in[g] || !linr_aligned16(a.in[g])) return a.in[g] ? LINR_EALIGN : LINR_EINVAL;
    a.W = W; a.bias = bwd ? nullptr : bias; a.cin = cin; a.cout = cout; a.gvalid = gch; a.flags = flags;
    a.pw_W = nullptr; a.pw_b = nullptr; a.pw_aux[0] = a.pw_aux[1] = nullptr; a.pw_out[0] = a.pw_out[1] = nullptr; a.pw_on = 0; a.pw_hi0 = 0;
    int epi = 0;
    if (pw && pw->mode != 0) {
        // the side path of a wide Inception layer: C = 2 h in {16, 32}
        epi = pw->mode;
        const int C = epi == 1 ? cin : epi == 2 ? 2 * cin : epi == 3 ? cin : 2 * cout;       // EPI 4: bwd conv0_0, cout = h
        const int h = C / 2, nhb = h / 8;
        if ((C != 16 && C != 32) || epi < 1 || epi > 4 || !pw->W) return LINR_EINVAL;
        if ((epi <= 2) == (bwd != 0)) return LINR_EINVAL;
        if (epi == 1 && (cout != h)) return LINR_EINVAL;
        if (epi == 2 && (cin != h || cout != h)) return LINR_EINVAL;
        if (epi == 3 && (cout != C)) return LINR_EINVAL;
        if (epi == 4 && (cin != C)) return LINR_EINVAL;
        if (epi != 1 && !pw->aux_h) return LINR_EINVAL;
        if (epi != 4 && !pw->out2_h) return LINR_EINVAL;
        a.pw_W = pw->W; a.pw_b = pw->b;
        for (int q = 0; q < nhb; ++q) {
            if (epi != 1) { a.pw_aux[q] = pw->aux_h[q]; if (!a.pw_aux[q] || !linr_aligned16(a.pw_aux[q])) return a.pw_aux[q] ? LINR_EALIGN : LINR_EINVAL; }
            if (epi != 4) { a.pw_out[q] = pw->out2_h[q]; if (!a.pw_out[q] || !linr_aligned16(a.pw_out[q])) return a.pw_out[q] ? LINR_EALIGN : LINR_EINVAL; }
        }
    }
    hipStream_t s = (hipStream_t)stream;
    linr_poison_hook(s, 16);
    for (int p0 = 0; p0 < npb; p0 += 2) {
        const int pb = npb - p0 >= 2 ? 2 : 1;
        for (int q = 0; q < 2; ++q) {
            a.out[q] = q < pb ? out_h[p0 + q] : nullptr;
            a.res[q] = (q < pb && res_h) ? res_h[p0 + q] : nullptr;
            a.act[q] = (q < pb && act_h) ? act_h[p0 + q] : nullptr;
            if (q < pb && (!a.out[q] || !linr_aligned16(a.out[q]))) return a.out[q] ? LINR_EALIGN : LINR_EINVAL;
            if (q < pb && (flags & LINR_RELU_MASK) && !a.act[q]) return LINR_EINVAL;
        }
        a.pb0 = p0;
        int e = epi;
        if (epi == 3) {          // gM rides in the launch that produces the upper half of the input gradient
            const int nh = npb / 2;
            a.pw_on = (p0 + pb > nh) ? 1 : 0;
            a.pw_hi0 = nh > p0 ? nh - p0 : 0;
            if (!a.pw_on) e = 0;
            else if (a.pw_hi0 != (gb == 2 ? 1 : 0)) return LINR_EINVAL;          // the kernel's compile-time layout
        }
        const int rc = bwd ? wc_dispatch<true>(gb, pb, e, a, lo, mask, ld, n, s) : wc_dispatch<false>(gb, pb, e, a, lo, mask, ld, n, s);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int linr_spconv_wide(int32_t bwd, const float* const* in_h, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                                const float* W, const float* bias, int32_t cin, int32_t cout, const float* const* res_h,
                                const float* const* act_h, float* const* out_h, uint32_t flags, void* stream) {
    return spconv_wide_impl(bwd, in_h, lo, mask, ld, n, W, bias, cin, cout, res_h, act_h, out_h, flags, nullptr, stream);
}

// The same convolution with a pointwise layer of the wide Inception layer fused into its epilogue (pw->mode, models/resnet.py:55-60):
//   1  forward conv0_0 (C -> h):   out2 = relu(in @ W10 + b10) of the row itself (conv1_0);            W = W10 [C][h], b = b10
//   2  forward conv1_1 (h -> h):   out2 = (the produced row) @ W12 + b12 + aux (conv1_2 + residual);     W = W12 [h][h], b = b12
//   3  backward of the tail conv (C <- C): out2 = ((produced upper half) @ W12^T) * (aux > 0)  (gM);      W = W12, aux = M
//   4  backward of conv0_0 (C <- h): produced += aux @ W10^T in front of the ReLU mask (aux = gH1);       W = W10
// aux_h / out2_h: HOST arrays of h / 8 block pointers.  Per output the fmaf chains of linr_linear_wide, so the fused and the two-launch
// forms give the same bits.
extern "C" int linr_spconv_wide_pw(int32_t bwd, const float* const* in_h, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                                   const float* W, const float* bias, int32_t cin, int32_t cout, const float* const* res_h,
                                   const float* const* act_h, float* const* out_h, uint32_t flags, const linr_wide_pw* pw, void* stream) {
    return spconv_wide_impl(bwd, in_h, lo, mask, ld, n, W, bias, cin, cout, res_h, act_h, out_h, flags, pw, stream);
}

// ---- weight gradient of a wide convolution -------------------------------------------------------------------------------------------
// gW[k][ci][co] = sum_r in[nbr_k(r)][ci] gout[r][co] block pair by block pair (input block bi, gradient block bo) with the 8-wide
// transposing kernel (csrc/fused.hip: spconv_wgrad_t_k), all pairs of the convolution as the groups of grouped launches (8 pairs per
// launch) into ONE slab, then one fixed-order reduction straight into the dense [27][cin][cout] kernel gradient and the bias gradient
// (round 3: a launch, a reduction and two copies per pair).
#define WW_PAIR 1736          // slab elements of a pair: [27][8][8] kernel block + 8 bias sums
__global__ __launch_bounds__(LINR_BLOCK) void wide_slab_reduce_k(const float* __restrict__ slab, int nblocks, int npairs, int cin, int cout,
                                                                 float* __restrict__ gW, float* __restrict__ gb) {
    // 16 elements per workgroup, 16 slices of the slab rows per element (each 4 interleaved partial sums), slices folded in order
    __shared__ float part[16][17];
    const int el = threadIdx.x % 16, sl = threadIdx.x / 16;
    const int total = 27 * cin * cout + cout;
    const int e = blockIdx.x * 16 + el;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (e < total) {
        const int nbo = cout / 8;
        int64_t src;
        if (e < 27 * cin * cout) {
            const int co = e % cout, ci = (e / cout) % cin, k = e / (cout * cin);
            const int bi = ci / 8, bo = co / 8, cw = cin - 8 * bi < 8 ? cin - 8 * bi : 8;
            src = (int64_t)(bi * nbo + bo) * WW_PAIR + (k * cw + (ci % 8)) * 8 + (co % 8);
        } else {
            const int co = e - 27 * cin * cout;
            src = (int64_t)(co / 8) * WW_PAIR + 1728 + (co % 8);          // the column sums of gradient block bo, from pair (0, bo)
        }
        const int64_t stride = (int64_t)npairs * WW_PAIR;
        int b = sl;
        for (; b + 48 < nblocks; b += 64) {
            s0 += slab[(int64_t)(b + 0) * stride + src];
            s1 += slab[(int64_t)(b + 16) * stride + src];
            s2 += slab[(int64_t)(b + 32) * stride + src];
            s3 += slab[(int64_t)(b + 48) * stride + src];
        }
        for (; b < nblocks; b += 16) s0 += slab[(int64_t)b * stride + src];
    }
    part[sl][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl != 0 || e >= total) return;
    float s = part[0][el];
#pragma unroll
    for (int q = 1; q < 16; ++q) s += part[q][el];
    if (e < 27 * cin * cout) gW[e] = s;
    else if (gb) gb[e - 27 * cin * cout] = s;
}


// One gather for ALL gradient blocks: spconv_wgrad_t_k<8, false> (csrc/fused.hip) with the transposed image of an input block's
// gathered rows multiplied with the 8-row tiles of NGB gradient blocks (group = input block, blockIdx.y) - the gathers, index loads
// and LDS transposes of a pair launch are paid once per input block instead of once per (input, gradient) pair; per accumulator the
// same v_mfma_f32_4x4x1 sequence in the same row order and the same wave fold as the pair kernel, so the slab holds the same bits.
#define WW_WAVES 4
#define WW_PITCH 288                  // bytes per tap in the LDS image: 8 rows x 32 B + 32 B
#define WW_TAPS 28                    // 27 taps + the dump slot of the 7th gather's unused lane group
struct WwArgs {
    const float* in[WC_MAXB];         // the groups' input blocks (zero row at [-8, 0))
    const float* g[WC_MAXB][WC_MAXB]; // per group: the gradient blocks [n][8] (one convolution: the same for every group)
    int cw[WC_MAXB];                  // live channels of a group's input block
    int64_t slab_off[WC_MAXB];        // per group: where its NGB pairs start in a slab row (one convolution: bi * nbo * WW_PAIR)
    float* slab; int64_t block_stride;
};

template <int NGB>
__global__ __launch_bounds__(WW_WAVES * 64) void wwgrad_k(WwArgs a, const int32_t* __restrict__ tile8t, int64_t n) {
    constexpr int NA = 32;                                  // accumulator floats per lane and gradient block
    constexpr int IMG_F4 = WW_TAPS * WW_PITCH / 16;
    constexpr int FOLD_F4 = 16 * (NA + 1);
    constexpr int SMEM_F4 = WW_WAVES * (IMG_F4 > FOLD_F4 ? IMG_F4 : FOLD_F4);
    __shared__ float4 smem[SMEM_F4];
    __shared__ float sbias[NGB][WW_WAVES][8];
    float* sacc = reinterpret_cast<float*>(smem);
    const int bi = blockIdx.y;
    const float* in = a.in[bi];
    const float* gsrc[NGB];
#pragma unroll
    for (int b = 0; b < NGB; ++b) gsrc[b] = a.g[bi][b];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane & 1, kk = lane >> 1, k = kk < 27 ? kk : 26;              // MFMA-side role: (tap, quad)
    const int gq = lane & 1, gu8 = (lane >> 1) & 7, gt = lane >> 4;             // gather-side role: (tap of 4, row of 8, quad)
    f32x4 acc[NGB][4][2];
#pragma unroll
    for (int b = 0; b < NGB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[b][c][h] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    int64_t per = (n + gridDim.x - 1) / gridDim.x;
    per = (per + 7) & ~(int64_t)7;
    const int64_t b0 = (int64_t)blockIdx.x * per;
    const int64_t b1 = (b0 + per < n) ? b0 + per : n;
    const char* ubase = reinterpret_cast<const char*>(in - 8);
    const uint32_t uoff = 32u + 16u * gq;
    const int gu = (lane >> 3) & 7, gc = lane & 7;           // the lane's element of an 8-row gradient tile
    const int32_t* tk = tile8t + (gt * 8 + gu8) * 8;
    char* img = reinterpret_cast<char*>(smem + wave * IMG_F4);
    uint32_t wofs[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int p = 4 * j + gt;
        wofs[j] = (uint32_t)((p < 27 ? LINR_TAP(p) : 27) * WW_PITCH + gu8 * 32 + gq * 16);
    }
    const uint32_t rd0 = (uint32_t)(k * WW_PITCH + q * 16);
    float bsum[NGB], gvn[NGB], gvc[NGB];
#pragma unroll
    for (int b = 0; b < NGB; ++b) bsum[b] = gvn[b] = gvc[b] = 0.0f;
    const int64_t g00 = b0 + 8 * wave;
    int4 ia = make_int4(-1, -1, -1, -1), ib = ia;
    float4 xg[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) xg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g00 < b1) {                        // wave-uniform; blocks behind the last row must not touch the tables at all
        const int4 a0 = *reinterpret_cast<const int4*>(tk + g00 * 32);
        const int4 c0 = *reinterpret_cast<const int4*>(tk + g00 * 32 + 4);
#pragma unroll
        for (int b = 0; b < NGB; ++b) gvc[b] = (g00 + gu < n) ? gsrc[b][(g00 + gu) * 8 + gc] : 0.0f;
        const int32_t i0[8] = {a0.x, a0.y, a0.z, a0.w, c0.x, c0.y, c0.z, c0.w};
#pragma unroll
        for (int j = 0; j < 7; ++j) xg[j] = *reinterpret_cast<const float4*>(ubase + (((uint32_t)i0[j] << 5) + uoff));
        const int64_t g1r = g00 + 8 * WW_WAVES;           // spare all -1 groups behind the last row group: no bounds check
        ia = *reinterpret_cast<const int4*>(tk + g1r * 32);
        ib = *reinterpret_cast<const int4*>(tk + g1r * 32 + 4);
#pragma unroll
        for (int b = 0; b < NGB; ++b) gvn[b] = (g1r + gu < n) ? gsrc[b][(g1r + gu) * 8 + gc] : 0.0f;
    }
    float4 xo[4];                          // rows 4 .. 7 of the previous group (zeros in front of the first: they add +0)
    float gvo[NGB];
#pragma unroll
    for (int u = 0; u < 4; ++u) xo[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int b = 0; b < NGB; ++b) gvo[b] = 0.0f;
    for (int64_t g0r = g00; g0r < b1; g0r += 8 * WW_WAVES) {
#pragma unroll
        for (int j = 0; j < 7; ++j) *reinterpret_cast<float4*>(img + wofs[j]) = xg[j];
        __builtin_amdgcn_sched_barrier(0);
        float gv[NGB];
#pragma unroll
        for (int b = 0; b < NGB; ++b) gv[b] = gvc[b];
        {
            const int32_t idn[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
#pragma unroll
            for (int j = 0; j < 7; ++j) xg[j] = *reinterpret_cast<const float4*>(ubase + (((uint32_t)idn[j] << 5) + uoff));
            const int64_t g2r = g0r + 16 * WW_WAVES;
            ia = *reinterpret_cast<const int4*>(tk + g2r * 32);
            ib = *reinterpret_cast<const int4*>(tk + g2r * 32 + 4);
#pragma unroll
            for (int b = 0; b < NGB; ++b) {
                gvc[b] = gvn[b];
                gvn[b] = (g2r + gu < n) ? gsrc[b][(g2r + gu) * 8 + gc] : 0.0f;
            }
        }
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(img + rd0 + (uint32_t)(u * 32));
#pragma unroll
        for (int b = 0; b < NGB; ++b) bsum[b] += gv[b];
        // rows 4 .. 7 of the PREVIOUS group first (operands in registers since the last iteration): they run while this group's
        // transposed reads are in flight; then rows 0 .. 3 of this group.  Per accumulator the rows keep their order.
        static_for<NGB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            static_for<4>([&](auto uc) {
                constexpr int u = 4 + decltype(uc)::value;
                static_for<2>([&](auto hc) {
                    constexpr int h = decltype(hc)::value;
                    constexpr int ab = u * 2 + h;           // the block holding g[row u][4h .. 4h+3]
                    acc[b][0][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gvo[b], xo[u - 4].x, acc[b][0][h], 4, ab, 0);
                    acc[b][1][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gvo[b], xo[u - 4].y, acc[b][1][h], 4, ab, 0);
                    acc[b][2][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gvo[b], xo[u - 4].z, acc[b][2][h], 4, ab, 0);
                    acc[b][3][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gvo[b], xo[u - 4].w, acc[b][3][h], 4, ab, 0);
                });
            });
        });
        static_for<NGB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            static_for<4>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                static_for<2>([&](auto hc) {
                    constexpr int h = decltype(hc)::value;
                    constexpr int ab = u * 2 + h;
                    acc[b][0][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv[b], x[u].x, acc[b][0][h], 4, ab, 0);
                    acc[b][1][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv[b], x[u].y, acc[b][1][h], 4, ab, 0);
                    acc[b][2][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv[b], x[u].z, acc[b][2][h], 4, ab, 0);
                    acc[b][3][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv[b], x[u].w, acc[b][3][h], 4, ab, 0);
                });
            });
        });
#pragma unroll
        for (int u = 0; u < 4; ++u) xo[u] = x[4 + u];
#pragma unroll
        for (int b = 0; b < NGB; ++b) gvo[b] = gv[b];
    }
    // the last group's rows 4 .. 7
    static_for<NGB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        static_for<4>([&](auto uc) {
            constexpr int u = 4 + decltype(uc)::value;
            static_for<2>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                constexpr int ab = u * 2 + h;
                acc[b][0][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gvo[b], xo[u - 4].x, acc[b][0][h], 4, ab, 0);
                acc[b][1][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gvo[b], xo[u - 4].y, acc[b][1][h], 4, ab, 0);
                acc[b][2][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gvo[b], xo[u - 4].z, acc[b][2][h], 4, ab, 0);
                acc[b][3][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gvo[b], xo[u - 4].w, acc[b][3][h], 4, ab, 0);
            });
        });
    });
    // bias sums: lane l holds the partial column sum of channel gc over rows gu, gu + 8, ...: the 8 row phases by a fixed xor tree
#pragma unroll
    for (int b = 0; b < NGB; ++b) {
        float t = bsum[b];
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) t += __shfl_xor(t, m, 64);
        if (lane < 8) sbias[b][wave][lane] = t;
    }
    const int cinv = a.cw[bi];
    const int per_k = cinv * 8, total = 27 * per_k;
    float* dst0 = a.slab + (int64_t)blockIdx.x * a.block_stride + a.slab_off[bi];
    static_for<NGB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        __syncthreads();                       // the images (b = 0) or the previous block's fold are done with
        {
            float* mine = sacc + (wave * 64 + lane) * (NA + 1);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mine[(c * 2 + h) * 4 + i] = acc[b][c][h][i];
        }
        __syncthreads();
        float* dst = dst0 + (int64_t)b * WW_PAIR;
        const int tid = threadIdx.x;
        if (tid < 8) {
            float t = sbias[b][0][tid];
            for (int w = 1; w < WW_WAVES; ++w) t += sbias[b][w][tid];
            dst[1728 + tid] = t;
        }
        for (int e = tid; e < total; e += WW_WAVES * 64) {
            const int kq = e / per_k, r = e - kq * per_k;
            const int ci = r >> 3, co = r & 7;
            const int idx = (2 * kq + (ci >> 2)) * (NA + 1) + ((ci & 3) * 2 + (co >> 2)) * 4 + (co & 3);
            constexpr int W = 64 * (NA + 1);
            float t = sacc[idx];
#pragma unroll
            for (int w = 1; w < WW_WAVES; ++w) t += sacc[w * W + idx];
            dst[e] = t;
        }
    });
}

extern "C" size_t linr_spconv_wgrad_wide_slab_bytes(int32_t cin, int32_t cout) {
    if (cin < 1 || cout < 8) return 0;
    return (size_t)LINR_WG_BLOCKS * ((cin + 7) / 8) * (cout / 8) * WW_PAIR * sizeof(float);
}

// in_h: HOST array of the ceil(cin / 8) input blocks (zero row in front), g_h: of the cout / 8 output-gradient blocks [n][8];
// nbr / tile8t: the frame's kernel map and its tiled copy (linr_kmap_tile8t); slab: linr_spconv_wgrad_wide_slab_bytes(cin, cout) bytes.
// Writes gW [27][cin][cout] and gb [cout] (gb may be NULL).
extern "C" int linr_spconv_wgrad_wide(const float* const* in_h, int32_t cin, const float* const* g_h, int32_t cout, const int32_t* nbr,
                                      const int32_t* tile8t, int64_t ld, int64_t n, float* slab, float* gW, float* gb, void* stream) {
    if (n < 0 || ld < n || cin < 1 || cin > 32 || cout < 8 || cout > 32 || cout % 8 || (cin > 8 && cin % 8)) return LINR_EINVAL;
    if (!in_h || !g_h || !nbr || !slab) return LINR_EINVAL;
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull) return LINR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    linr_poison_hook(s, 16);
    const int nbi = (cin + 7) / 8, nbo = cout / 8, npairs = nbi * nbo;
    if (n == 0) {
        if (!gW) return LINR_EINVAL;                     // deferred reductions (gW == NULL) need rows
        if (hipMemsetAsync(gW, 0, (size_t)27 * cin * cout * sizeof(float), s) != hipSuccess) return LINR_EINVAL;
        return gb ? linr_hip_rc(hipMemsetAsync(gb, 0, (size_t)cout * sizeof(float), s)) : 0;
    }
    for (int i = 0; i < nbi; ++i) if (!in_h[i] || !linr_aligned16(in_h[i])) return in_h[i] ? LINR_EALIGN : LINR_EINVAL;
    for (int i = 0; i < nbo; ++i) if (!g_h[i]) return LINR_EINVAL;
    int nblk = LINR_WG_BLOCKS;
    if (tile8t && linr_aligned16(tile8t) && (nbo == 1 || nbo == 2 || nbo == 4)) {
        // one launch: group = input block, all gradient blocks from its one gather
        WwArgs a;
        for (int i = 0; i < WC_MAXB; ++i) {
            a.in[i] = in_h[i < nbi ? i : 0]; a.cw[i] = 8; a.slab_off[i] = (int64_t)(i < nbi ? i : 0) * nbo * WW_PAIR;
            for (int b = 0; b < WC_MAXB; ++b) a.g[i][b] = g_h[b < nbo ? b : 0];
        }
        for (int i = 0; i < nbi; ++i) a.cw[i] = cin - 8 * i < 8 ? cin - 8 * i : 8;
        a.slab = slab; a.block_stride = (int64_t)npairs * WW_PAIR;
        // 256 persistent blocks per input block (sweep 128 .. 512 on loot10, profiles/r04_wide.txt: 16 -> 16 86 us at 256, 96 at 384, 108 at 512)
        nblk = 256;
        const dim3 grid(nblk, nbi);
        if (nbo == 1) wwgrad_k<1><<<grid, WW_WAVES * 64, 0, s>>>(a, tile8t, n);
        else if (nbo == 2) wwgrad_k<2><<<grid, WW_WAVES * 64, 0, s>>>(a, tile8t, n);
        else wwgrad_k<4><<<grid, WW_WAVES * 64, 0, s>>>(a, tile8t, n);
        const int rc = linr_launch_rc();
        if (rc) return rc;
    } else
    for (int p0 = 0; p0 < npairs; p0 += LINR_MAXG) {
        const int ng = npairs - p0 < LINR_MAXG ? npairs - p0 : LINR_MAXG;
        Grp gp = Grp();
        for (int g = 0; g < ng; ++g) {
            const int p = p0 + g, bi = p / nbo, bo = p % nbo;
            gp.in[g] = in_h[bi] - in_h[0];
            gp.res[g] = g_h[bo] - g_h[0];                       // linr_conv3_wgrad_mfma: res = the output gradient's offset
            gp.w[g] = (int64_t)p * WW_PAIR;
            gp.b[g] = (int64_t)p * WW_PAIR;
            gp.e2[g] = cin - 8 * bi < 8 ? cin - 8 * bi : 8;      // live input channels of the block
        }
        LinrWgradDst d = {slab, (int64_t)npairs * WW_PAIR, 0, 1728, 8};
        const int rc = linr_conv3_wgrad_mfma(in_h[0], 8, g_h[0], 8, nbr, ld, n, 8, 8, d, LINR_WG_BLOCKS, s, &gp, ng, tile8t);
        if (rc) return rc;
    }
    if (!gW) return linr_launch_rc();                   // partials only: the caller reduces them with linr_wide_reduce_many
    const int total = 27 * cin * cout + cout;
    wide_slab_reduce_k<<<linr_grid(total, 16), LINR_BLOCK, 0, s>>>(slab, nblk, npairs, cin, cout, gW, gb);
    return linr_launch_rc();
}

// Two convolutions of the SAME shape (cin = cout = h in {8, 16}: conv0_1 and conv1_1 of a wide Inception layer, which do not depend on
// each other) as ONE launch of 2 h / 8 groups: alone, an 8 -> 8 weight gradient is 256 workgroups - one wave per SIMD.  Partials only:
// a slab row holds convolution A's pairs, then B's (row stride 2 (h / 8)^2 WW_PAIR floats); reduce with linr_wide_reduce_many (kind 0,
// slab = the row's A or B part, ws_ci = the row stride).
extern "C" int linr_spconv_wgrad_wide2(const float* const* inA_h, const float* const* gA_h, const float* const* inB_h, const float* const* gB_h,
                                       int32_t h, const int32_t* tile8t, int64_t n, float* slab, void* stream) {
    if (n < 1 || (h != 8 && h != 16) || !inA_h || !gA_h || !inB_h || !gB_h || !tile8t || !slab) return LINR_EINVAL;
    if (!linr_aligned16(tile8t) || (uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull) return LINR_EINVAL;
    const int nb = h / 8, npairs = nb * nb;
    WwArgs a;
    for (int i = 0; i < WC_MAXB; ++i) {
        const int cv = (i / nb) & 1, bi = i % nb;                    // group i: convolution cv, its input block bi
        const float* const* in_h = cv ? inB_h : inA_h;
        const float* const* g_h = cv ? gB_h : gA_h;
        a.in[i] = in_h[bi]; a.cw[i] = 8;
        a.slab_off[i] = (int64_t)(cv * npairs + bi * nb) * WW_PAIR;
        if (!a.in[i]) return LINR_EINVAL;
        if (!linr_aligned16(a.in[i])) return LINR_EALIGN;
        for (int b = 0; b < WC_MAXB; ++b) { a.g[i][b] = g_h[b < nb ? b : 0]; if (!a.g[i][b]) return LINR_EINVAL; }
    }
    a.slab = slab; a.block_stride = (int64_t)2 * npairs * WW_PAIR;
    hipStream_t s = (hipStream_t)stream;
    linr_poison_hook(s, 16);
    const dim3 grid(256, 2 * nb);
    if (nb == 1) wwgrad_k<1><<<grid, WW_WAVES * 64, 0, s>>>(a, tile8t, n);
    else wwgrad_k<2><<<grid, WW_WAVES * 64, 0, s>>>(a, tile8t, n);
    return linr_launch_rc();
}

// slab rows linr_spconv_wgrad_wide writes (the `nblocks` of its deferred reduction): 256 with the tiled table and cout in {8, 16, 32}
extern "C" int32_t linr_spconv_wgrad_wide_blocks(int32_t cout, int32_t tiled) {
    const int nbo = cout / 8;
    return (tiled && (nbo == 1 || nbo == 2 || nbo == 4)) ? 256 : LINR_WG_BLOCKS;
}

// ---- pointwise layers on channel-blocked activations ---------------------------------------------------------------------------------
// MinkowskiConvolution(kernel_size=1) / nn.Linear of the wide network (conv1_0 C -> C/2, conv1_2 C/2 -> C/2: models/resnet.py:25-46;
// the head's Linear(C, 24): models/upsample.py:73-76) as ONE launch: a lane owns a row, reads every input block once, and writes every
// output block - the blocked form ran (Ci / 8)(Co / 8) launches of linear_k<8, 8> accumulating through memory.  The weight element
// (ci, co) lives at W[ci * ws_ci + co * ws_co] (wave-uniform: scalar loads), so ME's [cin][cout], torch's [cout][cin] and the
// backward-data pass (strides swapped) are the same kernel.  Per output: bias, then the inputs ascending (one fmaf chain), then
// + res, + old (LINR_ACCUM), * (act > 0) (LINR_RELU_MASK), ReLU - the epilogue order of linear_k (csrc/linear.hip).
struct WlArgs {
    const float* in[WC_MAXB];      // INB: cin / 8 blocks [n][8]; else in[0] = one [n][cin] matrix
    float* out[WC_MAXB];           // OUTB: cout / 8 blocks [n][8]; else out[0] = one [n][cout] matrix
    const float* res[WC_MAXB];     // laid out like out, or all nullptr
    const float* act[WC_MAXB];     // laid out like out (LINR_RELU_MASK)
    const float* W; int ws_ci, ws_co;
    const float* bias;             // [cout] or nullptr
    unsigned flags;
};

// COM: weight element (i, o) at W[i * COUT + o] (ME layout forward, torch layout backward-data), else at W[o * CIN + i]: compile-time
// offsets, so the scalar loads batch (with run-time strides every weight costs an address computation and its own s_load: the head's
// 16 -> 24 layer took 39 us instead of 12).
template <int CIN, int COUT, bool INB, bool OUTB, bool COM>
__global__ __launch_bounds__(LINR_BLOCK) void wlin_k(WlArgs a, int64_t n) {
    const int64_t row = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (row >= n) return;
    float x[CIN];
    if constexpr (INB) {
#pragma unroll
        for (int b = 0; b < CIN / 8; ++b) {
            const float4 t0 = *reinterpret_cast<const float4*>(a.in[b] + row * 8), t1 = *reinterpret_cast<const float4*>(a.in[b] + row * 8 + 4);
            x[8 * b] = t0.x; x[8 * b + 1] = t0.y; x[8 * b + 2] = t0.z; x[8 * b + 3] = t0.w;
            x[8 * b + 4] = t1.x; x[8 * b + 5] = t1.y; x[8 * b + 6] = t1.z; x[8 * b + 7] = t1.w;
        }
    } else {
        static_assert(CIN % 4 == 0, "unblocked rows are read as float4");
#pragma unroll
        for (int v = 0; v < CIN / 4; ++v) {
            const float4 t = *reinterpret_cast<const float4*>(a.in[0] + row * CIN + 4 * v);
            x[4 * v] = t.x; x[4 * v + 1] = t.y; x[4 * v + 2] = t.z; x[4 * v + 3] = t.w;
        }
    }
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = a.bias ? a.bias[o] : 0.0f;
#pragma unroll
    for (int i = 0; i < CIN; ++i)
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = fmaf(x[i], a.W[COM ? i * COUT + o : o * CIN + i], acc[o]);
    constexpr int OP = OUTB ? COUT / 8 : 1, OW = OUTB ? 8 : COUT;          // output pieces and their width
    static_assert(OW % 4 == 0, "output rows are written as float4");
#pragma unroll
    for (int b = 0; b < OP; ++b) {
        float* op = a.out[b] + row * OW;
        float* y = acc + b * OW;
        if (a.res[b]) {
#pragma unroll
            for (int v = 0; v < OW / 4; ++v) {
                const float4 t = *reinterpret_cast<const float4*>(a.res[b] + row * OW + 4 * v);
                y[4 * v] += t.x; y[4 * v + 1] += t.y; y[4 * v + 2] += t.z; y[4 * v + 3] += t.w;
            }
        }
        if (a.flags & LINR_ACCUM) {
#pragma unroll
            for (int v = 0; v < OW / 4; ++v) {
                const float4 t = *reinterpret_cast<const float4*>(op + 4 * v);
                y[4 * v] += t.x; y[4 * v + 1] += t.y; y[4 * v + 2] += t.z; y[4 * v + 3] += t.w;
            }
        }
        if (a.flags & LINR_RELU_MASK) {
#pragma unroll
            for (int v = 0; v < OW / 4; ++v) {
                const float4 t = *reinterpret_cast<const float4*>(a.act[b] + row * OW + 4 * v);
                y[4 * v] = t.x > 0.0f ? y[4 * v] : 0.0f; y[4 * v + 1] = t.y > 0.0f ? y[4 * v + 1] : 0.0f;
                y[4 * v + 2] = t.z > 0.0f ? y[4 * v + 2] : 0.0f; y[4 * v + 3] = t.w > 0.0f ? y[4 * v + 3] : 0.0f;
            }
        }
        if (a.flags & LINR_RELU) {
#pragma unroll
            for (int o = 0; o < OW; ++o) y[o] = fmaxf(y[o], 0.0f);
        }
#pragma unroll
        for (int v = 0; v < OW / 4; ++v) *reinterpret_cast<float4*>(op + 4 * v) = make_float4(y[4 * v], y[4 * v + 1], y[4 * v + 2], y[4 * v + 3]);
    }
}

// in_h / out_h / res_h / act_h: HOST arrays of device pointers (blocked: channels / 8 entries of [n][8] matrices; unblocked: one entry,
// a dense [n][channels] matrix).  Shapes: blocked -> blocked with (cin, cout) in {8, 16, 32} x {8, 16, 32}; blocked -> [n][24] and
// [n][24] -> blocked with the blocked side 16 or 32 (the head's first Linear and its backward-data pass).
extern "C" int linr_linear_wide(const float* const* in_h, int32_t cin, int32_t in_blocked, const float* W, int32_t ws_ci, int32_t ws_co,
                                const float* bias, int32_t cout, int32_t out_blocked, const float* const* res_h, const float* const* act_h,
                                float* const* out_h, int64_t n, uint32_t flags, void* stream) {
    if (n < 0 || cin < 1 || cout < 1 || cin > 32 || cout > 32 || !in_h || !out_h || !W) return LINR_EINVAL;
    if ((flags & LINR_RELU_MASK) && !act_h) return LINR_EINVAL;
    if (n == 0) return 0;
    const int ni = in_blocked ? cin / 8 : 1, no = out_blocked ? cout / 8 : 1;
    if ((in_blocked && cin % 8) || (out_blocked && cout % 8)) return LINR_EINVAL;
    WlArgs a;
    for (int i = 0; i < WC_MAXB; ++i) { a.in[i] = nullptr; a.out[i] = nullptr; a.res[i] = nullptr; a.act[i] = nullptr; }
    for (int i = 0; i < ni; ++i) {
        if (!in_h[i]) return LINR_EINVAL;
        if (!linr_aligned16(in_h[i])) return LINR_EALIGN;
        a.in[i] = in_h[i];
    }
    for (int i = 0; i < no; ++i) {
        if (!out_h[i]) return LINR_EINVAL;
        if (!linr_aligned16(out_h[i])) return LINR_EALIGN;
        a.out[i] = out_h[i];
        if (res_h && res_h[i]) { if (!linr_aligned16(res_h[i])) return LINR_EALIGN; a.res[i] = res_h[i]; }
        if (flags & LINR_RELU_MASK) {
            if (!act_h[i]) return LINR_EINVAL;
            if (!linr_aligned16(act_h[i])) return LINR_EALIGN;
            a.act[i] = act_h[i];
        }
    }
    a.W = W; a.ws_ci = ws_ci; a.ws_co = ws_co; a.bias = (flags & LINR_NO_BIAS) ? nullptr : bias; a.flags = flags;
    hipStream_t s = (hipStream_t)stream;
    linr_poison_hook(s, 16);
    const unsigned grid = linr_grid(n, LINR_BLOCK);
    const bool com = (ws_ci == cout && ws_co == 1);
    if (!com && !(ws_ci == 1 && ws_co == cin)) return LINR_EINVAL;          // dense [cin][cout] or dense [cout][cin]
#define WL_GO(CI, CO, IB, OB)                                                                         \
    if (cin == CI && cout == CO && (in_blocked != 0) == IB && (out_blocked != 0) == OB) {             \
        if (com) wlin_k<CI, CO, IB, OB, true><<<grid, LINR_BLOCK, 0, s>>>(a, n);                      \
        else wlin_k<CI, CO, IB, OB, false><<<grid, LINR_BLOCK, 0, s>>>(a, n);                         \
        return linr_launch_rc();                                                                      \
    }
    WL_GO(8, 8, true, true) WL_GO(8, 16, true, true) WL_GO(16, 8, true, true) WL_GO(16, 16, true, true)
    WL_GO(16, 32, true, true) WL_GO(32, 16, true, true) WL_GO(32, 32, true, true) WL_GO(8, 32, true, true) WL_GO(32, 8, true, true)
    WL_GO(16, 24, true, false) WL_GO(32, 24, true, false) WL_GO(24, 16, false, true) WL_GO(24, 32, false, true)
#undef WL_GO
    return LINR_EINVAL;
}

// Weight gradient of such a layer: gW(ci, co) = sum_r x[r][ci] g[r][co] at gW[ci * ws_ci + co * ws_co], gb[co] = sum_r g[r][co]
// (gb may be NULL; LINR_ACCUM adds to the destinations).  Every (input piece, gradient piece) pair is a group of ONE grouped launch of
// xtg_wgrad_k (csrc/linear.hip: the rows as the K dimension of v_mfma_f32_16x16x4_f32) into a dense [cin + 1][cout] partial per slab
// row, then one fixed-order reduction scattered to the strides - the blocked form ran a launch and a reduction per pair.
extern "C" size_t linr_linear_wgrad_wide_workspace_bytes(int64_t n, int32_t cin, int32_t cout) {
    if (n <= 0 || cin < 1 || cout < 1) return 0;
    return (size_t)linr_lin_blocks(n) * ((size_t)(cin + 2) * cout) * sizeof(float);
}

extern "C" int linr_linear_wgrad_wide(const float* const* in_h, int32_t cin, int32_t in_blocked, const float* const* g_h, int32_t cout,
                                      int32_t g_blocked, int64_t n, float* gW, int32_t ws_ci, int32_t ws_co, float* gb, uint32_t flags,
                                      void* ws, size_t ws_bytes, void* stream) {
    if (n < 0 || cin < 1 || cout < 1 || cin > 32 || cout > 32 || !in_h || !g_h) return LINR_EINVAL;
    if ((in_blocked && cin % 8) || (g_blocked && cout % 8) || (!in_blocked && cin > 31)) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!ws) return LINR_EINVAL;
    if (ws_bytes < linr_linear_wgrad_wide_workspace_bytes(n, cin, cout)) return LINR_ENOSPC;
    const int ni = in_blocked ? cin / 8 : 1, no = g_blocked ? cout / 8 : 1;
    const int mi = in_blocked ? 8 : cin, nn = g_blocked ? 8 : cout;              // channels per piece = its leading dimension
    if (ni * no > LINR_MAXG) return LINR_EINVAL;
    for (int i = 0; i < ni; ++i) if (!in_h[i]) return LINR_EINVAL;
    for (int i = 0; i < no; ++i) if (!g_h[i]) return LINR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    linr_poison_hook(s, 16);
    const int nb = linr_lin_blocks(n);
    const int64_t stride = (int64_t)(cin + 2) * cout;                             // [cin + 1][cout] + a dump row for the duplicate bias sums
    Grp gp = Grp();
    for (int pi = 0; pi < ni; ++pi)
        for (int po = 0; po < no; ++po) {
            const int g = pi * no + po;
            gp.in[g] = in_h[pi] - in_h[0];
            gp.res[g] = g_h[po] - g_h[0];
            gp.w[g] = (int64_t)(mi * pi) * cout + nn * po;
            gp.b[g] = (int64_t)(pi == 0 ? cin : cin + 1) * cout + nn * po;        // the column sums of g: kept from the first input piece
        }
    LinrLinDst d = {(float*)ws, stride, 0, cout, 1, 0};
    const int rc = linr_linear_wgrad_partial(in_h[0], mi, g_h[0], nn, n, mi, nn, d, nb, s, &gp, ni * no);
    if (rc || !gW) return rc;                            // gW == NULL: partials only (linr_wide_reduce_many)
    return linr_linear_slab_reduce_launch((const float*)ws, nb, stride, cin, cout, gW, ws_ci, ws_co, gb, flags, s);
}

extern "C" int32_t linr_linear_wgrad_wide_blocks(int64_t n) { return n > 0 ? linr_lin_blocks(n) : 0; }

// ---- many deferred reductions in one launch --------------------------------------------------------------------------------------------
// A backward pass of the wide network ends ~64 weight-gradient launches, each followed by a 5 us reduction of its slab; with gW = NULL
// linr_spconv_wgrad_wide / linr_linear_wgrad_wide leave the partials in their slabs and this entry reduces up to 32 of them per launch
// (the same fixed-order sums: 16 elements per workgroup, 16 slices of the slab rows, four interleaved partial sums per slice).
#define RD_MAX 32
struct RdItem { const float* slab; float* gW; float* gb; int kind, nblocks, cin, cout, ws_ci, ws_co, blk0; };
struct RdArgs { RdItem it[RD_MAX]; int n; };
__global__ __launch_bounds__(LINR_BLOCK) void wide_reduce_many_k(RdArgs A) {
    __shared__ float part[16][17];
    int ii = 0;
    for (int i = 1; i < A.n; ++i) ii += ((int)blockIdx.x >= A.it[i].blk0) ? 1 : 0;          // uniform: this workgroup's item
    const RdItem& d = A.it[ii];
    const float* slab = d.slab;
    const int cin = d.cin, cout = d.cout, nblocks = d.nblocks;
    const int el = threadIdx.x % 16, sl = threadIdx.x / 16;
    const int e = ((int)blockIdx.x - d.blk0) * 16 + el;
    const int total = d.kind == 0 ? 27 * cin * cout + cout : (cin + 1) * cout;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    float* dst = nullptr;
    if (e < total) {
        int64_t src, stride;
        if (d.kind == 0) {                      // slab of linr_spconv_wgrad_wide (wide_slab_reduce_k's addressing)
            const int nbo = cout / 8, npairs = ((cin + 7) / 8) * nbo;
            stride = d.ws_ci > 0 ? (int64_t)d.ws_ci : (int64_t)npairs * WW_PAIR;          // ws_ci: the slab's row stride when it is shared
            if (e < 27 * cin * cout) {
                const int co = e % cout, ci = (e / cout) % cin, k = e / (cout * cin);
                const int bi = ci / 8, bo = co / 8, cw = cin - 8 * bi < 8 ? cin - 8 * bi : 8;
                src = (int64_t)(bi * nbo + bo) * WW_PAIR + (k * cw + (ci % 8)) * 8 + (co % 8);
                dst = d.gW + e;
            } else {
                const int co = e - 27 * cin * cout;
                src = (int64_t)(co / 8) * WW_PAIR + 1728 + (co % 8);
                dst = d.gb ? d.gb + co : nullptr;
            }
        } else {                                // slab of linr_linear_wgrad_wide: rows of a dense [cin + 1][cout] partial + a dump row
            stride = (int64_t)(cin + 2) * cout;
            src = e;
            const int ci = e / cout, co = e % cout;
            dst = ci < cin ? d.gW + ci * d.ws_ci + co * d.ws_co : (d.gb ? d.gb + co : nullptr);
        }
        int b = sl;
        for (; b + 48 < nblocks; b += 64) {
            s0 += slab[(int64_t)(b + 0) * stride + src];
            s1 += slab[(int64_t)(b + 16) * stride + src];
            s2 += slab[(int64_t)(b + 32) * stride + src];
            s3 += slab[(int64_t)(b + 48) * stride + src];
        }
        for (; b < nblocks; b += 16) s0 += slab[(int64_t)b * stride + src];
    }
    part[sl][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl != 0 || dst == nullptr) return;
    float s = part[0][el];
#pragma unroll
    for (int q = 1; q < 16; ++q) s += part[q][el];
    *dst = s;
}

// items_h: HOST array of `count` reductions - kind 0: a convolution's slab (nblocks = linr_spconv_wgrad_wide_blocks) into gW [27][cin][cout]
// and gb [cout]; kind 1: a pointwise layer's workspace (nblocks = linr_linear_wgrad_wide_blocks(n)) into gW at the strides and gb.
extern "C" int linr_wide_reduce_many(const linr_wide_reduce* items_h, int32_t count, void* stream) {
    if (count < 0 || (count > 0 && !items_h)) return LINR_EINVAL;
    for (int i = 0; i < count; ++i) {
        const linr_wide_reduce& t = items_h[i];
        if ((t.kind != 0 && t.kind != 1) || !t.slab || !t.gW || t.nblocks < 1 || t.cin < 1 || t.cout < 1 || t.cin > 32 || t.cout > 32) return LINR_EINVAL;
        if (t.kind == 0 && t.cout % 8) return LINR_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    linr_poison_hook(s, 16);
    for (int i0 = 0; i0 < count; i0 += RD_MAX) {
        RdArgs A;
        A.n = count - i0 < RD_MAX ? count - i0 : RD_MAX;
        int blk = 0;
        for (int i = 0; i < A.n; ++i) {
            const linr_wide_reduce& t = items_h[i0 + i];
            const int total = t.kind == 0 ? 27 * t.cin * t.cout + t.cout : (t.cin + 1) * t.cout;
            A.it[i] = {t.slab, t.gW, t.gb, t.kind, t.nblocks, t.cin, t.cout, t.ws_ci, t.ws_co, blk};
            blk += (total + 15) / 16;
        }
        for (int i = A.n; i < RD_MAX; ++i) A.it[i] = A.it[0];
        wide_reduce_many_k<<<blk, LINR_BLOCK, 0, s>>>(A);
        const int rc = linr_launch_rc();
        if (rc) return rc;
    }
    return 0;
}

// ---- the occupancy head on channel-blocked activations -------------------------------------------------------------------------------
// CNP.basic_module behind the prune convolution (models/upsample.py:137-161): p = sigmoid(Linear(24, 1)(ReLU(Linear(C, 24)(c)))) and the
// stage's bits (model_core.py:72-81: BCELoss sums with the log clamp at -100) in ONE launch: a lane owns a row; per unit the fmaf chain
// of the pointwise kernels (bias first, inputs ascending), so p is what linr_linear_wide + linr_linear_fwd + linr_bce_bits_fwd gave.
// The hidden layer is not kept: the backward kernel recomputes it from c.
struct WhArgs {
    const float* c[WC_MAXB];
    const float* w1; const float* b1; const float* w2; const float* b2;      // torch layouts: W1 [24][C], b1 [24], w2 [24], b2 [1]
    const float* target; int target_ld;                                       // occupancy column or nullptr
    float* p; double* partial;                                                // partial[blockIdx.x]: the block's nats (target != nullptr)
};

template <int C>
__device__ __forceinline__ void whead_hidden(const float* const (&cb)[WC_MAXB], int64_t row, const float* __restrict__ w1,
                                             const float* __restrict__ b1, float (&c)[C], float (&h)[24]) {
#pragma unroll
    for (int b = 0; b < C / 8; ++b) {
        const float4 t0 = *reinterpret_cast<const float4*>(cb[b] + row * 8), t1 = *reinterpret_cast<const float4*>(cb[b] + row * 8 + 4);
        c[8 * b] = t0.x; c[8 * b + 1] = t0.y; c[8 * b + 2] = t0.z; c[8 * b + 3] = t0.w;
        c[8 * b + 4] = t1.x; c[8 * b + 5] = t1.y; c[8 * b + 6] = t1.z; c[8 * b + 7] = t1.w;
    }
#pragma unroll
    for (int j = 0; j < 24; ++j) h[j] = b1[j];
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
        for (int j = 0; j < 24; ++j) h[j] = fmaf(c[i], w1[j * C + i], h[j]);
}

template <int C>
__global__ __launch_bounds__(LINR_BLOCK) void whead_fwd_k(WhArgs a, int64_t n) {
    __shared__ double sred[LINR_BLOCK];
    const int64_t row = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    double nats = 0.0;
    if (row < n) {
        float c[C], h[24];
        whead_hidden<C>(a.c, row, a.w1, a.b1, c, h);
        float z = a.b2[0];
#pragma unroll
        for (int j = 0; j < 24; ++j) z = fmaf(fmaxf(h[j], 0.0f), a.w2[j], z);
        const float p = 1.0f / (1.0f + expf(-z));
        a.p[row] = p;
        if (a.target) {
            const float t = a.target[row * a.target_ld];
            const float lp = fmaxf(logf(p), -100.0f), lq = fmaxf(logf(1.0f - p), -100.0f);
            nats = (double)((t - 1.0f) * lq - t * lp);
        }
    }
    if (!a.partial) return;                                   // uniform
    sred[threadIdx.x] = nats;
    __syncthreads();
    for (int s = LINR_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sred[threadIdx.x] += sred[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.partial[blockIdx.x] = sred[0];
}

extern "C" size_t linr_head_wide_workspace_bytes(int64_t n) {
    return n > 0 ? (size_t)linr_grid(n, LINR_BLOCK) * sizeof(double) : 0;
}

// c_h: HOST array of the C / 8 blocks of the prune convolution's output; target: the stage's occupancy column (stride target_ld) or
// NULL; p [n]; bits_acc (double[1], += the stage's bits) or NULL; ws: linr_head_wide_workspace_bytes(n) bytes when bits are wanted.
extern "C" int linr_head_wide_fwd(const float* const* c_h, int32_t C, const float* w1, const float* b1, const float* w2, const float* b2,
                                  const float* target, int32_t target_ld, int64_t n, float* p, double* bits_acc, void* ws, size_t ws_bytes,
                                  void* stream) {
    if (n < 0 || (C != 16 && C != 32) || !c_h || !w1 || !b1 || !w2 || !b2 || !p) return LINR_EINVAL;
    if (bits_acc && (!target || target_ld < 1 || !ws)) return LINR_EINVAL;
    if (n == 0) return 0;
    const bool want_bits = bits_acc || (ws && target);           // ws without bits_acc: the block partials only (linr_bits_finish adds them up)
    if (want_bits && (target_ld < 1 || ws_bytes < linr_head_wide_workspace_bytes(n))) return target_ld < 1 ? LINR_EINVAL : LINR_ENOSPC;
    if (want_bits && (((uintptr_t)ws) & 7u)) return LINR_EALIGN;
    WhArgs a;
    for (int i = 0; i < WC_MAXB; ++i) a.c[i] = nullptr;
    for (int i = 0; i < C / 8; ++i) {
        if (!c_h[i]) return LINR_EINVAL;
        if (!linr_aligned16(c_h[i])) return LINR_EALIGN;
        a.c[i] = c_h[i];
    }
    a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.target = target; a.target_ld = target_ld; a.p = p;
    a.partial = want_bits ? (double*)ws : nullptr;
    hipStream_t s = (hipStream_t)stream;
    linr_poison_hook(s, 16);
    const unsigned nb = linr_grid(n, LINR_BLOCK);
    if (C == 16) whead_fwd_k<16><<<nb, LINR_BLOCK, 0, s>>>(a, n);
    else whead_fwd_k<32><<<nb, LINR_BLOCK, 0, s>>>(a, n);
    if (bits_acc) return linr_bits_finish_launch((const double*)ws, (int)nb, bits_acc, s);
    return linr_launch_rc();
}

// Backward of up to 8 heads in ONE grouped launch (blockIdx.y = stage; every stage's inputs - c, p, the occupancy column - exist once
// the forward has run): per row the hidden layer again from c, gz = d bits / d z from (p, t) (torch's binary_cross_entropy backward
// with its 1e-12 clamp, then the sigmoid's), gh = gz w2 [h > 0], gc = W1^T gh; the weight gradients
//   gW1[24][C] = sum_r gh[r] (x) c[r],  gb1 = sum_r gh[r]     as X^T G products with the rows as the K dimension of
//   v_mfma_f32_16x16x4_f32 (each wave turns its 64 rows of [gh | c | 1] into fragments through a wave-private LDS tile),
//   gw2[24] = sum_r gz[r] relu(h[r]),  gb2 = sum_r gz[r]       per lane, folded by a fixed shuffle tree and the waves in order.
// Persistent blocks: one slab row [stage][W1 | b1 | w2 | b2] per block, summed by the fixed-order reduction of the executor.
#define WH_MAXS 8
struct WhbArgs {
    const float* c[WH_MAXS][WC_MAXB]; float* gc[WH_MAXS][WC_MAXB];
    const float* p[WH_MAXS]; const float* target[WH_MAXS]; int target_ld;
    const float* w1[WH_MAXS]; const float* b1[WH_MAXS]; const float* w2[WH_MAXS];
    float gscale;
    float* slab; int64_t block_stride;
};

template <int C>
__global__ __launch_bounds__(LINR_BLOCK) void whead_bwd_k(WhbArgs A, int64_t n) {
    constexpr int LDW = 24 + C + 1;                         // [gh 24 | c C | 1]: 41 / 57 floats, odd
    constexpr int NT = (C + 1 + 15) / 16;                   // N tiles of the X^T G product (C + 1 columns)
    constexpr int PER = 24 * C + 49;                        // a stage's parameters: W1, b1, w2, b2
    __shared__ float sT[(LINR_BLOCK / 64) * 64 * LDW];
    __shared__ float sfold[64 * (8 * NT + 1)];
    __shared__ float sw2[(LINR_BLOCK / 64) * 25];
    const int st = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mm = lane & 15, rr = lane >> 4;
    const float* w1 = A.w1[st];
    const float* b1 = A.b1[st];
    const float* w2 = A.w2[st];
    const float* pp_ = A.p[st];
    const float* tg = A.target[st];
    // The per-row MLP runs on the matrix cores like the convolutions (v_mfma_f32_4x4x1 with CBSZ = 4: the A operand - four weights -
    // of block ABID broadcast, B = the lane's own scalar; K = 1 keeps the fmaf chains of the forward head: bias first, inputs
    // ascending).  A-operand images, block (lane >> 2) of register v = "combo" 16 v + block, j = lane & 3:
    //   wA: combo 6 i + hq -> W1[4 hq + j][i] (i < C); combos 6 C .. 6 C + 5 -> b1[4 (combo - 6 C) + j]
    //   wB: combo (C / 4) jj + q -> W1[jj][4 q + j]  (jj < 24, q < C / 4)
    //   wC: block hq < 6 -> w2[4 hq + j]
    constexpr int NWA = (6 * C + 6 + 15) / 16, NWB = 24 * (C / 4) / 16, CQ = C / 4;
    float wA[NWA], wB[NWB], wC;
    {
        const int blk = lane >> 2, j = lane & 3;
#pragma unroll
        for (int v = 0; v < NWA; ++v) {
            const int cb = 16 * v + blk;
            wA[v] = cb < 6 * C ? w1[(4 * (cb % 6) + j) * C + cb / 6] : (cb < 6 * C + 6 ? b1[4 * (cb - 6 * C) + j] : 0.0f);
        }
#pragma unroll
        for (int v = 0; v < NWB; ++v) {
            const int cb = 16 * v + blk;
            wB[v] = w1[(cb / CQ) * C + 4 * (cb % CQ) + j];
        }
        wC = blk < 6 ? w2[4 * blk + j] : 0.0f;
    }
    float* T = sT + wave * 64 * LDW;
    f32x4 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float gw2[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) gw2[j] = 0.0f;
    float gz_sum = 0.0f;
    const int64_t tiles = (n + LINR_BLOCK - 1) / LINR_BLOCK;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t row_raw = t * LINR_BLOCK + threadIdx.x;
        const bool live = row_raw < n;
        const int64_t row = live ? row_raw : n - 1;          // every lane stays in the MFMAs (they ignore EXEC)
        float c[C];
#pragma unroll
        for (int b = 0; b < C / 8; ++b) {
            const float4 t0 = *reinterpret_cast<const float4*>(A.c[st][b] + row * 8), t1 = *reinterpret_cast<const float4*>(A.c[st][b] + row * 8 + 4);
            c[8 * b] = t0.x; c[8 * b + 1] = t0.y; c[8 * b + 2] = t0.z; c[8 * b + 3] = t0.w;
            c[8 * b + 4] = t1.x; c[8 * b + 5] = t1.y; c[8 * b + 6] = t1.z; c[8 * b + 7] = t1.w;
        }
        float gz = 0.0f;
        if (live) {
            const float pp = pp_[row], tt = tg[row * A.target_ld];
            const float gp = A.gscale * (pp - tt) / fmaxf((1.0f - pp) * pp, 1e-12f);
            gz = gp * ((1.0f - pp) * pp);
        }
        // hpre = b1 + W1 c  (6 output quads; the bias through x = 1, then the inputs ascending)
        f32x4 hp[6];
        static_for<6>([&](auto hc) {
            constexpr int hq = decltype(hc)::value;
            constexpr int cb = 6 * C + hq;
            hp[hq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[cb / 16], 1.0f, (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, 4, cb % 16, 0);
        });
        static_for<C>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<6>([&](auto hc) {
                constexpr int hq = decltype(hc)::value;
                constexpr int cb = 6 * i + hq;
                hp[hq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[cb / 16], c[i], hp[hq], 4, cb % 16, 0);
            });
        });
        // gh = [hpre > 0] gz w2 ;  gw2 += gz relu(hpre)
        float gh[24];
        static_for<6>([&](auto hc) {
            constexpr int hq = decltype(hc)::value;
            const f32x4 g4 = __builtin_amdgcn_mfma_f32_4x4x1f32(wC, gz, (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, 4, hq, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float hv = hp[hq][j];
                gh[4 * hq + j] = (live && hv > 0.0f) ? g4[j] : 0.0f;
                gw2[4 * hq + j] = fmaf(gz, fmaxf(hv, 0.0f), gw2[4 * hq + j]);
            }
        });
        gz_sum += gz;
        // gc = W1^T gh  (C / 4 output quads, hidden units ascending)
        f32x4 gcq[CQ];
#pragma unroll
        for (int q = 0; q < CQ; ++q) gcq[q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        static_for<24>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            static_for<CQ>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int cb = CQ * jj + q;
                gcq[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(wB[cb / 16], gh[jj], gcq[q], 4, cb % 16, 0);
            });
        });
        if (live) {
#pragma unroll
            for (int q = 0; q < CQ; ++q)
                *reinterpret_cast<float4*>(A.gc[st][q / 2] + row * 8 + 4 * (q % 2)) = make_float4(gcq[q][0], gcq[q][1], gcq[q][2], gcq[q][3]);
        }
        float* Tr = T + lane * LDW;
#pragma unroll
        for (int j = 0; j < 24; ++j) Tr[j] = gh[j];                       // zero for rows beyond n (gz = 0)
#pragma unroll
        for (int i = 0; i < C; ++i) Tr[24 + i] = c[i];
        Tr[24 + C] = live ? 1.0f : 0.0f;
        // X^T G over this wave's 64 rows (wave-private tile: LDS operations of a wave execute in order, no block barrier)
#pragma unroll 4
        for (int s4 = 0; s4 < 16; ++s4) {
            const float* Tq = T + (4 * s4 + rr) * LDW;
            const float a0 = Tq[mm];
            const float a1 = (mm < 8) ? Tq[16 + mm] : 0.0f;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int col = 16 * b + mm;
                const float bv = (col < C + 1) ? Tq[24 + col] : 0.0f;
                acc[0][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][b], 0, 0, 0);
                acc[1][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][b], 0, 0, 0);
            }
        }
    }
    // fold the waves in wave order, then one partial per destination element (C/D map: row = (lane >> 4) * 4 + reg, col = lane & 15)
    float* mine = sfold + lane * (8 * NT + 1);
    for (int w = 0; w < LINR_BLOCK / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = (a * NT + b) * 4 + j;
                        mine[e] = (w == 0) ? acc[a][b][j] : mine[e] + acc[a][b][j];
                    }
        }
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) gz_sum += __shfl_xor(gz_sum, d, 64);
#pragma unroll
    for (int j = 0; j < 24; ++j) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) gw2[j] += __shfl_xor(gw2[j], d, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 24; ++j) sw2[wave * 25 + j] = gw2[j];
        sw2[wave * 25 + 24] = gz_sum;
    }
    __syncthreads();
    if (wave == 0) {
        float* dst = A.slab + (int64_t)blockIdx.x * A.block_stride + (int64_t)st * PER;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = 16 * a + rr * 4 + j, col = 16 * b + mm;
                    const float v = mine[(a * NT + b) * 4 + j];
                    if (m < 24) {
                        if (col < C) dst[m * C + col] = v;
                        else if (col == C) dst[24 * C + m] = v;
                    }
                }
        if (lane < 25) {
            const float v = ((sw2[lane] + sw2[25 + lane]) + sw2[50 + lane]) + sw2[75 + lane];
            dst[24 * C + 24 + lane] = v;                     // w2 [24], then b2
        }
    }
}

#define WH_BLOCKS 256
extern "C" size_t linr_head_wide_bwd_slab_bytes(int32_t C, int32_t nstages) {
    if ((C != 16 && C != 32) || nstages < 1 || nstages > WH_MAXS) return 0;
    return (size_t)WH_BLOCKS * nstages * (24 * C + 49) * sizeof(float);
}

// c_h / gc_h: HOST arrays [nstages][C / 8] of block pointers (stage-major); p_h / target_h / w1_h / b1_h / w2_h: [nstages].  grads: the
// stages' parameter gradients [nstages][W1 (24 x C) | b1 (24) | w2 (24) | b2 (1)] - the inner_mlps of consecutive stages are
// consecutive in the reference's parameter order - written, not accumulated.
extern "C" int linr_head_wide_bwd(const float* const* c_h, const float* const* p_h, const float* const* target_h, int32_t target_ld,
                                  const float* const* w1_h, const float* const* b1_h, const float* const* w2_h, int32_t C, int32_t nstages,
                                  float gscale, float* const* gc_h, int64_t n, float* slab, size_t slab_bytes, float* grads, void* stream) {
    if (n < 0 || (C != 16 && C != 32) || nstages < 1 || nstages > WH_MAXS || target_ld < 1) return LINR_EINVAL;
    if (!c_h || !p_h || !target_h || !w1_h || !b1_h || !w2_h || !gc_h || !slab || !grads) return LINR_EINVAL;
    if (slab_bytes < linr_head_wide_bwd_slab_bytes(C, nstages)) return LINR_ENOSPC;
    hipStream_t s = (hipStream_t)stream;
    linr_poison_hook(s, 16);
    const int64_t total = (int64_t)nstages * (24 * C + 49);
    if (n == 0) return linr_hip_rc(hipMemsetAsync(grads, 0, (size_t)total * sizeof(float), s));
    WhbArgs A;
    const int nb = C / 8;
    for (int k = 0; k < WH_MAXS; ++k) {
        const int kk = k < nstages ? k : 0;
        for (int b = 0; b < WC_MAXB; ++b) {
            const int bb = b < nb ? b : 0;
            A.c[k][b] = c_h[kk * nb + bb]; A.gc[k][b] = gc_h[kk * nb + bb];
            if (!A.c[k][b] || !A.gc[k][b]) return LINR_EINVAL;
            if (!linr_aligned16(A.c[k][b]) || !linr_aligned16(A.gc[k][b])) return LINR_EALIGN;
        }
        A.p[k] = p_h[kk]; A.target[k] = target_h[kk]; A.w1[k] = w1_h[kk]; A.b1[k] = b1_h[kk]; A.w2[k] = w2_h[kk];
        if (!A.p[k] || !A.target[k] || !A.w1[k] || !A.b1[k] || !A.w2[k]) return LINR_EINVAL;
    }
    A.target_ld = target_ld; A.gscale = gscale; A.slab = slab; A.block_stride = total;
    const dim3 grid(WH_BLOCKS, nstages);
    if (C == 16) whead_bwd_k<16><<<grid, LINR_BLOCK, 0, s>>>(A, n);
    else whead_bwd_k<32><<<grid, LINR_BLOCK, 0, s>>>(A, n);
    const int rc = linr_launch_rc();
    if (rc) return rc;
    return linr_slab_reduce_launch(slab, WH_BLOCKS, total, grads, s);
}

// dst (+)= src[0] + src[1] + ... over n floats, the sources added in list order (the gradient fan-in of x_glob: the seven stage priors'
// gradients in one pass instead of one read-modify-write pass each).
#define SM_MAX 8
struct SmArgs { const float* src[SM_MAX]; int count; };
__global__ __launch_bounds__(LINR_BLOCK) void sum_many_k(SmArgs a, int64_t n4, float* __restrict__ dst, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i >= n4) return;
    float4 t = accumulate ? reinterpret_cast<const float4*>(dst)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < a.count; ++k) {
        const float4 v = reinterpret_cast<const float4*>(a.src[k])[i];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    reinterpret_cast<float4*>(dst)[i] = t;
}

// src_h: HOST array of `count` (1..8) device pointers, each n floats (n a multiple of 4, everything 16-byte aligned).
extern "C" int linr_sum_many(const float* const* src_h, int32_t count, int64_t n, float* dst, int32_t accumulate, void* stream) {
    if (n < 0 || (n & 3) || count < 1 || count > SM_MAX || !src_h || !dst) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!linr_aligned16(dst)) return LINR_EALIGN;
    SmArgs a;
    a.count = count;
    for (int k = 0; k < SM_MAX; ++k) {
        a.src[k] = src_h[k < count ? k : 0];
        if (!a.src[k]) return LINR_EINVAL;
        if (!linr_aligned16(a.src[k])) return LINR_EALIGN;
    }
    linr_poison_hook((hipStream_t)stream, 16);
    sum_many_k<<<linr_grid(n / 4, LINR_BLOCK), LINR_BLOCK, 0, (hipStream_t)stream>>>(a, n / 4, dst, accumulate ? 1 : 0);
    return linr_launch_rc();
}

// bits_acc += (sum of `count` per-block nats partials) / ln 2 in a fixed order: the tail of linr_head_wide_fwd for callers that collected
// the partials of several stages (ws given, bits_acc NULL) and finish them in one launch.
extern "C" int linr_bits_finish(const double* partial, int64_t count, double* bits_acc, void* stream) {
    if (count < 0 || count > 0x7fffffff || !bits_acc || (count > 0 && !partial)) return LINR_EINVAL;
    if (count == 0) return 0;
    return linr_bits_finish_launch(partial, (int)count, bits_acc, (hipStream_t)stream);
}

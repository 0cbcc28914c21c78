// Executor-only kernels on the COMPRESSED kernel map (csrc/net.hip is their only caller).
//
// Compressed kernel map: the coordinate list is sorted x-major, so the up-to-three dz = -1,0,+1 neighbours of one
// (dx,dy) column are consecutive rows.  Per row: lo[q] = row of the first present neighbour of column q = (dx+1)+3(dy+1)
// (9 x int32) + a 27-bit presence mask = 40 B instead of 27 x int32 = 108 B.  The 36 MB table of a loot-like frame
// becomes 13.5 MB, i.e. one XCD's share (1.7 MB) stays L2-resident across the ~140 conv passes of a training step.
#include "common.h"
#include "conv_common.h"
#include "head_bwd.h"
#include <stdlib.h>
#define WG_WAVES 4          // waves per block of the MFMA weight-gradient kernels (same-box A/B: 2 -> 2.765, 4 -> 2.574, 8 -> 2.596 ms/step)

__global__ __launch_bounds__(LINR_BLOCK) void kmap_compress_k(const int32_t* __restrict__ nbr, int64_t nbr_ld, int64_t n,
                                                              int32_t* __restrict__ lo, uint32_t* __restrict__ mask,
                                                              int64_t ld) {
    const int64_t row = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (row >= n) return;
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int32_t a = nbr[(int64_t)q * nbr_ld + row];
        const int32_t b = nbr[(int64_t)(q + 9) * nbr_ld + row];
        const int32_t c = nbr[(int64_t)(q + 18) * nbr_ld + row];
        const uint32_t m3 = (a >= 0 ? 1u : 0u) | (b >= 0 ? 2u : 0u) | (c >= 0 ? 4u : 0u);
        lo[(int64_t)q * ld + row] = a >= 0 ? a : (b >= 0 ? b : (c >= 0 ? c : 0));
        m |= m3 << (3 * q);
    }
    mask[row] = m;
}

extern "C" int linr_kmap_compress(const int32_t* nbr, int64_t nbr_ld, int64_t n, int32_t* lo, uint32_t* mask, int64_t ld,
                                  void* stream) {
    if (n < 0 || nbr_ld < n || ld < n) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!nbr || !lo || !mask) return LINR_EINVAL;
    kmap_compress_k<<<linr_grid(n, LINR_BLOCK), LINR_BLOCK, 0, (hipStream_t)stream>>>(nbr, nbr_ld, n, lo, mask, ld);
    return linr_launch_rc();
}

// Plain conv3 forward / backward-data on the compressed map.  Same arithmetic order as spconv_gather_k (taps in LINR_TAP order,
// gathered channel ascending) => bit-identical results.  `in` must have the zero pad row at index -1.
//   BWD == false: acc[o] += x[i] * W[(k*GIN + i)*GOUT + o]
//   BWD == true : acc[o] += x[i] * W[(k*GOUT + o)*GIN + i]   (gathered rows come from the mirrored offset)
template <int GIN, int GOUT, bool BWD, int LOADW>
__global__ __launch_bounds__(LINR_BLOCK) void cconv_k(const float* __restrict__ in, int in_ld,
                                                      const int32_t* __restrict__ lo, const uint32_t* __restrict__ mask,
                                                      int64_t ld, int64_t n, const float* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ res,
                                                      int res_ld, const float* __restrict__ act, int act_ld,
                                                      float* __restrict__ out, int out_ld, unsigned flags) {
    const int64_t row = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (row >= n) return;
    const char* pad = reinterpret_cast<const char*>(in - in_ld);
    uint32_t off[27];
    decode_offsets<BWD>(lo, mask, ld, row, (uint32_t)in_ld * 4u, off);
    float acc[GOUT];
#pragma unroll
    for (int o = 0; o < GOUT; ++o) acc[o] = (bias != nullptr) ? bias[o] : 0.0f;
#pragma unroll
    for (int kk = 0; kk < 27; ++kk) {
        const int k = LINR_TAP(kk);
        float x[LOADW];
        RowLoadF<LOADW>::run(pad + off[k], x);
        const float* __restrict__ wk = W + k * GIN * GOUT;
#pragma unroll
        for (int i = 0; i < GIN; ++i) {
#pragma unroll
            for (int o = 0; o < GOUT; ++o) {
                const float w = BWD ? wk[o * GIN + i] : wk[i * GOUT + o];
                acc[o] = fmaf(x[i], w, acc[o]);
            }
        }
    }
    // epilogue order: + res, + old (ACCUM), * mask, ReLU
    if (res != nullptr) {
        const float* r = res + row * res_ld;
#pragma unroll
        for (int o = 0; o < GOUT; ++o) acc[o] += r[o];
    }
    float* op = out + row * out_ld;
    if (flags & LINR_ACCUM) {
#pragma unroll
        for (int o = 0; o < GOUT; ++o) acc[o] += op[o];
    }
    if (flags & LINR_RELU_MASK) {
        const float* a = act + row * act_ld;
#pragma unroll
        for (int o = 0; o < GOUT; ++o) acc[o] = a[o] > 0.0f ? acc[o] : 0.0f;
    }
    if (flags & LINR_RELU) {
#pragma unroll
        for (int o = 0; o < GOUT; ++o) acc[o] = fmaxf(acc[o], 0.0f);
    }
    if ((GOUT % 4 == 0) && (out_ld % 4 == 0)) {
#pragma unroll
        for (int v = 0; v < GOUT / 4; ++v)
            *reinterpret_cast<float4*>(op + 4 * v) = make_float4(acc[4 * v], acc[4 * v + 1], acc[4 * v + 2], acc[4 * v + 3]);
    } else {
#pragma unroll
        for (int o = 0; o < GOUT; ++o) op[o] = acc[o];
    }
}

// ---- the same convolution on the matrix cores ----------------------------------------------------------------------------
// v_mfma_f32_4x4x1_16b_f32 is 16 independent 4x4 outer products (K = 1); with CBSZ = 4 the A operand of block ABID is
// broadcast to all 16 blocks, so ONE instruction computes, for all 64 lanes at once,
//        acc[row(lane)][4*ABID + i] += W[ci][4*ABID + i] * x[row(lane)][ci]        i = 0..3
// i.e. 64 rows x 4 output channels x 1 input channel = 256 FMAs with NO padding (the 16-wide MFMA shapes waste half
// of their N dimension at Cout = 8).  The lane keeps the thread-per-row layout of the VALU kernel: B = the lane's
// gathered feature x[ci], D = the lane's 4 accumulators, A = the weight row held by lanes 0..GOUT-1 (read from an LDS
// copy of the whole [27][Cin][Cout] kernel).  K = 1 makes every instruction a single-rounding fmaf(x, w, acc), issued
// in the same order as the VALU kernel (taps in LINR_TAP order, ci ascending) => bit-identical results, at the MFMA rate
// (measured 91-119 TFLOP/s for this stream vs 52-71 TFLOP/s for v_pk_fma_f32; tools/mfma_probe.hip, valu_probe.hip).
// Measured dead ends for this kernel (kept out of the tree): hand-pinned software pipelines
// (2-3 offsets ahead, or a whole dz-plane of gathers in flight) and wave-cooperative staging of each (dx,dy) column's
// contiguous neighbour range through LDS were all slower than the compiler's own interleaving below.
// occupancy head fused behind the prune convolution (models/upsample.py:153-160 + models/model_core.py:76-81):
// z = w2 . relu(W1 c + b1) + b2, p = sigmoid(z), bits partial of the block.  EPI == 1 selects it.
struct HeadArgs {
    const float* w1;      // [24][8]  inner_mlps.k.0.0.weight
    const float* b1;      // [24]
    const float* w2;      // [24]     inner_mlps.k.0.2.weight
    const float* b2;      // [1]
    const float* target;  // occupancy column (stride target_ld) or nullptr (decoder: probabilities only)
    int target_ld;
    float* p_out;         // [n]
    double* partial;      // [gridDim.x] block partial sums of nats, or nullptr
};

#ifndef LINR_CONV_BLOCK
#define LINR_CONV_BLOCK 256
#endif
template <int GIN, int GOUT, bool BWD, int LOADW, int EPI = 0>
__global__ __launch_bounds__(LINR_CONV_BLOCK) void cconv_mfma_k(const float* __restrict__ in, int in_ld,
                                                           const int32_t* __restrict__ lo, const uint32_t* __restrict__ mask,
                                                           int64_t ld, int64_t n, const float* __restrict__ W,
                                                           const float* __restrict__ bias, const float* __restrict__ res,
                                                           int res_ld, const float* __restrict__ act, int act_ld,
                                                           float* __restrict__ out, int out_ld, unsigned flags,
                                                           HeadArgs hd = HeadArgs(), PwArgs pw = PwArgs(), Grp gp = Grp()) {
    static_assert(GOUT == 4 || GOUT == 8, "output channels must fill 1 or 2 MFMA blocks");
    {   // group offsets (all zero for a plain launch)
        const int gi = blockIdx.y;
        in += gp.in[gi]; W += gp.w[gi]; out += gp.out[gi];
        if (bias) bias += gp.b[gi];
        if (res) res += gp.res[gi];
        if (act) act += gp.act[gi];
        if constexpr (EPI == 1) {
            hd.w1 += gp.e0[gi]; hd.b1 += gp.e1[gi]; hd.w2 += gp.e2[gi]; hd.b2 += gp.e3[gi];
            if (hd.target) hd.target += gp.e4[gi];
            hd.p_out += gp.e5[gi];
            if (hd.partial) hd.partial += gp.e6[gi];
        }
        if constexpr (EPI == 2) { pw.w += gp.e0[gi]; pw.b += gp.e1[gi]; }
        if constexpr (EPI == 3) { pw.w += gp.e0[gi]; pw.aux += gp.e1[gi]; pw.aux_out += gp.e2[gi]; }
        if constexpr (EPI == 4) { pw.w += gp.e0[gi]; pw.aux += gp.e1[gi]; }
    }
    // All weights of the convolution live in registers for the whole kernel: the A operand of the 16-block MFMA is taken
    // from block ABID (an immediate), so ONE VGPR carries 16 different weight 4-vectors - block b of register wv[g][i]
    // holds W(k = g*KPV + b/HB, input i, outputs 4*(b%HB) .. +3).  27 x Cin x Cout floats = at most 32 VGPRs, loaded once;
    // the loop below has no weight traffic at all (an LDS copy + 4 ds_reads and waits per offset cost ~8 us per launch).
    constexpr int HB = GOUT / 4;                 // output halves (MFMA blocks) per offset
    constexpr int KPV = 16 / HB;                 // offsets packed per register
    constexpr int NG = (27 + KPV - 1) / KPV;
    const int lane = threadIdx.x & 63;
    float wv[NG][GIN];
    {
        const int blk = lane >> 2, j = lane & 3;
        const int kl = blk / HB, co = 4 * (blk % HB) + j;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int k = g * KPV + kl;
#pragma unroll
            for (int i = 0; i < GIN; ++i)
                wv[g][i] = (k < 27) ? (BWD ? W[(k * GOUT + co) * GIN + i] : W[(k * GIN + i) * GOUT + co]) : 0.0f;
        }
    }
    const int64_t row_raw = (int64_t)blockIdx.x * LINR_CONV_BLOCK + threadIdx.x;
    const bool live = row_raw < n;
    const int64_t row = live ? row_raw : n - 1;          // every lane stays in the MFMAs (they ignore EXEC)
    const char* pad = reinterpret_cast<const char*>(in - in_ld);
    const uint32_t rowbytes = (uint32_t)in_ld * 4u;
    uint32_t off[27];
    decode_offsets<BWD>(lo, mask, ld, row, rowbytes, off);
    f32x4 acc[GOUT / 4];
#pragma unroll
    for (int h = 0; h < GOUT / 4; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[h][j] = (bias != nullptr) ? bias[4 * h + j] : 0.0f;
    constexpr int PF = 4;
    float x[PF + 1][LOADW];
    // Left to itself hipcc waits (vmcnt(0)) right after every 16-byte gather - 54 serial round trips per wave.  The loop
    // is therefore pipelined by hand: the load of offset k+PF is issued before the MFMAs of offset k and sched_barrier
    // keeps it there, so the compiler's own counted waits leave PF rows in flight.
#pragma unroll
    for (int u = 0; u < PF; ++u) RowLoadF<LOADW>::run(pad + off[LINR_TAP(u)], x[u]);
    __builtin_amdgcn_sched_barrier(0);
    static_for<27>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;          // step; k = the tap it handles (common.h: LINR_TAP)
        constexpr int k = LINR_TAP(kk);
        constexpr int g = k / KPV, ab = (k % KPV) * HB;
        if constexpr (kk + PF < 27) RowLoadF<LOADW>::run(pad + off[LINR_TAP(kk + PF)], x[(kk + PF) % (PF + 1)]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < GIN; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], x[kk % (PF + 1)][i], acc[0], 4, ab, 0);
            if constexpr (GOUT == 8)
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], x[kk % (PF + 1)][i], acc[1], 4, ab + 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    float a[GOUT];
#pragma unroll
    for (int h = 0; h < GOUT / 4; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[4 * h + j] = acc[h][j];
    if constexpr (EPI == 1) {
        // ---- fused occupancy head: the conv output row a[0..8) is C_k --------------------------------------------
        float* op = out + row * out_ld;
        if (live) {
            *reinterpret_cast<float4*>(op) = make_float4(a[0], a[1], a[2], a[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(a[4], a[5], a[6], a[7]);
        }
        float z = hd.b2[0];
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            float hj = hd.b1[j];
#pragma unroll
            for (int i = 0; i < 8; ++i) hj = fmaf(a[i], hd.w1[j * 8 + i], hj);
            z = fmaf(fmaxf(hj, 0.0f), hd.w2[j], z);
        }
        const float p = 1.0f / (1.0f + expf(-z));
        if (live) hd.p_out[row] = p;
        if (hd.partial != nullptr) {          // wave-uniform (kernel argument)
            __shared__ double sred[LINR_CONV_BLOCK / 64];
            double nats = 0.0;
            if (live) {
                const float t = hd.target[row * hd.target_ld];
                nats = (double)((t - 1.0f) * fmaxf(logf(1.0f - p), -100.0f) - t * fmaxf(logf(p), -100.0f));
            }
            // fixed shuffle tree inside the wave, then the waves in order => bit-reproducible
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) nats += __shfl_xor(nats, d, 64);
            if (lane == 0) sred[threadIdx.x >> 6] = nats;
            __syncthreads();
            if (threadIdx.x == 0) {
                double tot = sred[0];
                for (int w = 1; w < LINR_CONV_BLOCK / 64; ++w) tot += sred[w];
                hd.partial[blockIdx.x] = tot;
            }
        }
        return;
    } else {
    if (!live) return;
    // epilogue order: + res, + old (ACCUM), [+ own-row pointwise term], * mask, ReLU
    if (res != nullptr) {
        const float* r = res + row * res_ld;
#pragma unroll
        for (int o = 0; o < GOUT; ++o) a[o] += r[o];
    }
    float* op = out + row * out_ld;
    if (flags & LINR_ACCUM) {
#pragma unroll
        for (int o = 0; o < GOUT; ++o) a[o] += op[o];
    }
    if constexpr (EPI == 4) {          // gA += gH[row][4:8] @ W10^T     (W10 [8][4]: gin[i] = sum_o g[o] * W10[i][o])
        const float4 g4 = *reinterpret_cast<const float4*>(pw.aux + row * 8 + 4);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = 0.0f;
#pragma unroll
            for (int o = 0; o < 4; ++o) t = fmaf(g[o], pw.w[i * 4 + o], t);
            a[i] += t;
        }
    }
    if constexpr (EPI == 3) {          // store gI, then gM = (gI[4:8] @ W12^T) * (M > 0)   (W12 [4][4])
        const float4 m4 = *reinterpret_cast<const float4*>(pw.aux + row * 4);
        const float mv[4] = {m4.x, m4.y, m4.z, m4.w};
        float gm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = 0.0f;
#pragma unroll
            for (int o = 0; o < 4; ++o) t = fmaf(a[4 + o], pw.w[i * 4 + o], t);
            gm[i] = mv[i] > 0.0f ? t : 0.0f;
        }
        *reinterpret_cast<float4*>(pw.aux_out + row * 4) = make_float4(gm[0], gm[1], gm[2], gm[3]);
    }
    if constexpr (EPI == 2) {          // conv0_0 half: ReLU, store; conv1_0 half: relu(in[row] @ W10 + b10)
        const char* self = reinterpret_cast<const char*>(in) + (uint32_t)row * ((uint32_t)in_ld * 4u);
        float xc[8];
        RowLoadF<8>::run(self, xc);
        float h1[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) h1[o] = pw.b[o];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int o = 0; o < 4; ++o) h1[o] = fmaf(xc[i], pw.w[i * 4 + o], h1[o]);
        *reinterpret_cast<float4*>(op) = make_float4(fmaxf(a[0], 0.0f), fmaxf(a[1], 0.0f), fmaxf(a[2], 0.0f), fmaxf(a[3], 0.0f));
        *reinterpret_cast<float4*>(op + 4) = make_float4(fmaxf(h1[0], 0.0f), fmaxf(h1[1], 0.0f), fmaxf(h1[2], 0.0f), fmaxf(h1[3], 0.0f));
        return;
    }
    if (flags & LINR_RELU_MASK) {
        const float* m = act + row * act_ld;
#pragma unroll
        for (int o = 0; o < GOUT; ++o) a[o] = m[o] > 0.0f ? a[o] : 0.0f;
    }
    if (flags & LINR_RELU) {
#pragma unroll
        for (int o = 0; o < GOUT; ++o) a[o] = fmaxf(a[o], 0.0f);
    }
    if (out_ld % 4 == 0) {
#pragma unroll
        for (int v = 0; v < GOUT / 4; ++v)
            *reinterpret_cast<float4*>(op + 4 * v) = make_float4(a[4 * v], a[4 * v + 1], a[4 * v + 2], a[4 * v + 3]);
    } else {
#pragma unroll
        for (int o = 0; o < GOUT; ++o) op[o] = a[o];
    }
    }
}

extern "C" int linr_spconv_cmap(int32_t bwd, const float* in, int32_t in_ld, const int32_t* lo, const uint32_t* mask, int64_t ld,
                                int64_t n, const float* W, const float* bias, int32_t cin, int32_t cout, const float* res,
                                int32_t res_ld, const float* act, int32_t act_ld, float* out, int32_t out_ld, uint32_t flags,
                                void* stream) {
    if (n < 0 || ld < n || (in_ld != 4 && in_ld != 8)) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!in || !lo || !mask || !W || !out) return LINR_EINVAL;
    if (!linr_aligned16(in) || !linr_aligned16(out)) return LINR_EALIGN;
    const int gin = bwd ? cout : cin, gout = bwd ? cin : cout;
    if (in_ld < gin || out_ld < gout || (gout != 4 && gout != 8)) return LINR_EINVAL;
    if ((flags & LINR_RELU_MASK) && (!act || act_ld < gout)) return LINR_EINVAL;
    if (res && res_ld < gout) return LINR_EINVAL;
    if ((uint64_t)(n + 1) * (uint64_t)in_ld * 4u >= 0xFFFFFFFFull) return LINR_EINVAL;     // 32-bit byte offsets
    return linr_cconv_launch(bwd != 0, in, in_ld, lo, mask, ld, n, W, bias, cin, cout, res, res_ld, act, act_ld, out, out_ld,
                             flags, (hipStream_t)stream, nullptr, 1);
}

// The executor's backward-weight kernel through its own entry: per-block partial sums of
//   gW[k][ci][co] = sum_r in[nbr_k(r)][ci] gout[r][co],  gb[co] = sum_r gout[r][co]
// into slab[b * (27 cin + 1) cout + (k cin + ci) cout + co] (bias row last), b < LINR_WG_BLOCKS (512); the caller sums the
// 512 partials in ascending order (the executor does so for all parameters at once in wgrad_reduce_k).
extern "C" int64_t linr_spconv_wgrad_cmap_blocks(void) { return LINR_WG_BLOCKS; }

extern "C" int linr_spconv_wgrad_cmap(const float* in, int32_t in_ld, const float* gout, int32_t gout_ld, const int32_t* nbr,
                                      const int32_t* tile8t, int64_t ld, int64_t n, int32_t cin, int32_t cout, float* slab,
                                      void* stream) {
    if (n < 0 || ld < n || in_ld != 8 || gout_ld < cout) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!in || !gout || !nbr || !slab) return LINR_EINVAL;
    if (!linr_aligned16(in)) return LINR_EALIGN;
    if (!((cin == 8 && (cout == 8 || cout == 4)) || (cin < 8 && cin >= 1 && cout == 8))) return LINR_EINVAL;
    if ((uint64_t)(n + 1) * (uint64_t)in_ld * 4u >= 0xFFFFFFFFull) return LINR_EINVAL;
    const int64_t elems = (int64_t)(27 * cin + 1) * cout;
    LinrWgradDst d = {slab, elems, 0, (int64_t)27 * cin * cout, cin};
    if (tile8t && !linr_aligned16(tile8t)) return LINR_EALIGN;
    return linr_conv3_wgrad_mfma(in, in_ld, gout, gout_ld, nbr, ld, n, cin, cout, d, LINR_WG_BLOCKS, (hipStream_t)stream, nullptr, 1, tile8t);
}

// prune conv 8->8 + head of stage k in one launch; partial: [linr_grid(n,256)] doubles or nullptr
int linr_cconv_head_launch(const float* in, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                           const float* W, const float* bias, float* c_out, const float* w1, const float* b1,
                           const float* w2, const float* b2, const float* target, int target_ld, float* p_out,
                           double* partial, hipStream_t s, const Grp* gp, int ngroups) {
    if (n == 0) return 0;
    const Grp g0 = gp ? *gp : Grp();
    HeadArgs hd = {w1, b1, w2, b2, target, target_ld, p_out, partial};
    cconv_mfma_k<8, 8, false, 8, 1><<<dim3(linr_grid(n, LINR_CONV_BLOCK), ngroups), LINR_CONV_BLOCK, 0, s>>>(
        in, 8, lo, mask, ld, n, W, bias, nullptr, 0, nullptr, 0, c_out, 8, 0, hd, PwArgs(), g0);
    return linr_launch_rc();
}

// ---- the two 4->4 convolutions of the Inception block as ONE pass --------------------------------------------------------
// forward (models/resnet.py:56-57): in = H [n][8];  I[:,0:4] = conv(H[:,0:4]; W01) + b01 + A[:,0:4]
//                                   M = relu(conv(H[:,4:8]; W11) + b11);  I[:,4:8] = M @ W12 + b12 + A[:,4:8]
// backward: gH = [bwd(gI[:,0:4]; W01) | bwd(gM; W11)] * (H > 0), gathered at the mirrored offsets from two matrices.
// One 32-byte (or 2 x 16-byte) gather per neighbour serves both convolutions; lanes 0-3 hold W01's tap, lanes 4-7 W11's.
struct DualArgs {
    const float* in2; int in2_ld;      // bwd: second gathered matrix (gM, ld 4); fwd: unused
    const float* w01; const float* w11;
    const float* b01; const float* b11;
    const float* a_res;                // fwd: A [n][8] (residual)
    const float* w12; const float* b12;
    float* m_out;                      // fwd: M [n][4]
    const float* act;                  // bwd: H [n][8] for the ReLU mask
};

template <bool BWD>
__global__ __launch_bounds__(LINR_BLOCK) void cconv_dual44_k(const float* __restrict__ in, int in_ld,
                                                             const int32_t* __restrict__ lo, const uint32_t* __restrict__ mask,
                                                             int64_t ld, int64_t n, DualArgs d, float* __restrict__ out,
                                                             Grp gp = Grp()) {
    {   // group offsets: in, e0 = in2, w = w01, e1 = w11, b = b01, e2 = b11, res = a_res, e3 = w12, e4 = b12, e5 = m_out, act, out
        const int gi = blockIdx.y;
        in += gp.in[gi]; out += gp.out[gi];
        if (d.in2) d.in2 += gp.e0[gi];
        d.w01 += gp.w[gi]; d.w11 += gp.e1[gi];
        if (d.b01) d.b01 += gp.b[gi];
        if (d.b11) d.b11 += gp.e2[gi];
        if (d.a_res) d.a_res += gp.res[gi];
        if (d.w12) d.w12 += gp.e3[gi];
        if (d.b12) d.b12 += gp.e4[gi];
        if (d.m_out) d.m_out += gp.e5[gi];
        if (d.act) d.act += gp.act[gi];
    }
    // register-resident weights (see cconv_mfma_k): block b of wv[g][i] holds, for offset k = g*8 + b/2, the tap of
    // W01 (b even) or W11 (b odd) for input i and outputs 0..3
    const int lane = threadIdx.x & 63;
    float wv[4][4];
    {
        const int blk = lane >> 2, j = lane & 3;
        const int kl = blk >> 1;
        const float* Wsel = (blk & 1) ? d.w11 : d.w01;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k = g * 8 + kl;
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[g][i] = (k < 27) ? (BWD ? Wsel[(k * 4 + j) * 4 + i] : Wsel[(k * 4 + i) * 4 + j]) : 0.0f;
        }
    }
    const int64_t row_raw = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    const bool live = row_raw < n;
    const int64_t row = live ? row_raw : n - 1;
    uint32_t off[27];
    decode_offsets<BWD>(lo, mask, ld, row, 1u, off);                  // row index + 1 (0 = pad row); scaled per matrix below
    const char* pad0 = reinterpret_cast<const char*>(in - in_ld);
    // row pitches are 16 or 32 bytes: scale by wave-uniform shifts (v_mul_lo_u32 is quarter rate)
    const uint32_t rb0 = __builtin_amdgcn_readfirstlane(in_ld == 8 ? 5u : 4u);
    const char* pad1 = BWD ? reinterpret_cast<const char*>(d.in2 - d.in2_ld) : pad0 + 16;     // fwd: second half of the same row
    const uint32_t rb1 = BWD ? __builtin_amdgcn_readfirstlane(d.in2_ld == 8 ? 5u : 4u) : rb0;
    f32x4 acc0, acc1;
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc0[j] = BWD ? 0.0f : d.b01[j]; acc1[j] = BWD ? 0.0f : d.b11[j]; }
    constexpr int PF = 3;                             // gathers run PF offsets ahead of the MFMAs (see cconv_mfma_k)
    float x0[PF + 1][4], x1[PF + 1][4];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        RowLoadF<4>::run(pad0 + (off[LINR_TAP(u)] << rb0), x0[u]);
        RowLoadF<4>::run(pad1 + (off[LINR_TAP(u)] << rb1), x1[u]);
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<27>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;           // step; k = weight tap (decode_offsets already mirrored `off` for BWD)
        constexpr int k = LINR_TAP(kk);
        constexpr int g = k / 8, ab = (k % 8) * 2;
        if constexpr (kk + PF < 27) {
            RowLoadF<4>::run(pad0 + (off[LINR_TAP(kk + PF)] << rb0), x0[(kk + PF) % (PF + 1)]);
            RowLoadF<4>::run(pad1 + (off[LINR_TAP(kk + PF)] << rb1), x1[(kk + PF) % (PF + 1)]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], x0[kk % (PF + 1)][i], acc0, 4, ab, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], x1[kk % (PF + 1)][i], acc1, 4, ab + 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if (!live) return;
    float* op = out + row * 8;
    if (BWD) {
        const float4 h0 = *reinterpret_cast<const float4*>(d.act + row * 8);
        const float4 h1 = *reinterpret_cast<const float4*>(d.act + row * 8 + 4);
        *reinterpret_cast<float4*>(op) = make_float4(h0.x > 0.0f ? acc0[0] : 0.0f, h0.y > 0.0f ? acc0[1] : 0.0f,
                                                     h0.z > 0.0f ? acc0[2] : 0.0f, h0.w > 0.0f ? acc0[3] : 0.0f);
        *reinterpret_cast<float4*>(op + 4) = make_float4(h1.x > 0.0f ? acc1[0] : 0.0f, h1.y > 0.0f ? acc1[1] : 0.0f,
                                                         h1.z > 0.0f ? acc1[2] : 0.0f, h1.w > 0.0f ? acc1[3] : 0.0f);
    } else {
        const float4 a0 = *reinterpret_cast<const float4*>(d.a_res + row * 8);
        const float4 a1 = *reinterpret_cast<const float4*>(d.a_res + row * 8 + 4);
        const float m[4] = {fmaxf(acc1[0], 0.0f), fmaxf(acc1[1], 0.0f), fmaxf(acc1[2], 0.0f), fmaxf(acc1[3], 0.0f)};
        *reinterpret_cast<float4*>(d.m_out + row * 4) = make_float4(m[0], m[1], m[2], m[3]);
        float i1[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) i1[o] = d.b12[o];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int o = 0; o < 4; ++o) i1[o] = fmaf(m[i], d.w12[i * 4 + o], i1[o]);
        *reinterpret_cast<float4*>(op) = make_float4(acc0[0] + a0.x, acc0[1] + a0.y, acc0[2] + a0.z, acc0[3] + a0.w);
        *reinterpret_cast<float4*>(op + 4) = make_float4(i1[0] + a1.x, i1[1] + a1.y, i1[2] + a1.z, i1[3] + a1.w);
    }
}

int linr_dual44_fwd_launch(const float* H, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, const float* w01,
                           const float* b01, const float* w11, const float* b11, const float* A, const float* w12,
                           const float* b12, float* M, float* I, hipStream_t s, const Grp* gp, int ngroups) {
    if (n == 0) return 0;
    const Grp g0 = gp ? *gp : Grp();
    DualArgs d = {nullptr, 0, w01, w11, b01, b11, A, w12, b12, M, nullptr};
    cconv_dual44_k<false><<<dim3(linr_grid(n, LINR_BLOCK), ngroups), LINR_BLOCK, 0, s>>>(H, 8, lo, mask, ld, n, d, I, g0);
    return linr_launch_rc();
}

int linr_dual44_bwd_launch(const float* gI, const float* gM, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                           const float* w01, const float* w11, const float* H, float* gH, hipStream_t s, const Grp* gp,
                           int ngroups) {
    if (n == 0) return 0;
    const Grp g0 = gp ? *gp : Grp();
    DualArgs d = {gM, 4, w01, w11, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, H};
    cconv_dual44_k<true><<<dim3(linr_grid(n, LINR_BLOCK), ngroups), LINR_BLOCK, 0, s>>>(gI, 8, lo, mask, ld, n, d, gH, g0);
    return linr_launch_rc();
}

// conv0_0 (8->4) + conv1_0 (1x1 8->4) forward with both ReLUs: H = [relu(conv3(A)) | relu(A @ W10 + b10)]
int linr_conv_pw_fwd_launch(const float* A, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, const float* w00,
                            const float* b00, const float* w10, const float* b10, float* H, hipStream_t s, const Grp* gp,
                            int ngroups) {
    if (n == 0) return 0;
    const Grp g0 = gp ? *gp : Grp();
    PwArgs pw = {w10, b10, nullptr, nullptr};
    cconv_mfma_k<8, 4, false, 8, 2><<<dim3(linr_grid(n, LINR_CONV_BLOCK), ngroups), LINR_CONV_BLOCK, 0, s>>>(
        A, 8, lo, mask, ld, n, w00, b00, nullptr, 0, nullptr, 0, H, 8, 0, HeadArgs(), pw, g0);
    return linr_launch_rc();
}

// backward of the block's tail conv: gI = bwd(gO; Wb) and gM = (gI[:,4:8] @ W12^T) * (M > 0)
int linr_conv_bwd_gm_launch(const float* gO, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, const float* wb,
                            const float* w12, const float* M, float* gI, float* gM, hipStream_t s, const Grp* gp,
                            int ngroups) {
    if (n == 0) return 0;
    const Grp g0 = gp ? *gp : Grp();
    PwArgs pw = {w12, nullptr, M, gM};
    cconv_mfma_k<8, 8, true, 8, 3><<<dim3(linr_grid(n, LINR_CONV_BLOCK), ngroups), LINR_CONV_BLOCK, 0, s>>>(
        gO, 8, lo, mask, ld, n, wb, nullptr, nullptr, 0, nullptr, 0, gI, 8, 0, HeadArgs(), pw, g0);
    return linr_launch_rc();
}

// gA = (bwd(gH[:,0:4]; W00) + gI (+ old gA: LINR_ACCUM) + gH[:,4:8] @ W10^T) (* (A > 0): LINR_RELU_MASK)
int linr_conv_bwd_ga_launch(const float* gH, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, const float* w00,
                            const float* w10, const float* gI, const float* A, float* gA, unsigned flags, hipStream_t s,
                            const Grp* gp, int ngroups) {
    if (n == 0) return 0;
    if ((flags & LINR_RELU_MASK) && !A) return LINR_EINVAL;
    const Grp g0 = gp ? *gp : Grp();
    PwArgs pw = {w10, nullptr, gH, nullptr};
    cconv_mfma_k<4, 8, true, 4, 4><<<dim3(linr_grid(n, LINR_CONV_BLOCK), ngroups), LINR_CONV_BLOCK, 0, s>>>(
        gH, 8, lo, mask, ld, n, w00, nullptr, gI, 8, A, 8, gA, 8, flags & (LINR_RELU_MASK | LINR_ACCUM), HeadArgs(), pw, g0);
    return linr_launch_rc();
}

// ---- first convolutions of the 7 outter blocks: one gather, 56 outputs -----------------------------------------------------
// Block b (1..7) starts with conv3(occ[:, :b] -> 8) + ReLU on the SAME occupancy rows (models/upsample.py:206-214), so the
// grouped forward gathers each neighbour's 8 occupancy floats once and feeds all 7 kernels: per offset 28 (block, input
// channel) pairs x 2 output quads = 56 MFMA blocks instead of 7 x 16 with zero-extended kernels, and 1/7 of the gathers.
// The weights sit in LDS as the A-operand image wl[k][v][lane]: block (lane >> 2) of register v is combo 16 v + block,
// combo c <-> pair p = c / 2 (block g = tri^-1(p), channel ci = p - g (g + 1) / 2), quad h = c % 2.  Per output the chain is
// bias, then taps in LINR_TAP order, ci ascending fmaf - the chain of cconv_mfma_k on that block alone, so the decoder's
// block-by-block forward gives the same bits.
struct Occ7Args { int64_t w[7], b[7], out[7]; };       // parameter offsets (kernel, bias) and output element offsets per block

__host__ __device__ constexpr int occ7_g(int p) { return p < 1 ? 0 : p < 3 ? 1 : p < 6 ? 2 : p < 10 ? 3 : p < 15 ? 4 : p < 21 ? 5 : 6; }

__global__ __launch_bounds__(LINR_CONV_BLOCK) void occ_conv7_k(const float* __restrict__ occ, const int32_t* __restrict__ lo,
                                                               const uint32_t* __restrict__ mask, int64_t ld, int64_t n,
                                                               const float* __restrict__ P, Occ7Args a,
                                                               float* __restrict__ out) {
    __shared__ float wl[27 * 4 * 64];
    {   // the weight image: a thread's 27 loads all in flight, then the LDS stores (as a rolled loop every element waited for two
        // dependent loads - the block's parameter offset out of the argument struct, then the weight: 54 round trips to the L2 in
        // front of the first tile).  Element e = threadIdx.x + 256 i: the combo (e & 255) >> 2 and j = e & 3 do not depend on i.
        const int c = (int)(threadIdx.x >> 2), j = (int)(threadIdx.x & 3);
        const int pr = c >> 1, h = c & 1, g = occ7_g(pr), ci = pr - g * (g + 1) / 2;
        int64_t wg = a.w[0];
#pragma unroll
        for (int q = 1; q < 7; ++q) wg = (g == q) ? a.w[q] : wg;          // constant indices: scalar loads + selects
        const float* src = P + wg + ci * 8 + 4 * h + j;
        const int kstride = (g + 1) * 8;
        float wv[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) wv[k] = c < 56 ? src[k * kstride] : 0.0f;
#pragma unroll
        for (int k = 0; k < 27; ++k) wl[k * 256 + threadIdx.x] = wv[k];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const char* pad = reinterpret_cast<const char*>(occ - 8);
    // the 27 KB weight image is built once per workgroup: the launch gives every workgroup several row tiles (linr_occ_conv7_launch)
    const int64_t tiles = (n + LINR_CONV_BLOCK - 1) / LINR_CONV_BLOCK;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t row_raw = tile * LINR_CONV_BLOCK + threadIdx.x;
    const bool live = row_raw < n;
    const int64_t row = live ? row_raw : n - 1;          // every lane stays in the MFMAs (they ignore EXEC)
    uint32_t off[27];
    decode_offsets<false>(lo, mask, ld, row, 32u, off);
    f32x4 acc[7][2];
#pragma unroll
    for (int g = 0; g < 7; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[g][h][j] = P[a.b[g] + 4 * h + j];
    constexpr int PF = 3;
    float x[PF + 1][8];
    float wr[2][4];
#pragma unroll
    for (int u = 0; u < PF; ++u) RowLoadF<8>::run(pad + off[LINR_TAP(u)], x[u]);
#pragma unroll
    for (int v = 0; v < 4; ++v) wr[0][v] = wl[(LINR_TAP(0) * 4 + v) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    static_for<27>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;            // step; tap LINR_TAP(kk)
        if constexpr (kk + PF < 27) RowLoadF<8>::run(pad + off[LINR_TAP(kk + PF)], x[(kk + PF) % (PF + 1)]);
        if constexpr (kk + 1 < 27) {
#pragma unroll
            for (int v = 0; v < 4; ++v) wr[(kk + 1) & 1][v] = wl[(LINR_TAP(kk + 1) * 4 + v) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        // input channel outermost: consecutive MFMAs write different accumulators (no back-to-back dependent issue), and
        // every output still sees its channels in ascending order
        static_for<7>([&](auto cic) {
            constexpr int ci = decltype(cic)::value;
            static_for<7 - ci>([&](auto gc) {
                constexpr int g = ci + decltype(gc)::value;
                static_for<2>([&](auto hc) {
                    constexpr int h = decltype(hc)::value;
                    constexpr int c = 2 * (g * (g + 1) / 2 + ci) + h;
                    acc[g][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[kk & 1][c / 16], x[kk % (PF + 1)][ci], acc[g][h], 4, c % 16, 0);
                });
            });
        });
        __builtin_amdgcn_sched_barrier(0);
    });
    if (live) {
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            float* op = out + a.out[g] + row * 8;
            *reinterpret_cast<float4*>(op) = make_float4(fmaxf(acc[g][0][0], 0.0f), fmaxf(acc[g][0][1], 0.0f),
                                                         fmaxf(acc[g][0][2], 0.0f), fmaxf(acc[g][0][3], 0.0f));
            *reinterpret_cast<float4*>(op + 4) = make_float4(fmaxf(acc[g][1][0], 0.0f), fmaxf(acc[g][1][1], 0.0f),
                                                             fmaxf(acc[g][1][2], 0.0f), fmaxf(acc[g][1][3], 0.0f));
        }
    }
    }
}

// occ: arena copy of the occupancy [n][8] with the zero pad row in front; w_off / b_off: parameter offsets of the 7 first
// convolutions (kernel [27][b][8] of block b) ; out + out_off[g]: A matrix of block g + 1
int linr_occ_conv7_launch(const float* occ, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, const float* P,
                          const int64_t* w_off, const int64_t* b_off, float* out, const int64_t* out_off, hipStream_t s) {
    if (n == 0) return 0;
    Occ7Args a;
    for (int g = 0; g < 7; ++g) { a.w[g] = w_off[g]; a.b[g] = b_off[g]; a.out[g] = out_off[g]; }
    // MFMA-bound (1512 per 64 rows): two workgroups per CU keep the matrix cores fed, and each amortises its weight image over
    // tiles / grid row tiles (same-box A/B at 337 k rows, ms/step: one tile per workgroup 1.732, three workgroups per CU 1.725,
    // two 1.716, one 1.720)
    constexpr int per_cu = 2;
    static const int cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
        return v;
    }();
    const int64_t tiles = linr_grid(n, LINR_CONV_BLOCK);
    const int64_t want = (int64_t)cus * per_cu;
    const int64_t per = (tiles + want - 1) / want;                 // tiles per workgroup
    const int64_t grid = (tiles + per - 1) / per;
    occ_conv7_k<<<(unsigned)grid, LINR_CONV_BLOCK, 0, s>>>(occ, lo, mask, ld, n, P, a, out);
    return linr_launch_rc();
}

// ---- backward-weight on the matrix cores -------------------------------------------------------------------------------------
// gW[k][ci][co] = sum_r x[nbr[k][r]][ci] * g[r][co].  Same wave-per-row-group organisation as spconv_wgrad_k (lane = one
// (offset k, channel quad q) pair, 16-byte gather of that quad for 8 rows at a time, persistent accumulators), but the
// 4 x COUT outer product per lane and row runs as v_mfma_f32_4x4x1_16b_f32 with the A operand broadcast from one block:
//     D[i] on lane l += A(lane 4*abid + i) * B(lane l)          (CBSZ = 4: one block feeds all 16; CBSZ = 3: one per half)
// B = component c of the lane's own gathered quad, A = the output gradient: a wave loads the g rows of its 8-row group
// with ONE coalesced dword load (lane l holds g[row l / COUT][l % COUT]), so block 2u + h (COUT 8) or u (COUT 4) already
// holds g[row u][4h .. 4h+3] and ABID selects it - no per-row gradient loads, no shuffles.  Register i of accumulator
// (c, h) on a lane is gW[k][4q + c][4h + i] of the lane's own pair.  DUAL (the two 4->4 convs of an Inception block):
// lanes 0..31 are conv 0, lanes 32..63 conv 1, each half loads its own gradient matrix and CBSZ = 3 keeps them apart.
// The bias gradient is the column sum of the same gradient tiles (each lane adds up the element it loads; a fixed shuffle
// tree and the waves in order finish it) - no per-row selects in the loop: VALU instructions run on the same FMA units as
// the f32 MFMAs, so every one of them is paid for.  K = 1 keeps exact fp32 FMAs in row order.
struct WgradSrc {
    const float* in; int in_ld;           // gathered matrix (quad q at column 4q)
    const float* g0; int g0_ld;           // output gradient (DUAL: of conv 0)
    const float* g1; int g1_ld;           // DUAL: output gradient of conv 1
};
struct WgradDual { int64_t w_off1, b_off1; };

// IDX: how a lane reads its neighbour indices from nbr[27][ld] - 0: scalar loads, 1: 16-byte loads (table 16-byte aligned, ld % 4 == 0).
// This is the direct-gather form: the fallback for frames without the transposed tiled table and the bitwise reference of
// spconv_wgrad_t_k below.
template <int XQ, int COUT, bool DUAL, int IDX>
__global__ __launch_bounds__(WG_WAVES * 64) void spconv_wgrad_mfma_k(WgradSrc S, const int32_t* __restrict__ nbr, int64_t nbr_ld,
                                                                    int64_t n, LinrWgradDst d, WgradDual dd, Grp gp = Grp()) {
    static_assert(!DUAL || (XQ == 2 && COUT == 4), "dual mode = two 4->4 convolutions");
    {   // group offsets: in, res = g0, act = g1, w/b = slab offsets of conv 0, e0/e1 = of conv 1, e2 = cin_valid override
        const int gi = blockIdx.y;
        S.in += gp.in[gi]; S.g0 += gp.res[gi];
        if (S.g1) S.g1 += gp.act[gi];
        d.w_off += gp.w[gi]; d.b_off += gp.b[gi];
        dd.w_off1 += gp.e0[gi]; dd.b_off1 += gp.e1[gi];
        if (gp.e2[gi] > 0) d.cin_valid = (int)gp.e2[gi];
    }
    constexpr int HB = COUT / 4;
    constexpr int NA = 4 * HB * 4;
    constexpr int CBSZ = DUAL ? 3 : 4;
    __shared__ float sacc[64 * (NA + 1)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = DUAL ? (lane >> 5) : (XQ == 2 ? (lane & 1) : 0);
    const int kk = DUAL ? (lane & 31) : (XQ == 2 ? (lane >> 1) : lane);          // >= 27: idle lanes
    const bool live = kk < 27;
    const int k = live ? kk : 26;
    f32x4 acc[4][HB];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < HB; ++h) acc[c][h] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    int64_t per = (n + gridDim.x - 1) / gridDim.x;
    per = (per + 7) & ~(int64_t)7;
    const int64_t b0 = (int64_t)blockIdx.x * per;
    const int64_t b1 = (b0 + per < n) ? b0 + per : n;
    const int32_t* nk = nbr + (int64_t)k * nbr_ld;
    const char* pad = reinterpret_cast<const char*>(S.in - S.in_ld) + 16 * q;
    const uint32_t rsh = __builtin_amdgcn_readfirstlane(S.in_ld == 8 ? 5u : 4u);      // 32- or 16-byte rows: a uniform shift, not a multiply
    // this lane's element of the 8-row gradient tile: row gu, channel gc of matrix gsel
    const float* gsel = (DUAL && q) ? S.g1 : S.g0;
    const int gld = (DUAL && q) ? S.g1_ld : S.g0_ld;
    const int gl = DUAL ? (lane & 31) : lane;
    const int gu = (gl / COUT) & 7, gc = gl % COUT;
    const int ncomp = __builtin_amdgcn_readfirstlane(d.cin_valid);      // live components per quad (>= 4: all)
    float bsum = 0.0f;
    for (int64_t g0r = b0 + 8 * wave; g0r < b1; g0r += 8 * WG_WAVES) {
        int32_t idx[8];
        if (IDX >= 1 && g0r + 8 <= n) {
            const int4 a = *reinterpret_cast<const int4*>(nk + g0r);
            const int4 b = *reinterpret_cast<const int4*>(nk + g0r + 4);
            idx[0] = a.x; idx[1] = a.y; idx[2] = a.z; idx[3] = a.w;
            idx[4] = b.x; idx[5] = b.y; idx[6] = b.z; idx[7] = b.w;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) idx[u] = (g0r + u < n) ? nk[g0r + u] : -1;
        }
        const float gv = (g0r + gu < n) ? gsel[(g0r + gu) * gld + gc] : 0.0f;
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = *reinterpret_cast<const float4*>(pad + ((uint32_t)(idx[u] + 1) << rsh));
        }
        bsum += gv;                      // bias gradient: column sums of the gradient rows (lanes beyond the 27 offsets
                                         // gather offset 26's rows again; their products are never written)
        static_for<8>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            static_for<HB>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                constexpr int ab = u * HB + h;           // the block holding g[row u][4h .. 4h+3]
                // first convs of the outter blocks (cin_valid = 1..7): component c of a quad is input channel >= c, so it
                // is dead in BOTH quads once c >= cin_valid (wave-uniform: cin_valid is a kernel argument)
                acc[0][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, x[u].x, acc[0][h], CBSZ, ab, 0);
                if (DUAL || ncomp > 1) acc[1][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, x[u].y, acc[1][h], CBSZ, ab, 0);
                if (DUAL || ncomp > 2) acc[2][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, x[u].z, acc[2][h], CBSZ, ab, 0);
                if (DUAL || ncomp > 3) acc[3][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, x[u].w, acc[3][h], CBSZ, ab, 0);
            });
        });
    }
    // fold waves in wave order (fixed => reproducible)
    float* mine = sacc + lane * (NA + 1);
    for (int w = 0; w < WG_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int h = 0; h < HB; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = (c * HB + h) * 4 + i;
                        mine[e] = (w == 0) ? acc[c][h][i] : mine[e] + acc[c][h][i];
                    }
        }
        __syncthreads();
    }
    // bias gradient: lane l holds the partial column sum of channel gc over rows gu, gu + 8, ...: add the 8 row phases with
    // a fixed xor tree, then the waves in order
    __shared__ float sbias[WG_WAVES][16];
    {
        float t = bsum;
#pragma unroll
        for (int m = COUT; m < 8 * COUT; m <<= 1) t += __shfl_xor(t, m, 64);
        const int slot = DUAL ? ((lane >> 5) * 4 + (lane & 3)) : (lane % COUT);          // lanes 0..COUT-1 (and 32..35 for DUAL)
        if ((lane & 31) < COUT && (DUAL || lane < 32)) sbias[wave][slot] = t;
        __syncthreads();
    }
    // register i of accumulator (c, h) on lane (kk, q)  <->  input channel 4q + c, output channel 4h + i of offset kk.  The
    // folded sums sit in LDS (sacc[lane][slot]); ALL threads copy them out in DESTINATION order, so the block's slab row is
    // written with coalesced dword stores (one lane-per-accumulator store per element would be 1,728 scattered 4-byte writes)
    {
        float* dst = d.base + (int64_t)blockIdx.x * d.block_stride;
        const int tid = threadIdx.x;
        if (tid < (DUAL ? 8 : COUT)) {
            float t = sbias[0][tid];
            for (int w = 1; w < WG_WAVES; ++w) t += sbias[w][tid];
            if (DUAL) dst[(tid < 4 ? d.b_off : dd.b_off1) + (tid & 3)] = t;
            else dst[d.b_off + tid] = t;
        }
        if constexpr (DUAL) {
            for (int e = tid; e < 2 * 432; e += WG_WAVES * 64) {
                const int t = e / 432, r = e - 432 * t;
                const int kq = r >> 4, slot = r & 15;                 // slot = ci * 4 + co (HB = 1)
                dst[(t ? dd.w_off1 : d.w_off) + r] = sacc[(32 * t + kq) * (NA + 1) + slot];
            }
        } else {
            const int cinv = d.cin_valid;
            const int per_k = cinv * COUT, total = 27 * per_k;
            for (int e = tid; e < total; e += WG_WAVES * 64) {
                const int kq = e / per_k, r = e - kq * per_k;
                const int ci = r / COUT, co = r - ci * COUT;
                const int ln = (XQ == 2) ? 2 * kq + (ci >> 2) : kq;
                const int slot = ((ci & 3) * HB + (co >> 2)) * 4 + (co & 3);
                dst[d.w_off + e] = sacc[ln * (NA + 1) + slot];
            }
        }
    }
}

// ---- weight gradients with COALESCED gathers and an LDS transpose --------------------------------------------------------------
// All weight-gradient kernels above take ~20 us per row pass whatever their MFMA count (8->8: 64 MFMAs per group, 8->4 and the
// dual 4->4: 32): they are bound by the L1 return path.  With lane = (tap, channel quad) a gather instruction delivers 54
// 16-byte pieces from ~20 different cache lines - about 55 % of the rate the convolutions reach with lane = row on the same
// bytes.  Here the gather of an 8-row group is laid out the convolutions' way - lane = (tap t of 4, row u of 8, quad q): one
// instruction fetches 4 taps x 8 CONSECUTIVE rows, i.e. four 256-byte runs - into a wave-private tap-major LDS image
// [tap][row][quad] with a tap pitch of 8 x 32 + 32 bytes, and every (tap, quad) lane reads its eight rows back with
// ds_read_b128: the pitch makes the 16-byte slot index (2 tap + quad + 2 row) mod 16 = (lane + 2 row) mod 16, distinct inside each of
// the hardware's 16-lane groups; the writes are 128 contiguous bytes per 8 lanes.  (SQ_LDS_BANK_CONFLICT still reads 3.4e5 cycles
// per launch, profiles/r02_pmc_wgrad_variants.txt: a few per cent of the LDS cycles, not attributed - the fold epilogue's
// stride-(NA + 1) accesses are the candidate, the row loop's accesses are conflict-free by construction.)  No block barrier (LDS operations of one wave
// execute in order).  Indices come from a second tiled table (linr_kmap_tile8t: [group][tap of 4][row][tap group j] so that a
// lane's seven indices are 32 contiguous bytes).  Pipeline: while the MFMAs of group t run, the gathers of group t+1 and the
// indices of group t+2 are in flight.  Same groups, same order, same MFMAs => same partial sums, bit for bit.
#define TW_PITCH 288                  // bytes per tap in the LDS image: 8 rows x 32 B + 32 B
#define TW_TAPS 28                    // 27 taps + one dump slot for the unused lane group of the 7th gather
template <int COUT, bool DUAL>
__global__ __launch_bounds__(WG_WAVES * 64) void spconv_wgrad_t_k(WgradSrc S, const int32_t* __restrict__ tile8t, int64_t n,
                                                                 LinrWgradDst d, WgradDual dd, Grp gp = Grp()) {
    static_assert(!DUAL || COUT == 4, "dual mode = two 4->4 convolutions");
    {
        const int gi = blockIdx.y;
        S.in += gp.in[gi]; S.g0 += gp.res[gi];
        if (S.g1) S.g1 += gp.act[gi];
        d.w_off += gp.w[gi]; d.b_off += gp.b[gi];
        dd.w_off1 += gp.e0[gi]; dd.b_off1 += gp.e1[gi];
        if (gp.e2[gi] > 0) d.cin_valid = (int)gp.e2[gi];
    }
    constexpr int HB = COUT / 4;
    constexpr int NA = 4 * HB * 4;
    constexpr int CBSZ = DUAL ? 3 : 4;
    // one LDS buffer: the four wave-private images during the row loop, the fold scratch afterwards (32 KB per block: four
    // blocks per CU)
    constexpr int IMG_F4 = TW_TAPS * TW_PITCH / 16;
    constexpr int FOLD_F4 = 16 * (NA + 1);                 // one wave's accumulators: 64 lanes x (NA + 1) floats
    constexpr int SMEM_F4 = WG_WAVES * (IMG_F4 > FOLD_F4 ? IMG_F4 : FOLD_F4);
    __shared__ float4 smem[SMEM_F4];
    float* sacc = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // MFMA-side role of the lane: (tap kk, quad q) as in spconv_wgrad_mfma_k
    const int q = DUAL ? (lane >> 5) : (lane & 1);
    const int kk = DUAL ? (lane & 31) : (lane >> 1);
    const int k = kk < 27 ? kk : 26;
    // gather-side role: (tap t of the instruction's 4, row u, quad gq)
    const int gq = lane & 1, gu8 = (lane >> 1) & 7, gt = lane >> 4;
    f32x4 acc[4][HB];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < HB; ++h) acc[c][h] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    int64_t per = (n + gridDim.x - 1) / gridDim.x;
    per = (per + 7) & ~(int64_t)7;
    const int64_t b0 = (int64_t)blockIdx.x * per;
    const int64_t b1 = (b0 + per < n) ? b0 + per : n;
    // gathers in saddr form: uniform base (the zero pad row) + a 32-bit lane offset ((index + 1) * 32 + 16 quad)
    const char* ubase = reinterpret_cast<const char*>(S.in - 8);
    const uint32_t uoff = 32u + 16u * gq;
    const float* gsel = (DUAL && q) ? S.g1 : S.g0;
    const int gld = (DUAL && q) ? S.g1_ld : S.g0_ld;
    const int gl = DUAL ? (lane & 31) : lane;
    const int gu = (gl / COUT) & 7, gc = gl % COUT;
    // the lane's 8 indices (7 used) of a group: tile8t[group][gt][gu8][0..7]
    const int32_t* tk = tile8t + (gt * 8 + gu8) * 8;
    char* img = reinterpret_cast<char*>(smem + wave * IMG_F4);
    // write position of gather j: the tap at position 4 j + gt of the slab-major sequence (position 27 = the dump slot); read
    // position of MFMA row u: tap k
    uint32_t wofs[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int p = 4 * j + gt;
        wofs[j] = (uint32_t)((p < 27 ? LINR_TAP(p) : 27) * TW_PITCH + gu8 * 32 + gq * 16);
    }
    const uint32_t rd0 = (uint32_t)(k * TW_PITCH + q * 16);
    float bsum = 0.0f;
    const int64_t g00 = b0 + 8 * wave;
    int4 ia = make_int4(-1, -1, -1, -1), ib = ia;
    float gvn = 0.0f, gvc = 0.0f;
    float4 xg[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) xg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g00 < b1) {                        // wave-uniform; blocks behind the last row must not touch the tables at all
        const int4 a0 = *reinterpret_cast<const int4*>(tk + g00 * 32);
        const int4 c0 = *reinterpret_cast<const int4*>(tk + g00 * 32 + 4);
        gvc = (g00 + gu < n) ? gsel[(g00 + gu) * gld + gc] : 0.0f;
        const int32_t i0[8] = {a0.x, a0.y, a0.z, a0.w, c0.x, c0.y, c0.z, c0.w};
#pragma unroll
        for (int j = 0; j < 7; ++j) xg[j] = *reinterpret_cast<const float4*>(ubase + (((uint32_t)i0[j] << 5) + uoff));
        const int64_t g1r = g00 + 8 * WG_WAVES;           // spare all -1 groups behind the last row group: no bounds check
        ia = *reinterpret_cast<const int4*>(tk + g1r * 32);
        ib = *reinterpret_cast<const int4*>(tk + g1r * 32 + 4);
        gvn = (g1r + gu < n) ? gsel[(g1r + gu) * gld + gc] : 0.0f;
    }
    for (int64_t g0r = g00; g0r < b1; g0r += 8 * WG_WAVES) {
        // (a) the gathered pieces of this group (requested one iteration ago) -> LDS image, tap-major
#pragma unroll
        for (int j = 0; j < 7; ++j) *reinterpret_cast<float4*>(img + wofs[j]) = xg[j];
        __builtin_amdgcn_sched_barrier(0);      // writes first: hoisting the next gathers above them costs 14 register-pair copies
        const float gv = gvc;
        // (b) next group's gathers (its indices arrived during the last MFMAs) and the indices of the group after it
        {
            const int32_t idn[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
#pragma unroll
            for (int j = 0; j < 7; ++j) xg[j] = *reinterpret_cast<const float4*>(ubase + (((uint32_t)idn[j] << 5) + uoff));
            gvc = gvn;
            const int64_t g2r = g0r + 16 * WG_WAVES;
            ia = *reinterpret_cast<const int4*>(tk + g2r * 32);
            ib = *reinterpret_cast<const int4*>(tk + g2r * 32 + 4);
            gvn = (g2r + gu < n) ? gsel[(g2r + gu) * gld + gc] : 0.0f;
        }
        // (c) transposed read: this lane's (tap, quad) for the 8 rows of the group
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(img + rd0 + (uint32_t)(u * 32));
        bsum += gv;
        static_for<8>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            static_for<HB>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                constexpr int ab = u * HB + h;
                // all four components unconditionally: the cin_valid skip of spconv_wgrad_mfma_k (scalar branches between the
                // MFMAs) costs this kernel more than the dead MFMAs of the three narrow first convs do (2.213 vs 2.227 ms/step);
                // accumulators of input channels >= cin_valid are never written out
                acc[0][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, x[u].x, acc[0][h], CBSZ, ab, 0);
                acc[1][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, x[u].y, acc[1][h], CBSZ, ab, 0);
                acc[2][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, x[u].z, acc[2][h], CBSZ, ab, 0);
                acc[3][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, x[u].w, acc[3][h], CBSZ, ab, 0);
            });
        });
    }
    __syncthreads();
    // Every wave parks its accumulators in its own LDS slice; the copy-out below adds the four slices in wave order
    // (((w0 + w1) + w2) + w3: the association of the former wave-by-wave fold, so the bits do not change) - one barrier instead
    // of four and no read-modify-write passes.
    {
        float* mine = sacc + (wave * 64 + lane) * (NA + 1);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < HB; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) mine[(c * HB + h) * 4 + i] = acc[c][h][i];
    }
    __shared__ float sbias[WG_WAVES][16];
    {
        float t = bsum;
#pragma unroll
        for (int m = COUT; m < 8 * COUT; m <<= 1) t += __shfl_xor(t, m, 64);
        const int slot = DUAL ? ((lane >> 5) * 4 + (lane & 3)) : (lane % COUT);
        if ((lane & 31) < COUT && (DUAL || lane < 32)) sbias[wave][slot] = t;
        __syncthreads();
    }
    {
        auto fold4 = [&](int e) {
            constexpr int W = 64 * (NA + 1);
            float t = sacc[e];
#pragma unroll
            for (int w = 1; w < WG_WAVES; ++w) t += sacc[w * W + e];
            return t;
        };
        float* dst = d.base + (int64_t)blockIdx.x * d.block_stride;
        const int tid = threadIdx.x;
        if (tid < (DUAL ? 8 : COUT)) {
            float t = sbias[0][tid];
            for (int w = 1; w < WG_WAVES; ++w) t += sbias[w][tid];
            if (DUAL) dst[(tid < 4 ? d.b_off : dd.b_off1) + (tid & 3)] = t;
            else dst[d.b_off + tid] = t;
        }
        if constexpr (DUAL) {
            for (int e = tid; e < 2 * 432; e += WG_WAVES * 64) {
                const int t = e / 432, r = e - 432 * t;
                dst[(t ? dd.w_off1 : d.w_off) + r] = fold4((32 * t + (r >> 4)) * (NA + 1) + (r & 15));
            }
        } else {
            const int cinv = d.cin_valid;
            if (cinv == 8) {          // all but the first convs of the outter blocks: constant divisors (a runtime division costs ~20 VALU ops)
                for (int e = tid; e < 27 * 8 * COUT; e += WG_WAVES * 64) {
                    const int kq = e / (8 * COUT), r = e % (8 * COUT);
                    const int ci = r / COUT, co = r % COUT;
                    dst[d.w_off + e] = fold4((2 * kq + (ci >> 2)) * (NA + 1) + ((ci & 3) * HB + (co >> 2)) * 4 + (co & 3));
                }
            } else {
                const int per_k = cinv * COUT, total = 27 * per_k;
                for (int e = tid; e < total; e += WG_WAVES * 64) {
                    const int kq = e / per_k, r = e - kq * per_k;
                    const int ci = r / COUT, co = r - ci * COUT;
                    dst[d.w_off + e] = fold4((2 * kq + (ci >> 2)) * (NA + 1) + ((ci & 3) * HB + (co >> 2)) * 4 + (co & 3));
                }
            }
        }
    }
}

// tile8t[g][t][u][j] = nbr[LINR_TAP(4 j + t)][8 g + u] (-1 for position 27, for j = 7 and beyond n): a lane (t, u) of the transposing
// kernel reads its seven indices of group g as two 16-byte loads.  Gather instruction j fetches the taps at positions 4 j .. 4 j + 3
// of the convolutions' slab-major tap sequence (common.h: LINR_TAP), i.e. neighbours that sit in the same few cache lines
__global__ __launch_bounds__(LINR_BLOCK) void kmap_tile8t_k(const int32_t* __restrict__ nbr, int64_t ld, int64_t n, int64_t groups,
                                                            int32_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (e >= groups * 256) return;
    const int64_t g = e >> 8;
    const int r = (int)(e & 255), t = r >> 6, u = (r >> 3) & 7, j = r & 7;
    const int p = 4 * j + t;                 // position in the gather sequence; its tap: the convolutions' slab-major order
    const int64_t row = 8 * g + u;
    out[e] = (j < 7 && p < 27 && row < n) ? nbr[(int64_t)LINR_TAP(p) * ld + row] : -1;
}

extern "C" size_t linr_kmap_tile8t_bytes(int64_t n) {
    if (n < 0) return 0;
    return (size_t)((n + 7) / 8 + 3 * WG_WAVES) * 256 * sizeof(int32_t);
}

extern "C" int linr_kmap_tile8t(const int32_t* nbr, int64_t ld, int64_t n, int32_t* tile8t, size_t tile8t_bytes, void* stream) {
    if (n < 0 || ld < n) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!nbr || !tile8t) return LINR_EINVAL;
    if (tile8t_bytes < linr_kmap_tile8t_bytes(n)) return LINR_ENOSPC;
    if (!linr_aligned16(tile8t)) return LINR_EALIGN;
    const int64_t groups = (n + 7) / 8 + 3 * WG_WAVES;      // spare all -1 groups: the kernel prefetches two strides ahead
    kmap_tile8t_k<<<linr_grid(groups * 256, LINR_BLOCK), LINR_BLOCK, 0, (hipStream_t)stream>>>(nbr, ld, n, groups, tile8t);
    return linr_launch_rc();
}

int linr_conv3_wgrad_mfma(const float* in, int in_ld, const float* gout, int gout_ld, const int32_t* nbr, int64_t nbr_ld,
                          int64_t n, int cin, int cout, LinrWgradDst d, int nblocks, hipStream_t s, const Grp* gp,
                          int ngroups, const int32_t* tile8t) {
    if (in_ld != 8 && in_ld != 4) return LINR_EINVAL;  // the kernels address gathered rows by a shift: 32- or 16-byte rows
    const int idx = (nbr_ld % 4 == 0 && linr_aligned16(nbr)) ? 1 : 0;
    const Grp g0 = gp ? *gp : Grp();
    const dim3 grid(nblocks, ngroups);
    WgradSrc S = {in, in_ld, gout, gout_ld, nullptr, 0};
    WgradDual dd = {0, 0};
    d.cin_valid = cin;
    // coalesced gathers + LDS transpose: 32-byte rows, the transposed tiled table
    if (tile8t && in_ld == 8 && cin <= 8 && (cout == 8 || (cout == 4 && cin == 8)) && linr_aligned16(in) && linr_aligned16(tile8t)) {
        if (cout == 8) spconv_wgrad_t_k<8, false><<<grid, WG_WAVES * 64, 0, s>>>(S, tile8t, n, d, dd, g0);
        else spconv_wgrad_t_k<4, false><<<grid, WG_WAVES * 64, 0, s>>>(S, tile8t, n, d, dd, g0);
        return linr_launch_rc();
    }
#define GO(XQ, CO)                                                                                                           \
    do {                                                                                                                     \
        if (idx == 1) spconv_wgrad_mfma_k<XQ, CO, false, 1><<<grid, WG_WAVES * 64, 0, s>>>(S, nbr, nbr_ld, n, d, dd, g0);   \
        else spconv_wgrad_mfma_k<XQ, CO, false, 0><<<grid, WG_WAVES * 64, 0, s>>>(S, nbr, nbr_ld, n, d, dd, g0);            \
        return linr_launch_rc();                                                                                             \
    } while (0)
    if (cin == 8 && cout == 8) GO(2, 8);
    if (cin == 8 && cout == 4) GO(2, 4);
    if (cin == 4 && cout == 4) GO(1, 4);
    if (cin < 8 && cout == 8 && in_ld >= 8) GO(2, 8);
#undef GO
    return LINR_EINVAL;
}

// both 4->4 convolutions of an Inception block: in = H [n][8]; conv 0 reads H[:,0:4] with gradient g0, conv 1 H[:,4:8] with g1
int linr_conv3_wgrad_dual44(const float* H, const float* g0, int g0_ld, const float* g1, int g1_ld, const int32_t* nbr,
                            int64_t nbr_ld, int64_t n, float* big, int64_t block_stride, int64_t w_off0, int64_t b_off0,
                            int64_t w_off1, int64_t b_off1, int nblocks, hipStream_t s, const Grp* gp, int ngroups,
                            const int32_t* tile8t) {
    const int idx = (nbr_ld % 4 == 0 && linr_aligned16(nbr)) ? 1 : 0;
    const Grp grp = gp ? *gp : Grp();
    const dim3 grid(nblocks, ngroups);
    WgradSrc S = {H, 8, g0, g0_ld, g1, g1_ld};
    LinrWgradDst d = {big, block_stride, w_off0, b_off0, 4};
    WgradDual dd = {w_off1, b_off1};
    if (tile8t && linr_aligned16(H) && linr_aligned16(tile8t)) {
        spconv_wgrad_t_k<4, true><<<grid, WG_WAVES * 64, 0, s>>>(S, tile8t, n, d, dd, grp);
        return linr_launch_rc();
    }
    if (idx == 1) spconv_wgrad_mfma_k<2, 4, true, 1><<<grid, WG_WAVES * 64, 0, s>>>(S, nbr, nbr_ld, n, d, dd, grp);
    else spconv_wgrad_mfma_k<2, 4, true, 0><<<grid, WG_WAVES * 64, 0, s>>>(S, nbr, nbr_ld, n, d, dd, grp);
    return linr_launch_rc();
}

// ---- fused backward of the occupancy head ------------------------------------------------------------------------------
// csrc/head_bwd.h holds the arithmetic (shared with the bf16 training executor); here: fp32 rows in and out, grouped launches.
struct HeadBwdArgs {
    const float* c;  const float* p;  const float* target; int target_ld;
    const float* w1; const float* b1; const float* w2;
    float gscale;                     // d loss / d nats
    float* gc;                        // [n][8]
    float* big; int64_t block_stride; int64_t off_w1, off_b1, off_w2, off_b2;
    int active;
};

__global__ __launch_bounds__(HB_WAVES * 64, 2) void head_bwd_k(HeadBwdArgs A, int64_t n, Grp gp = Grp()) {
    __shared__ float lds[HB_LDS_FLOATS];
    // group offsets: in = c, e0 = p, e1 = target, w = w1, b = b1, e2 = w2, out = gc, e3..e6 = slab offsets of w1, b1, w2, b2
    const int gi = blockIdx.y;
    const float* C = A.c + gp.in[gi];
    float* GC = A.gc + gp.out[gi];
    HbParams h;
    h.p = A.p + gp.e0[gi]; h.target = A.target + gp.e1[gi]; h.target_ld = A.target_ld;
    h.w1 = A.w1 + gp.w[gi]; h.b1 = A.b1 + gp.b[gi]; h.w2 = A.w2 + gp.e2[gi];
    h.gscale = A.gscale; h.n = n;
    h.dst = A.big + (int64_t)blockIdx.x * A.block_stride;
    h.off_w1 = A.off_w1 + gp.e3[gi]; h.off_b1 = A.off_b1 + gp.e4[gi]; h.off_w2 = A.off_w2 + gp.e5[gi]; h.off_b2 = A.off_b2 + gp.e6[gi];
    h.active = A.active;
    head_bwd_body<HbRaw32>(h,
        [&](int64_t row) { return HbRaw32{*reinterpret_cast<const f32x4*>(C + row * 8), *reinterpret_cast<const f32x4*>(C + row * 8 + 4)}; },
        [](const HbRaw32& r, float (&c)[8]) {
            c[0] = r.a[0]; c[1] = r.a[1]; c[2] = r.a[2]; c[3] = r.a[3]; c[4] = r.b[0]; c[5] = r.b[1]; c[6] = r.b[2]; c[7] = r.b[3];
        },
        [&](int64_t row, const float (&g)[8]) {
            *reinterpret_cast<float4*>(GC + row * 8) = make_float4(g[0], g[1], g[2], g[3]);
            *reinterpret_cast<float4*>(GC + row * 8 + 4) = make_float4(g[4], g[5], g[6], g[7]);
        }, lds);
}

static int hb_cus() {
    static const int v = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    return v;
}

// rows_written == nullptr: slab rows 0 .. nblocks - 1 are all written (rows beyond the active blocks get zeros); otherwise only the
// active blocks' rows are written and *rows_written tells the caller how many (its reduction must stop there)
int linr_head_bwd_launch(const float* c, const float* p, const float* target, int target_ld, const float* w1,
                         const float* b1, const float* w2, float gscale, float* gc, int64_t n, float* big,
                         int64_t block_stride, int64_t off_w1, int64_t off_b1, int64_t off_w2, int64_t off_b2,
                         hipStream_t s, const Grp* gp, int ngroups, int nblocks, int* rows_written) {
    if (rows_written) *rows_written = 0;
    if (n == 0) return 0;
    const Grp g0 = gp ? *gp : Grp();
    const int active = hb_blocks(n, ngroups, hb_cus(), nblocks);
    HeadBwdArgs A = {c, p, target, target_ld, w1, b1, w2, gscale, gc, big, block_stride, off_w1, off_b1, off_w2, off_b2, active};
    head_bwd_k<<<dim3(rows_written ? active : nblocks, ngroups), HB_WAVES * 64, 0, s>>>(A, n, g0);
    if (rows_written) *rows_written = active;
    return linr_launch_rc();
}

// executor entry: all matrices are arena matrices (16-byte aligned rows, ld in {4, 8}, pad row present)
int linr_cconv_launch(bool bwd, const float* in, int in_ld, const int32_t* lo, const uint32_t* mask, int64_t ld,
                      int64_t n, const float* W, const float* bias, int cin, int cout, const float* res, int res_ld,
                      const float* act, int act_ld, float* out, int out_ld, unsigned flags, hipStream_t s, const Grp* gp,
                      int ngroups) {
    if (n == 0) return 0;
    const Grp g0 = gp ? *gp : Grp();
    const dim3 grid(linr_grid(n, LINR_CONV_BLOCK), ngroups);
    static const int use_mfma = getenv("LINR_CONV_MFMA") ? atoi(getenv("LINR_CONV_MFMA")) : 1;   // 0: the VALU kernel (bitwise reference)
#define GO(GI, GO_, B, LW)                                                                                              \
    do {                                                                                                                \
        if (use_mfma && (GO_ == 4 || GO_ == 8))                                                                    \
            cconv_mfma_k<GI, (GO_ == 4 || GO_ == 8) ? GO_ : 8, B, LW><<<grid, LINR_CONV_BLOCK, 0, s>>>(                       \
                in, in_ld, lo, mask, ld, n, W, bias, res, res_ld, act, act_ld, out, out_ld, flags, HeadArgs(), PwArgs(), g0);  \
        else                                                                                                            \
            cconv_k<GI, GO_, B, LW><<<linr_grid(n, LINR_BLOCK), LINR_BLOCK, 0, s>>>(in, in_ld, lo, mask, ld, n, W, bias, res, res_ld, act,   \
                                                                act_ld, out, out_ld, flags);                            \
        return linr_launch_rc();                                                                                        \
    } while (0)
#define CASE(CI, CO)                                                      \
    if (cin == CI && cout == CO) {                                        \
        if (!bwd) GO(CI, CO, false, ((CI + 3) / 4 * 4));                  \
        else GO(CO, CI, true, ((CO + 3) / 4 * 4));                        \
    }
    CASE(8, 8) CASE(8, 4) CASE(4, 4)
    CASE(1, 8) CASE(2, 8) CASE(3, 8) CASE(4, 8) CASE(5, 8) CASE(6, 8) CASE(7, 8)
#undef CASE
#undef GO
    return LINR_EINVAL;
}


// ---- the executor's fused layers as stand-alone ops (include/linr_hip.h) ---------------------------------------------------
static bool cmap_ok(const void* in, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n) {
    return in && lo && mask && ld >= n;
}
#define HEAD_PARAMS 241            // inner_mlps.k.0: 0.weight [24][8], 0.bias [24], 2.weight [1][24], 2.bias [1]

extern "C" size_t linr_head_workspace_bytes(int64_t n) {
    if (n < 0) return 0;
    const size_t fwd = (size_t)linr_grid(n, LINR_CONV_BLOCK) * sizeof(double);
    const size_t bwd = (size_t)LINR_WG_BLOCKS * HEAD_PARAMS * sizeof(float);
    return (fwd > bwd ? fwd : bwd) + 64;
}

extern "C" int linr_head_fwd(const float* prior, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                             const float* Wp, const float* bp, const float* w1, const float* b1, const float* w2,
                             const float* b2, const float* target, int32_t target_ld, float* c_out, float* p_out,
                             double* bits_acc, void* ws, size_t ws_bytes, void* stream) {
    if (n < 0) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!cmap_ok(prior, lo, mask, ld, n) || !Wp || !bp || !w1 || !b1 || !w2 || !b2 || !c_out || !p_out) return LINR_EINVAL;
    if (bits_acc && (!target || target_ld < 1 || !ws)) return LINR_EINVAL;
    if (!linr_aligned16(prior) || !linr_aligned16(c_out)) return LINR_EALIGN;
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull) return LINR_EINVAL;
    double* part = nullptr;
    if (bits_acc) {
        if (ws_bytes < linr_head_workspace_bytes(n)) return LINR_ENOSPC;
        if (((uintptr_t)ws) & 7u) return LINR_EALIGN;
        part = (double*)ws;
    }
    int rc = linr_cconv_head_launch(prior, lo, mask, ld, n, Wp, bp, c_out, w1, b1, w2, b2, target, target_ld, p_out, part,
                                    (hipStream_t)stream);
    if (rc) return rc;
    if (bits_acc) return linr_bits_finish_launch(part, (int)linr_grid(n, LINR_CONV_BLOCK), bits_acc, (hipStream_t)stream);
    return 0;
}

extern "C" int linr_head_bwd(const float* c, const float* p, const float* target, int32_t target_ld, const float* w1,
                             const float* b1, const float* w2, float gscale, float* gc, int64_t n, float* ghead, void* ws,
                             size_t ws_bytes, void* stream) {
    if (n < 0 || target_ld < 1) return LINR_EINVAL;
    if (!ghead) return LINR_EINVAL;
    if (n == 0) return linr_hip_rc(hipMemsetAsync(ghead, 0, HEAD_PARAMS * sizeof(float), (hipStream_t)stream));
    if (!c || !p || !target || !w1 || !b1 || !w2 || !gc || !ws) return LINR_EINVAL;
    if (ws_bytes < linr_head_workspace_bytes(n)) return LINR_ENOSPC;
    if (!linr_aligned16(c) || !linr_aligned16(gc) || !linr_aligned16(ws)) return LINR_EALIGN;
    float* slab = (float*)ws;
    // gscale multiplies BITS (like linr_bce_bits_bwd); the kernel works in nats: d bits / d nats = 1 / ln 2
    int rc = linr_head_bwd_launch(c, p, target, target_ld, w1, b1, w2, gscale * 1.4426950408889634f, gc, n, slab, HEAD_PARAMS, 0, 192, 216, 240,
                                  (hipStream_t)stream);
    if (rc) return rc;
    return linr_slab_reduce_launch(slab, LINR_WG_BLOCKS, HEAD_PARAMS, ghead, (hipStream_t)stream);
}

static bool inc_ok(const linr_inception_params* q) {
    return q && q->w00 && q->b00 && q->w01 && q->b01 && q->w10 && q->b10 && q->w11 && q->b11 && q->w12 && q->b12;
}

extern "C" int linr_inception_fwd(const float* x, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                                  const linr_inception_params* q, float* H, float* M, float* I, void* stream) {
    if (n < 0) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!cmap_ok(x, lo, mask, ld, n) || !inc_ok(q) || !H || !M || !I) return LINR_EINVAL;
    if (!linr_aligned16(x) || !linr_aligned16(H) || !linr_aligned16(M) || !linr_aligned16(I)) return LINR_EALIGN;
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull) return LINR_EINVAL;
    int rc = linr_conv_pw_fwd_launch(x, lo, mask, ld, n, q->w00, q->b00, q->w10, q->b10, H, (hipStream_t)stream);
    if (rc) return rc;
    return linr_dual44_fwd_launch(H, lo, mask, ld, n, q->w01, q->b01, q->w11, q->b11, x, q->w12, q->b12, M, I, (hipStream_t)stream);
}

extern "C" int linr_inception_bwd_data(const float* gI, const float* x, const float* H, const float* M, const int32_t* lo,
                                       const uint32_t* mask, int64_t ld, int64_t n, const linr_inception_params* q, float* gM,
                                       float* gH, float* gX, uint32_t flags, void* stream) {
    if (n < 0) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!cmap_ok(gI, lo, mask, ld, n) || !inc_ok(q) || !H || !M || !gM || !gH || !gX) return LINR_EINVAL;
    if ((flags & LINR_RELU_MASK) && !x) return LINR_EINVAL;
    if (flags & ~(LINR_RELU_MASK | LINR_ACCUM)) return LINR_EINVAL;
    if (!linr_aligned16(gI) || !linr_aligned16(gM) || !linr_aligned16(gH) || !linr_aligned16(gX) || !linr_aligned16(H) ||
        !linr_aligned16(M)) return LINR_EALIGN;
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull) return LINR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // I[:,4:8] = M @ W12 + b12 + x[:,4:8], M = relu(.)  =>  gM = (gI[:,4:8] @ W12^T) * (M > 0)
    int rc = linr_linear_launch(gI + 4, 8, n, q->w12, 1, 4, nullptr, 4, 4, nullptr, 0, M, 4, gM, 4, LINR_RELU_MASK, s);
    if (rc) return rc;
    rc = linr_dual44_bwd_launch(gI, gM, lo, mask, ld, n, q->w01, q->w11, H, gH, s);
    if (rc) return rc;
    return linr_conv_bwd_ga_launch(gH, lo, mask, ld, n, q->w00, q->w10, gI, x, gX, flags, s);
}

extern "C" int linr_occ_conv7(const float* occ, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                              const float* params, const int64_t* w_off_h, const int64_t* b_off_h, float* out,
                              const int64_t* out_off_h, void* stream) {
    if (n < 0) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!cmap_ok(occ, lo, mask, ld, n) || !params || !w_off_h || !b_off_h || !out || !out_off_h) return LINR_EINVAL;
    if (!linr_aligned16(occ) || !linr_aligned16(out)) return LINR_EALIGN;
    for (int g = 0; g < 7; ++g)
        if (w_off_h[g] < 0 || b_off_h[g] < 0 || (out_off_h[g] & 3)) return LINR_EINVAL;
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull) return LINR_EINVAL;
    return linr_occ_conv7_launch(occ, lo, mask, ld, n, params, w_off_h, b_off_h, out, out_off_h, (hipStream_t)stream);
}

extern "C" int linr_spconv_wgrad_dual44(const float* H, const float* g0, int32_t g0_ld, const float* g1, int32_t g1_ld,
                                        const int32_t* nbr, const int32_t* tile8t, int64_t ld, int64_t n, float* slab, void* stream) {
    if (n < 0 || ld < n || g0_ld < 4 || g1_ld < 4) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!H || !g0 || !g1 || !nbr || !slab) return LINR_EINVAL;
    if (!linr_aligned16(H)) return LINR_EALIGN;
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull) return LINR_EINVAL;
    // per block: [W01 432 | b01 4 | W11 432 | b11 4]
    return linr_conv3_wgrad_dual44(H, g0, g0_ld, g1, g1_ld, nbr, ld, n, slab, 872, 0, 432, 436, 868, LINR_WG_BLOCKS,
                                   (hipStream_t)stream, nullptr, 1, tile8t);
}

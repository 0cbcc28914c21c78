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
};

template <int GB, int PB, bool BWD>
__global__ __launch_bounds__(LINR_CONV_BLOCK) void wconv_k(WcArgs a, const int32_t* __restrict__ lo, const uint32_t* __restrict__ mask,
                                                          int64_t ld, int64_t n) {
    constexpr int GIN = 8 * GB, NQ = 2 * PB, NV = GB * PB;          // NV = GIN * NQ / 16 A-operand registers per tap
    constexpr int PL = NV >= 4 ? NV / 4 : 1, PW = NV >= 4 ? 4 : NV;  // planes of PW registers: wl[k][plane][lane][PW]
    extern __shared__ float wl[];
    for (int e = threadIdx.x; e < 27 * NV * 64; e += LINR_CONV_BLOCK) {
        const int w = e % PW, lane = (e / PW) % 64, pl = (e / (PW * 64)) % PL, k = e / (NV * 64);
        const int v = pl * PW + w, blk = lane >> 2, j = lane & 3;
        const int c = 16 * v + blk, gi = c / NQ, oq = c % NQ, po = 8 * a.pb0 + 4 * oq + j;
        float val = 0.0f;
        if (gi < a.gvalid) val = BWD ? a.W[((int64_t)k * a.cin + po) * a.cout + gi] : a.W[((int64_t)k * a.cin + gi) * a.cout + po];
        wl[e] = val;
    }
    __syncthreads();
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
            for (int j = 0; j < 4; ++j) acc[q][j] = (!BWD && a.bias) ? a.bias[8 * a.pb0 + 4 * q + j] : 0.0f;
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
        // epilogue order of the 8-wide kernels: + res, + old (LINR_ACCUM), * (act > 0) (LINR_RELU_MASK), ReLU
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            float o[8];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) o[4 * h + j] = acc[2 * pb + h][j];
            if (a.res[pb]) {
                const float* r = a.res[pb] + row * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += r[j];
            }
            float* op = a.out[pb] + row * 8;
            if (a.flags & LINR_ACCUM) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += op[j];
            }
            if (a.flags & LINR_RELU_MASK) {
                const float* m = a.act[pb] + row * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = m[j] > 0.0f ? o[j] : 0.0f;
            }
            if (a.flags & LINR_RELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.0f);
            }
            *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

template <int GB, int PB, bool BWD>
static int wc_launch(const WcArgs& a, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, hipStream_t s) {
    constexpr size_t lds = (size_t)27 * GB * PB * 64 * 4;
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
    wconv_k<GB, PB, BWD><<<(unsigned)grid, LINR_CONV_BLOCK, lds, s>>>(a, lo, mask, ld, n);
    return linr_launch_rc();
}

template <bool BWD>
static int wc_dispatch(int gb, int pb, const WcArgs& a, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, hipStream_t s) {
    if (gb == 1 && pb == 1) return wc_launch<1, 1, BWD>(a, lo, mask, ld, n, s);
    if (gb == 1 && pb == 2) return wc_launch<1, 2, BWD>(a, lo, mask, ld, n, s);
    if (gb == 2 && pb == 1) return wc_launch<2, 1, BWD>(a, lo, mask, ld, n, s);
    if (gb == 2 && pb == 2) return wc_launch<2, 2, BWD>(a, lo, mask, ld, n, s);
    if (gb == 4 && pb == 1) return wc_launch<4, 1, BWD>(a, lo, mask, ld, n, s);
    if (gb == 4 && pb == 2) return wc_launch<4, 2, BWD>(a, lo, mask, ld, n, s);
    return LINR_EINVAL;
}

// in / out / res / act: HOST arrays of device pointers to the [rows][8] blocks (gathered blocks must have the zero row in front).
// fwd (bwd = 0): gathers cin channels in ceil(cin / 8) blocks (cin < 8: one block whose channels >= cin are ignored), produces cout / 8
// blocks.  bwd (bwd = 1): gathers the output gradient in cout / 8 blocks at the mirrored taps, produces the input gradient in cin / 8
// blocks (cin a multiple of 8).  flags: LINR_RELU, LINR_ACCUM, LINR_RELU_MASK as in linr_spconv_cmap.  Up to 32 channels either side.
extern "C" int linr_spconv_wide(int32_t bwd, const float* const* in_h, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                                const float* W, const float* bias, int32_t cin, int32_t cout, const float* const* res_h,
                                const float* const* act_h, float* const* out_h, uint32_t flags, void* stream) {
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
    hipStream_t s = (hipStream_t)stream;
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
        const int rc = bwd ? wc_dispatch<true>(gb, pb, a, lo, mask, ld, n, s) : wc_dispatch<false>(gb, pb, a, lo, mask, ld, n, s);
        if (rc) return rc;
    }
    return 0;
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
    if (!in_h || !g_h || !nbr || !slab || !gW) return LINR_EINVAL;
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull) return LINR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nbi = (cin + 7) / 8, nbo = cout / 8, npairs = nbi * nbo;
    if (n == 0) {
        if (hipMemsetAsync(gW, 0, (size_t)27 * cin * cout * sizeof(float), s) != hipSuccess) return LINR_EINVAL;
        return gb ? linr_hip_rc(hipMemsetAsync(gb, 0, (size_t)cout * sizeof(float), s)) : 0;
    }
    for (int i = 0; i < nbi; ++i) if (!in_h[i] || !linr_aligned16(in_h[i])) return in_h[i] ? LINR_EALIGN : LINR_EINVAL;
    for (int i = 0; i < nbo; ++i) if (!g_h[i]) return LINR_EINVAL;
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
    const int total = 27 * cin * cout + cout;
    wide_slab_reduce_k<<<linr_grid(total, 16), LINR_BLOCK, 0, s>>>(slab, LINR_WG_BLOCKS, npairs, cin, cout, gW, gb);
    return linr_launch_rc();
}

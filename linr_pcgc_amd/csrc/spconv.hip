// Sparse 3x3x3 convolution on a fixed coordinate set: forward, backward-data, backward-weight.
// Replaces MinkowskiConvolution fwd/bwd (see include/linr_hip.h for the reference call sites).
//
// v1 design (output-stationary gather, no atomics, bit-reproducible):
//   fwd / bwd_data : one lane owns one output row; it walks the 27 offsets in fixed order, gathers the
//                    neighbour's feature row (16/32 B vector loads) and FMAs against weights that are
//                    wave-uniform (scalar loads -> SGPR operands).  nbr is [27][rows] so the index loads of a
//                    wave are coalesced.  bwd_data is the same kernel through the mirror identity
//                    nbr[26-k][i] == j  <=>  nbr[k][j] == i  with the weight tile transposed.
//   bwd_weight     : persistent blocks stage a tile of gathered rows + output gradients in LDS, lane (k,ci)
//                    keeps its COUT accumulators in registers across all of its tiles, block partials go to a
//                    slab and a second kernel sums the slabs in fixed order (deterministic, no float atomics).
#include "common.h"
#include <stdlib.h>

template <int W> struct RowLoad {
    static __device__ __forceinline__ void run(const float* __restrict__ p, float* x) {
#pragma unroll
        for (int v = 0; v < W / 4; ++v) {
            const float4 t = *reinterpret_cast<const float4*>(p + 4 * v);
            x[4 * v] = t.x; x[4 * v + 1] = t.y; x[4 * v + 2] = t.z; x[4 * v + 3] = t.w;
        }
    }
};

// GIN: channels of the gathered operand, GOUT: channels produced per row.
// BWD == false: acc[o] += x[i] * W[(k*GIN + i)*GOUT + o], neighbour row nbr[k]
// BWD == true : acc[o] += x[i] * W[(k*GOUT + o)*GIN + i], neighbour row nbr[26-k]
// LOADW: floats fetched per gathered row with float4 loads (multiple of 4, >= GIN) or 0 for scalar loads.
// PAD: the caller guarantees a readable all-zero row at in[-in_ld .. -1] (row index -1), so an absent neighbour
//      needs neither a branch nor a select: the 27 index loads and 27 row gathers are straight-line code the
//      scheduler can keep in flight together.  Adding fmaf(0, w, acc) leaves acc bit-identical, so PAD and
//      non-PAD builds give the same bits (w is finite).
template <int GIN, int GOUT, bool BWD, int LOADW, bool PAD, bool WFIXED = false>
__global__ __launch_bounds__(LINR_BLOCK) void spconv_gather_k(
    const float* __restrict__ in, int in_ld, const int32_t* __restrict__ nbr, int64_t nbr_ld, int64_t n,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ res, int res_ld,
    const float* __restrict__ act, int act_ld, float* __restrict__ out, int out_ld, unsigned flags) {
    const int64_t row = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (row >= n) return;
    float acc[GOUT];
#pragma unroll
    for (int o = 0; o < GOUT; ++o) acc[o] = (bias != nullptr) ? bias[o] : 0.0f;
    int32_t idx[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) idx[k] = nbr[(int64_t)(BWD ? 26 - k : k) * nbr_ld + row];
#pragma unroll
    for (int kk = 0; kk < 27; ++kk) {
        const int k = LINR_TAP(kk);               // tap order of the conv family (common.h)
        const int32_t j = idx[k];
        if (PAD || j >= 0) {
            constexpr int XW = LOADW ? LOADW : GIN;
            float x[XW];
            const float* p = in + (int64_t)j * in_ld;
            if constexpr (LOADW != 0) {
                RowLoad<LOADW>::run(p, x);
            } else {
#pragma unroll
                for (int i = 0; i < GIN; ++i) x[i] = p[i];
            }
            const float* __restrict__ wk = W + (WFIXED ? 0 : k * GIN * GOUT);   // WFIXED: timing experiment only
#pragma unroll
            for (int i = 0; i < GIN; ++i) {
#pragma unroll
                for (int o = 0; o < GOUT; ++o) {
                    const float w = BWD ? wk[o * GIN + i] : wk[i * GOUT + o];
                    acc[o] = fmaf(x[i], w, acc[o]);
                }
            }
        }
    }
    // epilogue order (documented in include/linr_hip.h): + res, + old (ACCUM), * mask, ReLU
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
    if ((GOUT % 4 == 0) && (out_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) {
#pragma unroll
        for (int v = 0; v < GOUT / 4; ++v)
            *reinterpret_cast<float4*>(op + 4 * v) = make_float4(acc[4 * v], acc[4 * v + 1], acc[4 * v + 2], acc[4 * v + 3]);
    } else {
#pragma unroll
        for (int o = 0; o < GOUT; ++o) op[o] = acc[o];
    }
}

template <int GIN, int GOUT, bool BWD>
static int launch_gather(const float* in, int in_ld, const int32_t* nbr, int64_t nbr_ld, int64_t n, const float* W,
                         const float* bias, const float* res, int res_ld, const float* act, int act_ld, float* out,
                         int out_ld, unsigned flags, hipStream_t s) {
    const unsigned grid = linr_grid(n, LINR_BLOCK);
    constexpr int LW = (GIN + 3) / 4 * 4;
    // vector path: the whole LW-float window must lie inside the row and be 16-byte aligned
    const bool vec = linr_aligned16(in) && (in_ld % 4 == 0) && (LW <= in_ld);
    const bool pad = (flags & LINR_PAD_ROW) != 0;
#define LINR_GO(LWV, PADV)                                                                                          \
    spconv_gather_k<GIN, GOUT, BWD, LWV, PADV><<<grid, LINR_BLOCK, 0, s>>>(in, in_ld, nbr, nbr_ld, n, W, bias, res,  \
                                                                           res_ld, act, act_ld, out, out_ld, flags)
    if (vec && pad) LINR_GO(LW, true);
    else if (vec) LINR_GO(LW, false);
    else if (pad) LINR_GO(0, true);
    else LINR_GO(0, false);
#undef LINR_GO
    return linr_launch_rc();
}

// internal entry (also used by the network executor): fwd (bwd == false) or bwd-data (bwd == true, roles swapped:
// `in` is the output gradient with `cout` channels, `out` the input gradient with `cin` channels).
int linr_conv3_launch(bool bwd, const float* in, int in_ld, const int32_t* nbr, int64_t nbr_ld, int64_t n,
                      const float* W, const float* bias, int cin, int cout, const float* res, int res_ld,
                      const float* act, int act_ld, float* out, int out_ld, unsigned flags, hipStream_t s) {
    if (n == 0) return 0;
#define LINR_CASE(CI, CO)                                                                                              \
    if (cin == CI && cout == CO) {                                                                                     \
        if (!bwd) return launch_gather<CI, CO, false>(in, in_ld, nbr, nbr_ld, n, W, bias, res, res_ld, act, act_ld, out, \
                                                      out_ld, flags, s);                                               \
        return launch_gather<CO, CI, true>(in, in_ld, nbr, nbr_ld, n, W, bias, res, res_ld, act, act_ld, out, out_ld,   \
                                           flags, s);                                                                  \
    }
    LINR_CASE(8, 8) LINR_CASE(8, 4) LINR_CASE(4, 4)
    LINR_CASE(1, 8) LINR_CASE(2, 8) LINR_CASE(3, 8) LINR_CASE(4, 8) LINR_CASE(5, 8) LINR_CASE(6, 8) LINR_CASE(7, 8)
#undef LINR_CASE
    return LINR_EINVAL;
}

extern "C" int linr_spconv_fwd(const float* in, int32_t in_ld, const int32_t* nbr, int64_t nbr_ld, int64_t n,
                               const float* W, const float* bias, int32_t cin, int32_t cout, const float* res,
                               int32_t res_ld, float* out, int32_t out_ld, uint32_t flags, void* stream) {
    if (n < 0 || nbr_ld < n || in_ld < cin || out_ld < cout) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!in || !nbr || !W || !out || (!bias && !(flags & LINR_NO_BIAS))) return LINR_EINVAL;
    if (flags & LINR_RELU_MASK) return LINR_EINVAL;
    if (res && res_ld < cout) return LINR_EINVAL;
    return linr_conv3_launch(false, in, in_ld, nbr, nbr_ld, n, W, (flags & LINR_NO_BIAS) ? nullptr : bias, cin, cout, res,
                             res_ld, nullptr, 0, out, out_ld, flags, (hipStream_t)stream);
}

extern "C" int linr_spconv_bwd_data(const float* gout, int32_t gout_ld, const int32_t* nbr, int64_t nbr_ld, int64_t n,
                                    const float* W, int32_t cin, int32_t cout, const float* act, int32_t act_ld,
                                    float* gin, int32_t gin_ld, uint32_t flags, void* stream) {
    if (n < 0 || nbr_ld < n || gout_ld < cout || gin_ld < cin) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!gout || !nbr || !W || !gin) return LINR_EINVAL;
    if ((flags & LINR_RELU_MASK) && (!act || act_ld < cin)) return LINR_EINVAL;
    if (flags & LINR_RELU) return LINR_EINVAL;
    return linr_conv3_launch(true, gout, gout_ld, nbr, nbr_ld, n, W, nullptr, cin, cout, nullptr, 0, act, act_ld, gin,
                             gin_ld, flags, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// backward-weight
// ---------------------------------------------------------------------------------------------------------------
#define BW_TILE 32   // rows staged per tile

// Accumulator ownership: P = 27*CIN (k,ci) pairs + 1 pseudo-pair for the bias (x == 1).  G = 256 / (P+1) row
// groups run side by side; group g takes rows r == g (mod G) of every tile.  Slab layout per block:
// [ (P+1) * COUT ] floats, pair-major.
template <int CIN, int COUT>
__global__ __launch_bounds__(LINR_BLOCK) void spconv_bwd_weight_k(
    const float* __restrict__ in, int in_ld, const float* __restrict__ gout, int gout_ld,
    const int32_t* __restrict__ nbr, int64_t nbr_ld, int64_t n, float* __restrict__ slab) {
    constexpr int P = 27 * CIN;
    constexpr int PP = P + 1;                 // + bias pseudo-pair
    constexpr int G = LINR_BLOCK / PP > 0 ? LINR_BLOCK / PP : 1;
    constexpr int XS = PP | 1;                // odd row stride: conflict-free column writes
    __shared__ float sx[BW_TILE * XS];
    __shared__ float sg[BW_TILE * COUT];
    __shared__ float sred[(G > 1) ? (G - 1) * PP * COUT : 1];
    const int tid = threadIdx.x;
    const int pair = tid % PP;
    const int grp = tid / PP;
    const bool active = grp < G;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.0f;
    const int64_t tiles = (n + BW_TILE - 1) / BW_TILE;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t row0 = t * BW_TILE;
        __syncthreads();
        // stage gathered inputs: item = (k, r), r fastest so index loads are coalesced
        for (int it = tid; it < 27 * BW_TILE; it += LINR_BLOCK) {
            const int r = it % BW_TILE, k = it / BW_TILE;
            const int64_t row = row0 + r;
            int32_t j = -1;
            if (row < n) j = nbr[(int64_t)k * nbr_ld + row];
            float* dst = sx + r * XS + k * CIN;
            if (j >= 0) {
                const float* p = in + (int64_t)j * in_ld;
#pragma unroll
                for (int c = 0; c < CIN; ++c) dst[c] = p[c];
            } else {
#pragma unroll
                for (int c = 0; c < CIN; ++c) dst[c] = 0.0f;
            }
        }
        for (int it = tid; it < BW_TILE * COUT; it += LINR_BLOCK) {
            const int r = it / COUT, c = it % COUT;
            const int64_t row = row0 + r;
            sg[it] = row < n ? gout[row * gout_ld + c] : 0.0f;
            if (c == 0) sx[r * XS + P] = row < n ? 1.0f : 0.0f;
        }
        __syncthreads();
        if (active) {
            for (int r = grp; r < BW_TILE; r += G) {
                const float x = sx[r * XS + pair];
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = fmaf(x, sg[r * COUT + o], acc[o]);
            }
        }
    }
    // fold the G row groups in fixed order (group 0 + 1 + 2 ...)
    __syncthreads();
    if (active && grp > 0) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) sred[((grp - 1) * PP + pair) * COUT + o] = acc[o];
    }
    __syncthreads();
    if (grp == 0) {
        for (int g = 1; g < G; ++g) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) acc[o] += sred[((g - 1) * PP + pair) * COUT + o];
        }
        float* dst = slab + ((int64_t)blockIdx.x * PP + pair) * COUT;
#pragma unroll
        for (int o = 0; o < COUT; ++o) dst[o] = acc[o];
    }
}

// second pass: element e of the slab summed over blocks in a fixed association => bit-reproducible.  16 threads per element:
// thread (element, slice) sums the blocks slice, slice + 16, ... in four interleaved chains, the 16 slices are added in order
// through LDS.  (One thread per element - 7 workgroups of 512-step serial chains for a 8->8 kernel - took 41 us per call.)
#define SR_ELEMS 16
#define SR_SLICES (LINR_BLOCK / SR_ELEMS)
__global__ __launch_bounds__(LINR_BLOCK) void slab_reduce_k(const float* __restrict__ slab, int nblocks, int elems,
                                                            int split, float* __restrict__ dstA,
                                                            float* __restrict__ dstB, unsigned flags) {
    __shared__ float part[SR_SLICES][SR_ELEMS + 1];
    const int el = threadIdx.x % SR_ELEMS, sl = threadIdx.x / SR_ELEMS;
    const int e = blockIdx.x * SR_ELEMS + el;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (e < elems) {
        int b = sl;
        for (; b + 3 * SR_SLICES < nblocks; b += 4 * SR_SLICES) {
            s0 += slab[(int64_t)(b + 0 * SR_SLICES) * elems + e];
            s1 += slab[(int64_t)(b + 1 * SR_SLICES) * elems + e];
            s2 += slab[(int64_t)(b + 2 * SR_SLICES) * elems + e];
            s3 += slab[(int64_t)(b + 3 * SR_SLICES) * elems + e];
        }
        for (; b < nblocks; b += SR_SLICES) s0 += slab[(int64_t)b * elems + e];
    }
    part[sl][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl != 0 || e >= elems) return;
    float s = part[0][el];
#pragma unroll
    for (int q = 1; q < SR_SLICES; ++q) s += part[q][el];
    float* d = e < split ? (dstA ? dstA + e : nullptr) : (dstB ? dstB + (e - split) : nullptr);
    if (d == nullptr) return;
    *d = (flags & LINR_ACCUM) ? *d + s : s;
}

// the fixed-order sum of a [nblocks][elems] slab of per-block partials (what every weight-gradient entry returns): elements
// [0, split) go to dstA, the rest to dstB (kernel | bias)
extern "C" int linr_slab_reduce(const float* slab, int32_t nblocks, int32_t elems, int32_t split, float* dstA, float* dstB,
                                uint32_t flags, void* stream) {
    if (nblocks < 1 || elems < 1 || split < 0 || split > elems) return LINR_EINVAL;
    if (!slab || (!dstA && !dstB) || (flags & ~LINR_ACCUM)) return LINR_EINVAL;
    slab_reduce_k<<<linr_grid(elems, SR_ELEMS), LINR_BLOCK, 0, (hipStream_t)stream>>>(slab, nblocks, elems, split, dstA, dstB, flags);
    return linr_launch_rc();
}

// ---- backward-weight v3: no LDS staging, wave-per-row-group ---------------------------------------------------------
// gW[k][ci][co] = sum_r x[nbr[k][r]][ci] * g[r][co].  One WAVE owns a group of 8 consecutive rows at a time: lane
// (k, q) gathers the 4-channel quad q of the k-th neighbour's feature row (16 B) for all 8 rows up front (8 gathers in
// flight), the rows' output gradients g[r][0..COUT) are wave-uniform (scalar loads -> SGPR operands), so a lane does
// 4*COUT FMAs per vector load and keeps its 4 x COUT accumulators in registers over all rows the wave visits.
// Lane 27*XQ is the bias lane (x = (1,0,0,0)).  A block owns a contiguous row range, folds its waves through LDS in
// wave order and writes ONE partial per destination element; partials of all blocks are summed later in fixed order
// (deterministic; no float atomics).
#define WG_WAVES 8
template <int XQ, int COUT, bool PAD, bool VIDX>
__global__ __launch_bounds__(WG_WAVES * 64) void spconv_wgrad_k(
    const float* __restrict__ in, int in_ld, const float* __restrict__ gout, int gout_ld,
    const int32_t* __restrict__ nbr, int64_t nbr_ld, int64_t n, LinrWgradDst d) {
    constexpr int NI = 27 * XQ;            // lanes in use: one per (offset, input-channel quad)
    constexpr int NA = 4 * COUT;
    __shared__ float sacc[64 * (NA + 1)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane / XQ, q = lane % XQ;
    const int k = kk < 27 ? kk : 26;
    const bool live = lane < NI;
    const bool biasl = lane == NI;
    float acc[4][COUT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[c][o] = 0.0f;
    int64_t per = (n + gridDim.x - 1) / gridDim.x;
    per = (per + 7) & ~(int64_t)7;                       // 8-row groups never straddle blocks
    const int64_t b0 = (int64_t)blockIdx.x * per;
    const int64_t b1 = (b0 + per < n) ? b0 + per : n;
    const int32_t* nk = nbr + (int64_t)k * nbr_ld;
    for (int64_t g0 = b0 + 8 * wave; g0 < b1; g0 += 8 * WG_WAVES) {
        int32_t idx[8];
        if (VIDX && g0 + 8 <= n) {
            const int4 a = *reinterpret_cast<const int4*>(nk + g0);
            const int4 b = *reinterpret_cast<const int4*>(nk + g0 + 4);
            idx[0] = a.x; idx[1] = a.y; idx[2] = a.z; idx[3] = a.w;
            idx[4] = b.x; idx[5] = b.y; idx[6] = b.z; idx[7] = b.w;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) idx[u] = (g0 + u < n) ? nk[g0 + u] : -1;
        }
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (PAD) {
                x[u] = *reinterpret_cast<const float4*>(in + (int64_t)idx[u] * in_ld + 4 * q);
            } else {
                x[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (idx[u] >= 0) x[u] = *reinterpret_cast<const float4*>(in + (int64_t)idx[u] * in_ld + 4 * q);
            }
            if (biasl) x[u] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (g0 + u < n) {                             // wave-uniform
                const float* gp = gout + (g0 + u) * gout_ld;
                float g[COUT];
#pragma unroll
                for (int o = 0; o < COUT; ++o) g[o] = gp[o];
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    acc[0][o] = fmaf(x[u].x, g[o], acc[0][o]);
                    acc[1][o] = fmaf(x[u].y, g[o], acc[1][o]);
                    acc[2][o] = fmaf(x[u].z, g[o], acc[2][o]);
                    acc[3][o] = fmaf(x[u].w, g[o], acc[3][o]);
                }
            }
        }
    }
    // fold waves in wave order (fixed => reproducible)
    float* mine = sacc + lane * (NA + 1);
    for (int w = 0; w < WG_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    const float v = acc[c][o];
                    mine[c * COUT + o] = (w == 0) ? v : mine[c * COUT + o] + v;
                }
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* dst = d.base + (int64_t)blockIdx.x * d.block_stride;
        if (live) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ci = 4 * q + c;
                if (ci < d.cin_valid) {
#pragma unroll
                    for (int o = 0; o < COUT; ++o) dst[d.w_off + (kk * d.cin_valid + ci) * COUT + o] = mine[c * COUT + o];
                }
            }
        } else if (biasl) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) dst[d.b_off + o] = mine[o];
        }
    }
}

// internal: partial weight gradients of one conv3 into `d` (every one of `nblocks` blocks writes its partial).
// cin < 8 with cout == 8 runs the 8-wide kernel on the zero-extended 8-float row and drops channels >= cin.
int linr_conv3_wgrad_partial(const float* in, int in_ld, const float* gout, int gout_ld, const int32_t* nbr,
                             int64_t nbr_ld, int64_t n, int cin, int cout, LinrWgradDst d, int nblocks, unsigned flags,
                             hipStream_t s) {
    const bool vidx = (nbr_ld % 4 == 0) && linr_aligned16(nbr);
    const bool pad = (flags & LINR_PAD_ROW) != 0;
    d.cin_valid = cin;
#define LINR_GO(XQ, CO)                                                                                                  \
    do {                                                                                                                 \
        if (pad && vidx) spconv_wgrad_k<XQ, CO, true, true><<<nblocks, WG_WAVES * 64, 0, s>>>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, d);        \
        else if (pad) spconv_wgrad_k<XQ, CO, true, false><<<nblocks, WG_WAVES * 64, 0, s>>>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, d);          \
        else if (vidx) spconv_wgrad_k<XQ, CO, false, true><<<nblocks, WG_WAVES * 64, 0, s>>>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, d);         \
        else spconv_wgrad_k<XQ, CO, false, false><<<nblocks, WG_WAVES * 64, 0, s>>>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, d);                  \
        return linr_launch_rc();                                                                                         \
    } while (0)
    if (cin == 8 && cout == 8) LINR_GO(2, 8);
    if (cin == 8 && cout == 4) LINR_GO(2, 4);
    if (cin == 4 && cout == 4) LINR_GO(1, 4);
    if (cin < 8 && cout == 8 && in_ld >= 8) LINR_GO(2, 8);
#undef LINR_GO
    return LINR_EINVAL;
}

static int wg_blocks(int64_t n) {
    int64_t nb = (n + 63) / 64;            // at least ~64 rows per block
    if (nb > LINR_WG_BLOCKS) nb = LINR_WG_BLOCKS;
    return (int)(nb < 1 ? 1 : nb);
}

extern "C" size_t linr_spconv_bwd_weight_workspace_bytes(int64_t n, int32_t cin, int32_t cout) {
    if (n <= 0) return 0;
    const size_t v1 = (size_t)linr_reduce_blocks(n, BW_TILE) * (27 * cin + 1) * cout * sizeof(float);
    const size_t v3 = (size_t)wg_blocks(n) * (27 * cin + 1) * cout * sizeof(float);
    return v1 > v3 ? v1 : v3;
}

template <int CIN, int COUT>
static int launch_bwd_weight(const float* in, int in_ld, const float* gout, int gout_ld, const int32_t* nbr,
                             int64_t nbr_ld, int64_t n, float* gW, float* gb, unsigned flags, float* slab,
                             hipStream_t s) {
    const int nb = linr_reduce_blocks(n, BW_TILE);
    spconv_bwd_weight_k<CIN, COUT><<<nb, LINR_BLOCK, 0, s>>>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, slab);
    const int elems = (27 * CIN + 1) * COUT;
    slab_reduce_k<<<linr_grid(elems, SR_ELEMS), LINR_BLOCK, 0, s>>>(slab, nb, elems, 27 * CIN * COUT, gW, gb, flags);
    return linr_launch_rc();
}

extern "C" int linr_spconv_bwd_weight(const float* in, int32_t in_ld, const float* gout, int32_t gout_ld,
                                      const int32_t* nbr, int64_t nbr_ld, int64_t n, int32_t cin, int32_t cout,
                                      float* gW, float* gb, uint32_t flags, void* ws, size_t ws_bytes, void* stream) {
    if (n < 0 || nbr_ld < n || in_ld < cin || gout_ld < cout) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!in || !gout || !nbr || !gW || !ws) return LINR_EINVAL;
    if (ws_bytes < linr_spconv_bwd_weight_workspace_bytes(n, cin, cout)) return LINR_ENOSPC;
    if (((uintptr_t)ws) & 3u) return LINR_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    // fast path (v3): 16-byte gathers of channel quads; needs 16-byte aligned rows (cin < 8: full 8-float rows)
    const bool vec = linr_aligned16(in) && (in_ld % 4 == 0);
    if (vec && ((cin == 8 && (cout == 8 || cout == 4)) || (cin == 4 && cout == 4) || (cin < 8 && cout == 8 && in_ld >= 8))) {
        const int nb = wg_blocks(n);
        const int elems = (27 * cin + 1) * cout;
        LinrWgradDst d = {(float*)ws, elems, 0, 27 * cin * cout, cin};
        // rows with a zero pad row in front (LINR_PAD_ROW) take the executor's matrix-core kernel (csrc/fused.hip)
        int rc = ((flags & LINR_PAD_ROW) && (in_ld == 4 || in_ld == 8)) ? linr_conv3_wgrad_mfma(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, cin, cout, d, nb, s)
                                        : linr_conv3_wgrad_partial(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, cin, cout, d, nb, flags, s);
        if (rc) return rc;
        slab_reduce_k<<<linr_grid(elems, SR_ELEMS), LINR_BLOCK, 0, s>>>((const float*)ws, nb, elems, 27 * cin * cout, gW, gb, flags);
        return linr_launch_rc();
    }
#define LINR_BW_CASE(CI, CO) \
    if (cin == CI && cout == CO) return launch_bwd_weight<CI, CO>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, gW, gb, flags, (float*)ws, s);
    LINR_BW_CASE(8, 8) LINR_BW_CASE(8, 4) LINR_BW_CASE(4, 4)
    LINR_BW_CASE(1, 8) LINR_BW_CASE(2, 8) LINR_BW_CASE(3, 8) LINR_BW_CASE(4, 8)
    LINR_BW_CASE(5, 8) LINR_BW_CASE(6, 8) LINR_BW_CASE(7, 8)
#undef LINR_BW_CASE
    return LINR_EINVAL;
}

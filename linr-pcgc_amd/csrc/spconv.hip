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
template <int GIN, int GOUT, bool BWD, int LOADW, bool PAD>
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
    for (int k = 0; k < 27; ++k) {
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

// second pass: element e of the slab summed over blocks; consecutive lanes own consecutive elements (coalesced),
// four fixed interleaved partial sums keep loads in flight; the association is fixed => bit-reproducible.
__global__ __launch_bounds__(LINR_BLOCK) void slab_reduce_k(const float* __restrict__ slab, int nblocks, int elems,
                                                            int split, float* __restrict__ dstA,
                                                            float* __restrict__ dstB, unsigned flags) {
    const int e = blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (e >= elems) return;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int b = 0;
    for (; b + 4 <= nblocks; b += 4) {
        s0 += slab[(int64_t)(b + 0) * elems + e];
        s1 += slab[(int64_t)(b + 1) * elems + e];
        s2 += slab[(int64_t)(b + 2) * elems + e];
        s3 += slab[(int64_t)(b + 3) * elems + e];
    }
    for (; b < nblocks; ++b) s0 += slab[(int64_t)b * elems + e];
    const float s = (s0 + s1) + (s2 + s3);
    float* d = e < split ? (dstA ? dstA + e : nullptr) : (dstB ? dstB + (e - split) : nullptr);
    if (d == nullptr) return;
    *d = (flags & LINR_ACCUM) ? *d + s : s;
}

// ---- v2 backward-weight: no LDS staging --------------------------------------------------------------------------
// gW[k][ci][co] = sum_r x[nbr[k][r]][ci] * g[r][co].  One WAVE owns one row r at a time: lane (k, q) gathers the
// 4-channel quad q of the k-th neighbour's feature row (16 B), the row's output gradient g[r][0..COUT) is
// wave-uniform (scalar loads -> SGPR operands), so a lane does 4*COUT FMAs per 2 vector loads and keeps its
// 4 x COUT accumulators in registers over all rows the wave visits.  Lane 27*XQ is the bias lane (x = (1,0,0,0)).
// Waves of a block take rows round-robin; the block folds its waves through LDS in wave order and writes one slab.
#define BW2_WAVES 8
template <int XQ, int COUT, bool PAD>
__global__ __launch_bounds__(BW2_WAVES * 64) void spconv_bwd_weight2_k(
    const float* __restrict__ in, int in_ld, const float* __restrict__ gout, int gout_ld,
    const int32_t* __restrict__ nbr, int64_t nbr_ld, int64_t n, int cin_valid, float* __restrict__ slab) {
    constexpr int CIN = 4 * XQ;
    constexpr int NI = 27 * XQ;
    constexpr int NA = 4 * COUT;
    __shared__ float sacc[64 * (NA + 1)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = lane / XQ, q = lane % XQ;
    const bool gather = lane < NI;
    const bool biasl = lane == NI;
    float acc[4][COUT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[c][o] = 0.0f;
    // contiguous row range per block, rows dealt round-robin to its waves
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per;
    const int64_t b1 = (b0 + per < n) ? b0 + per : n;
    const int32_t* nk = nbr + (int64_t)(gather ? k : 0) * nbr_ld;
#pragma unroll 4
    for (int64_t r = b0 + wave; r < b1; r += BW2_WAVES) {
        float g[COUT];
        const float* gp = gout + r * gout_ld;            // r is wave-uniform: scalar loads
#pragma unroll
        for (int o = 0; o < COUT; ++o) g[o] = gp[o];
        float4 x = make_float4(biasl ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f);
        if (gather) {
            const int32_t j = nk[r];
            if (PAD || j >= 0) x = *reinterpret_cast<const float4*>(in + (int64_t)j * in_ld + 4 * q);
        }
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            acc[0][o] = fmaf(x.x, g[o], acc[0][o]);
            acc[1][o] = fmaf(x.y, g[o], acc[1][o]);
            acc[2][o] = fmaf(x.z, g[o], acc[2][o]);
            acc[3][o] = fmaf(x.w, g[o], acc[3][o]);
        }
    }
    // fold waves in wave order (fixed => reproducible)
    float* mine = sacc + lane * (NA + 1);
    for (int w = 0; w < BW2_WAVES; ++w) {
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
        float* dst = slab + (int64_t)blockIdx.x * ((27 * CIN + 1) * COUT);
        if (gather) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int o = 0; o < COUT; ++o) dst[(k * CIN + 4 * q + c) * COUT + o] = mine[c * COUT + o];
        } else if (biasl) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) dst[27 * CIN * COUT + o] = mine[o];
        }
    }
    (void)cin_valid;
}

// slab [nb][27*8+1][8] -> gW[27][cin][8] for cin < 8 (the 8-wide kernel ran on a zero-extended view of the input)
__global__ __launch_bounds__(LINR_BLOCK) void slab_reduce_narrow_k(const float* __restrict__ slab, int nblocks, int cin,
                                                                   float* __restrict__ gW, float* __restrict__ gb,
                                                                   unsigned flags) {
    const int e = blockIdx.x * LINR_BLOCK + threadIdx.x;      // element of the narrow result [(27*cin+1)*8]
    const int elems = (27 * cin + 1) * 8;
    if (e >= elems) return;
    const int o = e % 8, pc = e / 8;
    const int wide = (pc < 27 * cin) ? ((pc / cin) * 8 + (pc % cin)) * 8 + o : 27 * 64 + o;
    const int welems = (27 * 8 + 1) * 8;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int b = 0;
    for (; b + 4 <= nblocks; b += 4) {
        s0 += slab[(int64_t)(b + 0) * welems + wide];
        s1 += slab[(int64_t)(b + 1) * welems + wide];
        s2 += slab[(int64_t)(b + 2) * welems + wide];
        s3 += slab[(int64_t)(b + 3) * welems + wide];
    }
    for (; b < nblocks; ++b) s0 += slab[(int64_t)b * welems + wide];
    const float s = (s0 + s1) + (s2 + s3);
    float* d = (pc < 27 * cin) ? (gW ? gW + e : nullptr) : (gb ? gb + o : nullptr);
    if (d == nullptr) return;
    *d = (flags & LINR_ACCUM) ? *d + s : s;
}

static int bw2_blocks(int64_t n) {
    int64_t nb = (n + 63) / 64;            // at least ~64 rows per block
    if (nb > 512) nb = 512;               // 2 blocks x 8 waves per CU
    return (int)(nb < 1 ? 1 : nb);
}

extern "C" size_t linr_spconv_bwd_weight_workspace_bytes(int64_t n, int32_t cin, int32_t cout) {
    if (n <= 0) return 0;
    const size_t v1 = (size_t)linr_reduce_blocks(n, BW_TILE) * (27 * cin + 1) * cout * sizeof(float);
    const size_t v2 = (size_t)bw2_blocks(n) * (27 * 8 + 1) * 8 * sizeof(float);
    return v1 > v2 ? v1 : v2;
}

template <int CIN, int COUT>
static int launch_bwd_weight(const float* in, int in_ld, const float* gout, int gout_ld, const int32_t* nbr,
                             int64_t nbr_ld, int64_t n, float* gW, float* gb, unsigned flags, float* slab,
                             hipStream_t s) {
    const int nb = linr_reduce_blocks(n, BW_TILE);
    spconv_bwd_weight_k<CIN, COUT><<<nb, LINR_BLOCK, 0, s>>>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, slab);
    const int elems = (27 * CIN + 1) * COUT;
    slab_reduce_k<<<linr_grid(elems, LINR_BLOCK), LINR_BLOCK, 0, s>>>(slab, nb, elems, 27 * CIN * COUT, gW, gb, flags);
    return linr_launch_rc();
}

template <int XQ, int COUT>
static int launch_bwd_weight2(const float* in, int in_ld, const float* gout, int gout_ld, const int32_t* nbr,
                              int64_t nbr_ld, int64_t n, int cin, float* gW, float* gb, unsigned flags, float* slab,
                              hipStream_t s) {
    const int nb = bw2_blocks(n);
    if (flags & LINR_PAD_ROW)
        spconv_bwd_weight2_k<XQ, COUT, true><<<nb, BW2_WAVES * 64, 0, s>>>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, cin, slab);
    else
        spconv_bwd_weight2_k<XQ, COUT, false><<<nb, BW2_WAVES * 64, 0, s>>>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, cin, slab);
    if (cin == 4 * XQ) {
        const int elems = (27 * cin + 1) * COUT;
        slab_reduce_k<<<linr_grid(elems, LINR_BLOCK), LINR_BLOCK, 0, s>>>(slab, nb, elems, 27 * cin * COUT, gW, gb, flags);
    } else {
        const int elems = (27 * cin + 1) * 8;
        slab_reduce_narrow_k<<<linr_grid(elems, LINR_BLOCK), LINR_BLOCK, 0, s>>>(slab, nb, cin, gW, gb, flags);
    }
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
    // fast path: 16-byte gathers of channel quads (cin < 8 -> the 8-wide kernel on the full 8-float row, extra
    // channels discarded by the reduction), needs aligned rows of at least 4*XQ floats
    const bool vec = linr_aligned16(in) && (in_ld % 4 == 0);
    if (vec && cin == 8 && cout == 8) return launch_bwd_weight2<2, 8>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, 8, gW, gb, flags, (float*)ws, s);
    if (vec && cin == 8 && cout == 4) return launch_bwd_weight2<2, 4>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, 8, gW, gb, flags, (float*)ws, s);
    if (vec && cin == 4 && cout == 4) return launch_bwd_weight2<1, 4>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, 4, gW, gb, flags, (float*)ws, s);
    if (vec && cin < 8 && cout == 8 && in_ld >= 8) return launch_bwd_weight2<2, 8>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, cin, gW, gb, flags, (float*)ws, s);
#define LINR_BW_CASE(CI, CO) \
    if (cin == CI && cout == CO) return launch_bwd_weight<CI, CO>(in, in_ld, gout, gout_ld, nbr, nbr_ld, n, gW, gb, flags, (float*)ws, s);
    LINR_BW_CASE(8, 8) LINR_BW_CASE(8, 4) LINR_BW_CASE(4, 4)
    LINR_BW_CASE(1, 8) LINR_BW_CASE(2, 8) LINR_BW_CASE(3, 8) LINR_BW_CASE(4, 8)
    LINR_BW_CASE(5, 8) LINR_BW_CASE(6, 8) LINR_BW_CASE(7, 8)
#undef LINR_BW_CASE
    return LINR_EINVAL;
}

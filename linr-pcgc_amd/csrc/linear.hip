// Pointwise layers: MinkowskiConvolution(kernel_size=1) and nn.Linear (PointwiseMLP), forward / backward-data /
// backward-weight.  See include/linr_hip.h for the reference call sites.
// One lane owns one row; the weight matrix is wave-uniform (scalar loads).  Weight element (ci,co) lives at
// W[ci*ws_ci + co*ws_co] so the same kernels serve ME's [cin][cout] and torch's [cout][cin] layouts, and
// backward-data is the forward kernel with the strides swapped.
#include "common.h"

template <int CIN, int COUT>
__global__ __launch_bounds__(LINR_BLOCK) void linear_k(const float* __restrict__ in, int in_ld, int64_t n,
                                                       const float* __restrict__ W, int ws_ci, int ws_co,
                                                       const float* __restrict__ bias, const float* __restrict__ res,
                                                       int res_ld, const float* __restrict__ act, int act_ld,
                                                       float* __restrict__ out, int out_ld, unsigned flags) {
    const int64_t row = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (row >= n) return;
    float x[CIN];
    const float* p = in + row * in_ld;
    if ((CIN % 4 == 0) && (in_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0)) {
#pragma unroll
        for (int v = 0; v < CIN / 4; ++v) {
            const float4 t = *reinterpret_cast<const float4*>(p + 4 * v);
            x[4 * v] = t.x; x[4 * v + 1] = t.y; x[4 * v + 2] = t.z; x[4 * v + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CIN; ++i) x[i] = p[i];
    }
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = (bias != nullptr) ? bias[o] : 0.0f;
#pragma unroll
    for (int i = 0; i < CIN; ++i) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = fmaf(x[i], W[i * ws_ci + o * ws_co], acc[o]);
    }
    // epilogue order: + res, + old (ACCUM), * mask, ReLU
    if (res != nullptr) {
        const float* r = res + row * res_ld;
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] += r[o];
    }
    float* op = out + row * out_ld;
    if (flags & LINR_ACCUM) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] += op[o];
    }
    if (flags & LINR_RELU_MASK) {
        const float* a = act + row * act_ld;
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = a[o] > 0.0f ? acc[o] : 0.0f;
    }
    if (flags & LINR_RELU) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = fmaxf(acc[o], 0.0f);
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) op[o] = acc[o];
}

#define LINR_LINEAR_SHAPES(X) X(15, 16) X(16, 8) X(8, 24) X(24, 1) X(8, 4) X(4, 4) /* forward shapes */ \
                              X(8, 16) X(1, 24) X(24, 8) X(4, 8) X(16, 15)         /* transposed (bwd-data) */

int linr_linear_launch(const float* in, int in_ld, int64_t n, const float* W, int ws_ci, int ws_co,
                         const float* bias, int cin, int cout, const float* res, int res_ld, const float* act,
                         int act_ld, float* out, int out_ld, unsigned flags, hipStream_t s) {
    const unsigned grid = linr_grid(n, LINR_BLOCK);
#define LINR_CASE(CI, CO)                                                                                          \
    if (cin == CI && cout == CO) {                                                                                 \
        linear_k<CI, CO><<<grid, LINR_BLOCK, 0, s>>>(in, in_ld, n, W, ws_ci, ws_co, bias, res, res_ld, act, act_ld, \
                                                     out, out_ld, flags);                                          \
        return linr_launch_rc();                                                                                   \
    }
    LINR_LINEAR_SHAPES(LINR_CASE)
#undef LINR_CASE
    return LINR_EINVAL;
}

extern "C" int linr_linear_fwd(const float* in, int32_t in_ld, int64_t n, const float* W, int32_t ws_ci, int32_t ws_co,
                               const float* bias, int32_t cin, int32_t cout, const float* res, int32_t res_ld,
                               float* out, int32_t out_ld, uint32_t flags, void* stream) {
    if (n < 0 || in_ld < cin || out_ld < cout) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!in || !W || !out || (!bias && !(flags & LINR_NO_BIAS))) return LINR_EINVAL;
    if ((flags & LINR_RELU_MASK) || (res && res_ld < cout)) return LINR_EINVAL;
    return linr_linear_launch(in, in_ld, n, W, ws_ci, ws_co, (flags & LINR_NO_BIAS) ? nullptr : bias, cin, cout, res,
                         res_ld, nullptr, 0, out, out_ld, flags, (hipStream_t)stream);
}

extern "C" int linr_linear_bwd_data(const float* gout, int32_t gout_ld, int64_t n, const float* W, int32_t ws_ci,
                                    int32_t ws_co, int32_t cin, int32_t cout, const float* act, int32_t act_ld,
                                    float* gin, int32_t gin_ld, uint32_t flags, void* stream) {
    if (n < 0 || gout_ld < cout || gin_ld < cin) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!gout || !W || !gin) return LINR_EINVAL;
    if ((flags & LINR_RELU_MASK) && (!act || act_ld < cin)) return LINR_EINVAL;
    if (flags & LINR_RELU) return LINR_EINVAL;
    // gin[ci] = sum_co gout[co] * W(ci,co): forward kernel with roles and strides swapped
    return linr_linear_launch(gout, gout_ld, n, W, ws_co, ws_ci, nullptr, cout, cin, nullptr, 0, act, act_ld, gin, gin_ld,
                         flags, (hipStream_t)stream);
}

// ---- backward-weight: same two-pass slab scheme as the sparse convolution -------------------------------------
#define LBW_TILE 64

template <int CIN, int COUT>
__global__ __launch_bounds__(LINR_BLOCK) void linear_bwd_weight_k(const float* __restrict__ in, int in_ld,
                                                                  const float* __restrict__ gout, int gout_ld,
                                                                  int64_t n, float* __restrict__ slab) {
    constexpr int PP = CIN + 1;                       // + bias pseudo-input (x == 1)
    constexpr int G = LINR_BLOCK / PP;
    constexpr int XS = PP | 1;
    __shared__ float sx[LBW_TILE * XS];
    __shared__ float sg[LBW_TILE * COUT];
    __shared__ float sred[(G > 1) ? (G - 1) * PP * COUT : 1];
    const int tid = threadIdx.x;
    const int pair = tid % PP, grp = tid / PP;
    const bool active = grp < G;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.0f;
    const int64_t tiles = (n + LBW_TILE - 1) / LBW_TILE;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t row0 = t * LBW_TILE;
        __syncthreads();
        for (int it = tid; it < LBW_TILE * PP; it += LINR_BLOCK) {
            const int r = it / PP, c = it % PP;
            const int64_t row = row0 + r;
            float v = 0.0f;
            if (row < n) v = (c < CIN) ? in[row * in_ld + c] : 1.0f;
            sx[r * XS + c] = v;
        }
        for (int it = tid; it < LBW_TILE * COUT; it += LINR_BLOCK) {
            const int r = it / COUT, c = it % COUT;
            const int64_t row = row0 + r;
            sg[it] = row < n ? gout[row * gout_ld + c] : 0.0f;
        }
        __syncthreads();
        if (active) {
            for (int r = grp; r < LBW_TILE; r += G) {
                const float x = sx[r * XS + pair];
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = fmaf(x, sg[r * COUT + o], acc[o]);
            }
        }
    }
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

__global__ __launch_bounds__(LINR_BLOCK) void linear_slab_reduce_k(const float* __restrict__ slab, int nblocks, int cin,
                                                                   int cout, float* __restrict__ gW, int ws_ci,
                                                                   int ws_co, float* __restrict__ gb, unsigned flags) {
    const int e = blockIdx.x * LINR_BLOCK + threadIdx.x;
    const int elems = (cin + 1) * cout;
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
    const int ci = e / cout, co = e % cout;
    float* d = (ci < cin) ? (gW ? gW + ci * ws_ci + co * ws_co : nullptr) : (gb ? gb + co : nullptr);
    if (d == nullptr) return;
    *d = (flags & LINR_ACCUM) ? *d + s : s;
}

extern "C" size_t linr_linear_bwd_weight_workspace_bytes(int64_t n, int32_t cin, int32_t cout) {
    if (n <= 0) return 0;
    return (size_t)linr_reduce_blocks(n, LBW_TILE) * (cin + 1) * cout * sizeof(float);
}

extern "C" int linr_linear_bwd_weight(const float* in, int32_t in_ld, const float* gout, int32_t gout_ld, int64_t n,
                                      int32_t cin, int32_t cout, float* gW, int32_t ws_ci, int32_t ws_co, float* gb,
                                      uint32_t flags, void* ws, size_t ws_bytes, void* stream) {
    if (n < 0 || in_ld < cin || gout_ld < cout) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!in || !gout || !ws) return LINR_EINVAL;
    if (ws_bytes < linr_linear_bwd_weight_workspace_bytes(n, cin, cout)) return LINR_ENOSPC;
    hipStream_t s = (hipStream_t)stream;
    const int nb = linr_reduce_blocks(n, LBW_TILE);
    float* slab = (float*)ws;
#define LINR_CASE(CI, CO)                                                                                   \
    if (cin == CI && cout == CO) {                                                                          \
        linear_bwd_weight_k<CI, CO><<<nb, LINR_BLOCK, 0, s>>>(in, in_ld, gout, gout_ld, n, slab);            \
    } else
    LINR_CASE(15, 16) LINR_CASE(16, 8) LINR_CASE(8, 24) LINR_CASE(24, 1) LINR_CASE(8, 4) LINR_CASE(4, 4)
    return LINR_EINVAL;
#undef LINR_CASE
    linear_slab_reduce_k<<<linr_grid((cin + 1) * cout, LINR_BLOCK), LINR_BLOCK, 0, s>>>(slab, nb, cin, cout, gW, ws_ci,
                                                                                       ws_co, gb, flags);
    return linr_launch_rc();
}

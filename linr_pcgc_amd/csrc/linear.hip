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

#define LINR_LINEAR_SHAPES(X) X(15, 16) X(16, 8) X(8, 24) X(24, 1) X(8, 4) X(4, 4) X(8, 8) /* forward shapes; 8x8: the blocks of wider models */ \
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

// ---- backward-weight on the matrix cores -------------------------------------------------------------------------------
// gW[ci][co] = sum_r x[r][ci] * g[r][co] is X^T G with the reduction over rows: exactly the K dimension of
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, k-ordered => bit-reproducible).  A wave walks 16-row chunks of its
// block's contiguous row range; lane l feeds A[m = l&15][k = l>>4] = x[row0 + (l>>4)][m] and
// B[k = l>>4][n = l&15] = g[row0 + (l>>4)][n] straight from global memory (64-byte coalesced runs), no LDS.
// The bias gradient rides along as the pseudo input channel m == cin with x == 1.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define XTG_WAVES 4

template <int MT, int NT>
__global__ __launch_bounds__(XTG_WAVES * 64) void xtg_wgrad_k(const float* __restrict__ X, int x_ld, int M,
                                                             const float* __restrict__ G, int g_ld, int N, int64_t n,
                                                             LinrLinDst d, Grp gp = Grp()) {
    {   // group offsets: in = X, res = G, w/b = slab offsets
        const int gi = blockIdx.y;
        X += gp.in[gi]; G += gp.res[gi]; d.w_off += gp.w[gi]; d.b_off += gp.b[gi];
        if (gp.n[gi] > 0) n = gp.n[gi];
    }
    __shared__ float sacc[64 * (MT * NT * 4 + 1)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mm = lane & 15, rr = lane >> 4;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    int64_t per = (n + gridDim.x - 1) / gridDim.x;
    per = (per + 15) & ~(int64_t)15;
    const int64_t b0 = (int64_t)blockIdx.x * per;
    const int64_t b1 = (b0 + per < n) ? b0 + per : n;
    // two 16-row chunks per iteration: their loads are all in flight before the first MFMA (the kernel streams X and G once and
    // is bound by the latency of these 4-byte loads); chunks are accumulated in the same order as one by one
    for (int64_t c0 = b0 + 16 * wave; c0 < b1; c0 += 2 * 16 * XTG_WAVES) {
        float av[2][4][MT], bv[2][4][NT];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t cu = c0 + (int64_t)u * 16 * XTG_WAVES;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int64_t row = cu + 4 * s4 + rr;
                const bool ok = row < b1;
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    const int m = 16 * a + mm;
                    float v = 0.0f;
                    if (ok && m < M) v = X[row * x_ld + m];
                    if (ok && m == M) v = 1.0f;
                    av[u][s4][a] = v;
                }
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const int nn = 16 * b + mm;
                    bv[u][s4][b] = (ok && nn < N) ? G[row * g_ld + nn] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && c0 + 16 * XTG_WAVES >= b1) break;          // wave-uniform: the second chunk does not exist (zeros anyway)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s4][a], bv[u][s4][b], acc[a][b], 0, 0, 0);
        }
    }
    float* mine = sacc + lane * (MT * NT * 4 + 1);
    for (int w = 0; w < XTG_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
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
    if (wave == 0) {
        float* dst = d.base + (int64_t)blockIdx.x * d.block_stride;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = 16 * a + rr * 4 + j;          // C/D map: row = (lane>>4)*4 + reg, col = lane&15
                    const int nn = 16 * b + mm;
                    if (nn < N) {
                        const float v = mine[(a * NT + b) * 4 + j];
                        if (m < M) dst[d.w_off + m * d.ws_ci + nn * d.ws_co] = v;
                        else if (m == M) dst[d.b_off + nn] = v;
                    }
                }
    }
}

int linr_linear_wgrad_partial(const float* in, int in_ld, const float* gout, int gout_ld, int64_t n, int cin, int cout,
                              LinrLinDst d, int nblocks, hipStream_t s, const Grp* gp, int ngroups) {
    const int mt = (cin + 1 + 15) / 16, nt = (cout + 15) / 16;
    const Grp g0 = gp ? *gp : Grp();
    const dim3 grid(nblocks, ngroups);
#define LINR_GO(A, B) do { xtg_wgrad_k<A, B><<<grid, XTG_WAVES * 64, 0, s>>>(in, in_ld, cin, gout, gout_ld, cout, n, d, g0); return linr_launch_rc(); } while (0)
    if (mt == 1 && nt == 1) LINR_GO(1, 1);
    if (mt == 2 && nt == 1) LINR_GO(2, 1);
    if (mt == 1 && nt == 2) LINR_GO(1, 2);
    if (mt == 2 && nt == 2) LINR_GO(2, 2);
#undef LINR_GO
    return LINR_EINVAL;
}

#define LR_ELEMS 16
#define LR_SLICES (LINR_BLOCK / LR_ELEMS)
// 16 threads per element (block slices in four interleaved chains, slices added in order): as slab_reduce_k of spconv.hip
__global__ __launch_bounds__(LINR_BLOCK) void linear_slab_reduce_k(const float* __restrict__ slab, int nblocks, int cin,
                                                                   int cout, float* __restrict__ gW, int ws_ci,
                                                                   int ws_co, float* __restrict__ gb, unsigned flags, int64_t stride) {
    __shared__ float part[LR_SLICES][LR_ELEMS + 1];
    const int el = threadIdx.x % LR_ELEMS, sl = threadIdx.x / LR_ELEMS;
    const int e = blockIdx.x * LR_ELEMS + el;
    const int elems = (cin + 1) * cout;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (e < elems) {
        int b = sl;
        for (; b + 3 * LR_SLICES < nblocks; b += 4 * LR_SLICES) {
            s0 += slab[(int64_t)(b + 0 * LR_SLICES) * stride + e];
            s1 += slab[(int64_t)(b + 1 * LR_SLICES) * stride + e];
            s2 += slab[(int64_t)(b + 2 * LR_SLICES) * stride + e];
            s3 += slab[(int64_t)(b + 3 * LR_SLICES) * stride + e];
        }
        for (; b < nblocks; b += LR_SLICES) s0 += slab[(int64_t)b * stride + e];
    }
    part[sl][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl != 0 || e >= elems) return;
    float s = part[0][el];
#pragma unroll
    for (int q = 1; q < LR_SLICES; ++q) s += part[q][el];
    const int ci = e / cout, co = e % cout;
    float* d = (ci < cin) ? (gW ? gW + ci * ws_ci + co * ws_co : nullptr) : (gb ? gb + co : nullptr);
    if (d == nullptr) return;
    *d = (flags & LINR_ACCUM) ? *d + s : s;
}

int linr_lin_blocks(int64_t n) {
    int64_t nb = (n + 255) / 256;
    if (nb > LINR_WG_BLOCKS) nb = LINR_WG_BLOCKS;
    return (int)(nb < 1 ? 1 : nb);
}
static int lin_blocks(int64_t n) { return linr_lin_blocks(n); }

// slab rows of `stride` floats, each a dense [cin + 1][cout] partial (bias = row cin) in front: fixed-order sum, scattered to the strides
int linr_linear_slab_reduce_launch(const float* slab, int nblocks, int64_t stride, int cin, int cout, float* gW, int ws_ci, int ws_co,
                                   float* gb, unsigned flags, hipStream_t s) {
    linear_slab_reduce_k<<<linr_grid((cin + 1) * cout, LR_ELEMS), LINR_BLOCK, 0, s>>>(slab, nblocks, cin, cout, gW, ws_ci, ws_co, gb, flags,
                                                                                       stride);
    return linr_launch_rc();
}

extern "C" size_t linr_linear_bwd_weight_workspace_bytes(int64_t n, int32_t cin, int32_t cout) {
    if (n <= 0) return 0;
    return (size_t)lin_blocks(n) * (cin + 1) * cout * sizeof(float);
}

extern "C" int linr_linear_bwd_weight(const float* in, int32_t in_ld, const float* gout, int32_t gout_ld, int64_t n,
                                      int32_t cin, int32_t cout, float* gW, int32_t ws_ci, int32_t ws_co, float* gb,
                                      uint32_t flags, void* ws, size_t ws_bytes, void* stream) {
    if (n < 0 || in_ld < cin || gout_ld < cout || cin < 1 || cout < 1 || cin > 31 || cout > 32) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!in || !gout || !ws) return LINR_EINVAL;
    if (ws_bytes < linr_linear_bwd_weight_workspace_bytes(n, cin, cout)) return LINR_ENOSPC;
    hipStream_t s = (hipStream_t)stream;
    const int nb = lin_blocks(n);
    // dense [cin+1][cout] partial per block (bias = row cin); the reduce pass scatters to the caller's strides
    LinrLinDst d = {(float*)ws, (int64_t)(cin + 1) * cout, 0, cout, 1, (int64_t)cin * cout};
    int rc = linr_linear_wgrad_partial(in, in_ld, gout, gout_ld, n, cin, cout, d, nb, s);
    if (rc) return rc;
    return linr_linear_slab_reduce_launch((const float*)ws, nb, (int64_t)(cin + 1) * cout, cin, cout, gW, ws_ci, ws_co, gb, flags, s);
}

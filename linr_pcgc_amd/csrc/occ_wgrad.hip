// Weight gradients of the first convolutions of the 7 outter blocks from ONE gather of the occupancy rows (csrc/net.hip is the
// caller; the forward counterpart is occ_conv7_k in csrc/fused.hip).
//
// Block b (1..7) starts with conv3(occ[:, :b] -> 8) on the SAME occupancy rows (models/upsample.py:206-214), so the seven weight
// gradients
//     gW_b[k][ci][co] = sum_r occ[nbr(r, k)][ci] * g_b[r][co]          (ci < b),      gb_b[co] = sum_r g_b[r][co]
// share every gathered row.  As seven stand-alone launches (spconv_wgrad_t_k) they cost seven gather passes and 7 x 432 MFMAs
// per 64 rows although only the 28 (block, input channel) pairs of the triangle carry work.  Here a wave gathers the 27
// neighbours of its 64 rows once, parks them TRANSPOSED in a wave-private LDS image ([input channel][tap][row]) beside the rows'
// own gradients ([block, output channel][row]) and reduces over the rows on the matrix cores with the rows as the K dimension of
// v_mfma_f32_4x4x1 (one row per instruction, no operand broadcast): the 16 blocks of an instruction are 16 of the 56 (block, input
// channel, output quad) combos - four "sets" cover them -, A = the combo's g_b[r][4 h + i], B = occ[nbr(r, 4 q + j)][ci] for the four
// taps of tap quad q, D = gW_b[4 q + j][ci][4 h + i].  27 taps = 7 tap quads; the spare 28th slot holds ones, which makes
// D[ci = 0] of that slot the bias gradient.  1792 MFMAs per 64 rows (of which 1512 useful) instead of 3024, one gather pass instead
// of seven; both operands come out of LDS as 16-byte reads of four consecutive rows.
// One wave per SIMD (112 accumulator registers), long-lived workgroups (one round, like csrc/fused_bwd.hip), one slab row per
// workgroup in the reduction contract of every weight-gradient kernel.
#include "common.h"
#include "conv_common.h"

#define OW_WAVES 4
#define OW_PS 68                          // floats between two planes: 64 rows + 4 (16-byte reads of neighbouring planes: other banks)
#define OW_CS (8 * OW_PS + 16)            // floats between the plane groups of two input channels of the occupancy image
#define OW_OCC_F (8 * OW_CS)              // occupancy image of one chunk (8 taps): [ci][slot][row]
#define OW_G_F (56 * OW_PS)               // own gradients: [7 blocks x 8 channels][row]
#define OW_WAVE_F (OW_OCC_F + OW_G_F)     // 8288 floats = 33,152 bytes per wave
#define OW_ACC 113                        // fold: floats per lane (112 accumulators, odd stride)

struct OccWgArgs {
    const float* occ;             // [n][8], zero row at index -1
    const float* g;               // gA of outter block 1; block b at g + goff[b - 1]   ([n][8] each)
    int64_t goff[7];
    int64_t w_off[7], b_off[7];   // slab offsets of kernel [27][b][8] and bias [8] of block b
    float* big;
    int64_t block_stride;
    int tiles_per_wave;
};

__host__ __device__ constexpr int ow_g(int p) { return p < 1 ? 0 : p < 3 ? 1 : p < 6 ? 2 : p < 10 ? 3 : p < 15 ? 4 : p < 21 ? 5 : 6; }

__global__ __launch_bounds__(OW_WAVES * 64, 1) void occ_wgrad7_k(OccWgArgs a, const int32_t* __restrict__ lo,
                                                                 const uint32_t* __restrict__ mask, int64_t ld, int64_t n) {
    __shared__ float smem[OW_WAVES * OW_WAVE_F];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = lane >> 2, t = lane & 3;
    float* occT = smem + wave * OW_WAVE_F;
    float* gT = occT + OW_OCC_F;
    // the lane's operand planes per set: combo c = 16 s + blk = 2 pair + h, pair = (block g, input channel ci) of the triangle
    int aofs[4], bofs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 16 * s + blk, cc = c < 56 ? c : 0;          // combos 56..63 do not exist: they recompute combo 0, discarded
        const int p = cc >> 1, h = cc & 1, g = ow_g(p), ci = p - g * (g + 1) / 2;
        aofs[s] = (g * 8 + 4 * h + t) * OW_PS;
        bofs[s] = ci * OW_CS + t * OW_PS;
    }
    f32x4 acc[4][7];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 7; ++q) acc[s][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    const int64_t T64 = (n + 63) >> 6;
    const int64_t tb0 = (int64_t)blockIdx.x * (OW_WAVES * a.tiles_per_wave);
    const int64_t tb1 = (tb0 + OW_WAVES * a.tiles_per_wave < T64) ? tb0 + OW_WAVES * a.tiles_per_wave : T64;
    const char* pad = reinterpret_cast<const char*>(a.occ - 8);
    const char* lob = reinterpret_cast<const char*>(lo);
    const uint32_t ld4 = (uint32_t)ld << 2;
    auto idx_load = [&](int64_t row, int32_t (&raw)[10]) {
        const uint32_t rb = (uint32_t)row << 2;
        raw[9] = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(mask) + rb);
#pragma unroll
        for (int q = 0; q < 9; ++q) raw[q] = *reinterpret_cast<const int32_t*>(lob + (rb + (uint32_t)q * ld4));
    };
    auto idx_decode = [&](const int32_t (&raw)[10], uint32_t (&off)[27]) {          // forward direction: tap k = q + 9 j
        const uint32_t m = (uint32_t)raw[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const uint32_t L = ((uint32_t)raw[q] + 1u) << 5;
            const int m0 = __builtin_amdgcn_sbfe(m, 3 * q, 1), m1 = __builtin_amdgcn_sbfe(m, 3 * q + 1, 1),
                      m2 = __builtin_amdgcn_sbfe(m, 3 * q + 2, 1);
            const uint32_t t1 = L + (uint32_t)__mul24(m0, -32);
            const uint32_t t2 = t1 + (uint32_t)__mul24(m1, -32);
            off[q] = L & (uint32_t)m0; off[q + 9] = t1 & (uint32_t)m1; off[q + 18] = t2 & (uint32_t)m2;
        }
    };
    auto own_load = [&](int64_t row_raw, f32x4 (&G)[14]) {
        const bool live = row_raw < n;
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            const float* gp = a.g + a.goff[b] + row_raw * 8;
            G[2 * b] = live ? *reinterpret_cast<const f32x4*>(gp) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            G[2 * b + 1] = live ? *reinterpret_cast<const f32x4*>(gp + 4) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    int64_t tile = tb0 + wave;
    if (tile < tb1) {                            // wave-uniform
        uint32_t off[27], offn0[8];
        int32_t raw[10];
        f32x4 x[8][2];                           // the gathered taps of the next chunk
        f32x4 G[14];                             // own gradients of the next tile
        {
            const int64_t r = (tile << 6) + lane;
            idx_load(r < n ? r : n - 1, raw);
            idx_decode(raw, off);
            own_load(r, G);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u][0] = *reinterpret_cast<const f32x4*>(pad + off[u]);
                x[u][1] = *reinterpret_cast<const f32x4*>(pad + off[u] + 16);
            }
        }
        bool first = true;
        for (; tile < tb1; tile += OW_WAVES) {
            const int64_t ntile = (tile + OW_WAVES < tb1) ? tile + OW_WAVES : tile;          // the last tile "prefetches" itself
            const int64_t nrow_raw = (ntile << 6) + lane;
            if (!first) idx_decode(raw, off);          // the index words were loaded during the previous tile (wave-uniform branch)
            first = false;
            // own gradients of this tile -> [block, channel][row]  (rows >= n: zeros, so whatever the dead lanes gathered is inert)
#pragma unroll
            for (int q = 0; q < 56; ++q) gT[q * OW_PS + lane] = G[q >> 2][q & 3];
            static_for<4>([&](auto chc) {
                constexpr int ch = decltype(chc)::value;
                constexpr int ntap = ch < 3 ? 8 : 3, ntq = ch < 3 ? 2 : 1;
                // this chunk's taps -> [ci][slot][row]
#pragma unroll
                for (int u = 0; u < ntap; ++u)
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci) occT[ci * OW_CS + u * OW_PS + lane] = x[u][ci >> 2][ci & 3];
                if constexpr (ch == 3) {          // slot 27: ones (bias gradient)
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci) occT[ci * OW_CS + 3 * OW_PS + lane] = 1.0f;
                }
                // the next chunk's gathers fly during this chunk's MFMAs; the last chunk gathers the first one of the next tile
                if constexpr (ch == 1) idx_load(nrow_raw < n ? nrow_raw : n - 1, raw);
                if constexpr (ch == 3) {          // taps 0..7 of the next tile (column q, lowest plane): the rest is decoded at its top
                    own_load(nrow_raw, G);
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        offn0[q] = (((uint32_t)raw[q] + 1u) << 5) & (uint32_t)__builtin_amdgcn_sbfe((uint32_t)raw[9], 3 * q, 1);
                }
                {
                    constexpr int nt = ch < 2 ? 8 : (ch == 2 ? 3 : 8);
#pragma unroll
                    for (int u = 0; u < nt; ++u) {
                        const uint32_t o = ch < 3 ? off[ch < 3 ? 8 * (ch + 1) + u : 0] : offn0[u];
                        x[u][0] = *reinterpret_cast<const f32x4*>(pad + o);
                        x[u][1] = *reinterpret_cast<const f32x4*>(pad + o + 16);
                    }
                }
                // rows as the K dimension: 16 row quads x 4 rows x (4 sets x ntq tap quads)
                // operands of two sets (half a row quad) are read one half ahead
                float4 A[2][2], B[2][2][2];
                auto rd = [&](int rq, int half, int buf) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int s = 2 * half + s2;
                        A[buf][s2] = *reinterpret_cast<const float4*>(gT + aofs[s] + 4 * rq);
#pragma unroll
                        for (int u = 0; u < ntq; ++u)
                            B[buf][s2][u] = *reinterpret_cast<const float4*>(occT + bofs[s] + 4 * u * OW_PS + 4 * rq);
                    }
                };
                rd(0, 0, 0);
                static_for<32>([&](auto hc) {
                    constexpr int hh = decltype(hc)::value, half = hh & 1;          // row quad hh >> 1
                    __builtin_amdgcn_sched_barrier(0);          // keeps the reads of later row quads from being hoisted (registers)
                    if constexpr (hh + 1 < 32) rd((hh + 1) >> 1, (hh + 1) & 1, (hh + 1) & 1);
                    static_for<4>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        static_for<2>([&](auto sc) {
                            constexpr int s2 = decltype(sc)::value, s = 2 * half + s2;
                            const float4 av = A[hh & 1][s2];
                            const float af = e == 0 ? av.x : e == 1 ? av.y : e == 2 ? av.z : av.w;
                            static_for<ntq>([&](auto uc) {
                                constexpr int u = decltype(uc)::value;
                                const float4 bv = B[hh & 1][s2][u];
                                const float bf = e == 0 ? bv.x : e == 1 ? bv.y : e == 2 ? bv.z : bv.w;
                                acc[s][2 * ch + u] = __builtin_amdgcn_mfma_f32_4x4x1f32(af, bf, acc[s][2 * ch + u], 0, 0, 0);
                            });
                        });
                    });
                });
            });
        }
    }
    __syncthreads();
    // ---- fold: the four waves in order, one slab row per workgroup -----------------------------------------------------------------
    float* sacc = smem;                          // [wave][lane][OW_ACC]
    {
        float* mine = sacc + (wave * 64 + lane) * OW_ACC;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < 7; ++q)
#pragma unroll
                for (int v = 0; v < 4; ++v) mine[(s * 7 + q) * 4 + v] = acc[s][q][v];
    }
    // the blocks' slab offsets through LDS: indexed straight out of the argument struct they are a dependent vector load per element
    __shared__ int64_t s_off[14];
    if (threadIdx.x < 7) s_off[threadIdx.x] = a.w_off[threadIdx.x];
    else if (threadIdx.x < 14) s_off[threadIdx.x] = a.b_off[threadIdx.x - 7];
    __syncthreads();
    float* dst = a.big + (int64_t)blockIdx.x * a.block_stride;
    // outputs: for block g (cin = g + 1): kernel (27 x cin x 8) then bias (8): 28 x 8 x 27 + 56 elements
    for (int e = threadIdx.x; e < 28 * 8 * 28; e += OW_WAVES * 64) {
        const int co = e & 7, p = (e >> 3) % 28, tap = e / (28 * 8);          // tap 27 = the ones slot
        const int g = ow_g(p), ci = p - g * (g + 1) / 2;
        if (tap == 27 && ci != 0) continue;
        const int c = 2 * p + (co >> 2), s = c >> 4, bk = c & 15, q = tap >> 2, tt = tap & 3, v = co & 3;
        const int idx = (bk * 4 + tt) * OW_ACC + (s * 7 + q) * 4 + v;
        const float sum = ((sacc[idx] + sacc[64 * OW_ACC + idx]) + sacc[2 * 64 * OW_ACC + idx]) + sacc[3 * 64 * OW_ACC + idx];
        if (tap < 27) dst[s_off[g] + (tap * (g + 1) + ci) * 8 + co] = sum;
        else dst[s_off[7 + g] + co] = sum;
    }
}

// Grid: one round of long-lived workgroups (one per CU: registers and LDS), never more than the slab's nb rows.
int linr_occ_wgrad7_rows(int64_t n, int nb, int cus, int* tiles_per_wave) {
    const int64_t t64 = (n + 63) >> 6;
    int64_t target = cus < 1 ? 1 : cus;
    if (target > nb) target = nb;
    int64_t m = (t64 + OW_WAVES * target - 1) / (OW_WAVES * target);
    if (m < 1) m = 1;
    if (tiles_per_wave) *tiles_per_wave = (int)m;
    int64_t blocks = (t64 + OW_WAVES * m - 1) / (OW_WAVES * m);
    return (int)(blocks < 1 ? 1 : blocks);
}

// occ: occupancy [n][8] with the zero row in front; g[b]: gradient of block b + 1's first-conv output (after its ReLU mask) [n][8];
// w_off / b_off: slab offsets of the seven kernels [27][b + 1][8] / biases; *rows_written: slab rows written (the grid's blocks)
int linr_occ_wgrad7_launch(const float* occ, const float* const* g, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                           float* big, int64_t block_stride, const int64_t* w_off, const int64_t* b_off, int nb, hipStream_t s,
                           int* rows_written) {
    if (rows_written) *rows_written = 0;
    if (n == 0) return 0;
    static const int cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
        return v;
    }();
    OccWgArgs a;
    a.occ = occ; a.g = g[0]; a.big = big; a.block_stride = block_stride;
    for (int b = 0; b < 7; ++b) { a.goff[b] = g[b] - g[0]; a.w_off[b] = w_off[b]; a.b_off[b] = b_off[b]; }
    const int blocks = linr_occ_wgrad7_rows(n, nb, cus, &a.tiles_per_wave);
    if (rows_written) *rows_written = blocks;
    occ_wgrad7_k<<<blocks, OW_WAVES * 64, 0, s>>>(a, lo, mask, ld, n);
    return linr_launch_rc();
}

// Op-level entry (tests): slab [nblocks][6104] = for b = 1..7: kernel [27][b][8] then bias [8], block after block
extern "C" int linr_occ_wgrad7(const float* occ, const float* const* gout7_h, const int32_t* lo, const uint32_t* mask, int64_t ld,
                               int64_t n, float* slab, int32_t nblocks, int32_t* rows_written_h, void* stream) {
    if (n < 0 || ld < n || nblocks < 1 || !rows_written_h) return LINR_EINVAL;
    *rows_written_h = 0;
    if (n == 0) return 0;
    if (!occ || !gout7_h || !lo || !mask || !slab) return LINR_EINVAL;
    if (!linr_aligned16(occ)) return LINR_EALIGN;
    for (int b = 0; b < 7; ++b) {
        if (!gout7_h[b]) return LINR_EINVAL;
        if (!linr_aligned16(gout7_h[b])) return LINR_EALIGN;
    }
    if ((uint64_t)(n + 1) * 32u >= 0xFFFFFFFFull || ld >= ((int64_t)1 << 26)) return LINR_EINVAL;
    int64_t w_off[7], b_off[7], cur = 0;
    for (int b = 0; b < 7; ++b) { w_off[b] = cur; cur += 27 * (b + 1) * 8; b_off[b] = cur; cur += 8; }
    int rows = 0;
    const int rc = linr_occ_wgrad7_launch(occ, gout7_h, lo, mask, ld, n, slab, 6104, w_off, b_off, nblocks, (hipStream_t)stream, &rows);
    *rows_written_h = rows;
    return rc;
}

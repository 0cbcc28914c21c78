// Whole-network executor: scale-context MLP -> block_in -> 8 x (prune conv + head MLP + sigmoid/BCE, outter block)
// forward, and the hand-derived backward, as one stream-ordered launch sequence over a flat parameter buffer and a
// caller-owned activation arena.  Replaces LINR_PCGC_Model.logic_core/forward (models/model_core.py:38-81),
// CNP.forward (models/upsample.py:163-217), make_block (:88-97), InceptionResNet.forward (models/resnet.py:55-60)
// and the autograd graph main.py:315-316 differentiates.
//
// Arena: every activation / gradient matrix is [1 + rows][ld] with an all-zero row in FRONT (row index -1), so the
// sparse convolutions read absent neighbours from it (LINR_PAD_ROW) with no branch.
#include "common.h"
#include "conv_common.h"
#include "layout.h"
#include "sce.h"
#include "net_shared.h"
#include <math.h>
#include <stdlib.h>
#include <vector>
#include <mutex>
#include <atomic>

#define TRY(e) do { int rc_ = (e); if (rc_) return rc_; } while (0)

// ---- parameter layout (reference parameters() order) ------------------------------------------------------------
extern "C" int linr_abi_version(void) { return LINR_ABI_VERSION; }

extern "C" int64_t linr_param_count(int32_t scale_num, int32_t block_layers) {
    Layout L;
    return make_layout(L, scale_num, block_layers) ? L.total : (int64_t)LINR_EINVAL;
}

// ---- arena ------------------------------------------------------------------------------------------------------
struct Arena {
    int64_t rows;
    float* base;
    int64_t cur;                 // floats
    int64_t pad_off[200];        // float offset of every pad row
    int pad_w[200];
    int npad;
    // forward (saved for backward)
    float *X0, *OCC;
    float *A[8], *H[8], *M[8], *I[8], *O[8];
    float *Hx[MAX_BL - 1], *Mx[MAX_BL - 1], *Ix[MAX_BL - 1];        // Inception layers 1.. of block_in (block_layers > 1)
    float *gIx[MAX_BL - 1], *gMx[MAX_BL - 1], *gHx[MAX_BL - 1];
    float *C[8], *HH[8], *Z[8], *P[8];
    // backward scratch
    float *gZ, *gHH, *gXG, *gX0;
    float *gC[8], *gO[8], *gI[8], *gA[8], *gH[8], *gM[8];   // one set per stage / block (the grouped launches need them side by side)
    float* BIG;                  // [LINR_WG_BLOCKS][n_params] per-block partial weight gradients
    float* GSUM;                 // [n_params] their fixed-order sum (the gradient of this backward call)
    int64_t n_params;
    void* slab; size_t slab_bytes;
};

static float* arena_mat(Arena& a, int ld) {
    a.cur = (a.cur + 3) & ~(int64_t)3;                       // 16-byte alignment
    a.pad_off[a.npad] = a.cur; a.pad_w[a.npad] = ld; a.npad++;
    float* p = a.base ? a.base + a.cur + ld : nullptr;      // row 0 starts after the pad row
    a.cur += (a.rows + 1) * (int64_t)ld;
    return p;
}

static size_t slab_need(int64_t rows) {
    const size_t a = linr_bce_workspace_bytes(rows), b = (size_t)8 * linr_grid(rows, LINR_BLOCK) * sizeof(double);
    return (a > b ? a : b) + 64;
}

static void make_arena(Arena& a, int64_t rows, float* base, int64_t n_params, int block_layers = 1) {
    a.rows = rows; a.base = base; a.cur = 0; a.npad = 0;
    a.X0 = arena_mat(a, 8); a.OCC = arena_mat(a, 8);
    for (int b = 0; b < 8; ++b) {
        a.A[b] = arena_mat(a, 8); a.H[b] = arena_mat(a, 8); a.M[b] = arena_mat(a, 4);
        a.I[b] = arena_mat(a, 8); a.O[b] = arena_mat(a, 8);
    }
    for (int k = 0; k < 8; ++k) {
        a.C[k] = arena_mat(a, 8); a.HH[k] = arena_mat(a, 24); a.Z[k] = arena_mat(a, 1); a.P[k] = arena_mat(a, 1);
    }
    a.gZ = arena_mat(a, 1); a.gHH = arena_mat(a, 24); a.gXG = arena_mat(a, 8); a.gX0 = arena_mat(a, 8);
    for (int i = 0; i < 8; ++i) {
        a.gC[i] = arena_mat(a, 8); a.gO[i] = arena_mat(a, 8); a.gI[i] = arena_mat(a, 8); a.gA[i] = arena_mat(a, 8);
        a.gH[i] = arena_mat(a, 8); a.gM[i] = arena_mat(a, 4);
    }
    for (int l = 0; l + 1 < MAX_BL; ++l) {
        a.Hx[l] = a.Mx[l] = a.Ix[l] = a.gIx[l] = a.gMx[l] = a.gHx[l] = nullptr;
        if (l + 1 < block_layers) {
            a.Hx[l] = arena_mat(a, 8); a.Mx[l] = arena_mat(a, 4); a.Ix[l] = arena_mat(a, 8);
            a.gIx[l] = arena_mat(a, 8); a.gMx[l] = arena_mat(a, 4); a.gHx[l] = arena_mat(a, 8);
        }
    }
    a.n_params = n_params;
    a.cur = (a.cur + 15) & ~(int64_t)15;
    a.GSUM = base ? base + a.cur : nullptr; a.cur += (n_params + 15) & ~(int64_t)15;
    a.BIG = base ? base + a.cur : nullptr; a.cur += (int64_t)LINR_WG_BLOCKS * n_params;
    a.cur = (a.cur + 15) & ~(int64_t)15;                     // 64-byte alignment for the slab (doubles inside)
    a.slab = base ? (void*)(base + a.cur) : nullptr;
    a.slab_bytes = slab_need(rows);
    a.cur += (int64_t)((a.slab_bytes + 3) / 4);
}

extern "C" size_t linr_net_arena_bytes(int64_t rows, int32_t block_layers) {
    if (rows < 0) return 0;
    if (block_layers < 1) block_layers = 1;
    if (block_layers > MAX_BL) return 0;
    Arena a;
    Layout L;
    make_layout(L, MAX_SCALES, block_layers);                 // sized for the largest scale_num: one arena serves any model of this depth
    make_arena(a, rows, nullptr, L.total, block_layers);
    return (size_t)a.cur * sizeof(float) + 64;
}

// ghid[r] = (W2^T gx0[r]) * (hid[r] > 0)     (linear_k<8,16> with the ReLU mask, weights of r's scale)
__global__ __launch_bounds__(LINR_BLOCK) void sce_bwd_k(const float* __restrict__ P, SceArgs a, int64_t n,
                                                        const float* __restrict__ gx0, const float* __restrict__ hid,
                                                        float* __restrict__ ghid) {
    int s;
    const int64_t r = sce_row_of(a, (int)blockIdx.x, s);
    if (r < 0) return;
    const float* W2 = P + a.w2[s];
    const float4 g0 = *reinterpret_cast<const float4*>(gx0 + r * 8);
    const float4 g1 = *reinterpret_cast<const float4*>(gx0 + r * 8 + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = fmaf(g[i], W2[i * 16 + o], acc[o]);
    const float4* hp = reinterpret_cast<const float4*>(hid + r * 16);
    float4* op = reinterpret_cast<float4*>(ghid + r * 16);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float4 h = hp[v];
        op[v] = make_float4(h.x > 0.0f ? acc[4 * v] : 0.0f, h.y > 0.0f ? acc[4 * v + 1] : 0.0f,
                            h.z > 0.0f ? acc[4 * v + 2] : 0.0f, h.w > 0.0f ? acc[4 * v + 3] : 0.0f);
    }
}

// The whole backward of the scale context in ONE launch (grid: slab rows x scales): per row ghid = (W2^T gx0) * (hid > 0) with the
// scale's weights from the scalar cache (the fmaf chain of sce_bwd_k), and all four parameter gradients as X^T G products with the
// rows as the K dimension of v_mfma_f32_16x16x4_f32 (xtg_wgrad_k's scheme):
//   gW1[m][i] = sum_r ghid[r][m] * [emb | offset_feat | 1][r][i]   (column 15 = the bias gradient gb1),
//   gW2[o][i] = sum_r gx0[r][o] * hid[r][i],   gb2[o] = sum_r gx0[r][o]  (per-lane sums, fixed shuffle tree)
// Each wave passes its 64 rows through a wave-private LDS tile [row][ghid 16 | x 16 | gx0 8 | hid 16] to turn "lane = row" into
// the fragment layout.  Neither ghid nor the MLP input is written to memory (round 2/3: a 64 B/row matrix each, read back by two
// pointwise weight-gradient launches).  One slab row per workgroup and scale, the four waves folded in order.
#define SB_LD 57
#ifndef SB_LAB
#define SB_LAB 0            // lab builds (tools/lab_build.sh net <tag> -DSB_LAB=mask): 1 no X^T G loop, 2 no LDS tile writes, 4 no gh / h arithmetic, 8 one tile per wave only
#endif
#define SB_WAVES 4          // (8 waves per workgroup = one workgroup per CU: 63.6 instead of 50.3 us per step for the two scale-context kernels)
__global__ __launch_bounds__(SB_WAVES * 64) void sce_bwd_all_k(const float* __restrict__ P, const float* __restrict__ off, SceArgs a,
                                                            const float* __restrict__ gx0, const float* __restrict__ hid,
                                                            float* __restrict__ big, int64_t block_stride) {
    __shared__ float sT[SB_WAVES * 64 * SB_LD];
    __shared__ float sfold[64 * 9];
    __shared__ float sb2[SB_WAVES * 8];
    int s = 0;                                                 // scale of this workgroup: uniform
    for (int i = 1; i < a.n_scales; ++i) s += ((int)blockIdx.x >= a.wg_off[i]) ? 1 : 0;
    const int sb = (int)blockIdx.x - a.wg_off[s], nsb = a.wg_off[s + 1] - a.wg_off[s];          // slab row, rows of this scale
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mm = lane & 15, rr = lane >> 4;
    const float* emb = P + a.emb[s];
    const float* W2 = P + a.w2[s];
    const int64_t r0 = a.row_off[s], n = a.row_off[s + 1] - r0;
    int64_t per = (n + nsb - 1) / nsb;
    per = (per + 15) & ~(int64_t)15;
    const int64_t b0 = (int64_t)sb * per;
    const int64_t b1 = (b0 + per < n) ? b0 + per : n;
    float* T = sT + wave * 64 * SB_LD;
    // gh = g W2 on v_mfma_f32_4x4x1 with the weight 4-vector broadcast (CBSZ = 4), K = 1 - each instruction is one fmaf per output, output
    // gradients ascending from 0 like the loop it replaces (same bits): the 128 weights are TWO registers per lane (combo 4 i + oq ->
    // W2[i][4 oq + j], block (lane >> 2) of register v is combo 16 v + block) where rounds 2-5 pinned them into 128 vector registers
    // per lane (one workgroup per CU; now two, LDS-bound).  Worth ~1 us of the kernel's 21-22 (profiles/r06_sce_lab.txt).
    float wG[2];
    {
        const int blk = lane >> 2, j4 = lane & 3;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int cb = 16 * v + blk;
            wG[v] = W2[(cb >> 2) * 16 + 4 * (cb & 3) + j4];
        }
    }
    // sce_fwd_k's weight image of the first layer (csrc/sce.h: combo 4 i + hq -> W1[4 hq + j][i], 60 + hq -> b1[4 hq + j])
    float wA[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (hid == nullptr) {
        const float* W1 = P + a.w1[s];
        const float* b1p = P + a.b1[s];
        const int blk = lane >> 2, j4 = lane & 3;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int cb = 16 * v + blk;
            wA[v] = cb < 60 ? W1[(4 * (cb & 3) + j4) * 15 + (cb >> 2)] : b1p[4 * (cb - 60) + j4];
        }
    }
    f32x4 acc1 = {0.0f, 0.0f, 0.0f, 0.0f}, acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
    float bs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[j] = 0.0f;
    // The inputs of a wave's NEXT tile are loaded while the current one is worked on (pinned by a scheduling barrier; rows behind the
    // range read its last row, masked by `live`).  The kernel's time is mostly fixed cost per workgroup - one tile per wave instead of
    // 3.75 still takes 17 of the 22 us (607 workgroups of 63 KB LDS on 512 slots: two rounds of prologue, tile, fold) - see
    // profiles/r06_sce_lab.txt.
    float4 ng0, ng1;
    float noff[7];
    auto fetch = [&](int64_t c0) {
        const int64_t row = c0 + lane;
        const int64_t r = r0 + (row < b1 ? row : b1 - 1);
        ng0 = *reinterpret_cast<const float4*>(gx0 + r * 8); ng1 = *reinterpret_cast<const float4*>(gx0 + r * 8 + 4);
#pragma unroll
        for (int i = 0; i < 7; ++i) noff[i] = off[r * 7 + i];
    };
    if (b0 + 64 * wave < b1) fetch(b0 + 64 * wave);
    for (int64_t c0 = b0 + 64 * wave; c0 < b1; c0 += 64 * SB_WAVES) {
        const int64_t row = c0 + lane;
        const bool live = row < b1;
        const int64_t r = r0 + (live ? row : b1 - 1);
        const float g[8] = {ng0.x, ng0.y, ng0.z, ng0.w, ng1.x, ng1.y, ng1.z, ng1.w};
        float x[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = emb[i];
#pragma unroll
        for (int i = 0; i < 7; ++i) x[8 + i] = noff[i];
        x[15] = 1.0f;                                          // the bias gradient's pseudo input
        fetch(c0 + 64 * SB_WAVES);
        __builtin_amdgcn_sched_barrier(0);
        float h[16];
        if constexpr ((SB_LAB & 4) != 0) {
#pragma unroll
            for (int o = 0; o < 16; ++o) h[o] = x[o];
        } else
        if (hid != nullptr) {                                  // (uniform) the op-level entry hands the forward's hidden layer in
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 t = *reinterpret_cast<const float4*>(hid + r * 16 + 4 * v);
                h[4 * v] = t.x; h[4 * v + 1] = t.y; h[4 * v + 2] = t.z; h[4 * v + 3] = t.w;
            }
        } else {
            // the executors do not keep the hidden layer (64 bytes per row written by the forward and read back here): it is recomputed
            // with sce_fwd_k's own instruction sequence - same bits - from inputs this kernel loads anyway
            f32x4 hq4[4];
            static_for<4>([&](auto hc) {
                constexpr int hq = decltype(hc)::value;
                hq4[hq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[3], 1.0f, (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, 4, 12 + hq, 0);
            });
            static_for<15>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                static_for<4>([&](auto hc) {
                    constexpr int hq = decltype(hc)::value;
                    constexpr int cb = 4 * i + hq;
                    hq4[hq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[cb / 16], x[i], hq4[hq], 4, cb % 16, 0);
                });
            });
#pragma unroll
            for (int o = 0; o < 16; ++o) h[o] = fmaxf(hq4[o >> 2][o & 3], 0.0f);
        }
        float gh[16];
        if constexpr ((SB_LAB & 4) != 0) {
#pragma unroll
            for (int o = 0; o < 16; ++o) gh[o] = g[o & 7];
        } else {
            f32x4 ghq[4];
            static_for<4>([&](auto oc) {
                constexpr int oq = decltype(oc)::value;
                ghq[oq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wG[0], g[0], (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, 4, oq, 0);
            });
            static_for<7>([&](auto ic) {
                constexpr int i = decltype(ic)::value + 1;
                static_for<4>([&](auto oc) {
                    constexpr int oq = decltype(oc)::value;
                    constexpr int cb = 4 * i + oq;
                    ghq[oq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wG[cb / 16], g[i], ghq[oq], 4, cb % 16, 0);
                });
            });
#pragma unroll
            for (int o = 0; o < 16; ++o) gh[o] = ghq[o >> 2][o & 3];
        }
        float* Tr = T + lane * SB_LD;
        if constexpr ((SB_LAB & 2) == 0) {
#pragma unroll
        for (int o = 0; o < 16; ++o) Tr[o] = (live && h[o] > 0.0f) ? gh[o] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) Tr[16 + i] = live ? x[i] : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) Tr[32 + j] = live ? g[j] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) Tr[40 + i] = live ? h[i] : 0.0f;
        } else {
            float t = 0.0f;
#pragma unroll
            for (int o = 0; o < 16; ++o) t += gh[o] + h[o] + x[o];
            bs[0] += t;
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bs[j] += g[j];
        }
        // wave-private tile: LDS operations of a wave execute in order, no barrier.  (Round 6 lab, profiles/r06_sce_lab.txt: the tile
        // transposed so that an operand is four 16-byte reads instead of sixteen 4-byte ones - same time, other summation order: not kept.)
        if constexpr ((SB_LAB & 1) == 0)
#pragma unroll 4
        for (int s4 = 0; s4 < 16; ++s4) {
            const float* Tq = T + (4 * s4 + rr) * SB_LD;
            const float a1 = Tq[mm], b1v = Tq[16 + mm];
            const float a2 = (mm < 8) ? Tq[32 + mm] : 0.0f, b2v = Tq[40 + mm];
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1v, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2v, acc2, 0, 0, 0);
        }
        if constexpr ((SB_LAB & 8) != 0) break;
    }
    // fold the 4 waves in wave order, then one partial per destination element (C/D map: row = (lane >> 4) * 4 + reg, col = lane & 15)
    float* mine = sfold + lane * 9;
    for (int w = 0; w < SB_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mine[j] = (w == 0) ? acc1[j] : mine[j] + acc1[j];
                mine[4 + j] = (w == 0) ? acc2[j] : mine[4 + j] + acc2[j];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) bs[j] += __shfl_xor(bs[j], dd, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sb2[wave * 8 + j] = bs[j];
    }
    __syncthreads();
    if (wave == 0) {
        float* dst = big + (int64_t)sb * block_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = rr * 4 + j;
            if (mm < 15) dst[a.w1[s] + m * 15 + mm] = mine[j];
            else dst[a.b1[s] + m] = mine[j];
            if (m < 8) dst[a.w2[s] + m * 16 + mm] = mine[4 + j];
        }
        if (lane < 8) {
            float t = sb2[lane];
#pragma unroll
            for (int w = 1; w < SB_WAVES; ++w) t += sb2[8 * w + lane];
            dst[a.b2[s] + lane] = t;
        }
    }
}

__global__ __launch_bounds__(LINR_BLOCK) void sigmoid_k(const float* __restrict__ z, int64_t n, float* __restrict__ p) {
    const int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i < n) p[i] = 1.0f / (1.0f + expf(-z[i]));
}

// dst (+)= src over n floats
__global__ __launch_bounds__(LINR_BLOCK) void axpy_k(const float* __restrict__ src, int64_t n, float* __restrict__ dst,
                                                     int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i < n) dst[i] = accumulate ? dst[i] + src[i] : src[i];
}

extern "C" int linr_axpy(const float* src, int64_t n, float* dst, int32_t accumulate, void* stream) {
    if (n < 0) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!src || !dst) return LINR_EINVAL;
    axpy_k<<<linr_grid(n, LINR_BLOCK), LINR_BLOCK, 0, (hipStream_t)stream>>>(src, n, dst, accumulate ? 1 : 0);
    return linr_launch_rc();
}

// gemb[i] = sum_m gb1[m] * W1[m][i]   (scale-embedding gradient through Linear(15,16); the embedding row is a
// constant input of every row of its scale, so its gradient is W1[:, :8]^T applied to the bias gradient)
struct EmbArgs { int64_t gb1[MAX_SCALES], w1[MAX_SCALES], gemb[MAX_SCALES]; };
__global__ void sce_emb_grad_all_k(const float* __restrict__ P, float* __restrict__ gsum, EmbArgs a) {
    const int t = threadIdx.x, g = blockIdx.x;
    if (t < 8) {
        const float* gb1 = gsum + a.gb1[g];
        const float* W1 = P + a.w1[g];
        float s = 0.0f;
        for (int m = 0; m < 16; ++m) s = fmaf(gb1[m], W1[m * 15 + t], s);
        gsum[a.gemb[g] + t] = s;
    }
}

// gsum[p] = sum_b big[b][p] in a fixed association (RED_SPLIT threads per parameter, each 8 interleaved partial sums over
// its quarter of the slab rows in ascending order, quarters added in order) => bit-reproducible.  One thread per
// parameter alone would be 214 blocks of latency-bound streaming on 256 CUs.
#define RED_SPLIT 4       // threads per parameter: each sums nblocks / RED_SPLIT slab rows
// Parameters nobody wrote partials for - the scale embedding (its gradient is derived from the reduced sums afterwards) and
// the context MLPs of scales the frame does not contain - lie in [0, prefix): `zr` lists those ranges and the reduction
// writes 0 for them WITHOUT reading the slab, which therefore needs no clearing pass (a 2-D memset of nb rows per step).
struct ZeroRanges { int n; int64_t prefix; int64_t b[MAX_SCALES + 1], e[MAX_SCALES + 1]; };
// Parameter ranges whose producer (a fused backward launch: one round of long-lived blocks, csrc/fused_bwd.hip) wrote only the
// first rows[i] rows of the slab: the reduction stops there instead of having the producer fill the other rows with zeros.
#define MAX_SHORT 48
struct ShortRanges { int n; int64_t b[MAX_SHORT], e[MAX_SHORT]; int rows[MAX_SHORT]; };
__global__ __launch_bounds__(LINR_BLOCK) void wgrad_reduce_k(const float* __restrict__ big, int nblocks, int64_t total,
                                                             float* __restrict__ gsum, ZeroRanges zr, ShortRanges sr) {
    __shared__ float part[RED_SPLIT][LINR_BLOCK / RED_SPLIT];
    const int lp = threadIdx.x % (LINR_BLOCK / RED_SPLIT), q = threadIdx.x / (LINR_BLOCK / RED_SPLIT);
    const int64_t p = (int64_t)blockIdx.x * (LINR_BLOCK / RED_SPLIT) + lp;
    float s = 0.0f;
    bool skip = false;
    if (p < zr.prefix)
        for (int i = 0; i < zr.n; ++i) skip = skip || (p >= zr.b[i] && p < zr.e[i]);
    if (p < total && !skip) {
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = 0.0f;
        const int per = nblocks / RED_SPLIT;                 // nblocks is a multiple of 8 * RED_SPLIT
        int rows = nblocks;
        for (int i = 0; i < sr.n; ++i)
            if (p >= sr.b[i] && p < sr.e[i]) rows = sr.rows[i];
        const int hi = (q + 1) * per < rows ? (q + 1) * per : rows;
        if (per == 64) {
            // the usual slab (256 rows): the thread's 64 loads are all in flight before the first add (the rolled loop waits
            // for memory eight times); same adds in the same order
            const float* src = big + (int64_t)(q * 64) * total + p;
            float v[64];
#pragma unroll
            for (int j = 0; j < 64; ++j) v[j] = (q * 64 + j < hi) ? src[(int64_t)j * total] : 0.0f;
#pragma unroll
            for (int j = 0; j < 64; ++j)
                if (q * 64 + j < hi) a[j & 7] += v[j];
        } else {
            for (int b = q * per; b < hi; b += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (b + i < hi) a[i] += big[(int64_t)(b + i) * total + p];
            }
        }
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    part[q][lp] = s;
    __syncthreads();
    if (q == 0 && p < total) gsum[p] = ((part[0][lp] + part[1][lp]) + part[2][lp]) + part[3][lp];
}

struct Ctx {
    const linr_frame* f;
    const float* P;
    Arena A;
    Layout L;
    hipStream_t s;           // the caller's stream
    int64_t R;
    int64_t nbr_ld;
    int nb;                  // persistent blocks of the weight-gradient kernels = partial rows of the slab for this frame
    struct Short { int64_t b, e; int rows; };
    std::vector<Short> shortr;      // parameter ranges of this backward pass that hold fewer than nb slab rows (fused launches)
    // a fused launch writes only `rows` slab rows for parameters [b, e) (no zero fill): the final reduction stops there.  The
    // table holds every range of a backward pass: 8 blocks x 2 + the prune convs + block_in's first conv + the 7 outter first
    // convs + up to 16 scale-context MLPs = 41
    void note_short(int64_t b, int64_t e, int rows) {
        if (rows >= nb) return;
        if ((int)shortr.size() >= MAX_SHORT) abort();          // cannot happen (see above); a silent drop would read unwritten rows
        shortr.push_back({b, e, rows});
    }
};

// Persistent blocks per weight-gradient launch (multiples of 32: wgrad_reduce_k's association).  Every block ends with a fold
// over its waves and writes one slab row per parameter, and the final reduction reads nb x n_params floats: fixed costs that
// grow with nb, while the grouped launches (8 groups since block_in joined them) bring nb x 8 blocks anyway.  Measured,
// ms/step with the joined schedule: 337 k rows (loot10): 512 -> 2.291, 384 -> 2.277, 256 -> 2.271, 192 -> 2.305, 160 -> 2.293,
// 128 -> 2.285; 373 k rows (andrew10): 512 -> 2.626, 256 -> 2.614, 160 -> 2.656; 54 k rows (sphere8): 384 -> 0.566, 256 -> 0.533,
// 192 -> 0.532, 128 -> 0.506, 96 -> 0.509, 64 -> 0.531 (profiles/r02_ab_wg_blocks.txt).  The block count decides how the partial
// sums associate, i.e. the rounding of the gradients; nothing else depends on it (tests: test_block_count_changes_only_the_rounding).
static int wg_blocks_for(int64_t rows) {
    static const int forced = getenv("LINR_WG_BLOCKS") ? atoi(getenv("LINR_WG_BLOCKS")) : 0;
    if (forced >= 32 && forced <= LINR_WG_BLOCKS && forced % 32 == 0) return forced;
    return rows >= 100000 ? 256 : 128;
}
int linr_wg_blocks_for(int64_t rows) { return wg_blocks_for(rows); }

// ---- live kernel timing for bench.py's roofline (include/linr_hip.h: linr_prof_*) --------------------------------------
// While enabled, the launches of a training step are bracketed by an event pair on their stream, by kernel class (the list is
// in include/linr_hip.h); `passes` counts the row passes (groups) of a launch.  Measurement aid only: mutex-guarded, nothing is
// recorded (and no lock is taken) when disabled.
#define LINR_PROF_MAX 4096
enum { PK_FUSED88 = 0, PK_CONV88 = 1, PK_FUSED_DUAL = 2, PK_FUSED_C00 = 3, PK_HEAD_FWD = 4, PK_CONVPW_FWD = 5, PK_DUAL_FWD = 6,
       PK_OCC7 = 7, PK_HEAD_BWD = 8, PK_WGRAD = 9, PK_LIN_WGRAD = 10, PK_SCE = 11, PK_MISC = 12, PK_BWD_DATA = 13 };
struct ProfRec { hipEvent_t e0, e1; int passes; };
static std::atomic<bool> g_prof_on{false};
static std::atomic<uint32_t> g_prof_mask{3u};
static std::mutex g_prof_mu;                    // guards the vectors below
static std::vector<ProfRec> g_prof[LINR_PROF_KINDS];          // used records
static std::vector<ProfRec> g_prof_free;        // pre-created event pairs (creating events in the hot path costs ~20 us each)

// ---- test hook: on-chip state poisoning (include/linr_hip.h: linr_debug_poison) ------------------------------------------
// A kernel must never read LDS (or rely on register contents) it did not write itself: what is left there belongs to whatever
// ran on the CU before - on a GPU shared with another process that can be a NaN pattern, and 0 x NaN poisons a gradient that
// 0 x (own finite leftovers) never would.  While enabled, every launch of the executor is preceded by a kernel that fills the
// LDS of every CU (and most of the vector registers) with 0xFFFFFFFF; tests then demand bitwise unchanged results.
static std::atomic<uint32_t> g_poison{0u};          // bit k: poison in front of the launches of kernel class k (linr_prof_* classes)
__global__ __launch_bounds__(1024) void poison_onchip_k(uint32_t pattern, uint32_t* sink) {
    extern __shared__ uint32_t pl[];
    for (int i = threadIdx.x; i < 40960; i += 1024) pl[i] = pattern;
    // v8 .. v127 of every wave: 16 waves x 128 registers = the four SIMDs' 512-row register files
    asm volatile(
                 "v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n"
                 "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n"
                 "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n"
                 "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n"
                 "v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n"
                 "v_mov_b32 v48, %0\n v_mov_b32 v49, %0\n v_mov_b32 v50, %0\n v_mov_b32 v51, %0\n v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n"
                 "v_mov_b32 v56, %0\n v_mov_b32 v57, %0\n v_mov_b32 v58, %0\n v_mov_b32 v59, %0\n v_mov_b32 v60, %0\n v_mov_b32 v61, %0\n v_mov_b32 v62, %0\n v_mov_b32 v63, %0\n"
                 "v_mov_b32 v64, %0\n v_mov_b32 v65, %0\n v_mov_b32 v66, %0\n v_mov_b32 v67, %0\n v_mov_b32 v68, %0\n v_mov_b32 v69, %0\n v_mov_b32 v70, %0\n v_mov_b32 v71, %0\n"
                 "v_mov_b32 v72, %0\n v_mov_b32 v73, %0\n v_mov_b32 v74, %0\n v_mov_b32 v75, %0\n v_mov_b32 v76, %0\n v_mov_b32 v77, %0\n v_mov_b32 v78, %0\n v_mov_b32 v79, %0\n"
                 "v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n v_mov_b32 v82, %0\n v_mov_b32 v83, %0\n v_mov_b32 v84, %0\n v_mov_b32 v85, %0\n v_mov_b32 v86, %0\n v_mov_b32 v87, %0\n"
                 "v_mov_b32 v88, %0\n v_mov_b32 v89, %0\n v_mov_b32 v90, %0\n v_mov_b32 v91, %0\n v_mov_b32 v92, %0\n v_mov_b32 v93, %0\n v_mov_b32 v94, %0\n v_mov_b32 v95, %0\n"
                 "v_mov_b32 v96, %0\n v_mov_b32 v97, %0\n v_mov_b32 v98, %0\n v_mov_b32 v99, %0\n v_mov_b32 v100, %0\n v_mov_b32 v101, %0\n v_mov_b32 v102, %0\n v_mov_b32 v103, %0\n"
                 "v_mov_b32 v104, %0\n v_mov_b32 v105, %0\n v_mov_b32 v106, %0\n v_mov_b32 v107, %0\n v_mov_b32 v108, %0\n v_mov_b32 v109, %0\n v_mov_b32 v110, %0\n v_mov_b32 v111, %0\n"
                 "v_mov_b32 v112, %0\n v_mov_b32 v113, %0\n v_mov_b32 v114, %0\n v_mov_b32 v115, %0\n v_mov_b32 v116, %0\n v_mov_b32 v117, %0\n v_mov_b32 v118, %0\n v_mov_b32 v119, %0\n"
                 "v_mov_b32 v120, %0\n v_mov_b32 v121, %0\n v_mov_b32 v122, %0\n v_mov_b32 v123, %0\n v_mov_b32 v124, %0\n v_mov_b32 v125, %0\n v_mov_b32 v126, %0\n v_mov_b32 v127, %0\n"
                 :: "v"(pattern) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    __syncthreads();
    const uint32_t t = pl[(threadIdx.x * 37u) % 40960u];
    if (t == 0x12345u && sink) *sink = t;             // never true for the pattern used: keeps the LDS stores alive
}
static void poison_onchip(hipStream_t s) {          // failures are ignored: a test hook
    static const bool ok = hipFuncSetAttribute((const void*)poison_onchip_k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840) == hipSuccess;
    if (!ok) return;
    poison_onchip_k<<<1024, 1024, 163840, s>>>(0xFFFFFFFFu, nullptr);
}
extern "C" int linr_debug_poison(uint32_t kind_mask) { g_poison = kind_mask; return 0; }
extern "C" int linr_debug_poison_now(void* stream) { poison_onchip((hipStream_t)stream); return linr_launch_rc(); }
void linr_poison_hook(hipStream_t s, int kind) {          // common.h: for the executors outside this file
    if ((g_poison.load(std::memory_order_relaxed) >> kind) & 1u) poison_onchip(s);
}

struct ProfScope {
    hipStream_t s; int kind; bool live;
    ProfRec r;
    ProfScope(hipStream_t s_, int kind_, int passes, bool want = true) : s(s_), kind(kind_), live(false) {
        if ((g_poison.load(std::memory_order_relaxed) >> kind_) & 1u) poison_onchip(s_);
        if (!want || !g_prof_on.load(std::memory_order_relaxed) || !((g_prof_mask.load(std::memory_order_relaxed) >> kind_) & 1u)) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof_free.empty()) return;
        r = g_prof_free.back();
        g_prof_free.pop_back();
        r.passes = passes;
        live = hipEventRecord(r.e0, s) == hipSuccess;
        if (!live) g_prof_free.push_back(r);
    }
    ~ProfScope() {
        if (!live) return;
        (void)hipEventRecord(r.e1, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof[kind].push_back(r);
    }
};

LinrProf::LinrProf(hipStream_t s, int kind, int passes) : impl(new ProfScope(s, kind, passes)) {}
LinrProf::~LinrProf() { delete static_cast<ProfScope*>(impl); }

extern "C" int linr_prof_mask(uint32_t mask) { g_prof_mask = mask; return 0; }

extern "C" int linr_prof_enable(int32_t mode) {          // 0: stop (records kept), 1: clear + start, 2: resume
    g_prof_on = false;
    if (mode == 0) return 0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    size_t used = 0;
    for (int k = 0; k < LINR_PROF_KINDS; ++k) {
        if (mode == 1) {
            for (auto& r : g_prof[k]) g_prof_free.push_back(r);
            g_prof[k].clear();
        }
        used += g_prof[k].size();
    }
    while (g_prof_free.size() + used < LINR_PROF_MAX) {
        ProfRec r;
        r.passes = 0;
        if (hipEventCreate(&r.e0) != hipSuccess) break;
        if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); break; }
        g_prof_free.push_back(r);
    }
    g_prof_on = true;
    return 0;
}

extern "C" int linr_prof_read(int32_t kind, double* total_ms, int64_t* launches, int64_t* passes) {
    if (kind < 0 || kind >= LINR_PROF_KINDS || !total_ms || !launches || !passes) return LINR_EINVAL;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double t = 0.0;
    int64_t np = 0;
    for (auto& r : g_prof[kind]) {
        TRY(linr_hip_rc(hipEventSynchronize(r.e1)));
        float ms = 0.0f;
        TRY(linr_hip_rc(hipEventElapsedTime(&ms, r.e0, r.e1)));
        t += ms;
        np += r.passes;
    }
    *total_ms = t; *launches = (int64_t)g_prof[kind].size(); *passes = np;
    return 0;
}

// index source of the conv kernels: the compressed map (the full neighbour table is 5 % slower also after the shift
// addressing of round 2: 2.565 vs 2.438 ms/step, profiles/r02_ab_conv_table.txt)
static const int32_t* clo(const Ctx& c) { return c.f->nbr_lo; }
static const uint32_t* cmk(const Ctx& c) { return c.f->nbr_mask; }

// grouped launches need the matrix-core conv kernel (the VALU fallback of LINR_CONV_MFMA=0 is a single-layer kernel)
static bool grouped_enabled() {
    static const int batched = getenv("LINR_BATCHED") ? atoi(getenv("LINR_BATCHED")) : 1;
    static const int conv_mfma = getenv("LINR_CONV_MFMA") ? atoi(getenv("LINR_CONV_MFMA")) : 1;
    return batched != 0 && conv_mfma != 0;
}

// block_in joins the grouped launches of the outter blocks when it has their shape (one Inception layer) - LINR_JOIN_BLOCK_IN=0
// keeps its layers as single launches (the round-1/2 schedule; same kernels, same bits)
static bool join_block_in(const Ctx& c) {
    static const int v = getenv("LINR_JOIN_BLOCK_IN") ? atoi(getenv("LINR_JOIN_BLOCK_IN")) : 1;
    return v != 0 && grouped_enabled() && c.f->nbr_lo && c.f->nbr_mask && c.L.block_in.nl == 1;
}

// backward-data and weight gradient of the 8->8 convolutions from ONE gather (csrc/fused_bwd.hip); LINR_FUSED_BWD=0 restores the
// two-kernel schedule (same input gradients bit for bit, weight gradients in another summation order)
static bool fused_bwd(const Ctx& c) {
    static const int v = getenv("LINR_FUSED_BWD") ? atoi(getenv("LINR_FUSED_BWD")) : 1;
    // (the fused kernels address the compressed map with 32-bit byte offsets: 9 x ld x 4 B < 2^32; larger maps take the two-kernel path)
    return v != 0 && c.f->nbr_lo && c.f->nbr_mask && c.nbr_ld < ((int64_t)1 << 26);
}

static int conv3(Ctx& c, bool bwd, const float* in, int in_ld, const float* W, const float* bias, int cin, int cout,
                 const float* res, int res_ld, const float* act, int act_ld, float* out, int out_ld, unsigned flags) {
    if (c.f->nbr_lo && c.f->nbr_mask) {
        ProfScope ps(c.s, bwd ? PK_BWD_DATA : PK_CONV88, 1);
        return linr_cconv_launch(bwd, in, in_ld, clo(c), cmk(c), c.nbr_ld, c.R, W, bias, cin, cout, res, res_ld,
                                 act, act_ld, out, out_ld, flags, c.s);
    }
    return linr_conv3_launch(bwd, in, in_ld, c.f->nbr, c.nbr_ld, c.R, W, bias, cin, cout, res, res_ld, act, act_ld, out, out_ld,
                             flags | LINR_PAD_ROW, c.s);
}

// transposed tiled table of the stand-alone weight-gradient kernels (csrc/fused.hip: spconv_wgrad_t_k), or NULL: indices from nbr
static const int32_t* wg_t8t(const Ctx& c) { return c.f->nbr8t; }
static int conv3_wgrad(Ctx& c, const float* in, int in_ld, const float* gout, int gout_ld, int cin, int cout,
                       int64_t w_off, int64_t b_off) {
    LinrWgradDst d = {c.A.BIG, c.L.total, w_off, b_off, cin};
    ProfScope ps(c.s, PK_WGRAD, 1);
    return linr_conv3_wgrad_mfma(in, in_ld, gout, gout_ld, c.f->nbr, c.nbr_ld, c.R, cin, cout, d, c.nb, c.s, nullptr, 1, wg_t8t(c));
}

static int linear(Ctx& c, const float* in, int in_ld, int64_t n, const float* W, int ws_ci, int ws_co, const float* bias,
                  int cin, int cout, const float* res, int res_ld, const float* act, int act_ld, float* out, int out_ld,
                  unsigned flags) {
    return linr_linear_launch(in, in_ld, n, W, ws_ci, ws_co, bias, cin, cout, res, res_ld, act, act_ld, out, out_ld, flags,
                              c.s);
}

static int linear_wgrad(Ctx& c, const float* in, int in_ld, const float* gout, int gout_ld, int64_t n, int cin, int cout,
                        int64_t w_off, int ws_ci, int ws_co, int64_t b_off) {
    LinrLinDst d = {c.A.BIG, c.L.total, w_off, ws_ci, ws_co, b_off};
    ProfScope ps(c.s, PK_LIN_WGRAD, 1);
    return linr_linear_wgrad_partial(in, in_ld, gout, gout_ld, n, cin, cout, d, c.nb, c.s);
}

// Per-layer matrices of block slot b (0 = block_in, 1..7 = outter blocks): layer 0 uses the slot's own H/M/I, the extra
// Inception layers of block_in (block_layers > 1, models/resnet.py:156-162) the Hx/Mx/Ix sets.
struct LayerBufs { float *H, *M, *I, *gI, *gM, *gH; };
static LayerBufs layer_bufs(Arena& a, int b, int l) {
    if (l == 0) return {a.H[b], a.M[b], a.I[b], a.gI[b], a.gM[b], a.gH[b]};
    return {a.Hx[l - 1], a.Mx[l - 1], a.Ix[l - 1], a.gIx[l - 1], a.gMx[l - 1], a.gHx[l - 1]};
}

// make_block (models/upsample.py:88-97): conv3(cin->8)+ReLU -> ResNetBlock(nl x Inception, extra skip if nl > 1) ->
// conv3(8->8) (+ res)
static int block_fwd(Ctx& c, const BlockP& bp, const float* in, int in_ld, int b, const float* res) {
    Arena& a = c.A;
    const float* P = c.P;
    TRY(conv3(c, false, in, in_ld, P + bp.a_w, P + bp.a_b, bp.cin, 8, nullptr, 0, nullptr, 0, a.A[b], 8, LINR_RELU));
    const float* X = a.A[b];                          // input of the current Inception layer
    for (int l = 0; l < bp.nl; ++l) {
        const IncP& q = bp.inc[l];
        const LayerBufs t = layer_bufs(a, b, l);
        if (c.f->nbr_lo && c.f->nbr_mask) {
            // Inception layer in two launches (csrc/fused.hip): [conv0_0 | conv1_0 centre tap] -> H, then the two 4->4 convs
            // as one pass with conv1_2 and the residual in the epilogue -> M, I
            TRY(linr_conv_pw_fwd_launch(X, clo(c), cmk(c), c.nbr_ld, c.R, P + q.c00_w, P + q.c00_b, P + q.c10_w,
                                        P + q.c10_b, t.H, c.s));
            TRY(linr_dual44_fwd_launch(t.H, clo(c), cmk(c), c.nbr_ld, c.R, P + q.c01_w, P + q.c01_b, P + q.c11_w,
                                       P + q.c11_b, X, P + q.c12_w, P + q.c12_b, t.M, t.I, c.s));
        } else {
            // path 0: H[:,0:4] = relu(conv3 8->4 (X));  path 1: H[:,4:8] = relu(X @ conv1_0)
            TRY(conv3(c, false, X, 8, P + q.c00_w, P + q.c00_b, 8, 4, nullptr, 0, nullptr, 0, t.H, 8, LINR_RELU));
            TRY(linear(c, X, 8, c.R, P + q.c10_w, 4, 1, P + q.c10_b, 8, 4, nullptr, 0, nullptr, 0, t.H + 4, 8, LINR_RELU));
            // I[:,0:4] = conv3 4->4 (H0) + X[:,0:4]
            TRY(conv3(c, false, t.H, 8, P + q.c01_w, P + q.c01_b, 4, 4, X, 8, nullptr, 0, t.I, 8, 0));
            // M = relu(conv3 4->4 (H1)); I[:,4:8] = M @ conv1_2 + X[:,4:8]
            TRY(conv3(c, false, t.H + 4, 8, P + q.c11_w, P + q.c11_b, 4, 4, nullptr, 0, nullptr, 0, t.M, 4, LINR_RELU));
            TRY(linear(c, t.M, 4, c.R, P + q.c12_w, 4, 1, P + q.c12_b, 4, 4, X + 4, 8, nullptr, 0, t.I + 4, 8, 0));
        }
        X = t.I;
    }
    float* Il = layer_bufs(a, b, bp.nl - 1).I;
    if (bp.nl > 1)           // ResNetBlock.forward: out += x when it chains more than one layer (resnet.py:160-161)
        axpy_k<<<linr_grid(c.R * 8, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(a.A[b], c.R * 8, Il, 1);
    TRY(conv3(c, false, Il, 8, P + bp.b_w, P + bp.b_b, 8, 8, res, 8, nullptr, 0, a.O[b], 8, 0));
    return 0;
}

// gO: gradient w.r.t. the block output [R,8].  gin != nullptr: also produce the input gradient (block_in only).
static int block_bwd(Ctx& c, const BlockP& bp, const float* in, int in_ld, int b, const float* gO, float* gin) {
    Arena& a = c.A;
    const float* P = c.P;
    const bool cm = c.f->nbr_lo && c.f->nbr_mask;
    const int nl = bp.nl;
    const LayerBufs last = layer_bufs(a, b, nl - 1);
    // O = conv3(I_last; b)
    TRY(conv3_wgrad(c, last.I, 8, gO, 8, 8, 8, bp.b_w, bp.b_b));
    for (int l = nl - 1; l >= 0; --l) {
        const IncP& q = bp.inc[l];
        const LayerBufs t = layer_bufs(a, b, l);
        const float* X = l == 0 ? a.A[b] : layer_bufs(a, b, l - 1).I;         // the layer's input
        float* gX = l == 0 ? a.gA[b] : layer_bufs(a, b, l - 1).gI;           // where its input gradient goes
        // gI of this layer: from the block's tail conv (last layer) or written by layer l+1 as its input gradient
        if (l == nl - 1) {
            if (cm) {   // gI = bwd(gO; Wb) with gM = (gI[:,4:8] @ W12^T) * (M > 0) in the epilogue (csrc/fused.hip)
                TRY(linr_conv_bwd_gm_launch(gO, clo(c), cmk(c), c.nbr_ld, c.R, P + bp.b_w, P + q.c12_w, t.M, t.gI, t.gM, c.s));
            } else {
                TRY(conv3(c, true, gO, 8, P + bp.b_w, nullptr, 8, 8, nullptr, 0, nullptr, 0, t.gI, 8, 0));
            }
        }
        if (!(cm && l == nl - 1))   // I[:,4:8] = M @ c12 + b12 + X[:,4:8]  =>  gM = (gI[:,4:8] @ W12^T) * (M > 0)
            TRY(linear(c, t.gI + 4, 8, c.R, P + q.c12_w, 1, 4, nullptr, 4, 4, nullptr, 0, t.M, 4, t.gM, 4, LINR_RELU_MASK));
        TRY(linear_wgrad(c, t.M, 4, t.gI + 4, 8, c.R, 4, 4, q.c12_w, 4, 1, q.c12_b));
        if (cm) {
            TRY(linr_conv3_wgrad_dual44(t.H, t.gI, 8, t.gM, 4, c.f->nbr, c.nbr_ld, c.R, a.BIG, c.L.total, q.c01_w, q.c01_b,
                                        q.c11_w, q.c11_b, c.nb, c.s, nullptr, 1, wg_t8t(c)));
            TRY(linr_dual44_bwd_launch(t.gI, t.gM, clo(c), cmk(c), c.nbr_ld, c.R, P + q.c01_w, P + q.c11_w, t.H, t.gH, c.s));
        } else {
            // I[:,0:4] = conv3(H0; c01) + X[:,0:4]
            TRY(conv3_wgrad(c, t.H, 8, t.gI, 8, 4, 4, q.c01_w, q.c01_b));
            TRY(conv3(c, true, t.gI, 8, P + q.c01_w, nullptr, 4, 4, nullptr, 0, t.H, 8, t.gH, 8, LINR_RELU_MASK));
            // M = relu(conv3(H1; c11))
            TRY(conv3_wgrad(c, t.H + 4, 8, t.gM, 4, 4, 4, q.c11_w, q.c11_b));
            TRY(conv3(c, true, t.gM, 4, P + q.c11_w, nullptr, 4, 4, nullptr, 0, t.H + 4, 8, t.gH + 4, 8, LINR_RELU_MASK));
        }
        // H0 = relu(conv3(X; c00)), H1 = relu(X @ c10)
        TRY(conv3_wgrad(c, X, 8, t.gH, 8, 8, 4, q.c00_w, q.c00_b));
        TRY(linear_wgrad(c, X, 8, t.gH + 4, 8, c.R, 8, 4, q.c10_w, 4, 1, q.c10_b));
        // input gradient: gX = bwd(gH[:,0:4]; W00) + gI (the layer's own residual) + gH[:,4:8] @ W10^T; layer 0's input is
        // A = relu(.) so it is masked by (A > 0), after the ResNetBlock's extra skip (nl > 1: + gI of the last layer)
        const bool skip = (l == 0 && nl > 1);
        if (skip)
            TRY(linr_hip_rc(hipMemcpyAsync(gX, last.gI, (size_t)c.R * 8 * sizeof(float), hipMemcpyDeviceToDevice, c.s)));
        if (cm) {
            TRY(linr_conv_bwd_ga_launch(t.gH, clo(c), cmk(c), c.nbr_ld, c.R, P + q.c00_w, P + q.c10_w, t.gI, l == 0 ? a.A[b] : nullptr,
                                        gX, (l == 0 ? LINR_RELU_MASK : 0u) | (skip ? LINR_ACCUM : 0u), c.s));
        } else {
            TRY(conv3(c, true, t.gH, 8, P + q.c00_w, nullptr, 8, 4, t.gI, 8, nullptr, 0, gX, 8, skip ? LINR_ACCUM : 0u));
            TRY(linear(c, t.gH + 4, 8, c.R, P + q.c10_w, 1, 4, nullptr, 4, 8, nullptr, 0, l == 0 ? a.A[b] : nullptr, 8, gX, 8,
                       LINR_ACCUM | (l == 0 ? LINR_RELU_MASK : 0u)));
        }
    }
    // A = relu(conv3(in; a))
    TRY(conv3_wgrad(c, in, in_ld, a.gA[b], 8, bp.cin, 8, bp.a_w, bp.a_b));
    if (gin) TRY(conv3(c, true, a.gA[b], 8, P + bp.a_w, nullptr, bp.cin, 8, nullptr, 0, nullptr, 0, gin, 8, 0));
    return 0;
}

static int check_frame(const linr_frame* f, const void* params, const void* arena, size_t arena_bytes, Ctx& c) {
    if (!f || !params || !arena) return LINR_EINVAL;
    if (f->rows < 0 || f->n_scales < 1 || f->n_scales > MAX_SCALES || !f->row_off_h || !f->scale_idx_h) return LINR_EINVAL;
    if (f->rows > 0 && (!f->nbr || !f->offset_feat || !f->occ)) return LINR_EINVAL;
    // the conv kernels address gathered rows with 32-bit byte offsets: (rows + 1) * 32 B must stay below 2^32
    if (f->rows >= ((int64_t)1 << 27) - 1 || f->nbr_ld < f->rows) return LINR_EINVAL;
    if (!make_layout(c.L, f->model_scale_num, f->block_layers < 1 ? 1 : f->block_layers)) return LINR_EINVAL;
    if (f->row_off_h[0] != 0 || f->row_off_h[f->n_scales] != f->rows) return LINR_EINVAL;
    for (int s = 0; s < f->n_scales; ++s) {
        if (f->row_off_h[s + 1] < f->row_off_h[s]) return LINR_EINVAL;
        if (f->scale_idx_h[s] < 0 || f->scale_idx_h[s] >= f->model_scale_num) return LINR_EINVAL;
    }
    if (arena_bytes < linr_net_arena_bytes(f->rows, c.L.BL)) return LINR_ENOSPC;
    if (!linr_aligned16(arena)) return LINR_EALIGN;
    c.f = f;
    c.P = (const float*)params;
    c.R = f->rows;
    c.nbr_ld = f->nbr_ld;
    c.nb = wg_blocks_for(f->rows);
    make_arena(c.A, f->rows, (float*)arena, c.L.total, c.L.BL);
    if (f->flags & LINR_FRAME_OCC_PADDED) {        // the caller's occupancy buffer has the zero row in front: use it in place
        if (!f->occ || !linr_aligned16(f->occ)) return LINR_EINVAL;
        c.A.OCC = const_cast<float*>(f->occ);
    }
    return 0;
}

static SceArgs sce_args(const Ctx& c) {
    SceArgs a;
    a.n_scales = c.f->n_scales;
    for (int s = 0; s < c.f->n_scales; ++s) {
        const int si = c.f->scale_idx_h[s];
        a.row_off[s] = c.f->row_off_h[s];
        a.emb[s] = c.L.emb + si * 8; a.w1[s] = c.L.m0_w[si]; a.b1[s] = c.L.m0_b[si]; a.w2[s] = c.L.m2_w[si]; a.b2[s] = c.L.m2_b[si];
    }
    a.row_off[c.f->n_scales] = c.f->rows;
    a.blk_off[0] = 0;
    for (int s = 0; s < c.f->n_scales; ++s)
        a.blk_off[s + 1] = a.blk_off[s] + (int)linr_grid(a.row_off[s + 1] - a.row_off[s], LINR_BLOCK);
    return a;
}

static void goffs(int64_t* dst, const float* const* ptrs, int n) {
    for (int i = 0; i < n; ++i) dst[i] = ptrs[i] - ptrs[0];
}

// Teacher-forced forward of all 8 stages with the 7 outter blocks and the 8 heads as grouped launches (their inputs -
// the ground-truth occupancy and x_glob - are all known up front).  Same kernels and per-row arithmetic as the staged
// path below, so the decoder reproduces these probabilities bit for bit.
// part 1: the occupancy-only layers of the outter blocks (first conv, conv0_0 | conv1_0, both 4->4 convs: they do not need
// x_glob); part 2: everything that does (tail conv + x_glob, the 8 heads).  3 = both.
static int forward_batched(Ctx& c, float* probs, double* bits_acc, int part = 3, bool join = false) {
    Arena& a = c.A;
    const float* P = c.P;
    const Layout& L = c.L;
    const int32_t* lo = clo(c);
    const uint32_t* mk = cmk(c);
    const int64_t nblk = linr_grid(c.R, LINR_BLOCK);
    // join: block_in (arena slot 0; its first conv has already run) rides as group 0 of the Inception-layer launches - the
    // eight blocks have the same structure behind their first conv (models/upsample.py:88-97) and do not depend on each
    // other until prior_k = x_glob + outter_k.  g0 = first slot in the grouped launches, ng = their group count; the
    // occupancy conv and the tail conv (which needs x_glob as residual) always cover slots 1..7 = arrays + o7.
    const int g0 = join ? 0 : 1, ng = 8 - g0, o7 = 1 - g0;
    const float *pA[8], *pH[8], *pM[8], *pI[8], *pO[8], *p_ab[8], *p_c00w[8], *p_c00b[8], *p_c10w[8], *p_c10b[8], *p_c01w[8],
        *p_c01b[8], *p_c11w[8], *p_c11b[8], *p_c12w[8], *p_c12b[8], *p_bw[8], *p_bb[8];
    for (int g = 0; g < ng; ++g) {
        const int b = g0 + g;
        const BlockP& bp = b == 0 ? L.block_in : L.outter[b - 1];
        pA[g] = a.A[b]; pH[g] = a.H[b]; pM[g] = a.M[b]; pI[g] = a.I[b]; pO[g] = a.O[b];
        p_ab[g] = P + bp.a_b; p_c00w[g] = P + bp.inc[0].c00_w; p_c00b[g] = P + bp.inc[0].c00_b; p_c10w[g] = P + bp.inc[0].c10_w; p_c10b[g] = P + bp.inc[0].c10_b;
        p_c01w[g] = P + bp.inc[0].c01_w; p_c01b[g] = P + bp.inc[0].c01_b; p_c11w[g] = P + bp.inc[0].c11_w; p_c11b[g] = P + bp.inc[0].c11_b;
        p_c12w[g] = P + bp.inc[0].c12_w; p_c12b[g] = P + bp.inc[0].c12_b; p_bw[g] = P + bp.b_w; p_bb[g] = P + bp.b_b;
    }
    if (part & 1) {
    {   // first conv of every outter block: A[b] = relu(conv3(occ[:, :b]; a) + a_b), one shared gather (csrc/fused.hip)
        int64_t w_off[7], b_off[7], o_off[7];
        for (int g = 0; g < 7; ++g) { w_off[g] = L.outter[g].a_w; b_off[g] = L.outter[g].a_b; o_off[g] = a.A[g + 1] - a.A[1]; }
        ProfScope ps(c.s, PK_OCC7, 7);
        TRY(linr_occ_conv7_launch(a.OCC, lo, mk, c.nbr_ld, c.R, P, w_off, b_off, a.A[1], o_off, c.s));
    }
    {   // H = [relu(conv0_0(A)) | relu(conv1_0(A))]
        Grp gp = Grp();
        goffs(gp.in, pA, ng); goffs(gp.w, p_c00w, ng); goffs(gp.b, p_c00b, ng); goffs(gp.out, pH, ng);
        goffs(gp.e0, p_c10w, ng); goffs(gp.e1, p_c10b, ng);
        ProfScope ps(c.s, PK_CONVPW_FWD, ng);
        TRY(linr_conv_pw_fwd_launch(pA[0], lo, mk, c.nbr_ld, c.R, p_c00w[0], p_c00b[0], p_c10w[0], p_c10b[0], a.H[g0], c.s, &gp, ng));
    }
    {   // both 4->4 convs + conv1_2 + residual -> M, I
        Grp gp = Grp();
        goffs(gp.in, pH, ng); goffs(gp.w, p_c01w, ng); goffs(gp.b, p_c01b, ng); goffs(gp.e1, p_c11w, ng); goffs(gp.e2, p_c11b, ng);
        goffs(gp.res, pA, ng); goffs(gp.e3, p_c12w, ng); goffs(gp.e4, p_c12b, ng); goffs(gp.e5, pM, ng); goffs(gp.out, pI, ng);
        ProfScope ps(c.s, PK_DUAL_FWD, ng);
        TRY(linr_dual44_fwd_launch(pH[0], lo, mk, c.nbr_ld, c.R, p_c01w[0], p_c01b[0], p_c11w[0], p_c11b[0], pA[0], p_c12w[0],
                                   p_c12b[0], a.M[g0], a.I[g0], c.s, &gp, ng));
    }
    }
    if (!(part & 2)) return linr_launch_rc();
    if (join)   // x_glob = O[0] = conv3(I[0]; b) of block_in: the one tail conv the others wait for
        TRY(conv3(c, false, a.I[0], 8, P + L.block_in.b_w, P + L.block_in.b_b, 8, 8, nullptr, 0, nullptr, 0, a.O[0], 8, 0));
    {   // O[b] = conv3(I; b) + x_glob
        Grp gp = Grp();
        goffs(gp.in, pI + o7, 7); goffs(gp.w, p_bw + o7, 7); goffs(gp.b, p_bb + o7, 7); goffs(gp.out, pO + o7, 7);
        ProfScope ps(c.s, PK_CONV88, 7);
        TRY(linr_cconv_launch(false, pI[o7], 8, lo, mk, c.nbr_ld, c.R, p_bw[o7], p_bb[o7], 8, 8, a.O[0], 8, nullptr, 0, a.O[1], 8, 0,
                              c.s, &gp, 7));
    }
    {   // the 8 occupancy heads
        const float *hO[8], *hC[8], *hP[8], *h_prw[8], *h_prb[8], *h_w1[8], *h_b1[8], *h_w2[8], *h_b2[8];
        for (int k = 0; k < 8; ++k) {
            hO[k] = a.O[k]; hC[k] = a.C[k]; hP[k] = a.P[k]; h_prw[k] = P + L.pr_w[k]; h_prb[k] = P + L.pr_b[k];
            h_w1[k] = P + L.h0_w[k]; h_b1[k] = P + L.h0_b[k]; h_w2[k] = P + L.h2_w[k]; h_b2[k] = P + L.h2_b[k];
        }
        Grp gp = Grp();
        goffs(gp.in, hO, 8); goffs(gp.w, h_prw, 8); goffs(gp.b, h_prb, 8); goffs(gp.out, hC, 8);
        goffs(gp.e0, h_w1, 8); goffs(gp.e1, h_b1, 8); goffs(gp.e2, h_w2, 8); goffs(gp.e3, h_b2, 8); goffs(gp.e5, hP, 8);
        for (int k = 0; k < 8; ++k) { gp.e4[k] = k; gp.e6[k] = (int64_t)k * nblk; }
        ProfScope ps(c.s, PK_HEAD_FWD, 8);
        TRY(linr_cconv_head_launch(a.O[0], lo, mk, c.nbr_ld, c.R, h_prw[0], h_prb[0], a.C[0], h_w1[0], h_b1[0], h_w2[0], h_b2[0],
                                   a.OCC, 8, a.P[0], bits_acc ? (double*)a.slab : nullptr, c.s, &gp, 8));
    }
    if (bits_acc) {
        ProfScope ps(c.s, PK_MISC, 0);
        TRY(linr_bits_finish_launch((const double*)a.slab, (int)(8 * nblk), bits_acc, c.s));
    }
    if (probs)
        for (int k = 0; k < 8; ++k)
            TRY(linr_hip_rc(hipMemcpyAsync(probs + (int64_t)k * c.R, a.P[k], (size_t)c.R * sizeof(float), hipMemcpyDeviceToDevice, c.s)));
    return linr_launch_rc();
}

extern "C" int linr_net_forward(const linr_frame* f, const float* params, float* arena, size_t arena_bytes,
                                int32_t stage_begin, int32_t stage_end, float* probs, double* bits_acc, void* stream) {
    Ctx c;
    TRY(check_frame(f, params, arena, arena_bytes, c));
    if (stage_begin < 0 || stage_end > 8 || stage_begin >= stage_end) return LINR_EINVAL;
    c.s = (hipStream_t)stream;
    if (c.R == 0) return 0;
    Arena& a = c.A;
    const float* P = c.P;
    // ground-truth / decoded occupancy (the decoder updates one column per call): gathered in place when the caller's buffer
    // has the zero row in front (check_frame points a.OCC at it), else copied into the padded arena matrix
    if (!(f->flags & LINR_FRAME_OCC_PADDED))
        TRY(linr_hip_rc(hipMemcpyAsync(a.OCC, f->occ, (size_t)c.R * 8 * sizeof(float), hipMemcpyDeviceToDevice, c.s)));
    const bool batched = grouped_enabled();
    const bool all_grouped = batched && stage_begin == 0 && stage_end == 8 && f->nbr_lo && f->nbr_mask;
    if (stage_begin == 0) {
        PadList pl;
        pl.n = a.npad;
        for (int i = 0; i < a.npad; ++i) { pl.off[i] = a.pad_off[i]; pl.w[i] = a.pad_w[i]; }
        {   // scale context: one small MLP per scale (model_core.py:48-53), all scales in one launch; its spare blocks clear the pad rows
            ProfScope ps(c.s, PK_SCE, 1);
            const SceArgs sa = sce_args(c);
            sce_fwd_k<float><<<sa.blk_off[sa.n_scales] + (a.npad + LINR_BLOCK / 32 - 1) / (LINR_BLOCK / 32), LINR_BLOCK, 0, c.s>>>(
                P, f->offset_feat, sa, c.R, nullptr, nullptr, a.X0, a.base, pl);      // (no hidden layer kept: sce_bwd_all_k recomputes it)
        }
        if (all_grouped && join_block_in(c)) {
            // block_in's first conv only: its Inception layer runs as group 0 of the outter blocks' launches (forward_batched)
            const BlockP& bi = c.L.block_in;
            TRY(conv3(c, false, a.X0, 8, P + bi.a_w, P + bi.a_b, bi.cin, 8, nullptr, 0, nullptr, 0, a.A[0], 8, LINR_RELU));
        } else {
            TRY(block_fwd(c, c.L.block_in, a.X0, 8, 0, nullptr));       // O[0] = x_glob
        }
    }
    if (all_grouped) return forward_batched(c, probs, bits_acc, 3, join_block_in(c));
    const int64_t nblk = linr_grid(c.R, LINR_BLOCK);
    bool fused_bits = false;
    for (int k = stage_begin; k < stage_end; ++k) {
        // prior_k = x_glob + outter_blocks[k-1](occ[:, :k])   (upsample.py:206-214; always the original x_glob)
        if (k > 0) TRY(block_fwd(c, c.L.outter[k - 1], a.OCC, 8, k, a.O[0]));
        if (c.f->nbr_lo && c.f->nbr_mask) {
            // prune conv + MLP + sigmoid + BCE partials in one launch (csrc/fused.hip)
            double* part = bits_acc ? (double*)a.slab + (int64_t)k * nblk : nullptr;
            TRY(linr_cconv_head_launch(a.O[k], clo(c), cmk(c), c.nbr_ld, c.R, P + c.L.pr_w[k], P + c.L.pr_b[k],
                                       a.C[k], P + c.L.h0_w[k], P + c.L.h0_b[k], P + c.L.h2_w[k], P + c.L.h2_b[k],
                                       a.OCC + k, 8, a.P[k], part, c.s));
            if (bits_acc) fused_bits = true;
        } else {
            TRY(conv3(c, false, a.O[k], 8, P + c.L.pr_w[k], P + c.L.pr_b[k], 8, 8, nullptr, 0, nullptr, 0, a.C[k], 8, 0));
            TRY(linear(c, a.C[k], 8, c.R, P + c.L.h0_w[k], 1, 8, P + c.L.h0_b[k], 8, 24, nullptr, 0, nullptr, 0, a.HH[k], 24,
                       LINR_RELU));
            TRY(linear(c, a.HH[k], 24, c.R, P + c.L.h2_w[k], 1, 24, P + c.L.h2_b[k], 24, 1, nullptr, 0, nullptr, 0, a.Z[k], 1, 0));
            if (bits_acc) {
                TRY(linr_bce_bits_fwd(a.Z[k], a.OCC + k, 8, c.R, a.P[k], bits_acc, a.slab, a.slab_bytes, c.s));
            } else {
                sigmoid_k<<<linr_grid(c.R, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(a.Z[k], c.R, a.P[k]);
            }
        }
        if (probs)
            TRY(linr_hip_rc(hipMemcpyAsync(probs + (int64_t)k * c.R, a.P[k], (size_t)c.R * sizeof(float),
                                           hipMemcpyDeviceToDevice, c.s)));
    }
    if (fused_bits)   // all stages' block partials in one fixed-order pass
        TRY(linr_bits_finish_launch((const double*)a.slab + (int64_t)stage_begin * nblk, (int)((stage_end - stage_begin) * nblk),
                                    bits_acc, c.s));
    return linr_launch_rc();
}

// ---- scale context as stand-alone ops (the launches linr_net_forward / _backward make for it) ---------------------------
static int sce_frame_check(const linr_frame* f, Layout& L) {
    if (!f || f->rows < 0 || f->n_scales < 1 || f->n_scales > MAX_SCALES || !f->row_off_h || !f->scale_idx_h) return LINR_EINVAL;
    if (!make_layout(L, f->model_scale_num, f->block_layers < 1 ? 1 : f->block_layers)) return LINR_EINVAL;
    if (f->row_off_h[0] != 0 || f->row_off_h[f->n_scales] != f->rows) return LINR_EINVAL;
    for (int s = 0; s < f->n_scales; ++s) {
        if (f->row_off_h[s + 1] < f->row_off_h[s]) return LINR_EINVAL;
        if (f->scale_idx_h[s] < 0 || f->scale_idx_h[s] >= f->model_scale_num) return LINR_EINVAL;
    }
    return 0;
}

extern "C" int linr_sce_fwd(const float* params, const linr_frame* f, float* mix, float* hid, float* x0, void* stream) {
    Ctx c;
    TRY(sce_frame_check(f, c.L));
    if (f->rows == 0) return 0;
    if (!params || !f->offset_feat || !hid || !x0) return LINR_EINVAL;
    if ((mix && !linr_aligned16(mix)) || !linr_aligned16(hid) || !linr_aligned16(x0)) return LINR_EALIGN;
    c.f = f;
    const SceArgs sa = sce_args(c);
    sce_fwd_k<float><<<sa.blk_off[sa.n_scales], LINR_BLOCK, 0, (hipStream_t)stream>>>(params, f->offset_feat, sa, f->rows, mix, hid, x0, nullptr,
                                                                              PadList{{}, {}, 0});
    return linr_launch_rc();
}

extern "C" int linr_sce_bwd(const float* params, const linr_frame* f, const float* gx0, const float* hid, float* ghid,
                            void* stream) {
    Ctx c;
    TRY(sce_frame_check(f, c.L));
    if (f->rows == 0) return 0;
    if (!params || !gx0 || !hid || !ghid) return LINR_EINVAL;
    if (!linr_aligned16(gx0) || !linr_aligned16(hid) || !linr_aligned16(ghid)) return LINR_EALIGN;
    c.f = f;
    const SceArgs sa = sce_args(c);
    sce_bwd_k<<<sa.blk_off[sa.n_scales], LINR_BLOCK, 0, (hipStream_t)stream>>>(params, sa, f->rows, gx0, hid, ghid);
    return linr_launch_rc();
}

// The whole backward of the scale context as one call (what linr_net_backward launches for it): the gradients of scale_emb and of
// every scale MLP of the frame into grads[0 .. linr_sce_param_count) - the scale context's parameters lead the flat layout whatever
// the width of the rest of the network - from gx0 [rows][8] and the hid [rows][16] that linr_sce_fwd kept.  The MLPs of scales the
// frame does not contain and their embedding rows get zeros.  slab: linr_sce_bwd_params_slab_bytes(model_scale_num) bytes.
#define SCE_SLAB_ROWS 256
extern "C" int64_t linr_sce_param_count(int32_t model_scale_num) {
    Layout L;
    return make_layout(L, model_scale_num, 1) ? L.block_in.a_w : (int64_t)LINR_EINVAL;
}
extern "C" size_t linr_sce_bwd_params_slab_bytes(int32_t model_scale_num) {
    const int64_t t = linr_sce_param_count(model_scale_num);
    return t > 0 ? (size_t)SCE_SLAB_ROWS * (size_t)t * sizeof(float) : 0;
}
extern "C" int linr_sce_bwd_params(const float* params, const linr_frame* f, const float* gx0, const float* hid, float* slab,
                                   size_t slab_bytes, float* grads, void* stream) {
    Ctx c;
    TRY(sce_frame_check(f, c.L));
    if (!params || !gx0 || !hid || !slab || !grads) return LINR_EINVAL;
    if (f->rows > 0 && !f->offset_feat) return LINR_EINVAL;
    if (!linr_aligned16(gx0) || !linr_aligned16(hid)) return LINR_EALIGN;
    const int64_t total = c.L.block_in.a_w;
    if (slab_bytes < (size_t)SCE_SLAB_ROWS * (size_t)total * sizeof(float)) return LINR_ENOSPC;
    c.f = f;
    hipStream_t s = (hipStream_t)stream;
    const int nb = SCE_SLAB_ROWS;
    SceArgs sa = sce_args(c);
    ZeroRanges zr;
    zr.n = 0; zr.prefix = total;
    zr.b[zr.n] = c.L.emb; zr.e[zr.n] = c.L.emb + (int64_t)c.L.S * 8; ++zr.n;
    ShortRanges sr;
    sr.n = 0;
    bool present[MAX_SCALES] = {};
    EmbArgs ea;
    int ns = 0;
    sa.wg_off[0] = 0;
    for (int j = 0; j < f->n_scales; ++j) {
        const int64_t nj = f->row_off_h[j + 1] - f->row_off_h[j];
        int64_t wg = nj > 0 ? (nj + LINR_BLOCK - 1) / LINR_BLOCK : 0;
        if (wg > nb) wg = nb;
        sa.wg_off[j + 1] = sa.wg_off[j] + (int)wg;
        if (wg == 0) continue;
        const int si = f->scale_idx_h[j];
        if (present[si]) return LINR_EINVAL;                 // two row ranges of one scale would share slab rows
        present[si] = true;
        if (wg < nb) { sr.b[sr.n] = c.L.m0_w[si]; sr.e[sr.n] = c.L.m2_b[si] + 8; sr.rows[sr.n] = (int)wg; ++sr.n; }
        ea.gb1[ns] = c.L.m0_b[si]; ea.w1[ns] = c.L.m0_w[si]; ea.gemb[ns] = c.L.emb + si * 8; ++ns;
    }
    for (int si = 0; si < c.L.S; ++si)
        if (!present[si]) {
            zr.b[zr.n] = c.L.m0_w[si];
            zr.e[zr.n] = si + 1 < c.L.S ? c.L.m0_w[si + 1] : total;
            ++zr.n;
        }
    if (ns > 0) sce_bwd_all_k<<<sa.wg_off[f->n_scales], SB_WAVES * 64, 0, s>>>(params, f->offset_feat, sa, gx0, hid, slab, total);
    wgrad_reduce_k<<<linr_grid(total, LINR_BLOCK / RED_SPLIT), LINR_BLOCK, 0, s>>>(slab, nb, total, grads, zr, sr);
    if (ns > 0) sce_emb_grad_all_k<<<ns, LINR_WAVE, 0, s>>>(params, grads, ea);
    return linr_launch_rc();
}

// fixed-order sum of the [nblocks][total] partial slab (shared with the op-level entries of csrc/fused.hip)
int linr_slab_reduce_launch(const float* big, int nblocks, int64_t total, float* gsum, hipStream_t s) {
    if (total <= 0) return 0;
    ZeroRanges zr;
    zr.n = 0; zr.prefix = 0;
    ShortRanges sr;
    sr.n = 0;
    wgrad_reduce_k<<<linr_grid(total, LINR_BLOCK / RED_SPLIT), LINR_BLOCK, 0, s>>>(big, nblocks, total, gsum, zr, sr);
    return linr_launch_rc();
}

struct Ptr8 { const float* p[8]; };
// dst = ((((((s7 + s6) + s5) + s4) + s3) + s2) + s1) + s0: the accumulation order of the stage-by-stage backward
__global__ __launch_bounds__(LINR_BLOCK) void sum8_k(Ptr8 src, int64_t n, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i >= n) return;
    float t = src.p[7][i];
#pragma unroll
    for (int k = 6; k >= 0; --k) t = t + src.p[k][i];
    dst[i] = t;
}

static void goffs_i(int64_t* dst, const int64_t* v, int n) {
    for (int i = 0; i < n; ++i) dst[i] = v[i] - v[0];
}

// Backward of the 8 heads and the 7 outter blocks as grouped launches (one launch per layer, gridDim.y = group); every
// kernel, its per-row arithmetic and the slab rows it writes are those of the stage-by-stage path, so gradients are
// bitwise the same.
static int backward_batched(Ctx& c, float gz_scale, bool join) {
    Arena& a = c.A;
    const float* P = c.P;
    const Layout& L = c.L;
    const int32_t* lo = clo(c);
    const uint32_t* mk = cmk(c);
    {
        const float *hC[8], *hP[8], *hO[8], *h_gC[8], *h_gO[8], *h_prw[8], *h_w1[8], *h_b1[8], *h_w2[8];
        int64_t o_w1[8], o_b1[8], o_w2[8], o_b2[8], o_prw[8], o_prb[8];
        for (int k = 0; k < 8; ++k) {
            hC[k] = a.C[k]; hP[k] = a.P[k]; hO[k] = a.O[k]; h_gC[k] = a.gC[k]; h_gO[k] = a.gO[k]; h_prw[k] = P + L.pr_w[k];
            h_w1[k] = P + L.h0_w[k]; h_b1[k] = P + L.h0_b[k]; h_w2[k] = P + L.h2_w[k];
            o_w1[k] = L.h0_w[k]; o_b1[k] = L.h0_b[k]; o_w2[k] = L.h2_w[k]; o_b2[k] = L.h2_b[k];
            o_prw[k] = L.pr_w[k]; o_prb[k] = L.pr_b[k];
        }
        {   // heads: gC and the four head-parameter gradients
            Grp gp = Grp();
            goffs(gp.in, hC, 8); goffs(gp.e0, hP, 8); goffs(gp.w, h_w1, 8); goffs(gp.b, h_b1, 8); goffs(gp.e2, h_w2, 8);
            goffs(gp.out, h_gC, 8);
            for (int k = 0; k < 8; ++k) gp.e1[k] = k;
            goffs_i(gp.e3, o_w1, 8); goffs_i(gp.e4, o_b1, 8); goffs_i(gp.e5, o_w2, 8); goffs_i(gp.e6, o_b2, 8);
            ProfScope ps(c.s, PK_HEAD_BWD, 8);
            int hrows = 0;
            TRY(linr_head_bwd_launch(a.C[0], a.P[0], a.OCC, 8, h_w1[0], h_b1[0], h_w2[0], gz_scale, a.gC[0], c.R, a.BIG, L.total,
                                     o_w1[0], o_b1[0], o_w2[0], o_b2[0], c.s, &gp, 8, c.nb, &hrows));
            c.note_short(L.h0_w[0], L.h2_b[7] + 1, hrows);          // the heads' parameters are one contiguous range
        }
        if (fused_bwd(c)) {   // C = conv3(prior_k; prune_k): gO[k] = bwd(gC[k]) and the weight gradients from one gather of gC
            Grp gp = Grp();
            goffs(gp.in, h_gC, 8); goffs(gp.res, hO, 8); goffs(gp.w, h_prw, 8); goffs(gp.out, h_gO, 8);
            goffs_i(gp.e3, o_prw, 8); goffs_i(gp.e4, o_prb, 8);
            LinrWgradDst d = {a.BIG, L.total, o_prw[0], o_prb[0], 8};
            ProfScope ps(c.s, PK_FUSED88, 8);
            int rows = 0;
            TRY(linr_conv88_bwd_wgrad_launch(a.gC[0], a.O[0], lo, mk, c.nbr_ld, c.R, h_prw[0], a.gO[0], nullptr, d, c.nb, c.s, &gp, 8, &rows));
            c.note_short(L.pr_w[0], L.pr_b[7] + 8, rows);
        } else {
        {   // C = conv3(prior_k; prune_k): weight gradients ...
            Grp gp = Grp();
            goffs(gp.in, hO, 8); goffs(gp.res, h_gC, 8); goffs_i(gp.w, o_prw, 8); goffs_i(gp.b, o_prb, 8);
            LinrWgradDst d = {a.BIG, L.total, o_prw[0], o_prb[0], 8};
            ProfScope ps(c.s, PK_WGRAD, 8);
            TRY(linr_conv3_wgrad_mfma(a.O[0], 8, a.gC[0], 8, c.f->nbr, c.nbr_ld, c.R, 8, 8, d, c.nb, c.s, &gp, 8, wg_t8t(c)));
        }
        {   // ... and gO[k] = bwd(gC[k])
            Grp gp = Grp();
            goffs(gp.in, h_gC, 8); goffs(gp.w, h_prw, 8); goffs(gp.out, h_gO, 8);
            ProfScope ps(c.s, PK_BWD_DATA, 8);
            TRY(linr_cconv_launch(true, a.gC[0], 8, lo, mk, c.nbr_ld, c.R, h_prw[0], nullptr, 8, 8, nullptr, 0, nullptr, 0, a.gO[0],
                                  8, 0, c.s, &gp, 8));
        }
        }
        Ptr8 src;
        for (int k = 0; k < 8; ++k) src.p[k] = a.gO[k];
        ProfScope ps(c.s, PK_MISC, 0);
        sum8_k<<<linr_grid(c.R * 8, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(src, c.R * 8, a.gXG);
    }
    // outter blocks 1..7 (slot b = block b; gO[b] is the gradient of the block output) and, with `join`, block_in as slot 0
    // (output gradient gXG = the sum above): one group per slot, g0 = first slot, ng = group count
    const int g0 = join ? 0 : 1, ng = 8 - g0, o7 = 1 - g0;
    const float *pA[8], *pH[8], *pM[8], *pI[8], *p_gO[8], *p_gI[8], *p_gM[8], *p_gH[8], *p_gA[8], *p_bw[8], *p_c12w[8], *p_c01w[8],
        *p_c11w[8], *p_c00w[8], *p_c10w[8], *p_in[8];
    int64_t o_bw[8], o_bb[8], o_c12w[8], o_c12b[8], o_c01w[8], o_c01b[8], o_c11w[8], o_c11b[8], o_c00w[8], o_c00b[8], o_c10w[8],
        o_c10b[8], o_aw[8], o_ab[8];
    for (int g = 0; g < ng; ++g) {
        const int b = g0 + g;
        const BlockP& bp = b == 0 ? L.block_in : L.outter[b - 1];
        pA[g] = a.A[b]; pH[g] = a.H[b]; pM[g] = a.M[b]; pI[g] = a.I[b]; p_gO[g] = b == 0 ? a.gXG : a.gO[b]; p_gI[g] = a.gI[b];
        p_gM[g] = a.gM[b]; p_gH[g] = a.gH[b]; p_gA[g] = a.gA[b]; p_in[g] = b == 0 ? a.X0 : a.OCC;
        p_bw[g] = P + bp.b_w; p_c12w[g] = P + bp.inc[0].c12_w; p_c01w[g] = P + bp.inc[0].c01_w; p_c11w[g] = P + bp.inc[0].c11_w;
        p_c00w[g] = P + bp.inc[0].c00_w; p_c10w[g] = P + bp.inc[0].c10_w;
        o_bw[g] = bp.b_w; o_bb[g] = bp.b_b; o_c12w[g] = bp.inc[0].c12_w; o_c12b[g] = bp.inc[0].c12_b; o_c01w[g] = bp.inc[0].c01_w; o_c01b[g] = bp.inc[0].c01_b;
        o_c11w[g] = bp.inc[0].c11_w; o_c11b[g] = bp.inc[0].c11_b; o_c00w[g] = bp.inc[0].c00_w; o_c00b[g] = bp.inc[0].c00_b; o_c10w[g] = bp.inc[0].c10_w;
        o_c10b[g] = bp.inc[0].c10_b; o_aw[g] = bp.a_w; o_ab[g] = bp.a_b;
    }
    (void)o7;
    if (fused_bwd(c)) {   // O = conv3(I; b): gI = bwd(gO; b), gM = (gI[:,4:8] @ W12^T) * (M > 0) and the weight gradient, one gather of gO
        Grp gp = Grp();
        goffs(gp.in, p_gO, ng); goffs(gp.res, pI, ng); goffs(gp.w, p_bw, ng); goffs(gp.out, p_gI, ng);
        goffs(gp.e0, p_c12w, ng); goffs(gp.e1, pM, ng); goffs(gp.e2, p_gM, ng); goffs_i(gp.e3, o_bw, ng); goffs_i(gp.e4, o_bb, ng);
        goffs_i(gp.e5, o_c12w, ng); goffs_i(gp.e6, o_c12b, ng);
        LinrWgradDst d = {a.BIG, L.total, o_bw[0], o_bb[0], 8};
        PwArgs pw = {p_c12w[0], nullptr, pM[0], a.gM[g0]};
        ProfScope ps(c.s, PK_FUSED88, ng);
        int rows = 0;
        TRY(linr_conv88_bwd_wgrad_launch(p_gO[0], pI[0], lo, mk, c.nbr_ld, c.R, p_bw[0], a.gI[g0], &pw, d, c.nb, c.s, &gp, ng, &rows,
                                         o_c12w[0], o_c12b[0]));
        // everything of a block behind its first conv comes from fused launches over the same groups (conv1_2 rides in this one,
        // conv0_0 / conv0_1 / conv1_0 / conv1_1 come below): one contiguous range of `rows` slab rows per block
        for (int g = 0; g < ng; ++g) c.note_short(o_c00w[g], o_bb[g] + 8, rows);
    } else {
    {   // O = conv3(I; b): weight gradient
        Grp gp = Grp();
        goffs(gp.in, pI, ng); goffs(gp.res, p_gO, ng); goffs_i(gp.w, o_bw, ng); goffs_i(gp.b, o_bb, ng);
        LinrWgradDst d = {a.BIG, L.total, o_bw[0], o_bb[0], 8};
        ProfScope ps(c.s, PK_WGRAD, ng);
        TRY(linr_conv3_wgrad_mfma(pI[0], 8, p_gO[0], 8, c.f->nbr, c.nbr_ld, c.R, 8, 8, d, c.nb, c.s, &gp, ng, wg_t8t(c)));
    }
    {   // gI = bwd(gO; b), gM = (gI[:,4:8] @ W12^T) * (M > 0)
        Grp gp = Grp();
        goffs(gp.in, p_gO, ng); goffs(gp.w, p_bw, ng); goffs(gp.out, p_gI, ng); goffs(gp.e0, p_c12w, ng); goffs(gp.e1, pM, ng);
        goffs(gp.e2, p_gM, ng);
        ProfScope ps(c.s, PK_BWD_DATA, ng);
        TRY(linr_conv_bwd_gm_launch(p_gO[0], lo, mk, c.nbr_ld, c.R, p_bw[0], p_c12w[0], pM[0], a.gI[g0], a.gM[g0], c.s, &gp, ng));
    }
    }
    if (!fused_bwd(c)) {   // conv1_2 weight gradient: M^T gI[:,4:8]  (the fused tail-conv launch above produces it on the side)
        Grp gp = Grp();
        goffs(gp.in, pM, ng); goffs(gp.res, p_gI, ng); goffs_i(gp.w, o_c12w, ng); goffs_i(gp.b, o_c12b, ng);
        LinrLinDst d = {a.BIG, L.total, o_c12w[0], 4, 1, o_c12b[0]};
        ProfScope ps(c.s, PK_LIN_WGRAD, ng);
        TRY(linr_linear_wgrad_partial(pM[0], 4, p_gI[0] + 4, 8, c.R, 4, 4, d, c.nb, c.s, &gp, ng));
    }
    if (fused_bwd(c)) {   // both 4->4 convs: gH and the two kernel / bias gradients from one gather of [gI[:,0:4] | gM]
        Grp gp = Grp();
        goffs(gp.in, p_gI, ng); goffs(gp.e5, p_gM, ng); goffs(gp.res, pH, ng); goffs(gp.w, p_c01w, ng); goffs(gp.e6, p_c11w, ng);
        goffs(gp.out, p_gH, ng); goffs_i(gp.e3, o_c01w, ng); goffs_i(gp.e4, o_c01b, ng); goffs_i(gp.e0, o_c11w, ng); goffs_i(gp.e1, o_c11b, ng);
        ProfScope ps(c.s, PK_FUSED_DUAL, ng);
        int rows_inc = 0;
        TRY(linr_dual44_bwd_wgrad_launch(p_gI[0], p_gM[0], pH[0], lo, mk, c.nbr_ld, c.R, p_c01w[0], p_c11w[0], a.gH[g0], a.BIG, L.total,
                                         o_c01w[0], o_c01b[0], o_c11w[0], o_c11b[0], c.nb, c.s, &gp, ng, &rows_inc));
        if (rows_inc != linr_fused_bwd_rows(c.R, c.nb, ng)) return LINR_EINVAL;        // (its parameters were registered with the tail conv)
    } else {   // both 4->4 convs: weight gradients, then gH
        Grp gp = Grp();
        goffs(gp.in, pH, ng); goffs(gp.res, p_gI, ng); goffs(gp.act, p_gM, ng); goffs_i(gp.w, o_c01w, ng); goffs_i(gp.b, o_c01b, ng);
        goffs_i(gp.e0, o_c11w, ng); goffs_i(gp.e1, o_c11b, ng);
        {
            ProfScope ps(c.s, PK_WGRAD, ng);
            TRY(linr_conv3_wgrad_dual44(pH[0], p_gI[0], 8, p_gM[0], 4, c.f->nbr, c.nbr_ld, c.R, a.BIG, L.total, o_c01w[0], o_c01b[0],
                                        o_c11w[0], o_c11b[0], c.nb, c.s, &gp, ng, wg_t8t(c)));
        }
        Grp gq = Grp();
        goffs(gq.in, p_gI, ng); goffs(gq.out, p_gH, ng); goffs(gq.e0, p_gM, ng); goffs(gq.w, p_c01w, ng); goffs(gq.e1, p_c11w, ng);
        goffs(gq.act, pH, ng);
        ProfScope ps(c.s, PK_BWD_DATA, ng);
        TRY(linr_dual44_bwd_launch(p_gI[0], p_gM[0], lo, mk, c.nbr_ld, c.R, p_c01w[0], p_c11w[0], pH[0], a.gH[g0], c.s, &gq, ng));
    }
    if (!fused_bwd(c)) {   // conv1_0 (1x1 8->4) weight gradient (the fused conv0_0 launch below produces it on the side)
        Grp gq = Grp();
        goffs(gq.in, pA, ng); goffs(gq.res, p_gH, ng); goffs_i(gq.w, o_c10w, ng); goffs_i(gq.b, o_c10b, ng);
        LinrLinDst dl = {a.BIG, L.total, o_c10w[0], 4, 1, o_c10b[0]};
        ProfScope ps(c.s, PK_LIN_WGRAD, ng);
        TRY(linr_linear_wgrad_partial(pA[0], 8, p_gH[0] + 4, 8, c.R, 8, 4, dl, c.nb, c.s, &gq, ng));
    }
    if (fused_bwd(c)) {   // conv0_0 (8->4): gA = (bwd(gH[:,0:4]; W00) + gI + gH[:,4:8] @ W10^T) * (A > 0) and its weight gradient, one gather
        Grp gp = Grp();
        goffs(gp.in, p_gH, ng); goffs(gp.res, pA, ng); goffs(gp.w, p_c00w, ng); goffs(gp.act, p_gI, ng); goffs(gp.out, p_gA, ng);
        goffs(gp.e0, p_c10w, ng); goffs_i(gp.e3, o_c00w, ng); goffs_i(gp.e4, o_c00b, ng); goffs_i(gp.e1, o_c10w, ng); goffs_i(gp.e2, o_c10b, ng);
        LinrWgradDst d = {a.BIG, L.total, o_c00w[0], o_c00b[0], 8};
        ProfScope ps(c.s, PK_FUSED_C00, ng);
        int rows_c00 = 0;
        TRY(linr_conv84_bwd_wgrad_launch(p_gH[0], pA[0], p_gI[0], lo, mk, c.nbr_ld, c.R, p_c00w[0], p_c10w[0], a.gA[g0], LINR_RELU_MASK, d,
                                         o_c10w[0], o_c10b[0], c.nb, c.s, &gp, ng, &rows_c00));
        if (rows_c00 != linr_fused_bwd_rows(c.R, c.nb, ng)) return LINR_EINVAL;       // (registered with the tail conv)
    } else {
    {   // conv0_0 (8->4) weight gradient
        Grp gp = Grp();
        goffs(gp.in, pA, ng); goffs(gp.res, p_gH, ng); goffs_i(gp.w, o_c00w, ng); goffs_i(gp.b, o_c00b, ng);
        LinrWgradDst d = {a.BIG, L.total, o_c00w[0], o_c00b[0], 8};
        ProfScope ps(c.s, PK_WGRAD, ng);
        TRY(linr_conv3_wgrad_mfma(pA[0], 8, p_gH[0], 8, c.f->nbr, c.nbr_ld, c.R, 8, 4, d, c.nb, c.s, &gp, ng, wg_t8t(c)));
    }
    {   // gA = (bwd(gH[:,0:4]; W00) + gI + gH[:,4:8] @ W10^T) * (A > 0)
        Grp gp = Grp();
        goffs(gp.in, p_gH, ng); goffs(gp.w, p_c00w, ng); goffs(gp.res, p_gI, ng); goffs(gp.act, pA, ng); goffs(gp.out, p_gA, ng);
        goffs(gp.e0, p_c10w, ng); goffs(gp.e1, p_gH, ng);
        ProfScope ps(c.s, PK_BWD_DATA, ng);
        TRY(linr_conv_bwd_ga_launch(p_gH[0], lo, mk, c.nbr_ld, c.R, p_c00w[0], p_c10w[0], p_gI[0], pA[0], a.gA[g0], LINR_RELU_MASK, c.s, &gp, ng));
    }
    }
    {   // A = relu(conv3(in; a)): weight gradient on the first b channels of the occupancy rows (outter block b) or on all 8
        // channels of the scale context x0 (block_in)
        // (with the fused backward block_in's first conv - slot 0 - gets its weight gradient from the launch that also produces
        // its input gradient, backward_core; the grouped launch then covers the outter blocks only)
        const int s0 = (join && fused_bwd(c)) ? 1 : 0, nq = ng - s0;
        bool same_in = nq == 7 && g0 + s0 == 1;
        for (int g = 0; g < nq && same_in; ++g) same_in = p_in[s0 + g] == a.OCC;
        if (same_in && fused_bwd(c)) {
            // the 7 outter blocks read the SAME occupancy rows: all seven weight gradients from one gather (csrc/occ_wgrad.hip)
            ProfScope ps(c.s, PK_WGRAD, nq);
            int rows = 0;
            TRY(linr_occ_wgrad7_launch(a.OCC, p_gA + s0, lo, mk, c.nbr_ld, c.R, a.BIG, L.total, o_aw + s0, o_ab + s0, c.nb, c.s, &rows));
            for (int g = 0; g < nq; ++g) c.note_short(o_aw[s0 + g], o_ab[s0 + g] + 8, rows);
            return 0;
        }
        Grp gp = Grp();
        goffs(gp.in, p_in + s0, nq); goffs(gp.res, p_gA + s0, nq); goffs_i(gp.w, o_aw + s0, nq); goffs_i(gp.b, o_ab + s0, nq);
        for (int g = 0; g < nq; ++g) gp.e2[g] = (g0 + s0 + g == 0) ? 8 : g0 + s0 + g;
        LinrWgradDst d = {a.BIG, L.total, o_aw[s0], o_ab[s0], 1};
        ProfScope ps(c.s, PK_WGRAD, nq);
        TRY(linr_conv3_wgrad_mfma(p_in[s0], 8, p_gA[s0], 8, c.f->nbr, c.nbr_ld, c.R, 1, 8, d, c.nb, c.s, &gp, nq, wg_t8t(c)));
    }
    return 0;
}

// The tail of every backward pass (fp32 executor: backward_core below; bf16 training executor: csrc/train_bf16.hip): the scale
// context's backward from gx0 [rows][8] fp32 and the hid [rows][16] of its forward (NULL: recomputed) (ghid and all four parameter gradients of every
// scale's context MLP in one launch), the fixed-order reduction of the [nb][total] slab `big` into gsum - `sh` lists the parameter
// ranges whose producers wrote fewer than nb slab rows - and the scale-embedding gradients derived from the reduced sums.
int linr_bwd_tail_launch(const linr_frame* f, const Layout& L, const float* P, const float* gx0, const float* hid, float* big,
                         float* gsum, int nb, const LinrShortRange* sh, int nsh, hipStream_t stream) {
    Ctx c;
    c.f = f; c.L = L; c.s = stream; c.nb = nb;
    for (int i = 0; i < nsh; ++i) c.note_short(sh[i].b, sh[i].e, sh[i].rows);
    int ns = 0, sl[MAX_SCALES];
    for (int s = 0; s < f->n_scales; ++s)
        if (f->row_off_h[s + 1] > f->row_off_h[s]) sl[ns++] = s;
    if (ns >= 1) {          // ghid and all four parameter gradients of every scale's context MLP in one launch
        ProfScope ps(c.s, PK_SCE, 1);
        SceArgs sa = sce_args(c);
        // slab rows per scale: one workgroup per 256 rows, at most nb; a scale with fewer leaves a short range for the reduction
        sa.wg_off[0] = 0;
        for (int j = 0; j < f->n_scales; ++j) {
            const int64_t nj = f->row_off_h[j + 1] - f->row_off_h[j];
            int64_t wg = nj > 0 ? (nj + LINR_BLOCK - 1) / LINR_BLOCK : 0;
            if (wg > c.nb) wg = c.nb;
            sa.wg_off[j + 1] = sa.wg_off[j] + (int)wg;
            if (wg > 0) {
                const int si = f->scale_idx_h[j];
                c.note_short(c.L.m0_w[si], c.L.m2_b[si] + 8, (int)wg);
            }
        }
        sce_bwd_all_k<<<sa.wg_off[f->n_scales], SB_WAVES * 64, 0, c.s>>>(P, f->offset_feat, sa, gx0, hid, big, c.L.total);
    }
    // one pass sums every parameter's per-block partials in fixed order
    ProfScope ps_tail(c.s, PK_MISC, 0);
    {   // the scale embedding and the context MLPs of absent scales get no partials: the reduction writes their zeros itself
        ZeroRanges zr;
        zr.n = 0; zr.prefix = c.L.block_in.a_w;
        zr.b[zr.n] = c.L.emb; zr.e[zr.n] = c.L.emb + (int64_t)c.L.S * 8; ++zr.n;
        bool present[MAX_SCALES] = {};
        for (int j = 0; j < ns; ++j) present[f->scale_idx_h[sl[j]]] = true;
        for (int si = 0; si < c.L.S; ++si)
            if (!present[si]) {
                zr.b[zr.n] = c.L.m0_w[si];
                zr.e[zr.n] = si + 1 < c.L.S ? c.L.m0_w[si + 1] : c.L.block_in.a_w;
                ++zr.n;
            }
        ShortRanges sr;
        sr.n = (int)c.shortr.size();
        for (int i = 0; i < sr.n; ++i) { sr.b[i] = c.shortr[i].b; sr.e[i] = c.shortr[i].e; sr.rows[i] = c.shortr[i].rows; }
        wgrad_reduce_k<<<linr_grid(c.L.total, LINR_BLOCK / RED_SPLIT), LINR_BLOCK, 0, c.s>>>(big, c.nb, c.L.total, gsum, zr, sr);
    }
    if (ns > 0) {
        EmbArgs ea;
        for (int j = 0; j < ns; ++j) {
            const int si = f->scale_idx_h[sl[j]];
            ea.gb1[j] = c.L.m0_b[si]; ea.w1[j] = c.L.m0_w[si]; ea.gemb[j] = c.L.emb + si * 8;
        }
        sce_emb_grad_all_k<<<ns, LINR_WAVE, 0, c.s>>>(P, gsum, ea);
    }
    return linr_launch_rc();
}

// backward of gscale * bits: leaves the parameter gradient of THIS call in arena GSUM (flat, parameters() order)
static int backward_core(Ctx& c, float gscale) {
    const linr_frame* f = c.f;
    Arena& a = c.A;
    const float* P = c.P;
    const float gz_scale = gscale * 1.4426950408889634f;       // d(bits)/d(nats) = 1/ln 2
    const bool batched = grouped_enabled();
    const bool grouped = batched && c.f->nbr_lo && c.f->nbr_mask;
    const bool join = grouped && join_block_in(c);
    if (grouped) TRY(backward_batched(c, gz_scale, join));
    for (int k = grouped ? -1 : 7; k >= 0; --k) {
        if (c.f->nbr_lo && c.f->nbr_mask) {
            // recompute the hidden layer, gC and the four head-parameter gradients in one launch (csrc/fused.hip)
            TRY(linr_head_bwd_launch(a.C[k], a.P[k], a.OCC + k, 8, P + c.L.h0_w[k], P + c.L.h0_b[k], P + c.L.h2_w[k], gz_scale,
                                     a.gC[k], c.R, a.BIG, c.L.total, c.L.h0_w[k], c.L.h0_b[k], c.L.h2_w[k], c.L.h2_b[k], c.s, nullptr, 1, c.nb));
        } else {
            TRY(linr_bce_bits_bwd(a.P[k], a.OCC + k, 8, c.R, gz_scale, a.gZ, c.s));
            // z = HH @ h2 + b ; HH = relu(C @ h0 + b)
            TRY(linear_wgrad(c, a.HH[k], 24, a.gZ, 1, c.R, 24, 1, c.L.h2_w[k], 1, 24, c.L.h2_b[k]));
            TRY(linear(c, a.gZ, 1, c.R, P + c.L.h2_w[k], 24, 1, nullptr, 1, 24, nullptr, 0, a.HH[k], 24, a.gHH, 24, LINR_RELU_MASK));
            TRY(linear_wgrad(c, a.C[k], 8, a.gHH, 24, c.R, 8, 24, c.L.h0_w[k], 1, 8, c.L.h0_b[k]));
            TRY(linear(c, a.gHH, 24, c.R, P + c.L.h0_w[k], 8, 1, nullptr, 24, 8, nullptr, 0, nullptr, 0, a.gC[k], 8, 0));
        }
        // C = conv3(prior_k; prune_k)
        TRY(conv3_wgrad(c, a.O[k], 8, a.gC[k], 8, 8, 8, c.L.pr_w[k], c.L.pr_b[k]));
        TRY(conv3(c, true, a.gC[k], 8, P + c.L.pr_w[k], nullptr, 8, 8, nullptr, 0, nullptr, 0, a.gO[k], 8, 0));
        // prior_k = x_glob (+ outter block k-1): both receive gO
        axpy_k<<<linr_grid(c.R * 8, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(a.gO[k], c.R * 8, a.gXG, k == 7 ? 0 : 1);
        if (k > 0) TRY(block_bwd(c, c.L.outter[k - 1], a.OCC, 8, k, a.gO[k], nullptr));
    }
    if (join && fused_bwd(c)) {      // first conv of block_in: input gradient and weight gradient from one gather of gA[0]
        const BlockP& bi = c.L.block_in;
        LinrWgradDst d = {a.BIG, c.L.total, bi.a_w, bi.a_b, 8};
        ProfScope ps(c.s, PK_FUSED88, 1);
        int rows = 0;
        TRY(linr_conv88_bwd_wgrad_launch(a.gA[0], a.X0, clo(c), cmk(c), c.nbr_ld, c.R, P + bi.a_w, a.gX0, nullptr, d, c.nb, c.s, nullptr, 1, &rows));
        c.note_short(bi.a_w, bi.a_b + 8, rows);
    } else if (join) {      // everything but the input gradient of its first conv was part of the grouped launches
        const BlockP& bi = c.L.block_in;
        TRY(conv3(c, true, a.gA[0], 8, P + bi.a_w, nullptr, bi.cin, 8, nullptr, 0, nullptr, 0, a.gX0, 8, 0));
    } else {
        TRY(block_bwd(c, c.L.block_in, a.X0, 8, 0, a.gXG, a.gX0));
    }
    std::vector<LinrShortRange> sh(c.shortr.size());
    for (size_t i = 0; i < sh.size(); ++i) sh[i] = {c.shortr[i].b, c.shortr[i].e, c.shortr[i].rows};
    return linr_bwd_tail_launch(f, c.L, P, a.gX0, nullptr, a.BIG, a.GSUM, c.nb, sh.data(), (int)sh.size(), c.s);
}

extern "C" int linr_net_backward(const linr_frame* f, const float* params, float* arena, size_t arena_bytes, float gscale,
                                 float* grads, void* stream) {
    Ctx c;
    TRY(check_frame(f, params, arena, arena_bytes, c));
    if (!grads) return LINR_EINVAL;
    c.s = (hipStream_t)stream;
    if (c.R == 0) return 0;
    TRY(backward_core(c, gscale));
    axpy_k<<<linr_grid(c.L.total, LINR_BLOCK), LINR_BLOCK, 0, c.s>>>(c.A.GSUM, c.L.total, grads, 1);
    return linr_launch_rc();
}

// decoded byte column -> float column k of the occupancy matrix [rows][8]
__global__ __launch_bounds__(LINR_BLOCK) void occ_col_from_u8_k(const uint8_t* __restrict__ sym, int64_t n, float* __restrict__ occ_col) {
    const int64_t r = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (r < n) occ_col[r * 8] = (float)sym[r];
}

extern "C" int linr_net_decode_stages(const linr_frame* f, const float* params, const uint8_t* codes, float min_param,
                                      float max_param, void* arena, size_t arena_bytes, const uint8_t* const* streams_h,
                                      const int64_t* stream_len_h, float* probs, float* p_pinned, uint8_t* s_pinned,
                                      uint8_t* s_dev, void* stream) {
    if (!f || !arena || !streams_h || !stream_len_h || !probs || !p_pinned || !s_pinned || !s_dev || !f->occ) return LINR_EINVAL;
    if (!codes && !params) return LINR_EINVAL;
    const int64_t R = f->rows;
    if (R == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    float* occ = const_cast<float*>(f->occ);
    for (int k = 0; k < 8; ++k) {
        if (codes) TRY(linr_net_forward_bf16(f, codes, min_param, max_param, arena, arena_bytes, k, k + 1, probs, nullptr, stream));
        else TRY(linr_net_forward(f, params, (float*)arena, arena_bytes, k, k + 1, probs, nullptr, stream));
        TRY(linr_hip_rc(hipMemcpyAsync(p_pinned, probs + (int64_t)k * R, (size_t)R * sizeof(float), hipMemcpyDeviceToHost, s)));
        TRY(linr_hip_rc(hipStreamSynchronize(s)));
        for (int i = 0; i < f->n_scales; ++i) {
            const int64_t r0 = f->row_off_h[i], n = f->row_off_h[i + 1] - r0;
            if (n <= 0) continue;
            const int rc = linr_ac_decode_binary(p_pinned + r0, n, streams_h[i * 8 + k], stream_len_h[i * 8 + k], s_pinned + r0);
            if (rc) return rc;
        }
        TRY(linr_hip_rc(hipMemcpyAsync(s_dev, s_pinned, (size_t)R, hipMemcpyHostToDevice, s)));
        occ_col_from_u8_k<<<linr_grid(R, LINR_BLOCK), LINR_BLOCK, 0, s>>>(s_dev, R, occ + k);
        TRY(linr_launch_rc());
    }
    return linr_hip_rc(hipStreamSynchronize(s));       // s_pinned / p_pinned may be reused by the caller right away
}

// torch.optim.Adam's step over the flat parameter buffer with the per-scale step counters of the scale-context MLPs (shared with
// csrc/train_bf16.hip): bias corrections in double, like torch.optim.Adam's Python scalars
int linr_adam_step_launch(const Layout& L, float* params, const float* gsum, float* exp_avg, float* exp_avg_sq, double lr, int64_t step,
                          const int64_t* scale_steps_h, double beta1, double beta2, double eps, double weight_decay, hipStream_t s) {
    LinrAdamRanges rg;
    rg.count = 0; rg.begin = L.m0_w[0]; rg.len = L.S > 1 ? L.m0_w[1] - L.m0_w[0] : L.block_in.a_w - L.m0_w[0];
    if (scale_steps_h) {
        rg.count = L.S;
        // scale_steps_h[s] = updates applied to the context MLP of scale s INCLUDING this one; 0 = it has never had a gradient and
        // is skipped (torch.optim.Adam skips .grad None; torch 1.13's zero_grad() leaves zeros afterwards, so a started scale is
        // updated on every step - with the zero gradient the reduction writes for a scale this frame lacks)
        for (int sc = 0; sc < L.S; ++sc) {
            const int64_t t = scale_steps_h[sc];
            rg.active[sc] = t >= 1 ? 1 : 0;
            rg.step_size[sc] = t >= 1 ? (float)(lr / (1.0 - pow(beta1, (double)t))) : 0.0f;
            rg.bc2_sqrt[sc] = t >= 1 ? (float)sqrt(1.0 - pow(beta2, (double)t)) : 1.0f;
        }
    }
    return linr_adam_launch(params, gsum, exp_avg, exp_avg_sq, L.total, lr / (1.0 - pow(beta1, (double)step)),
                            sqrt(1.0 - pow(beta2, (double)step)), beta1, beta2, eps, weight_decay,
                            scale_steps_h ? &rg : nullptr, s);
}

extern "C" int linr_net_train_step(const linr_frame* f, float* params, float* arena, size_t arena_bytes, float gscale,
                                   float* exp_avg, float* exp_avg_sq, double lr, int64_t step, const int64_t* scale_steps_h,
                                   double beta1, double beta2, double eps, double weight_decay, double* bits_acc,
                                   void* stream) {
    if (!exp_avg || !exp_avg_sq || !bits_acc || step < 1) return LINR_EINVAL;
    Ctx c;
    TRY(check_frame(f, params, arena, arena_bytes, c));
    if (scale_steps_h) {          // checked before anything is launched
        for (int s = 0; s < c.L.S; ++s)
            if (scale_steps_h[s] < 0) return LINR_EINVAL;
        for (int j = 0; j < f->n_scales; ++j)          // a scale of this frame cannot be "never started"
            if (f->row_off_h[j + 1] > f->row_off_h[j] && scale_steps_h[f->scale_idx_h[j]] < 1) return LINR_EINVAL;
    }
    TRY(linr_net_forward(f, params, arena, arena_bytes, 0, 8, nullptr, bits_acc, stream));
    c.s = (hipStream_t)stream;
    if (c.R == 0) return 0;
    TRY(backward_core(c, gscale));
    ProfScope ps(c.s, PK_MISC, 0);
    return linr_adam_step_launch(c.L, params, c.A.GSUM, exp_avg, exp_avg_sq, lr, step, scale_steps_h, beta1, beta2, eps, weight_decay, c.s);
}

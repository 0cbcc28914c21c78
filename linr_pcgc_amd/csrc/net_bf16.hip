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
#include "bf16_common.h"
#include <stdlib.h>

#define TRY(e) do { int rc_ = (e); if (rc_) return rc_; } while (0)

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
    if (a.pads.n >= BPADS_MAX) abort();
    a.pads.off[a.pads.n] = (a.cur - (int64_t)(reinterpret_cast<char*>(a.mats) - a.base)) / 2;
    a.pads.w[a.pads.n++] = 8;
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

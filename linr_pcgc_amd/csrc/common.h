// Internal helpers shared by the gfx950 kernels.  Not part of the C-ABI (include/linr_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/linr_hip.h"

#define LINR_BLOCK 256
#define LINR_WAVE 64

static inline int linr_hip_rc(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }
static inline int linr_launch_rc() { return linr_hip_rc(hipGetLastError()); }
// Order in which every 3x3x3 forward / backward-data convolution kernel visits its 27 taps: step kk handles tap
// LINR_TAP(kk) = dx-index (kk / 9) + 3 * dy-index ((kk / 3) % 3) + 9 * dz-index (kk % 3), i.e. x-slab by x-slab, inside a slab
// (dx,dy) column by column, the three dz taps of a column back to back.  In the x-major row order (z fastest) the dz neighbours
// of a column are consecutive rows and the three dy columns of a slab lie within a few dozen rows, so a 64-row tile reads
// each of its three ~3 KB neighbour regions once and then hits in the L1 for the other eight taps of the slab (pure gathers:
// 19.2 us per pass in ascending tap order, 17.6 column by column, 17.0 slab by slab - tools/gather_probe.hip; training step
// 2.216 -> 2.166 -> 2.135 ms).  The order is part of the arithmetic (fp32 accumulation order of every output): ALL kernels of
// that family use it - MFMA, VALU, gather-table reference, dual 4->4, shared occupancy conv, bf16 - which keeps them
// bit-identical to each other and the decoder to the encoder.
#define LINR_TAP(kk) (((kk) / 9) + 3 * (((kk) / 3) % 3) + 9 * ((kk) % 3))

static inline bool linr_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline unsigned linr_grid(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// number of persistent blocks used by the two-pass reductions: enough to fill 256 CUs several times over,
// few enough that the partial slabs stay small.
static inline int linr_reduce_blocks(int64_t n, int rows_per_tile) {
    int64_t tiles = (n + rows_per_tile - 1) / rows_per_tile;
    int64_t nb = tiles < 512 ? tiles : 512;
    return (int)(nb < 1 ? 1 : nb);
}

// Destination of per-block partial weight gradients: element e of block b lands at base[b*block_stride + off + e].
// The network executor points this at one big [LINR_WG_BLOCKS][n_params] slab indexed by flat parameter offset, so
// ONE final pass sums every parameter's partials in fixed order (deterministic) - fused with Adam in the train step.
struct LinrWgradDst {
    float* base;
    int64_t block_stride;
    int64_t w_off;       // weight tensor offset
    int64_t b_off;       // bias offset
    int cin_valid;       // conv3 only: input channels actually present (<= kernel width)
};
struct LinrLinDst {      // pointwise layers: element (ci,co) at w_off + ci*ws_ci + co*ws_co, bias co at b_off + co
    float* base;
    int64_t block_stride;
    int64_t w_off;
    int ws_ci, ws_co;
    int64_t b_off;
};
// per-range Adam schedule (csrc/loss_optim.hip: adam_k); 16 = MAX_SCALES of the executor
struct LinrAdamRanges {
    int count;               // 0: none
    int64_t begin, len;      // range r = [begin + r*len, begin + (r+1)*len)
    int active[16];
    float step_size[16], bc2_sqrt[16];
};
// torch.optim.Adam's single-tensor update, the ONE definition every kernel that applies it uses (adam_k; the tail-fold experiments of
// round 4 used it too): the operations
// are pinned - separate multiplies and adds for the moments, one fused multiply-add for the step (what adam_k has compiled to since
// round 1) - so that the fused and the stand-alone path update parameters bit-identically whatever the surrounding code looks like.
__device__ __forceinline__ float linr_adam_update(float p, float grad, float& m, float& v, float step_size, float bc2_sqrt, float beta1,
                                                  float omb1, float beta2, float omb2, float eps, float wd) {
#pragma clang fp contract(off)
    const float g = fmaf(wd, p, grad);
    const float mi = m * beta1 + omb1 * g;           // exp_avg.mul_(beta1).add_(grad, alpha=1-beta1)
    const float vi = v * beta2 + (omb2 * g) * g;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    return fmaf(-step_size, mi / denom, p);
}
__attribute__((visibility("hidden")))
int linr_adam_launch(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double step_size,
                     double bc2_sqrt, double beta1, double beta2, double eps, double weight_decay,
                     const LinrAdamRanges* rg, hipStream_t s);
#define LINR_WG_BLOCKS 512   // persistent blocks of every weight-gradient kernel (2 per CU; sweep: tools/wg_blocks_sweep.sh)

// Grouped launches: independent layers of equal shape (the 7 outter blocks, the 8 occupancy heads, whose inputs are the
// ground-truth occupancy and x_glob during overfitting / encoding) run as ONE launch with gridDim.y = groups.  Group g adds
// these ELEMENT offsets to the kernel's base pointers; a plain launch passes all zeros.  Per-row arithmetic is identical
// in grouped and plain launches, so the staged decoder (plain) reproduces the encoder (grouped) bit for bit.
#define LINR_MAXG 8
struct Grp {
    int64_t in[LINR_MAXG], w[LINR_MAXG], b[LINR_MAXG], res[LINR_MAXG], act[LINR_MAXG], out[LINR_MAXG];
    int64_t e0[LINR_MAXG], e1[LINR_MAXG], e2[LINR_MAXG], e3[LINR_MAXG], e4[LINR_MAXG], e5[LINR_MAXG], e6[LINR_MAXG];
    int64_t n[LINR_MAXG];      // > 0: the group's own row count (groups of unequal size: the scales of a frame)
};


// ---- internal launchers shared with the network executor (C++ linkage, not exported) ---------------------------
// epilogue order of both: acc (+ bias) -> + res -> + old (LINR_ACCUM) -> * (act > 0) (LINR_RELU_MASK) -> ReLU
__attribute__((visibility("hidden")))
int linr_conv3_launch(bool bwd, const float* in, int in_ld, const int32_t* nbr, int64_t nbr_ld, int64_t n,
                      const float* W, const float* bias, int cin, int cout, const float* res, int res_ld,
                      const float* act, int act_ld, float* out, int out_ld, unsigned flags, hipStream_t s);
__attribute__((visibility("hidden")))
int linr_linear_launch(const float* in, int in_ld, int64_t n, const float* W, int ws_ci, int ws_co, const float* bias,
                       int cin, int cout, const float* res, int res_ld, const float* act, int act_ld, float* out,
                       int out_ld, unsigned flags, hipStream_t s);
__attribute__((visibility("hidden")))
int linr_conv3_wgrad_partial(const float* in, int in_ld, const float* gout, int gout_ld, const int32_t* nbr,
                             int64_t nbr_ld, int64_t n, int cin, int cout, LinrWgradDst d, int nblocks, unsigned flags,
                             hipStream_t s);
__attribute__((visibility("hidden")))
int linr_linear_wgrad_partial(const float* in, int in_ld, const float* gout, int gout_ld, int64_t n, int cin, int cout,
                              LinrLinDst d, int nblocks, hipStream_t s, const Grp* gp = nullptr, int ngroups = 1);
__attribute__((visibility("hidden"))) int linr_lin_blocks(int64_t n);
__attribute__((visibility("hidden")))
int linr_linear_slab_reduce_launch(const float* slab, int nblocks, int64_t stride, int cin, int cout, float* gW, int ws_ci, int ws_co,
                                   float* gb, unsigned flags, hipStream_t s);
__attribute__((visibility("hidden")))
int linr_cconv_launch(bool bwd, const float* in, int in_ld, const int32_t* lo, const uint32_t* mask, int64_t ld,
                      int64_t n, const float* W, const float* bias, int cin, int cout, const float* res, int res_ld,
                      const float* act, int act_ld, float* out, int out_ld, unsigned flags, hipStream_t s,
                      const Grp* gp = nullptr, int ngroups = 1);
__attribute__((visibility("hidden")))
int linr_cconv_head_launch(const float* in, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                           const float* W, const float* bias, float* c_out, const float* w1, const float* b1,
                           const float* w2, const float* b2, const float* target, int target_ld, float* p_out,
                           double* partial, hipStream_t s, const Grp* gp = nullptr, int ngroups = 1);
__attribute__((visibility("hidden")))
int linr_head_bwd_launch(const float* c, const float* p, const float* target, int target_ld, const float* w1,
                         const float* b1, const float* w2, float gscale, float* gc, int64_t n, float* big,
                         int64_t block_stride, int64_t off_w1, int64_t off_b1, int64_t off_w2, int64_t off_b2,
                         hipStream_t s, const Grp* gp = nullptr, int ngroups = 1, int nblocks = LINR_WG_BLOCKS,
                         int* rows_written = nullptr);
__attribute__((visibility("hidden")))
int linr_slab_reduce_launch(const float* big, int nblocks, int64_t total, float* gsum, hipStream_t s);
__attribute__((visibility("hidden")))
int linr_bits_finish_launch(const double* partial, int count, double* bits_acc, hipStream_t s);
__attribute__((visibility("hidden")))
int linr_dual44_fwd_launch(const float* H, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, const float* w01,
                           const float* b01, const float* w11, const float* b11, const float* A, const float* w12,
                           const float* b12, float* M, float* I, hipStream_t s, const Grp* gp = nullptr, int ngroups = 1);
__attribute__((visibility("hidden")))
int linr_dual44_bwd_launch(const float* gI, const float* gM, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                           const float* w01, const float* w11, const float* H, float* gH, hipStream_t s,
                           const Grp* gp = nullptr, int ngroups = 1);
__attribute__((visibility("hidden")))
int linr_conv_pw_fwd_launch(const float* A, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, const float* w00,
                            const float* b00, const float* w10, const float* b10, float* H, hipStream_t s,
                            const Grp* gp = nullptr, int ngroups = 1);
__attribute__((visibility("hidden")))
int linr_conv_bwd_gm_launch(const float* gO, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, const float* wb,
                            const float* w12, const float* M, float* gI, float* gM, hipStream_t s,
                            const Grp* gp = nullptr, int ngroups = 1);
__attribute__((visibility("hidden")))
int linr_conv_bwd_ga_launch(const float* gH, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, const float* w00,
                            const float* w10, const float* gI, const float* A, float* gA, unsigned flags, hipStream_t s,
                            const Grp* gp = nullptr, int ngroups = 1);
__attribute__((visibility("hidden")))
int linr_occ_conv7_launch(const float* occ, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n, const float* P,
                          const int64_t* w_off, const int64_t* b_off, float* out, const int64_t* out_off, hipStream_t s);
__attribute__((visibility("hidden")))
int linr_conv3_wgrad_mfma(const float* in, int in_ld, const float* gout, int gout_ld, const int32_t* nbr, int64_t nbr_ld,
                          int64_t n, int cin, int cout, LinrWgradDst d, int nblocks, hipStream_t s, const Grp* gp = nullptr,
                          int ngroups = 1, const int32_t* tile8t = nullptr);
struct PwArgs;
__attribute__((visibility("hidden")))
int linr_fused_bwd_rows(int64_t n, int nb, int ngroups);
// backward-data + weight gradient of a conv 8->8 from one gather (csrc/fused_bwd.hip); pw != nullptr: gM epilogue
__attribute__((visibility("hidden")))
int linr_conv88_bwd_wgrad_launch(const float* g, const float* xin, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                                 const float* W, float* out, const PwArgs* pw, LinrWgradDst d, int nb, hipStream_t s,
                                 const Grp* gp = nullptr, int ngroups = 1, int* rows_written = nullptr, int64_t w12_off = 0,
                                 int64_t b12_off = 0);
// ... of the two 4->4 convolutions of an Inception layer (gH, masked by H > 0) and of conv0_0 8->4 (gA with cconv_mfma_k's EPI 4)
__attribute__((visibility("hidden")))
int linr_dual44_bwd_wgrad_launch(const float* gI, const float* gM, const float* H, const int32_t* lo, const uint32_t* mask,
                                 int64_t ld, int64_t n, const float* w01, const float* w11, float* gH, float* big,
                                 int64_t block_stride, int64_t w_off0, int64_t b_off0, int64_t w_off1, int64_t b_off1, int nb,
                                 hipStream_t s, const Grp* gp = nullptr, int ngroups = 1, int* rows_written = nullptr);
__attribute__((visibility("hidden")))
int linr_conv84_bwd_wgrad_launch(const float* gH, const float* A, const float* gI, const int32_t* lo, const uint32_t* mask,
                                 int64_t ld, int64_t n, const float* w00, const float* w10, float* gA, unsigned flags,
                                 LinrWgradDst d, int64_t w10_off, int64_t b10_off, int nb, hipStream_t s, const Grp* gp = nullptr,
                                 int ngroups = 1, int* rows_written = nullptr);
__attribute__((visibility("hidden")))
int linr_conv3_wgrad_dual44(const float* H, const float* g0, int g0_ld, const float* g1, int g1_ld, const int32_t* nbr,
                            int64_t nbr_ld, int64_t n, float* big, int64_t block_stride, int64_t w_off0, int64_t b_off0,
                            int64_t w_off1, int64_t b_off1, int nblocks, hipStream_t s, const Grp* gp = nullptr,
                            int ngroups = 1, const int32_t* tile8t = nullptr);

// test hook (include/linr_hip.h: linr_debug_poison): poisons LDS and vector registers of every CU on `s` when bit `kind` of the
// mask is set; kinds 0..13 = the linr_prof_* classes of the fp32 executor, 14 = the bf16 executor, 15 = the decoder's own kernels
__attribute__((visibility("hidden"))) void linr_poison_hook(hipStream_t s, int kind);
// csrc/occ_wgrad.hip: weight gradients of the first convolutions of the 7 outter blocks from one gather of the occupancy rows
__attribute__((visibility("hidden")))
int linr_occ_wgrad7_launch(const float* occ, const float* const* g, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                           float* big, int64_t block_stride, const int64_t* w_off, const int64_t* b_off, int nb, hipStream_t s,
                           int* rows_written);

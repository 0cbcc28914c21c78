/* linr_hip.h - C-ABI of the MI355X (gfx950) coding-network engine for LINR-PCGC.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference (100 % Python) reaches its
 * native code through MinkowskiEngine 0.5.4 and torchac 0.9.3 wheels; each entry point below names the
 * reference call site(s) it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _h (host); the caller (PyTorch's caching
 *     allocator) owns all memory; nothing is allocated, freed or retained across calls.  The library keeps no mutable
 *     global state on the data path: calls on different streams may run from different host threads (the GOP decoder
 *     does).  The two optional process-wide aids - the live kernel timing of linr_prof_* (mutex-guarded) and the
 *     on-chip poison test hook of linr_debug_poison (an atomic mask) - are off by default.
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered, no host synchronisation.
 *   - return value: 0 on success, >0 = hipError_t, <0 = argument error (LINR_E*).  No exceptions cross.
 *   - feature matrices are row-major float32 [rows, ld] with an explicit leading dimension `*_ld` so that a
 *     channel slice (ME.cat / merge_two_frames) is a pointer offset, not a copy.
 *   - the kernel map is int32 nbr[27][ld]: nbr[k*ld + j] = row of coord[j] + delta_k or -1, with
 *     k = (dx+1) + 3*(dy+1) + 9*(dz+1)  (MinkowskiEngine hypercube region, first axis fastest).
 */
#ifndef LINR_HIP_H
#define LINR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LINR_ABI_VERSION 11
#define LINR_API __attribute__((visibility("default")))

#define LINR_EINVAL   (-1)   /* bad argument (null pointer, negative size, unsupported channel count) */
#define LINR_ENOSPC   (-2)   /* workspace / arena / output buffer too small */
#define LINR_EALIGN   (-3)   /* pointer or leading dimension violates the documented alignment */

/* epilogue / mode flags */
#define LINR_RELU      1u    /* out = max(out, 0)                                   (MinkowskiReLU, nn.ReLU) */
#define LINR_ACCUM     2u    /* out += result instead of out = result               (gradient fan-in)        */
#define LINR_RELU_MASK 4u    /* result *= (act > 0): backward of a ReLU whose OUTPUT is `act`                */
#define LINR_NO_BIAS   8u
#define LINR_PAD_ROW  16u    /* the gathered operand has a readable all-zero row at index -1 (in - in_ld):
                                absent neighbours are read from it instead of being branched around      */

LINR_API int linr_abi_version(void);
/* number of float parameters of LINR_PCGC_Model(scale_num, hidden=8, block_layers, outstage=8, instage=1)
 * in parameters() order (models/model_core.py:31-35, models/upsample.py:43-76; block_layers = main.py:521, the
 * Inception layers of block_in's ResNetBlock, 1..4 - the outter blocks always have one, upsample.py:72-76).
 * 54,712 for scale_num = 7, block_layers = 1.  This count and the whole-network entry points (linr_net_*) are the
 * hidden_channel_conv = 8 model (main.py:520 default): their kernels are specialised for 8-wide rows.  Widths 16 / 32 run on
 * the channel-blocked executor of the host mirror (linr_pcgc_amd/wide_net.py), which drives the op-level entries below on
 * 8-wide channel blocks and keeps its own parameter count. */
LINR_API int64_t linr_param_count(int32_t scale_num, int32_t block_layers);

/* ---- kernel map -------------------------------------------------------------------------------------------
 * Replaces MinkowskiEngine's CoordinateManager insert + kernel-map generation that every ME.SparseTensor /
 * MinkowskiConvolution call triggers (models/function_utils.py:13-18,58-69,92-93; models/upsample.py:20-23,
 * 90-97; models/resnet.py:15-51).  coords: int32 [n,3], unique, sorted by the x-major ravel key (the order
 * qscTensor guarantees, models/module_utils.py:246-256), each coordinate in [0, 2^20).
 * Writes nbr[k*ld + row_base + j] = row_base + (index of neighbour) or -1 for j < n.
 * ws: at least linr_kmap_workspace_bytes(n) bytes, 8-byte aligned. */
LINR_API size_t linr_kmap_workspace_bytes(int64_t n);
LINR_API int linr_kmap_build(const int32_t* coords, int64_t n, int32_t* nbr, int64_t ld, int64_t row_base,
                    void* ws, size_t ws_bytes, void* stream);
/* Compressed form of the same map for x-major sorted coordinates: the dz = -1,0,+1 neighbours of a (dx,dy) column are
 * consecutive rows, so lo[q*ld + j] = row of the first present neighbour of column q = (dx+1)+3(dy+1) and bit
 * q*3 + (dz+1) of mask[j] says which are present: 40 B/row instead of 108.  Rows of nbr must hold global row ids. */
LINR_API int linr_kmap_compress(const int32_t* nbr, int64_t nbr_ld, int64_t n, int32_t* lo, uint32_t* mask, int64_t ld,
                       void* stream);
/* The 7-neighbour occupancy features of the scale context (qscTensor.set_offset_tensor, models/module_utils.py:201-224,
 * offsets of glob_params.py:3) read off the kernel map: out[j*7 + i] = 1.0f if neighbour i of voxel row_base + j exists.
 * Replaces 7 QuickSearchCoord.search calls per scale in decoder.decode_one_frame (decoder.py:160-166). */
LINR_API int linr_kmap_offset_feat(const int32_t* nbr, int64_t ld, int64_t row_base, int64_t n, float* out, void* stream);
/* Child occupancy of an octree level (octree_level.forward, models/module_utils.py:86-110; the 8 x [N,1] `occ_lst` of
 * datautils/custom_dataset.py:201-206): child [m,3] int32 sorted x-major and unique, parent [n,3] = the sorted unique
 * floor(child / 2); writes occ[j*8 + 4dx+2dy+dz] = 1.0f iff child 2*parent[j]+(dx,dy,dz) exists.  Same sorted-key binary
 * search as the kernel map; ws: linr_kmap_workspace_bytes(m) bytes, 8-byte aligned. */
LINR_API int linr_octree_occupancy(const int32_t* child, int64_t m, const int32_t* parent, int64_t n, float* occ, void* ws,
                          size_t ws_bytes, void* stream);
/* Sorted unique coordinate list, optionally of the parents: the torch.unique(dim=0) of datautils/custom_dataset.py:271-282 (the
 * input cloud: shift 0) and of octree_level.forward (models/module_utils.py:92,103: parent = unique(floor(child / 2)): shift 1) as one
 * call - compact x-major keys of ((coords - origin) >> shift), radix sort over 3 (coord_bits - shift) bits, unique, decode.  coords:
 * int32 [n,3] with (coords - origin) in [0, 2^coord_bits), coord_bits <= 20, any order, duplicates allowed; origin: DEVICE int32 [3]
 * or NULL (zero); out: int32 [n,3] (room for n rows); *count: DEVICE int64 =
 * number of distinct rows written.  ws: at least linr_sort_unique_workspace_bytes(n) bytes, 256-byte aligned. */
LINR_API size_t linr_sort_unique_workspace_bytes(int64_t n);
LINR_API int linr_coords_sort_unique(const int32_t* coords, int64_t n, const int32_t* origin, int32_t shift, int32_t coord_bits, int32_t* out,
                                     int64_t* count, void* ws, size_t ws_bytes, void* stream);
/* Per-axis minimum / maximum of a coordinate list (the coord_data_min of custom_dataset.py:276-279 and the span that fixes the
 * octree depth): coords int32 [n,3], n >= 1; out: DEVICE int32 [6] = min x, y, z, max x, y, z. */
LINR_API int linr_coords_minmax(const int32_t* coords, int64_t n, int32_t* out, void* stream);
/* One octree level as one call (octree_level.forward, models/module_utils.py:86-110: parents AND their child occupancy; the dataset
 * calls it once per scale, datautils/custom_dataset.py:289-344): child int32 [m,3] sorted x-major and unique, coordinates in
 * [0, 2^coord_bits); parent [m,3] and occ [m,8] have room for m rows, *count (DEVICE int64) = the number of parents written.
 * ws: linr_sort_unique_workspace_bytes(m) bytes, 256-byte aligned. */
LINR_API int linr_octree_level(const int32_t* child, int64_t m, int32_t coord_bits, int32_t* parent, float* occ, int64_t* count, void* ws,
                               size_t ws_bytes, void* stream);
/* ALL octree levels of a frame as one call, without a sort (csrc/octree.hip; the loop of datautils/custom_dataset.py:289-344 over
 * octree_level.forward, models/module_utils.py:86-110): the parents of a sorted unique child list are the set bits of a bitmap over
 * their compact x-major keys, read in word order.  child: int32 [m,3] sorted x-major and unique, coordinates in [0, 2^coord_bits),
 * 2 <= coord_bits <= 11 (12-bit clouds and deeper: linr_octree_level per level); m_dev: NULL, or a DEVICE int64 holding the live row
 * count (<= m: the count linr_coords_sort_unique left on the device - no host read between the two calls).  Level l = 0 ..
 * linr_octree_levels_count(coord_bits, max_levels) - 1 has child coordinates of coord_bits - l bits; its parents (int32 rows) and their
 * child occupancy (float32 [.,8], column 4 dx + 2 dy + dz) are written BACK TO BACK into parents / occ, level after level, each
 * buffer with room for linr_octree_levels_rows(m, coord_bits, max_levels) rows; counts: DEVICE int64 [levels] = rows of every level
 * (the caller reads them once, behind the call, and slices).  Bit-identical to linr_octree_level applied level by level.
 * ws: linr_octree_levels_workspace_bytes(m, coord_bits, max_levels) bytes, 256-byte aligned. */
LINR_API int32_t linr_octree_levels_count(int32_t coord_bits, int32_t max_levels);
LINR_API int64_t linr_octree_levels_rows(int64_t m, int32_t coord_bits, int32_t max_levels);
LINR_API size_t linr_octree_levels_workspace_bytes(int64_t m, int32_t coord_bits, int32_t max_levels);
LINR_API int linr_octree_levels(const int32_t* child, int64_t m, const int64_t* m_dev, int32_t coord_bits, int32_t max_levels,
                                int32_t* parents, float* occ, int64_t* counts, void* ws, size_t ws_bytes, void* stream);
/* sets *bad (device int32, pre-zeroed by the caller) to non-zero if coords are not sorted/unique/in range */
LINR_API int linr_kmap_validate(const int32_t* coords, int64_t n, int32_t* bad, void* stream);

/* ---- sparse 3x3x3 convolution on a fixed coordinate set ------------------------------------------------------
 * Replaces ME.MinkowskiConvolution(kernel_size=3, stride=1, bias=True).forward and its autograd backward
 * (call sites: models/upsample.py:17-23,90-97,153,171,210; models/resnet.py:15-51,56-57).
 * W: [27][cin][cout] (the ME `.kernel` tensor), bias: [cout] (`.bias` is [1,cout]).
 * fwd:        out[j, :cout] = bias + sum_k in[nbr[k][j], :cin] @ W[k]  (+ res[j, :cout]) (ReLU)
 * bwd_data:   gin[i, :cin] (+)= sum_k gout[nbr[26-k][i], :cout] @ W[k]^T   (* (act[i] > 0))
 * bwd_weight: gW[k] (+)= sum_j in[nbr[k][j]]^T gout[j] ; gb (+)= sum_j gout[j]   (deterministic two-pass)
 * cin in 1..8, cout in {4, 8}.  `res`/`act` may be NULL when their flag is absent. */
LINR_API int linr_spconv_fwd(const float* in, int32_t in_ld, const int32_t* nbr, int64_t nbr_ld, int64_t n,
                    const float* W, const float* bias, int32_t cin, int32_t cout,
                    const float* res, int32_t res_ld, float* out, int32_t out_ld, uint32_t flags, void* stream);
LINR_API int linr_spconv_bwd_data(const float* gout, int32_t gout_ld, const int32_t* nbr, int64_t nbr_ld, int64_t n,
                         const float* W, int32_t cin, int32_t cout,
                         const float* act, int32_t act_ld, float* gin, int32_t gin_ld, uint32_t flags,
                         void* stream);
LINR_API size_t linr_spconv_bwd_weight_workspace_bytes(int64_t n, int32_t cin, int32_t cout);
LINR_API int linr_spconv_bwd_weight(const float* in, int32_t in_ld, const float* gout, int32_t gout_ld,
                           const int32_t* nbr, int64_t nbr_ld, int64_t n, int32_t cin, int32_t cout,
                           float* gW, float* gb, uint32_t flags, void* ws, size_t ws_bytes, void* stream);

/* The same convolution on the COMPRESSED kernel map (linr_kmap_compress) and the matrix cores
 * (v_mfma_f32_4x4x1_16b_f32 with broadcast weights: 64 rows x 4 output channels x 1 input channel per instruction, no
 * padding; K = 1 keeps every product a single-rounding fmaf in the order of the conv family (the three dz taps of a (dx,dy)
 * column back to back, column by column inside an x-slab - LINR_TAP in csrc/common.h; channel ascending), so results are
 * bit-identical to linr_spconv_fwd / _bwd_data).  This is what the network executor launches.  Requirements: `in`
 * 16-byte aligned with in_ld in {4, 8} and a zero row at index -1 (LINR_PAD_ROW contract), cout (fwd) / cin (bwd) in {4, 8}.
 * bwd != 0 selects backward-data with `in` = output gradient, `out` = input gradient. */
LINR_API int linr_spconv_cmap(int32_t bwd, const float* in, int32_t in_ld, const int32_t* lo, const uint32_t* mask,
                     int64_t ld, int64_t n, const float* W, const float* bias, int32_t cin, int32_t cout,
                     const float* res, int32_t res_ld, const float* act, int32_t act_ld, float* out, int32_t out_ld,
                     uint32_t flags, void* stream);
/* The same convolution for channel-BLOCKED activations wider than 8 (--hidden_channel_conv 16 / 32, main.py:520; models/upsample.py:
 * 38-76, models/resnet.py:12-51): a C-wide matrix is C / 8 blocks [rows][8], each with its zero row in front.  ONE gather of all
 * input blocks per tap feeds every output channel (csrc/wide.hip).  in_h / out_h / res_h / act_h: HOST arrays of device pointers to
 * the blocks.  fwd: gathers ceil(cin / 8) blocks (cin < 8: channels >= cin of the one block are ignored), produces cout / 8 blocks;
 * bwd: gathers the output gradient's cout / 8 blocks at the mirrored taps, produces cin / 8 blocks.  cin, cout <= 32 (8, 16, 32;
 * cin < 8 forward only).  W [27][cin][cout], bias [cout] or NULL; flags: LINR_RELU, LINR_ACCUM, LINR_RELU_MASK. */
LINR_API int linr_spconv_wide(int32_t bwd, const float* const* in_h, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                              const float* W, const float* bias, int32_t cin, int32_t cout, const float* const* res_h,
                              const float* const* act_h, float* const* out_h, uint32_t flags, void* stream);
/* ... with a pointwise layer of the wide Inception layer (models/resnet.py:55-60) fused into the epilogue - pw->mode:
 *   1 forward conv0_0 (C -> h = C / 2): out2 = relu(in @ W10 + b10) of the row itself (conv1_0)
 *   2 forward conv1_1 (h -> h): out2 = (the produced row) @ W12 + b12 + aux (conv1_2 and the residual's upper half)
 *   3 backward of the block's tail convolution (C <- C): out2 = ((produced upper half) @ W12^T) * (aux > 0), aux = M
 *   4 backward of conv0_0 (C <- h): produced += aux @ W10^T in front of the ReLU mask, aux = the gradient of H1
 * W [cin_pw][cout_pw] (ME layout), b or NULL; aux_h / out2_h: HOST arrays of h / 8 block pointers; C in {16, 32}.  The fmaf chains are
 * those of linr_linear_wide: the fused and the two-launch forms give the same bits. */
typedef struct {
    int32_t mode;
    const float* W;
    const float* b;
    const float* const* aux_h;
    float* const* out2_h;
} linr_wide_pw;
LINR_API int linr_spconv_wide_pw(int32_t bwd, const float* const* in_h, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                                 const float* W, const float* bias, int32_t cin, int32_t cout, const float* const* res_h,
                                 const float* const* act_h, float* const* out_h, uint32_t flags, const linr_wide_pw* pw, void* stream);
/* Weight gradient of that convolution: gW [27][cin][cout], gb [cout] (may be NULL) from the ceil(cin / 8) input blocks in_h (zero row in
 * front) and the cout / 8 output-gradient blocks g_h: one launch whose groups are the input blocks - the gathered, LDS-transposed rows
 * of a block multiply the tiles of ALL gradient blocks (cout 8 / 16 / 32 with the tiled table; else block pair by block pair) - into
 * `slab` (linr_spconv_wgrad_wide_slab_bytes(cin, cout) bytes), then ONE fixed-order reduction into the dense tensors.
 * nbr / tile8t: the frame's kernel map [27][ld] and its tiled copy (linr_kmap_tile8t). */
LINR_API size_t linr_spconv_wgrad_wide_slab_bytes(int32_t cin, int32_t cout);
LINR_API int linr_spconv_wgrad_wide(const float* const* in_h, int32_t cin, const float* const* g_h, int32_t cout, const int32_t* nbr,
                                    const int32_t* tile8t, int64_t ld, int64_t n, float* slab, float* gW, float* gb, void* stream);

/* Pointwise layers of the wide network on blocked activations (MinkowskiConvolution kernel_size 1: models/resnet.py:25-46 conv1_0,
 * conv1_2; the head's nn.Linear(C, 24): models/upsample.py:73-76) as ONE launch: out = [ReLU]([mask]((bias + in @ W) + res + old)).
 * in_h / out_h / res_h / act_h: HOST arrays of device pointers - blocked (channels / 8 matrices [n][8]) or, with *_blocked = 0, ONE
 * dense [n][channels] matrix (the head's 24 hidden units); res / act are laid out like out.  Weight element (ci, co) at
 * W[ci * ws_ci + co * ws_co] of a DENSE matrix: (ws_ci, ws_co) = (cout, 1) (ME layout) or (1, cin) (torch layout), nothing else;
 * backward-data = the same call with the roles of cin / cout and the two strides swapped (as linr_linear_bwd_data).  Shapes: blocked -> blocked with cin, cout in {8, 16, 32}; blocked (16, 32)
 * -> dense 24 and dense 24 -> blocked (16, 32).  flags: LINR_RELU, LINR_ACCUM, LINR_RELU_MASK, LINR_NO_BIAS. */
LINR_API int linr_linear_wide(const float* const* in_h, int32_t cin, int32_t in_blocked, const float* W, int32_t ws_ci, int32_t ws_co,
                              const float* bias, int32_t cout, int32_t out_blocked, const float* const* res_h, const float* const* act_h,
                              float* const* out_h, int64_t n, uint32_t flags, void* stream);
/* Their weight gradient: gW(ci, co) = sum_r x[r][ci] g[r][co] written at gW[ci * ws_ci + co * ws_co], gb[co] = sum_r g[r][co] (NULL:
 * skipped); LINR_ACCUM adds to the destinations.  All (input piece, gradient piece) pairs in one grouped launch + one fixed-order
 * reduction; ws: linr_linear_wgrad_wide_workspace_bytes(n, cin, cout) bytes.  Dense sides: up to 31 channels. */
LINR_API size_t linr_linear_wgrad_wide_workspace_bytes(int64_t n, int32_t cin, int32_t cout);
LINR_API int linr_linear_wgrad_wide(const float* const* in_h, int32_t cin, int32_t in_blocked, const float* const* g_h, int32_t cout,
                                    int32_t g_blocked, int64_t n, float* gW, int32_t ws_ci, int32_t ws_co, float* gb, uint32_t flags,
                                    void* ws, size_t ws_bytes, void* stream);

/* Deferred reductions: with gW = NULL linr_spconv_wgrad_wide / linr_linear_wgrad_wide only write their per-block partials (slab / ws stay
 * in use); linr_wide_reduce_many then sums up to any number of them, 32 per launch, in the same fixed order as the entries' own
 * reductions.  kind 0: a convolution (nblocks = linr_spconv_wgrad_wide_blocks(cout, tiled table given), gW [27][cin][cout], gb [cout] or
 * NULL; ws_ci > 0: the slab's row stride in floats when several convolutions share the rows); kind 1: a pointwise layer (nblocks = linr_linear_wgrad_wide_blocks(n), gW at the strides ws_ci / ws_co, gb or NULL). */
typedef struct {
    int32_t kind, nblocks, cin, cout, ws_ci, ws_co;
    const float* slab;
    float* gW;
    float* gb;
} linr_wide_reduce;
LINR_API int32_t linr_spconv_wgrad_wide_blocks(int32_t cout, int32_t tiled);
/* The weight gradients of TWO convolutions of the same shape h -> h (h in {8, 16}: conv0_1 and conv1_1 of a wide Inception layer) as one
 * launch - partials only: a slab row (2 (h / 8)^2 x 1736 floats; 256 rows) holds A's block pairs, then B's; reduce each with
 * linr_wide_reduce_many: kind 0, slab = the start of its part, nblocks 256, ws_ci = the row stride in floats. */
LINR_API int linr_spconv_wgrad_wide2(const float* const* inA_h, const float* const* gA_h, const float* const* inB_h, const float* const* gB_h,
                                     int32_t h, const int32_t* tile8t, int64_t n, float* slab, void* stream);
LINR_API int32_t linr_linear_wgrad_wide_blocks(int64_t n);
LINR_API int linr_wide_reduce_many(const linr_wide_reduce* items_h, int32_t count, void* stream);

/* The occupancy head of the wide network behind the prune convolution (CNP.basic_module, models/upsample.py:137-161):
 * p = sigmoid(Linear(24, 1)(ReLU(Linear(C, 24)(c)))) and the stage's bits (model_core.py:72-81) in one launch.  c_h: HOST array of the
 * C / 8 blocks [n][8] (C = 16 / 32); w1 [24][C], b1 [24], w2 [24], b2 [1] (torch layouts); target: the occupancy column (stride
 * target_ld) or NULL; p [n]; bits_acc (double[1], += bits) or NULL; ws: linr_head_wide_workspace_bytes(n) bytes when bits are wanted. */
LINR_API size_t linr_head_wide_workspace_bytes(int64_t n);
/* (with bits_acc = NULL but target and ws given the stage's per-block partials stay in ws - linr_head_wide_workspace_bytes(n) / 8 doubles -
 * and linr_bits_finish adds the partials of any number of stages to bits_acc in one launch) */
LINR_API int linr_bits_finish(const double* partial, int64_t count, double* bits_acc, void* stream);
LINR_API int linr_head_wide_fwd(const float* const* c_h, int32_t C, const float* w1, const float* b1, const float* w2, const float* b2,
                                const float* target, int32_t target_ld, int64_t n, float* p, double* bits_acc, void* ws, size_t ws_bytes,
                                void* stream);
/* Backward of nstages (<= 8) such heads in ONE grouped launch: gc = d (gscale * bits) / d c per stage and the parameter gradients
 * grads[nstages][W1 (24 x C) | b1 (24) | w2 (24) | b2 (1)] (written, not accumulated; the inner_mlps of consecutive stages are
 * consecutive in the reference's parameter order).  c_h / gc_h: HOST arrays [nstages][C / 8] of block pointers (stage-major);
 * p_h / target_h / w1_h / b1_h / w2_h: [nstages].  slab: linr_head_wide_bwd_slab_bytes(C, nstages) bytes of scratch. */
LINR_API size_t linr_head_wide_bwd_slab_bytes(int32_t C, int32_t nstages);
LINR_API int linr_head_wide_bwd(const float* const* c_h, const float* const* p_h, const float* const* target_h, int32_t target_ld,
                                const float* const* w1_h, const float* const* b1_h, const float* const* w2_h, int32_t C, int32_t nstages,
                                float gscale, float* const* gc_h, int64_t n, float* slab, size_t slab_bytes, float* grads, void* stream);

/* Backward-weight of the same convolution as a stand-alone kernel (the executor uses it for the first convolutions of the outter
 * blocks, whose inputs need no gradient, and for the schedules without the fused backward below): lane = (offset, channel quad),
 * gathered quad x broadcast gradient tile on v_mfma_f32_4x4x1, row order fixed => reproducible.
 * Writes linr_spconv_wgrad_cmap_blocks() (= 512) per-block partials: slab[b][(27 cin + 1) cout], kernel gradient
 * [27][cin][cout] first, bias gradient [cout] last; their ascending sum over b is MinkowskiConvolution's kernel / bias
 * gradient (ME autograd of the call sites above).  `in`: [n][8] floats, 16-byte aligned, zero row at index -1.
 * tile8t: linr_kmap_tile8t's table or NULL.  With it the gathers are laid out like the convolutions' - lane = (tap of 4, row of
 * 8, quad): four 256-byte runs per instruction - into a wave-private LDS image that every (tap, quad) lane reads back
 * transposed (spconv_wgrad_t_k, the executor's choice); without it the indices come from nbr[27][ld] and every lane gathers its
 * own (tap, quad) (spconv_wgrad_mfma_k).  Same partial sums, bit for bit. */
LINR_API int64_t linr_spconv_wgrad_cmap_blocks(void);
LINR_API int linr_spconv_wgrad_cmap(const float* in, int32_t in_ld, const float* gout, int32_t gout_ld, const int32_t* nbr,
                           const int32_t* tile8t, int64_t ld, int64_t n, int32_t cin, int32_t cout, float* slab, void* stream);
/* Fixed-order sum over the rows of a [nblocks][elems] slab of per-block partials (the form linr_spconv_wgrad_cmap, _wgrad_dual44,
 * _bwd_fused, linr_occ_wgrad7 ... return): elements [0, split) to dstA, [split, elems) to dstB (either may be NULL);
 * LINR_ACCUM adds to the destination.  16 threads per element, slices added in order: bit-reproducible. */
LINR_API int linr_slab_reduce(const float* slab, int32_t nblocks, int32_t elems, int32_t split, float* dstA, float* dstB,
                     uint32_t flags, void* stream);
/* The tiled copy of the kernel map in the lane order of the transposing kernel: tile8t[g][t][u][j] = nbr[tap(4 j + t)][8 g + u]
 * (-1 for position 27, j = 7, beyond n), tap(p) = p / 9 + 3 * ((p / 3) % 3) + 9 * (p % 3): the conv family's slab-major tap
 * sequence, so that the four taps of one gather instruction are neighbours in memory.  tile8t: linr_kmap_tile8t_bytes(n) bytes,
 * 16-byte aligned. */
LINR_API size_t linr_kmap_tile8t_bytes(int64_t n);
LINR_API int linr_kmap_tile8t(const int32_t* nbr, int64_t ld, int64_t n, int32_t* tile8t, size_t tile8t_bytes, void* stream);
/* Backward of a convolution 8 -> 8 from ONE gather of the output gradient (csrc/fused_bwd.hip; what the executor launches for
 * the prune convolutions, the blocks' tail convolutions and block_in's first convolution, i.e. the autograd nodes of
 * upsample.py:20-23,88-97):  gin = backward-data of linr_spconv_cmap (bit-identical to it) and per-block partials of the
 * kernel / bias gradient, slab[b][27 * 64 + 8], b < nblocks (every row is written; their ascending sum over b is
 * MinkowskiConvolution's kernel / bias gradient).  gW[k] = sum_i in[i]^T gout[nbr(i, 26 - k)] - the rows backward-data gathers
 * anyway - is accumulated from a wave-private LDS image of the gathered rows.  gout: [n][8], 16-byte aligned, zero row at
 * index -1; in: [n][8] (the convolution's input); gin: [n][8], 16-byte aligned. */
LINR_API int linr_spconv_bwd_fused(const float* gout, const float* in, const int32_t* lo, const uint32_t* mask, int64_t ld,
                          int64_t n, const float* W, float* gin, float* slab, int32_t nblocks, void* stream);
/* Measurement aid for bench.py's roofline: while enabled, the launches of a training step inside linr_net_forward /
 * _backward / _train_step are bracketed by HIP event pairs on the stream they are launched on, by kernel class:
 *   0 fused backward 8->8 (conv_bwd_wgrad_k<0>)   1 conv 8->8 forward, plain epilogue   2 fused backward of the two 4->4 convs
 *   3 fused backward of conv0_0 8->4              4 prune conv + head forward           5 conv0_0 | conv1_0 forward
 *   6 both 4->4 convs forward                     7 shared occupancy conv               8 head backward
 *   9 first-conv / stand-alone weight gradients  10 pointwise weight gradients         11 scale context (forward, backward)
 *  12 sums, reduction, Adam                      13 stand-alone backward-data convolutions (schedules without the fused backward)
 * and of the bf16 training executor (linr_net_train_step_bf16; 14..16 are poison-only classes, see linr_debug_poison):
 *  17 fused backward 8->8 (bbwd_k<0>)            18 fused backward of the two 4->4 convs 19 fused backward of conv0_0 8->4
 *  20 forward convolutions (bconv_k)             21 head backward                       22 first-conv weight gradients
 *  23 scale context forward, sums, conversions, Adam
 * linr_prof_mask selects the classes that are recorded (default: 0 and 1; an event pair costs a few microseconds of stream
 * time).  linr_prof_read waits for the recorded events and returns their summed elapsed time, the number of launches and the
 * number of row passes (a grouped launch over g layers counts g).  mode 1 = clear the records and start, 2 = resume,
 * 0 = stop (records are kept).  Mutex-guarded; 4096 launches in total. */
#define LINR_PROF_KINDS 24
LINR_API int linr_prof_mask(uint32_t mask);
/* Test hook (tests/test_gpu_parity.py::test_results_do_not_depend_on_leftover_onchip_state): every launch of
 * linr_net_forward / _backward / _train_step whose kernel class (the list above) has its bit set in kind_mask is preceded by a kernel that fills the LDS of every CU with 0xFFFFFFFF (a NaN
 * pattern) - results must not change: a kernel may not read on-chip state it did not write (on a GPU shared with another
 * process the leftovers are not one's own finite numbers). */
LINR_API int linr_debug_poison(uint32_t kind_mask);          /* bits 0..13: the classes above; 14: bf16 executor; 15: linr_decode_scale; 16: the entries of csrc/wide.hip */
LINR_API int linr_debug_poison_now(void* stream);           /* the same poisoning, once, on `stream` (in front of an op-level call) */
LINR_API int linr_prof_enable(int32_t mode);
LINR_API int linr_prof_read(int32_t kind, double* total_ms, int64_t* launches, int64_t* passes);

/* ---- pointwise layers ---------------------------------------------------------------------------------------
 * Replaces ME.MinkowskiConvolution(kernel_size=1) (models/resnet.py:31-37,45-51) and nn.Linear inside
 * PointwiseMLP (models/module_utils.py:42-81).  Element (ci,co) of the weight is W[ci*ws_ci + co*ws_co]:
 * ME kernel [cin][cout] -> (cout, 1); nn.Linear weight [cout][cin] -> (1, cin).
 * fwd: out = in @ W + bias (+res)(ReLU); bwd_data: gin (+)= gout @ W^T (*mask); bwd_weight: gW, gb. */
LINR_API int linr_linear_fwd(const float* in, int32_t in_ld, int64_t n, const float* W, int32_t ws_ci, int32_t ws_co,
                    const float* bias, int32_t cin, int32_t cout, const float* res, int32_t res_ld,
                    float* out, int32_t out_ld, uint32_t flags, void* stream);
LINR_API int linr_linear_bwd_data(const float* gout, int32_t gout_ld, int64_t n, const float* W, int32_t ws_ci,
                         int32_t ws_co, int32_t cin, int32_t cout, const float* act, int32_t act_ld,
                         float* gin, int32_t gin_ld, uint32_t flags, void* stream);
LINR_API size_t linr_linear_bwd_weight_workspace_bytes(int64_t n, int32_t cin, int32_t cout);
LINR_API int linr_linear_bwd_weight(const float* in, int32_t in_ld, const float* gout, int32_t gout_ld, int64_t n,
                           int32_t cin, int32_t cout, float* gW, int32_t ws_ci, int32_t ws_co, float* gb,
                           uint32_t flags, void* ws, size_t ws_bytes, void* stream);
/* dst[i] (+)= src[i] over n floats: the residual / gradient fan-in adds of the width-generic executor (linr_pcgc_amd/wide_net.py);
 * torch's `x + y` in models/resnet.py:59,161 and autograd's accumulation. */
LINR_API int linr_axpy(const float* src, int64_t n, float* dst, int32_t accumulate, void* stream);
/* dst (+)= src[0] + ... + src[count - 1] over n floats (count 1..8, n a multiple of 4, 16-byte aligned pointers; src_h: HOST array),
 * the sources added in list order: a gradient fan-in (e.g. x_glob's: upsample.py:206-214 under autograd) in one pass. */
LINR_API int linr_sum_many(const float* const* src_h, int32_t count, int64_t n, float* dst, int32_t accumulate, void* stream);

/* ---- occupancy head loss -------------------------------------------------------------------------------------
 * Replaces sigmoid + nn.BCELoss(reduction='sum') / ln 2 (models/upsample.py:160, models/model_core.py:14,76-81).
 * fwd: p[j] = sigmoid(z[j]); partial sums of -(t log p + (1-t) log(1-p))/ln2 (logs clamped at -100) are ADDED
 * into bits_acc (double[1], device) deterministically (fixed-order two-pass); t = target[j*target_ld].
 * bwd: gz[j] = gscale * d bits / d z[j] following torch's binary_cross_entropy_backward + sigmoid backward. */
LINR_API size_t linr_bce_workspace_bytes(int64_t n);
LINR_API int linr_bce_bits_fwd(const float* z, const float* target, int32_t target_ld, int64_t n, float* p,
                      double* bits_acc, void* ws, size_t ws_bytes, void* stream);
LINR_API int linr_bce_bits_bwd(const float* p, const float* target, int32_t target_ld, int64_t n, float gscale,
                      float* gz, void* stream);

/* ---- optimiser -----------------------------------------------------------------------------------------------
 * Replaces torch.optim.Adam(...).step() over 189 tensors (main.py:231-237,319) with one launch over the flat
 * parameter buffer.  step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) are computed by the host in
 * double like torch does (hyper-parameters cross the ABI as double and are rounded to float once, like the
 * Python scalars torch passes to its kernels); L2 weight decay is folded into the gradient. */
LINR_API int linr_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                   double step_size, double bc2_sqrt, double beta1, double beta2, double eps, double weight_decay,
                   void* stream);

/* ---- whole-network executor ----------------------------------------------------------------------------------
 * One frame = all scales of one point cloud concatenated into a single row space (finest first); CNP weights are
 * shared by all scales, only the scale-context MLP differs (models/model_core.py:31-35).
 * Replaces LINR_PCGC_Model.logic_core/forward (models/model_core.py:38-81) + CNP.forward
 * (models/upsample.py:163-217) and the autograd backward that main.py:315-316 runs. */
#define LINR_FRAME_OCC_PADDED 1
typedef struct linr_frame {
    int64_t rows;                 /* total rows over all scales, < 2^27 - 1 (32-bit byte offsets of the gathers)  */
    int32_t n_scales;             /* scales present in this frame (<= model scale_num)                      */
    int32_t model_scale_num;      /* LINR_PCGC_Model scale_num (fixes the parameter layout)                 */
    int32_t block_layers;         /* Inception layers of block_in (main.py:521); 0 is read as 1              */
    int32_t flags;                /* 0, or LINR_FRAME_OCC_PADDED: `occ` is row 0 of a [rows + 1][8] buffer whose row -1 is all
                                   * zero - the executor then gathers the occupancy in place instead of keeping a padded
                                   * copy in its arena (one device copy per call less) */
    const int64_t* row_off_h;     /* HOST [n_scales+1] first row of each scale                              */
    const int32_t* scale_idx_h;   /* HOST [n_scales]  which scale embedding / scale MLP each scale uses     */
    const int32_t* nbr;           /* [27][nbr_ld] kernel map with global row ids                            */
    int64_t nbr_ld;               /* leading dimension of nbr (>= rows; a multiple of 4 enables 16-byte index loads) */
    const int32_t* nbr_lo;        /* [9][nbr_ld]  compressed kernel map (linr_kmap_compress), or NULL            */
    const uint32_t* nbr_mask;     /* [nbr_ld]     27-bit presence masks of the compressed map, or NULL           */
    const float*   offset_feat;   /* [rows][7]  7-neighbour occupancy (qscTensor.set_offset_tensor)         */
    const float*   occ;           /* [rows][8]  child occupancy ground truth (occ_lst concatenated)         */
    const int32_t* nbr8t;         /* linr_kmap_tile8t over nbr (the tiled copy in gather-lane order), or NULL             */
} linr_frame;

LINR_API size_t linr_net_arena_bytes(int64_t rows, int32_t block_layers);
/* stages [stage_begin, stage_end) of the 8-stage head; stage_begin == 0 also runs scale context + block_in.
 * The encoder calls (0, 8); the decoder calls (k, k+1) after writing decoded occupancy column k-1 into
 * frame->occ, which executes the identical launches => bitwise identical probabilities (models/upsample.py:249-295).
 * probs: [8][rows] (stage-major) or NULL.  bits_acc: double[1], accumulated into (caller zeroes). */
LINR_API int linr_net_forward(const linr_frame* f, const float* params, float* arena, size_t arena_bytes,
                     int32_t stage_begin, int32_t stage_end, float* probs, double* bits_acc, void* stream);
/* gradient of gscale * bits w.r.t. all parameters, ADDED into grads (flat, linr_param_count floats);
 * requires a preceding linr_net_forward(…, 0, 8, …) on the same arena. */
LINR_API int linr_net_backward(const linr_frame* f, const float* params, float* arena, size_t arena_bytes,
                      float gscale, float* grads, void* stream);

/* One iteration of main.py:305-321 in a single call: forward (bits added into bits_acc), backward of
 * gscale * bits (gscale = 1/point_num), deterministic gradient reduction and the fused Adam update of `params`
 * (torch.optim.Adam with L2 weight decay; `step` = this update's 1-based count, bias corrections computed in double).
 * scale_steps_h: HOST [model_scale_num] update counts of the per-scale context MLPs INCLUDING this update, or NULL.  With it the
 * update follows torch.optim.Adam (the reference pins torch 1.13.1, enviroment.yaml:30): an MLP whose count is 0 - no frame so far
 * contained its scale (custom_dataset.py:325), its .grad is still None - is skipped entirely; every other MLP is updated with
 * its own step count, with a zero gradient when this frame lacks the scale (optimizer.zero_grad() of main.py:320 leaves zero
 * tensors, so weight decay and moment decay keep acting).  A scale of this frame with count 0 is an error.  NULL updates every
 * parameter with `step`.  Nothing synchronises with the host. */
LINR_API int linr_net_train_step(const linr_frame* f, float* params, float* arena, size_t arena_bytes, float gscale,
                        float* exp_avg, float* exp_avg_sq, double lr, int64_t step, const int64_t* scale_steps_h,
                        double beta1, double beta2, double eps, double weight_decay, double* bits_acc, void* stream);

/* ---- bf16 / uint8-weight inference executor (BASELINE config[4]) ------------------------------------------------
 * The codec codes the geometry with the DE-QUANTISED model (encoder.py:101-103, decoder.py:87): w = q/255*(max-min)+min
 * for the uint8 codes q of quant_uniform2 (model_compression/model_size_est.py:72-91).  This entry takes the codes
 * themselves as the model (`codes`: device uint8 [linr_param_count], parameters() order; min_param / max_param: the two
 * floats of side_info.json) and de-quantises inside the kernels; features are bf16 (16-byte rows), every 3x3x3
 * convolution runs on v_mfma_f32_4x4x4_16b_bf16 with fp32 accumulation, biases / pointwise layers / MLPs stay fp32.
 * Inference only: same stage semantics as linr_net_forward (the encoder calls (0, 8), the decoder (k, k+1) after writing
 * occupancy column k-1 into frame->occ; both run identical per-row arithmetic => bit-identical probabilities).
 * probs: [8][rows] stage-major (required); bits_acc as in linr_net_forward.  Needs the compressed kernel map.
 * arena: linr_net_bf16_arena_bytes(rows, block_layers) bytes, 64-byte aligned; stages of one frame must share it.
 * Tolerance against the fp32 path on the same de-quantised weights (tests/test_gpu_bf16.py): logits |d| <= 5e-2,
 * bits within 1 %. */
LINR_API size_t linr_net_bf16_arena_bytes(int64_t rows, int32_t block_layers);
LINR_API int linr_net_forward_bf16(const linr_frame* f, const uint8_t* codes, float min_param, float max_param, void* arena,
                          size_t arena_bytes, int32_t stage_begin, int32_t stage_end, float* probs, double* bits_acc,
                          void* stream);

/* ---- bf16 training executor (BASELINE config[4]: "bf16 SparseConv"; beside the fp32 step above, never instead of it) --
 * The overfit iteration of main.py:305-321 - forward of models/model_core.py:38-81 / models/upsample.py:88-97,137-217 /
 * models/resnet.py:12-60, the autograd backward of main.py:315-316, Adam of main.py:231-237,319 - with bf16 feature AND gradient
 * rows ([1 + rows][8], 16-byte rows: one gather per tap), fp32 master parameters (`params`, updated by Adam in fp32), fp32
 * accumulation.  ONE rounding rule: every matrix written to memory is rounded to bf16 (RNE) and every consumer sees the stored
 * value; the 3x3x3 kernels are rounded to bf16 inside the kernels; biases, 1x1 convolutions, scale-context and head MLPs, sigmoid
 * and the bits are fp32 (bits accumulate in double).  oracle/network_bf16.py emulates these roundings with autograd; tolerances
 * are in tests/test_gpu_bf16_train.py.  block_layers 1 and the compressed kernel map only.
 * arena: linr_net_train_bf16_arena_bytes(rows, 1) bytes, 64-byte aligned, shared by the calls of one step.
 * occ_bf16: NULL, or the frame's occupancy converted once by linr_occ_to_bf16 ([1 + rows][8] bf16, zero row in front; the
 * pointer passed is that of the ZERO row) - a frame's occupancy does not change over the epochs.
 * linr_net_forward_train_bf16: teacher-forced forward of all 8 stages, every activation the backward needs kept in the arena;
 *   probs [8][rows] stage-major or NULL; bits_acc double[1] (accumulated into) or NULL.
 * linr_net_backward_bf16: gradient of gscale * bits, ADDED into grads (fp32, linr_param_count floats); needs the forward above on
 *   the same arena.
 * linr_net_train_step_bf16: forward + backward + fixed-order reduction + Adam, arguments as linr_net_train_step. */
LINR_API size_t linr_net_train_bf16_arena_bytes(int64_t rows, int32_t block_layers);
LINR_API int linr_occ_to_bf16(const float* occ, int64_t rows, uint16_t* out_padded, void* stream);
LINR_API int linr_net_forward_train_bf16(const linr_frame* f, const float* params, void* arena, size_t arena_bytes,
                                const uint16_t* occ_bf16, float* probs, double* bits_acc, void* stream);
LINR_API int linr_net_backward_bf16(const linr_frame* f, const float* params, void* arena, size_t arena_bytes,
                           const uint16_t* occ_bf16, float gscale, float* grads, void* stream);
LINR_API int linr_net_train_step_bf16(const linr_frame* f, float* params, void* arena, size_t arena_bytes, const uint16_t* occ_bf16,
                             float gscale, float* exp_avg, float* exp_avg_sq, double lr, int64_t step,
                             const int64_t* scale_steps_h, double beta1, double beta2, double eps, double weight_decay,
                             double* bits_acc, void* stream);
/* The backward of ONE convolution 8->8 of that executor as a stand-alone op (ME.MinkowskiConvolution's backward, models/resnet.py:15-51
 * under autograd): gin = bwd-data(gout; W) rounded to bf16 AND the kernel / bias gradient from one gather of gout.  gout / in / gin:
 * bf16 [n][8] whose row -1 exists (gout's must be zero); W fp32 [27][8][8] (rounded to bf16 in-kernel for backward-data);
 * slab: [nblocks][1736] floats = per-block partials [kernel 1728 | bias 8]; *rows_written of its rows are written (sum them). */
LINR_API int linr_spconv_bwd_fused_bf16(const uint16_t* gout, const uint16_t* in, const int32_t* lo, const uint32_t* mask, int64_t ld,
                               int64_t n, const float* W, uint16_t* gin, float* slab, int32_t nblocks, int32_t* rows_written,
                               void* stream);

/* Staged DECODE of one frame object (decoder.decode_one_frame, decoder.py:153-176 + CNP.decode, models/upsample.py:249-295) as
 * ONE call: for stage k = 0..7 { linr_net_forward[_bf16](k, k+1); probabilities of the stage -> pinned host buffer; the range
 * decoder on every scale's stream k (linr_ac_decode_binary, host); decoded symbols -> column k of f->occ } - the loop the
 * reference runs in Python with 16 .cpu() round trips per scale.  The call returns after the last stage (it synchronises
 * the stream once per stage by construction) and does not hold the Python GIL, so a host thread per frame scales.
 * streams_h / stream_len_h: HOST arrays [f->n_scales][8] of the per-stage streams (pack_bitstream payloads) and their byte
 * lengths; f->occ must be writable device memory (the frame's occupancy, zeroed by the caller; LINR_FRAME_OCC_PADDED honoured);
 * probs [8][rows] device scratch (holds all 8 stages' probabilities afterwards); p_pinned [rows] floats and s_pinned [rows]
 * bytes of page-locked host memory; s_dev [rows] bytes of device scratch.  codes == NULL: fp32 executor with `params`;
 * otherwise the bf16 / uint8-weight executor with codes, min_param, max_param (then `arena` is a linr_net_bf16_arena_bytes one). */
LINR_API int linr_net_decode_stages(const linr_frame* f, const float* params, const uint8_t* codes, float min_param,
                           float max_param, void* arena, size_t arena_bytes, const uint8_t* const* streams_h,
                           const int64_t* stream_len_h, float* probs, float* p_pinned, uint8_t* s_pinned, uint8_t* s_dev,
                           void* stream);

/* One scale of the decoder as one call - the body of decoder.decode_one_frame's loop (decoder.py:153-176): kernel map of the
 * level's coordinates, the 7-neighbour features read off it (instead of qscTensor.set_offset_tensor's 7 searches), the 8 decode
 * stages of linr_net_decode_stages, and octree_level.upper_layer (models/module_utils.py:117-127): the coordinates of the next finer
 * level = children 2 p + (dx, dy, dz) of every decoded octant 4 dx + 2 dy + dz, sorted x-major (radix sort of child_bits-bit-per-
 * axis keys).  coord: DEVICE int32 [n][3], sorted x-major, unique; streams_h / stream_len_h: the 8 stage streams of this scale
 * (HOST); ws: DEVICE workspace of linr_decode_scale_ws_bytes(n, block_layers, codes != NULL) bytes, 256-byte aligned; p_pinned /
 * s_pinned: pinned HOST buffers of n floats / n bytes; child_xyz: DEVICE int32 [child_cap][3] (8 n is always enough); *child_n_h
 * receives the number of children.  params (fp32 executor) or codes + min / max (bf16 / uint8-weight executor).  Synchronises the
 * stream.  Between two scales the caller only allocates the next workspace. */
LINR_API size_t linr_decode_scale_ws_bytes(int64_t n, int32_t block_layers, int32_t bf16);
LINR_API int linr_decode_scale(const int32_t* coord, int64_t n, int32_t scale_idx, int32_t model_scale_num, int32_t block_layers,
                      int32_t child_bits, const float* params, const uint8_t* codes, float min_param, float max_param,
                      const uint8_t* const* streams_h, const int64_t* stream_len_h, void* ws, size_t ws_bytes,
                      float* p_pinned, uint8_t* s_pinned, int32_t* child_xyz, int64_t child_cap, int64_t* child_n_h,
                      void* stream);

/* ---- the executor's fused layers as stand-alone ops ------------------------------------------------------------
 * What linr_net_forward / _backward launch for one layer, callable (and testable) on its own.  All of them work on the
 * compressed kernel map (linr_kmap_compress) and follow the LINR_PAD_ROW contract: every matrix that a kernel GATHERS
 * from (named below) needs a readable all-zero row at index -1 and 16-byte aligned rows.
 *
 * Scale context (LINR_PCGC_Model.logic_core, models/model_core.py:48-53): x0[r] = W2 relu(W1 [emb_s | offset_feat[r]] + b1) + b2
 * with the weights of row r's scale; `f` supplies rows, row_off_h, scale_idx_h, model_scale_num, block_layers (parameter
 * layout) and offset_feat.  mix [rows][16], hid [rows][16], x0 [rows][8].  bwd: ghid = (W2^T gx0) * (hid > 0). */
LINR_API int linr_sce_fwd(const float* params, const linr_frame* f, float* mix, float* hid, float* x0, void* stream);
LINR_API int linr_sce_bwd(const float* params, const linr_frame* f, const float* gx0, const float* hid, float* ghid,
                 void* stream);
/* The complete backward of the scale context (model_core.py:48-53 under autograd) as one call: d loss / d (scale_emb, every scale MLP
 * of the frame) into grads[0 .. linr_sce_param_count(model_scale_num)) - these parameters lead the flat parameter order
 * (scale_emb, scale_mlp.{s}.{0,2}) whatever the width of the rest of the network, so `params` / `grads` may be the flat buffers of a
 * hidden_channel_conv 16 / 32 model.  gx0 [rows][8], hid [rows][16] as kept by linr_sce_fwd (whose mix may be NULL); scales the frame
 * does not contain get zero gradients.  slab: linr_sce_bwd_params_slab_bytes(model_scale_num) bytes of scratch. */
LINR_API int64_t linr_sce_param_count(int32_t model_scale_num);
LINR_API size_t linr_sce_bwd_params_slab_bytes(int32_t model_scale_num);
LINR_API int linr_sce_bwd_params(const float* params, const linr_frame* f, const float* gx0, const float* hid, float* slab,
                                 size_t slab_bytes, float* grads, void* stream);

/* Occupancy head of one stage = CNP.basic_module + the stage's BCE term (models/upsample.py:137-161,
 * models/model_core.py:76-81): c = conv3(prior; Wp, bp) (gathers `prior` [n][8]), z = w2 . relu(W1 c + b1) + b2,
 * p = sigmoid(z); if bits_acc != NULL the stage's bits (target = occupancy column, stride target_ld) are ADDED into it
 * (fixed-order block partials in ws).  bwd: gc = d(gscale * bits)/dc and the head-MLP gradients
 * ghead[241] = [gW1 24x8 | gb1 24 | gw2 24 | gb2 1] (parameters() order of inner_mlps.k.0), deterministic.
 * ws: linr_head_workspace_bytes(n) bytes, 16-byte aligned. */
LINR_API size_t linr_head_workspace_bytes(int64_t n);
LINR_API int linr_head_fwd(const float* prior, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                  const float* Wp, const float* bp, const float* w1, const float* b1, const float* w2, const float* b2,
                  const float* target, int32_t target_ld, float* c_out, float* p_out, double* bits_acc, void* ws,
                  size_t ws_bytes, void* stream);
LINR_API int linr_head_bwd(const float* c, const float* p, const float* target, int32_t target_ld, const float* w1,
                  const float* b1, const float* w2, float gscale, float* gc, int64_t n, float* ghead, void* ws,
                  size_t ws_bytes, void* stream);

/* One InceptionResNet layer (models/resnet.py:55-60) in two launches: H = [relu(conv0_0(x)) | relu(conv1_0(x))],
 * M = relu(conv1_1(H[:,4:8])), I = [conv0_1(H[:,0:4]) | conv1_2(M)] + x.  x, H are gathered.  All [n][8] except M [n][4].
 * bwd_data: from gI (gathered) the launches of the backward data chain: gM = (gI[:,4:8] W12^T) * (M > 0) (gathered),
 * gH = [bwd(gI[:,0:4]; W01) | bwd(gM; W11)] * (H > 0) (gathered), gX = bwd(gH[:,0:4]; W00) + gI + gH[:,4:8] W10^T
 * (+ old gX: LINR_ACCUM) (* (x > 0): LINR_RELU_MASK, x = the ReLU output that fed the layer). */
typedef struct linr_inception_params {
    const float *w00, *b00;   /* conv0_0 [27][8][4], [4] */
    const float *w01, *b01;   /* conv0_1 [27][4][4], [4] */
    const float *w10, *b10;   /* conv1_0 [8][4], [4]     */
    const float *w11, *b11;   /* conv1_1 [27][4][4], [4] */
    const float *w12, *b12;   /* conv1_2 [4][4], [4]     */
} linr_inception_params;
LINR_API int linr_inception_fwd(const float* x, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                       const linr_inception_params* q, float* H, float* M, float* I, void* stream);
LINR_API int linr_inception_bwd_data(const float* gI, const float* x, const float* H, const float* M, const int32_t* lo,
                            const uint32_t* mask, int64_t ld, int64_t n, const linr_inception_params* q, float* gM,
                            float* gH, float* gX, uint32_t flags, void* stream);
/* weight gradients of the layer's two 4->4 convolutions in one pass (conv0_1 on H[:,0:4] with gradient g0, conv1_1 on
 * H[:,4:8] with g1): 512 per-block partials slab[b][872] = [gW01 432 | gb01 4 | gW11 432 | gb11 4], summed in ascending b. */
LINR_API int linr_spconv_wgrad_dual44(const float* H, const float* g0, int32_t g0_ld, const float* g1, int32_t g1_ld,
                             const int32_t* nbr, const int32_t* tile8t, int64_t ld, int64_t n, float* slab, void* stream);

/* The backward of an Inception layer's two convolution pairs in the same form (models/resnet.py:55-60; gM comes from the
 * caller: the tail convolution's epilogue or a pointwise op):
 *   gH = [bwd(gI[:, 0:4]; W01) | bwd(gM; W11)] * (H > 0)  with dW01, db01, dW11, db11  (one gather of [gI[:, 0:4] | gM]), then
 *   gX = (bwd(gH[:, 0:4]; W00) + gI (+ old gX: LINR_ACCUM) + gH[:, 4:8] @ W10^T) (* (x > 0): LINR_RELU_MASK)  with dW00, db00 and,
 *        on lanes the 11-tap last chunk leaves idle, dW10 = x^T gH[:, 4:8], db10 of the 1x1 conv1_0.
 * gH and gX are bit-identical to linr_inception_bwd_data's.  slab: [nblocks][1776] per-block partials
 * [W00 864 | b00 4 | W01 432 | b01 4 | W11 432 | b11 4 | W10 32 | b10 4], every row written; sum over the rows = the gradients.
 * conv1_2's gradient (M^T gI[:, 4:8]) is linr_linear_bwd_weight's job.  gI, gH, gX, x, H: [n][8], gM: [n][4], all
 * 16-byte aligned; gI, gM and gH need the zero row at index -1. */
LINR_API int linr_inception_bwd_fused(const float* gI, const float* gM, const float* x, const float* H, const int32_t* lo,
                             const uint32_t* mask, int64_t ld, int64_t n, const linr_inception_params* q, float* gH, float* gX,
                             uint32_t flags, float* slab, int32_t nblocks, void* stream);

/* Weight gradients of the FIRST convolutions of the 7 outter blocks (block b = 1..7: conv3(occ[:, :b] -> 8) on the same occupancy
 * rows, models/upsample.py:206-214; ME: seven MinkowskiConvolution backward-weight calls) from ONE gather of the occupancy rows:
 *   gW_b[k][ci][co] = sum_r occ[nbr(r, k)][ci] * gout7[b - 1][r][co]  (ci < b),   gb_b[co] = sum_r gout7[b - 1][r][co].
 * occ [n][8] with the zero row at index -1; gout7_h: HOST array of 7 device pointers [n][8] (the gradients behind the ReLU);
 * everything 16-byte aligned.  slab: [nblocks][6104] per-block partials, for b = 1..7: kernel [27][b][8] then bias [8]; only the
 * first *rows_written_h (<= nblocks) rows are written - sum over those rows = the gradients. */
LINR_API int linr_occ_wgrad7(const float* occ, const float* const* gout7_h, const int32_t* lo, const uint32_t* mask, int64_t ld,
                    int64_t n, float* slab, int32_t nblocks, int32_t* rows_written_h, void* stream);
/* First convolutions of the 7 outter blocks (models/upsample.py:206-214 -> make_block's first conv + ReLU): block g + 1
 * computes relu(conv3(occ[:, :g+1]; kernel [27][g+1][8]) + bias) on the SAME gathered occupancy rows, so one gather feeds
 * all seven.  occ [n][8] (gathered); kernel / bias of block g at params + w_off_h[g] / b_off_h[g]; result of block g at
 * out + out_off_h[g] ([n][8], element offsets, multiples of 4). */
LINR_API int linr_occ_conv7(const float* occ, const int32_t* lo, const uint32_t* mask, int64_t ld, int64_t n,
                   const float* params, const int64_t* w_off_h, const int64_t* b_off_h, float* out,
                   const int64_t* out_off_h, void* stream);

/* ---- arithmetic-coder feed (host side) -----------------------------------------------------------------------
 * Replaces torchac.encode_float_cdf / decode_float_cdf as used by BinaryArithmeticCoding
 * (models/module_utils.py:8-40; callers models/upsample.py:224-237,275; models/model_core.py:204-208) and by the
 * model stream (model_compression/model_size_est.py:470-482,545-563).  Follows torchac 0.9.3's published
 * coder (un-vendored, pinned in enviroment.yaml:32); the one known-answer vector the reference ships (the model stream of
 * loot/gop_32_62: 282,642 bits) is consistent with this coder in two ways that cannot be told apart offline - 35,319 bytes with
 * the CPU's Laplace pdf plus a 90-bit header of an older revision, or 35,320 bytes plus today's 82-bit header if the CUDA pdf
 * the reference builds its CDF from differs in the last bit (tests/test_oracle_golden.py::test_model_stream_known_answer).
 * Consequence for interop: a model.bin whose CDF was built on another device decodes only if that device's expf agrees to the
 * last bit on the 256 pdf values.  Real torchac is not installable here, so byte-compatibility beyond that vector is by
 * construction, not by test.
 * All pointers here are HOST pointers.  Return: bytes written (>= 0) or a negative LINR_E* code. */
LINR_API int64_t linr_ac_encode_binary(const float* prob_h, const uint8_t* sym_h, int64_t n, uint8_t* out_h, int64_t cap);
LINR_API int     linr_ac_decode_binary(const float* prob_h, int64_t n, const uint8_t* in_h, int64_t in_len, uint8_t* sym_h);
LINR_API int64_t linr_ac_encode_cdf16(const uint16_t* cdf_h, int32_t lp, int32_t cdf_shared, const int16_t* sym_h,
                             int64_t n, uint8_t* out_h, int64_t cap);
LINR_API int     linr_ac_decode_cdf16(const uint16_t* cdf_h, int32_t lp, int32_t cdf_shared, int64_t n,
                             const uint8_t* in_h, int64_t in_len, int16_t* sym_h);
/* n_streams independent binary streams coded on a thread pool (8 stages x scales are independent streams) */
LINR_API int linr_ac_encode_binary_batch(const float* const* prob_h, const uint8_t* const* sym_h, const int64_t* n,
                                int32_t n_streams, uint8_t* const* out_h, const int64_t* cap,
                                int64_t* out_len, int32_t n_threads);

/* ---- frame input (host) ----------------------------------------------------------------------------------------------
 * Body of an ASCII PLY (datautils/custom_dataset.py:9-14 read_ply_o3d - open3d's C++ reader - followed by :263-269, which keeps
 * the rounded x, y, z).  text_h / len: the bytes BEHIND "end_header\n"; n_rows vertices of n_cols whitespace-separated numbers,
 * one vertex per line (empty lines between vertices tolerated); cx, cy, cz: the columns of x, y, z.  xyz_h [n_rows][3] receives
 * the values rounded to the nearest integer (ties to even, like numpy.rint).  Returns 0, or LINR_EINVAL for a short / malformed
 * line, a non-finite coordinate or bad arguments; *rows_parsed_h (optional) = the vertices completed before the error.
 * Host pointers only; no GPU involved; thread-safe (linr_pcgc_amd/ply.py reads the frames of a GOP on a thread pool). */
LINR_API int linr_ply_parse_ascii(const char* text_h, size_t len, int64_t n_rows, int32_t n_cols, int32_t cx, int32_t cy,
                         int32_t cz, int64_t* xyz_h, int64_t* rows_parsed_h);

#ifdef __cplusplus
}
#endif
#endif /* LINR_HIP_H */

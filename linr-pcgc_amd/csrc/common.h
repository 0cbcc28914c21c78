// Internal helpers shared by the gfx950 kernels.  Not part of the C-ABI (include/linr_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/linr_hip.h"

#define LINR_BLOCK 256
#define LINR_WAVE 64

static inline int linr_hip_rc(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }
static inline int linr_launch_rc() { return linr_hip_rc(hipGetLastError()); }
static inline bool linr_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline unsigned linr_grid(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// number of persistent blocks used by the two-pass reductions: enough to fill 256 CUs several times over,
// few enough that the partial slabs stay small.
static inline int linr_reduce_blocks(int64_t n, int rows_per_tile) {
    int64_t tiles = (n + rows_per_tile - 1) / rows_per_tile;
    int64_t nb = tiles < 512 ? tiles : 512;
    return (int)(nb < 1 ? 1 : nb);
}

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

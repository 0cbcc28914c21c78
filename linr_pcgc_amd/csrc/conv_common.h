// Device helpers shared by the compressed-map convolution kernels (csrc/fused.hip, csrc/fused_bwd.hip).  Not part of the C-ABI.
#pragma once
#include "common.h"
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class F, int... Ks>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Ks...>) {
    (f(std::integral_constant<int, Ks>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }


// byte offsets (from the pad row) of the 27 neighbours of `row`; absent -> 0 (the pad row itself).  Rows are 4 << sh bytes
// wide: the scaling is a shift by a wave-uniform amount (v_mul_lo_u32 is a quarter-rate instruction and there would be 27 of
// them per lane), and the nine column bases are read through uniform base pointers + ONE 32-bit row offset (global_load
// saddr form) instead of nine 64-bit address computations.
template <bool BWD>
__device__ __forceinline__ void decode_offsets(const int32_t* __restrict__ lo, const uint32_t* __restrict__ mask,
                                               int64_t ld, int64_t row, uint32_t rowbytes, uint32_t (&off)[27]) {
    const uint32_t sh = __builtin_amdgcn_readfirstlane(rowbytes >= 32u ? 5u : (rowbytes >= 16u ? 4u : (rowbytes >= 4u ? 2u : 0u)));
    const uint32_t r32 = (uint32_t)row;
    if (mask == nullptr) {            // `lo` is the full table nbr[27][ld]: no decode arithmetic, 27 coalesced index loads
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const int32_t* lk = lo + (int64_t)k * ld;
            off[BWD ? 26 - k : k] = (uint32_t)(lk[r32] + 1) << sh;
        }
        return;
    }
    // Byte offsets in 32 bits (9 * ld * 4 < 2^32; maps with ld >= 2^26 take the plain 64-bit path) make every one of the ten
    // loads a saddr-form global_load (the uniform table base in SGPRs + one VGPR offset, one v_add per column) - left to
    // itself hipcc folds the column stride into the lane's 64-bit address and spends two quarter-rate v_mad_u64_u32 per
    // column on it.  Per column:
    // L = (lo + 1) << sh; a tap's offset is L advanced by one row per present tap below it and zeroed when absent:
    //   m_j = -bit_j (v_bfe_i32), o_j = t_j & m_j, t_{j+1} = t_j - m_j * R (one v_mad_i32_i24)   => 9 VALU ops per column
    const uint32_t rb = r32 << 2;
    const uint32_t m = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(mask) + rb);
    const int negR = -(int)(1u << sh);
    const char* lob = reinterpret_cast<const char*>(lo);
    const uint32_t ld4 = (uint32_t)ld << 2;
    const bool small = ld < ((int64_t)1 << 26);            // wave-uniform
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int32_t lv = small ? *reinterpret_cast<const int32_t*>(lob + (rb + (uint32_t)q * ld4)) : (lo + (int64_t)q * ld)[r32];
        const uint32_t L = ((uint32_t)lv + 1u) << sh;             // +1: row index -> offset from the pad row
        const int m0 = __builtin_amdgcn_sbfe(m, 3 * q, 1), m1 = __builtin_amdgcn_sbfe(m, 3 * q + 1, 1),
                  m2 = __builtin_amdgcn_sbfe(m, 3 * q + 2, 1);
        const uint32_t t1 = L + (uint32_t)__mul24(m0, negR);
        const uint32_t t2 = t1 + (uint32_t)__mul24(m1, negR);
        const uint32_t o0 = L & (uint32_t)m0, o1 = t1 & (uint32_t)m1, o2 = t2 & (uint32_t)m2;
        // forward uses offset k, backward-data the mirrored offset 26-k  (k = q + 9*dz)
        if (!BWD) { off[q] = o0; off[q + 9] = o1; off[q + 18] = o2; }
        else      { off[26 - q] = o0; off[26 - (q + 9)] = o1; off[26 - (q + 18)] = o2; }
    }
}

template <int W> struct RowLoadF {
    static __device__ __forceinline__ void run(const char* __restrict__ p, float* x) {
#pragma unroll
        for (int v = 0; v < W / 4; ++v) {
            const float4 t = *reinterpret_cast<const float4*>(p + 16 * v);
            x[4 * v] = t.x; x[4 * v + 1] = t.y; x[4 * v + 2] = t.z; x[4 * v + 3] = t.w;
        }
    }
};


// pointwise side paths of the Inception block fused into conv epilogues (models/resnet.py:55-60):
//   EPI == 2 (fwd 8->4, conv0_0): also out[4:8] = relu(in[row] @ W10 + b10)                    (conv1_0 is a centre tap)
//   EPI == 3 (bwd 8->8, block tail conv): also gM = (gI[4:8] @ W12^T) * (M > 0)                 (backward of conv1_2 + ReLU)
//   EPI == 4 (bwd of conv0_0, gathered 4 -> produced 8): a += res; a += gH[row][4:8] @ W10^T; a *= (A > 0)
struct PwArgs {
    const float* w;       // the 1x1 kernel [cin][cout] (ME layout)
    const float* b;       // its bias (EPI 2) or nullptr
    const float* aux;     // EPI 3: M [n][4];  EPI 4: gH [n][8] (own-row gradient of H)
    float* aux_out;       // EPI 3: gM [n][4]
};


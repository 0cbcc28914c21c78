// Wave-specialised form of conv_bwd_wgrad_single_k (csrc/fused_bwd.hip includes this file) and the kernel the launchers use: the same
// arithmetic in the same order - every input gradient and every slab partial bit-identical (tests/test_gpu_fused_ops.py) - issued from
// TWO instruction streams per SIMD instead of one.
//
// conv_bwd_wgrad_single_k keeps ~400 registers alive, so one wave per SIMD is resident and its single in-order stream has to issue the
// 880 matrix instructions of a tile AND the ~1250 other instructions (index decode, gathers, LDS traffic, waits) that feed them: a late
// gather stalls the matrix pipe and a busy matrix pipe delays the next gather (matrix pipe 53 % busy).  Here a block is FOUR PAIRS of
// waves; the waves w and w + 4 of a pair share a SIMD and a tile:
//   producer (wave w)      index decode, the gather ring, backward-data MFMAs, the LDS image of the gathered rows, the epilogue
//                          of the input gradient (and the 1x1 convolutions riding on it)                      200-230 registers
//   consumer (wave w + 4)  transposed reads of the image, the weight-gradient MFMAs against the rows' own inputs
// Both fit the 256 registers of two waves per SIMD, and the SIMD's arbiter issues one wave's vector / memory instructions under the
// other's MFMAs.  The image stays double-buffered by chunk (8 taps; KIND 2: 16): while the producer gathers chunk c of a tile, the
// consumer multiplies chunk c - 1 - the schedule of conv_bwd_wgrad_single_k, made explicit by ONE block barrier per chunk (s_barrier;
// the producer waits for its LDS writes only, the gathers in flight stay in flight).  Every wave of the block runs the same number
// of chunk phases (waves without a tile only execute the barriers).
// Measured (profiles/r05_split_lab.txt): 8-group launch of the conv 8->8 202 -> 185 us, step 1.62 -> 1.58 ms; the gathers + decode
// alone take 117 us and the 880 fp32 MFMAs have a floor of 121 us, so the two streams overlap them better but not perfectly.  Giving
// ALL matrix instructions to one wave (a pure data mover beside it) was tried and is slower (236 us): one stream cannot keep the pipe
// busy across its LDS reads.
#define FS_THREADS (2 * FB_WAVES * 64)
#ifndef FS_LAB
#define FS_LAB 0                         // kernel-floor experiments (tools/split_lab.sh): 1 no gathers, 2 no weight-gradient MFMAs, 4 no
#endif                                   // backward-data MFMAs, 8 no LDS traffic, 32 no phase barriers

__device__ __forceinline__ void fs_barrier_producer() { if constexpr (!(FS_LAB & 32)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// The consumer's reads of image buffer (pc & 1) must have RETURNED before the producer overwrites that buffer in the next phase.  Every
// such read feeds an MFMA of this phase, and the compiler's lgkmcnt wait sits in front of that MFMA - but MFMAs are not memory
// operations, so the asm's "memory" clobber alone would not stop a scheduler from sinking one (and its wait) below the barrier.  The
// scheduling barrier in FRONT of the asm makes the order structural: nothing moves across it (no change in the emitted code).
__device__ __forceinline__ void fs_barrier_consumer() {
    if constexpr (!(FS_LAB & 32)) {
#ifndef FS_NO_CONSUMER_SB                 // (lab knob: the code as it shipped in round 5, for the same-box A/B of profiles/r06_*)
        __builtin_amdgcn_sched_barrier(0);
#endif
        asm volatile("s_barrier" ::: "memory");
    }
}

template <int KIND, int EPI>
__global__ __launch_bounds__(FS_THREADS, 1) void conv_bwd_wgrad_k(FbArgs a, const int32_t* __restrict__ lo,
                                                                       const uint32_t* __restrict__ mask, int64_t ld, int64_t n,
                                                                       PwArgs pw, LinrWgradDst d, FbDst2 d2, Grp gp) {
    using T = FbT<KIND>;
    constexpr int CT = T::CT, NCH = T::NCH, XN = T::XN, WGM = T::WGM;
    __shared__ float4 smem[FB_WAVES * FB_WAVE_BYTES / 16];
    __shared__ float sbias[FB_WAVES][8];
    __shared__ float s12[FB_WAVES][20];
    {   // group offsets: as in conv_bwd_wgrad_single_k
        const int gi = blockIdx.y;
        a.g += gp.in[gi]; a.xin += gp.res[gi]; a.W += gp.w[gi]; a.out += gp.out[gi];
        if constexpr (KIND == 1) { a.g1 += gp.e5[gi]; a.W1 += gp.e6[gi]; d2.w_off1 += gp.e0[gi]; d2.b_off1 += gp.e1[gi]; }
        if constexpr (KIND == 2) { if (a.res) a.res += gp.act[gi]; pw.w += gp.e0[gi]; d2.w_off1 += gp.e1[gi]; d2.b_off1 += gp.e2[gi]; }
        if constexpr (EPI == 3) { pw.w += gp.e0[gi]; pw.aux += gp.e1[gi]; pw.aux_out += gp.e2[gi]; d2.w_off1 += gp.e5[gi]; d2.b_off1 += gp.e6[gi]; }
        d.w_off += gp.e3[gi]; d.b_off += gp.e4[gi];
    }
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave8 & 3, role = wave8 >> 2;
    char* img = reinterpret_cast<char*>(smem) + pair * FB_WAVE_BYTES;
    const int64_t T64 = (n + 63) >> 6;
    const int64_t tb0 = (int64_t)blockIdx.x * (FB_WAVES * a.tiles_per_wave);
    const int64_t tb1 = (tb0 + FB_WAVES * a.tiles_per_wave < T64) ? tb0 + FB_WAVES * a.tiles_per_wave : T64;
    const int iters = a.tiles_per_wave;
    float* dst = d.base + (int64_t)blockIdx.x * d.block_stride;
    if (role == 0) {
        // the consumer's first chunk phase multiplies "the last chunk of the previous tile" with zero inputs: finite numbers there
        for (int o = lane * 16; o < FB_BUF; o += 64 * 16)
            *reinterpret_cast<float4*>(img + ((NCH - 1) & 1) * FB_BUF + o) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (role == 0) {
        // ================================================ producer ================================================================
#ifdef FS_PRIO_P
        __builtin_amdgcn_s_setprio(FS_PRIO_P);
#endif
        constexpr int WI = KIND == 0 ? 8 : 4;
        float wv[4][WI];                        // backward-data weights as A-operand images (conv_bwd_wgrad_single_k)
        {
            const int blk = lane >> 2, j = lane & 3;
            const int kl = blk >> 1, co = 4 * (blk & 1) + j;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k = g * 8 + kl;
#pragma unroll
                for (int i = 0; i < WI; ++i) {
                    float w = 0.0f;
                    if (k < 27) {
                        if constexpr (KIND == 0) w = a.W[(k * 8 + co) * 8 + i];
                        if constexpr (KIND == 1) w = ((blk & 1) ? a.W1 : a.W)[(k * 4 + j) * 4 + i];
                        if constexpr (KIND == 2) w = a.W[(k * 8 + co) * 4 + i];
                    }
                    wv[g][i] = w;
                }
            }
        }
        char* imgW = img + lane * 16;
        float bsum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] = 0.0f;
        const char* pad = reinterpret_cast<const char*>(a.g - 8);
        const char* pad1 = KIND == 1 ? reinterpret_cast<const char*>(a.g1 - 4) : nullptr;
#ifndef FS_PF
#define FS_PF 8
#endif
        constexpr int PF = FS_PF, RING = FS_PF + 1;
        static_assert(27 % RING == 0 && PF + 1 == RING, "ring slots must not depend on the tile");
        static_assert(27 - PF > 10, "the next tile's offsets are decoded at step 10 and first used at step 27 - PF");
        const char* lob = reinterpret_cast<const char*>(lo);
        const uint32_t ld4 = (uint32_t)ld << 2;
        auto idx_load = [&](int64_t row, int32_t (&raw)[10]) {
            const uint32_t rb = (uint32_t)row << 2;
            raw[9] = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(mask) + rb);
#pragma unroll
            for (int q = 0; q < 9; ++q) raw[q] = *reinterpret_cast<const int32_t*>(lob + (rb + (uint32_t)q * ld4));
        };
        auto idx_decode = [&](const int32_t (&raw)[10], uint32_t (&off)[27]) {
            const uint32_t m = (uint32_t)raw[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const uint32_t L = ((uint32_t)raw[q] + 1u) << 5;
                const int m0 = __builtin_amdgcn_sbfe(m, 3 * q, 1), m1 = __builtin_amdgcn_sbfe(m, 3 * q + 1, 1),
                          m2 = __builtin_amdgcn_sbfe(m, 3 * q + 2, 1);
                const uint32_t t1 = L + (uint32_t)__mul24(m0, -32);
                const uint32_t t2 = t1 + (uint32_t)__mul24(m1, -32);
                off[26 - q] = L & (uint32_t)m0; off[26 - (q + 9)] = t1 & (uint32_t)m1; off[26 - (q + 18)] = t2 & (uint32_t)m2;
            }
        };
        auto gather = [&](uint32_t o, f32x4 (&xr)[XN]) {
            xr[0] = *reinterpret_cast<const f32x4*>(pad + o);
            if constexpr (KIND == 0) xr[1] = *reinterpret_cast<const f32x4*>(pad + o + 16);
            if constexpr (KIND == 1) xr[1] = *reinterpret_cast<const f32x4*>(pad1 + (o >> 1));
        };
        float w12[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) w12[j] = 0.0f;
        if constexpr (EPI == 3) {
#pragma unroll
            for (int j = 0; j < 16; ++j) w12[j] = pw.w[j];
        }
        float g12[20];
#pragma unroll
        for (int j = 0; j < 20; ++j) g12[j] = 0.0f;
        float w10[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) w10[j] = 0.0f;
        if constexpr (KIND == 2) {
#pragma unroll
            for (int j = 0; j < 32; ++j) w10[j] = pw.w[j];
        }
        int64_t tile = tb0 + pair;
        uint32_t off[27], offn[27];
        int32_t raw[10];
        f32x4 x[RING][XN];
        if (tile < tb1) {
            const int64_t r = (tile << 6) + lane;
            idx_load(r < n ? r : n - 1, raw);
            idx_decode(raw, off);
#pragma unroll
            for (int u = 0; u < PF; ++u) gather(off[LINR_TAP(u)], x[u]);
        }
        for (int it = 0; it < iters; ++it, tile += FB_WAVES) {
            if (tile >= tb1) {                                   // wave-uniform: keep the block's barrier count
#pragma unroll
                for (int c = 0; c < NCH; ++c) fs_barrier_producer();
                continue;
            }
            const int64_t row_raw = (tile << 6) + lane;
            const bool live = row_raw < n;
            const int64_t row = live ? row_raw : n - 1;
            const int64_t ntile = (tile + FB_WAVES < tb1) ? tile + FB_WAVES : tile;
            const int64_t nrow_raw = (ntile << 6) + lane;
            const int64_t nrow = nrow_raw < n ? nrow_raw : n - 1;
            f32x4 acc[2] = {(f32x4){0.0f, 0.0f, 0.0f, 0.0f}, (f32x4){0.0f, 0.0f, 0.0f, 0.0f}};
            float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0, e2 = e0, e3 = e0, e4 = e0;
            __builtin_amdgcn_sched_barrier(0);
            static_for<27>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                constexpr int k = LINR_TAP(kk);
                constexpr int g = k / 8, ab = (k % 8) * 2;
                constexpr int ch = kk / CT, slot = kk % CT;
                if constexpr (!(FS_LAB & 1)) gather(kk + PF < 27 ? off[LINR_TAP((kk + PF) % 27)] : offn[LINR_TAP((kk + PF) % 27)], x[(kk + PF) % RING]);
                if constexpr (kk == 1) idx_load(nrow, raw);
                if constexpr (kk == 10) idx_decode(raw, offn);
                if constexpr (kk == 5) {
                    if constexpr (EPI == 3) e0 = *reinterpret_cast<const float4*>(pw.aux + row * 4);
                    if constexpr (KIND == 1) {
                        e0 = *reinterpret_cast<const float4*>(a.xin + row * 8);
                        e1 = *reinterpret_cast<const float4*>(a.xin + row * 8 + 4);
                    }
                    if constexpr (KIND == 2) {
                        e0 = *reinterpret_cast<const float4*>(a.res + row * 8);
                        e1 = *reinterpret_cast<const float4*>(a.res + row * 8 + 4);
                        e2 = *reinterpret_cast<const float4*>(a.g + row * 8 + 4);
                        if (a.flags & LINR_RELU_MASK) {
                            e3 = *reinterpret_cast<const float4*>(a.xin + row * 8);
                            e4 = *reinterpret_cast<const float4*>(a.xin + row * 8 + 4);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4(&xk)[XN] = x[kk % RING];
                if constexpr (XN == 2) asm volatile("" : "+v"(xk[0]), "+v"(xk[1]));
                else asm volatile("" : "+v"(xk[0]));
                if constexpr (FS_LAB & 8) {
                } else if constexpr (KIND == 2) {
                    *reinterpret_cast<f32x4*>(imgW + (ch & 1) * FB_BUF + slot * FB_HP) = xk[0];
                } else {
                    *reinterpret_cast<f32x4*>(imgW + (ch & 1) * FB_BUF + slot * FB_TP) = xk[0];
                    *reinterpret_cast<f32x4*>(imgW + (ch & 1) * FB_BUF + slot * FB_TP + FB_HP) = xk[XN - 1];
                }
                constexpr int NBW = KIND == 0 ? 8 : 4;
                static_for<NBW>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (FS_LAB & 4) {
                    } else if constexpr (KIND == 0) {
                        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[i / 4][i % 4], acc[0], 4, ab, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[i / 4][i % 4], acc[1], 4, ab + 1, 0);
                    } else if constexpr (KIND == 1) {
                        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[0][i], acc[0], 4, ab, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[1][i], acc[1], 4, ab + 1, 0);
                    } else {
                        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[0][i], acc[0], 4, ab, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][i], xk[0][i], acc[1], 4, ab + 1, 0);
                    }
                });
                if constexpr (KIND == 2 && kk == 26) {           // conv1_0's weight gradient rides in slot 11 of the last chunk
                    *reinterpret_cast<float4*>(imgW + ((NCH - 1) & 1) * FB_BUF + 11 * FB_HP) = e2;
                    if (live) { bsum[4] += e2.x; bsum[5] += e2.y; bsum[6] += e2.z; bsum[7] += e2.w; }
                }
                if constexpr (k == 13) {                         // the centre tap is the row's own gradient: bias gradient
                    if (live) {
#pragma unroll
                        for (int j = 0; j < 4 * XN; ++j) bsum[j] += xk[j / 4][j % 4];
                    }
                }
                if constexpr (slot == CT - 1 || kk == 26) fs_barrier_producer();          // the chunk's image is complete
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int j = 0; j < 27; ++j) off[j] = offn[j];
            if (live) {
                float o[8];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[4 * h + j] = acc[h][j];
                if constexpr (EPI == 3) {          // gM = (gin[4:8] @ W12^T) * (M > 0)   (W12 [4][4])
                    const float mv[4] = {e0.x, e0.y, e0.z, e0.w};
                    float gm[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t = 0.0f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) t = fmaf(o[4 + q], w12[i * 4 + q], t);
                        gm[i] = mv[i] > 0.0f ? t : 0.0f;
                    }
                    *reinterpret_cast<float4*>(pw.aux_out + row * 4) = make_float4(gm[0], gm[1], gm[2], gm[3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) g12[i * 4 + q] = fmaf(mv[i], o[4 + q], g12[i * 4 + q]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) g12[16 + q] += o[4 + q];
                }
                if constexpr (KIND == 1) {         // gH = [bwd(gI[:, 0:4]; W01) | bwd(gM; W11)] * (H > 0)
                    const float hv[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = hv[j] > 0.0f ? o[j] : 0.0f;
                }
                float* op = a.out + row * 8;
                if constexpr (KIND == 2) {         // + gI, + old (ACCUM), + gH[4:8] @ W10^T, * (A > 0)
                    const float rv[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += rv[j];
                    if (a.flags & LINR_ACCUM) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] += op[j];
                    }
                    const float gq[4] = {e2.x, e2.y, e2.z, e2.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float t = 0.0f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) t = fmaf(gq[q], w10[i * 4 + q], t);
                        o[i] += t;
                    }
                    if (a.flags & LINR_RELU_MASK) {
                        const float av[8] = {e3.x, e3.y, e3.z, e3.w, e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = av[j] > 0.0f ? o[j] : 0.0f;
                    }
                }
                *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
        // ---- fold (producer half): bias gradients, conv1_2's gradients, the slab rows beyond the grid -------------------------
        __syncthreads();                                          // S0: the images are dead (the consumers reuse them)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = bsum[j];
#pragma unroll
            for (int dd = 32; dd > 0; dd >>= 1) t += __shfl_xor(t, dd, 64);
            if (lane == 0) sbias[pair][j] = t;
        }
        if constexpr (EPI == 3) {
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                float t = g12[j];
#pragma unroll
                for (int dd = 32; dd > 0; dd >>= 1) t += __shfl_xor(t, dd, 64);
                if (lane == 0) s12[pair][j] = t;
            }
        }
        __syncthreads();                                          // S1
        const int tid = threadIdx.x;                              // 0 .. 255
        if constexpr (EPI == 3) {
            if (tid < 20) {
                const float t = ((s12[0][tid] + s12[1][tid]) + s12[2][tid]) + s12[3][tid];
                dst[(tid < 16 ? d2.w_off1 + tid : d2.b_off1 + (tid - 16))] = t;
            }
        }
        if (tid < (KIND == 2 ? 8 : 4 * XN)) {
            const float t = ((sbias[0][tid] + sbias[1][tid]) + sbias[2][tid]) + sbias[3][tid];
            if constexpr (KIND == 0) dst[d.b_off + tid] = t;
            else dst[(tid < 4 ? d.b_off : d2.b_off1) + (tid & 3)] = t;
        }
        for (int64_t r = (int64_t)blockIdx.x + gridDim.x; r < a.nb_slab; r += gridDim.x) {
            float* z = d.base + r * d.block_stride;
            for (int e = tid; e < T::WELEMS; e += FB_WAVES * 64) {
                z[d.w_off + e] = 0.0f;
                if constexpr (KIND == 1) z[d2.w_off1 + e] = 0.0f;
            }
            if (tid < T::BELEMS) {
                z[d.b_off + tid] = 0.0f;
                if constexpr (KIND != 0) z[d2.b_off1 + tid] = 0.0f;
            }
            if constexpr (KIND == 2) { if (tid < 32) z[d2.w_off1 + tid] = 0.0f; }
            if constexpr (EPI == 3) { if (tid < 20) z[(tid < 16 ? d2.w_off1 + tid : d2.b_off1 + (tid - 16))] = 0.0f; }
        }
#pragma unroll
        for (int c = 0; c < 2 * NCH - 1; ++c) __syncthreads();   // the consumers' fold rounds
    } else {
        // ================================================ consumer ================================================================
#ifdef FS_PRIO_C
        __builtin_amdgcn_s_setprio(FS_PRIO_C);
#endif
        // weight-gradient roles of the lane: as in conv_bwd_wgrad_single_k
        const int Q = lane >> 4;
        const int wq = KIND == 0 ? (lane & 1) : KIND == 1 ? ((lane >> 3) & 1) : 0;
        const int wslot = KIND == 0 ? ((lane & 15) >> 1) : KIND == 1 ? (lane & 7) : (lane & 15);
        const int ablk = KIND == 1 ? ((lane >> 2) & 1) : ((lane >> 2) & 3), ai = lane & 3;
        const char* imgR = img + (KIND == 2 ? wslot * FB_HP : wslot * FB_TP + wq * FB_HP) + (16 * Q) * 16;
        const int E8 = lane >> 3, wq3 = lane & 1, wslot3 = (lane & 7) >> 1, ablk3 = (lane >> 2) & 1;
        const char* imgR3 = img + FB_BUF + wslot3 * FB_TP + wq3 * FB_HP + (8 * E8) * 16;
        f32x4 wacc[NCH][8];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int c = 0; c < 8; ++c) wacc[ch][c] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        float xa[8], xl[8], xlp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { xa[j] = 0.0f; xl[j] = 0.0f; xlp[j] = 0.0f; }
        auto wg_mfma = [&](auto pcc, auto r0c, auto mc, const float4* b, const float (&XA)[8]) {
            constexpr int pc = decltype(pcc)::value, r0 = decltype(r0c)::value, m = decltype(mc)::value;
            constexpr int j = m / WGM, c = m % 4, r = r0 + j;
            const float B = c == 0 ? b[j].x : c == 1 ? b[j].y : c == 2 ? b[j].z : b[j].w;
            if constexpr (KIND == 1) {
                wacc[pc][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(XA[r >> 1], B, wacc[pc][c], 1, r & 1, 0);
            } else {
                constexpr int h = (m % 8) / 4;
                if constexpr (T::sets(pc) == 8)
                    wacc[pc][c * 2 + h] = __builtin_amdgcn_mfma_f32_4x4x1f32(XA[2 * (r >> 1) + h], B, wacc[pc][c * 2 + h], 1, r & 1, 0);
                else
                    wacc[pc][c * 2 + h] = __builtin_amdgcn_mfma_f32_4x4x1f32(XA[2 * (r >> 2) + h], B, wacc[pc][c * 2 + h], 2, r & 3, 0);
            }
        };
        auto xa_load = [&](int64_t row0, float (&XA)[8]) {
            if constexpr (KIND == 1) {
#pragma unroll
                for (int rq = 0; rq < 8; ++rq) {
                    const int64_t r = row0 + 16 * Q + 2 * rq + ablk;
                    XA[rq] = (r < n) ? a.xin[r * 8 + 4 * wq + ai] : 0.0f;
                }
            } else {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int64_t r = row0 + 16 * Q + 4 * rq + ablk;
#pragma unroll
                    for (int h = 0; h < 2; ++h) XA[2 * rq + h] = (r < n) ? a.xin[r * 8 + 4 * h + ai] : 0.0f;
                }
            }
        };
        auto xl_load = [&](int64_t row0, float (&XL)[8]) {         // KIND 0: the eighths' layout
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int64_t r = row0 + 8 * E8 + 2 * rq + ablk3;
#pragma unroll
                for (int h = 0; h < 2; ++h) XL[2 * rq + h] = (r < n) ? a.xin[r * 8 + 4 * h + ai] : 0.0f;
            }
        };
        int64_t tile = tb0 + pair;
        const bool any = tile < tb1;
        if (any) {
            xa_load(tile << 6, xa);
            if constexpr (KIND == 0) xl_load(tile << 6, xl);
        }
        for (int it = 0; it < iters; ++it, tile += FB_WAVES) {
            if (tile >= tb1) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) fs_barrier_consumer();
                continue;
            }
            const int64_t ntile = (tile + FB_WAVES < tb1) ? tile + FB_WAVES : tile;
            float xan[8], xln[8];
            float4 bq[2][6];
            __builtin_amdgcn_sched_barrier(0);
            static_for<27>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                constexpr int ch = kk / CT, slot = kk % CT;
                constexpr int pc = (ch + NCH - 1) % NCH;           // the chunk whose rows are multiplied in this phase
                constexpr int nst = T::ntaps(ch);
                constexpr int r0 = (slot * T::prows(pc)) / nst, nr = ((slot + 1) * T::prows(pc)) / nst - r0;
                constexpr int kn = (kk + 1) % 27, chn = kn / CT, slotn = kn % CT, pcn = (chn + NCH - 1) % NCH, nstn = T::ntaps(chn);
                constexpr int r0n = (slotn * T::prows(pcn)) / nstn, nrn = ((slotn + 1) * T::prows(pcn)) / nstn - r0n;
                if constexpr (slot == 0 && !(FS_LAB & 8)) {        // first step of a phase: behind the barrier
#pragma unroll
                    for (int j = 0; j < nr; ++j)
                        bq[kk & 1][j] = (KIND == 0 && pc == NCH - 1) ? *reinterpret_cast<const float4*>(imgR3 + (r0 + j) * 16)
                                                                     : *reinterpret_cast<const float4*>(imgR + (pc & 1) * FB_BUF + (r0 + j) * 16);
                }
                if constexpr (kk == 3) xa_load(ntile << 6, xan);
                if constexpr (KIND == 0 && kk == 4) xl_load(ntile << 6, xln);
                if constexpr (slotn != 0 && !(FS_LAB & 8)) {       // the next step's rows, one step ahead (same phase only)
#pragma unroll
                    for (int j = 0; j < nrn; ++j)
                        bq[kn & 1][j] = (KIND == 0 && pcn == NCH - 1) ? *reinterpret_cast<const float4*>(imgR3 + (r0n + j) * 16)
                                                                      : *reinterpret_cast<const float4*>(imgR + (pcn & 1) * FB_BUF + (r0n + j) * 16);
                }
                static_for<(FS_LAB & 2) ? 0 : nr * WGM>([&](auto mc) {
                    if constexpr (ch == 0) wg_mfma(std::integral_constant<int, pc>{}, std::integral_constant<int, r0>{}, mc, bq[kk & 1], xlp);
                    else wg_mfma(std::integral_constant<int, pc>{}, std::integral_constant<int, r0>{}, mc, bq[kk & 1], xa);
                });
                if constexpr (slot == CT - 1 || kk == 26) fs_barrier_consumer();
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (KIND == 0) { xlp[j] = xl[j]; xl[j] = xln[j]; } else xlp[j] = xa[j];
                xa[j] = xan[j];
            }
        }
        if (any) {   // the last chunk of the pair's last tile
            constexpr int pr = T::prows(NCH - 1);
            float4 b[pr];
#pragma unroll
            for (int j = 0; j < pr; ++j)
                b[j] = *reinterpret_cast<const float4*>((KIND == 0 ? imgR3 : imgR + ((NCH - 1) & 1) * FB_BUF) + j * 16);
            static_for<pr * WGM>([&](auto mc) {
                wg_mfma(std::integral_constant<int, NCH - 1>{}, std::integral_constant<int, 0>{}, mc, b, xlp);
            });
        }
        // ---- fold (consumer half): the (pair, row set) partials of every element in fixed order, one slab row per block ----------
        __syncthreads();                                          // S0
        __syncthreads();                                          // S1 (the producers' bias round)
        float* sacc = reinterpret_cast<float*>(smem);             // [pair][lane][33]
        const int tid = threadIdx.x - FB_WAVES * 64;              // 0 .. 255
        static_for<NCH>([&](auto chc) {
            constexpr int ch = decltype(chc)::value;
            constexpr int ntaps = T::ntaps(ch);
            constexpr int sets = T::sets(ch), width = 64 / sets;
            if constexpr (ch > 0) __syncthreads();                // the previous round's reads
            float* mine = sacc + (pair * 64 + lane) * 33;
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) mine[c * 4 + i] = wacc[ch][c][i];
            __syncthreads();
            constexpr int per_tap = KIND == 0 ? 64 : 32;
            constexpr int nslots = ntaps + ((KIND == 2 && ch == NCH - 1) ? 1 : 0);
            for (int e = tid; e < nslots * per_tap; e += FB_WAVES * 64) {
                const int slot = e / per_tap, r = e % per_tap;
                int kp, idx, dofs;
                const int kk = CT * ch + slot;
                const int k = kk / 9 + 3 * ((kk / 3) % 3) + 9 * (kk % 3);
                if constexpr (KIND == 0) {
                    const int ci = r >> 3, co = r & 7;
                    kp = 2 * slot + (co >> 2); idx = ((co & 3) * 2 + (ci >> 2)) * 4 + (ci & 3);
                    dofs = (int)d.w_off + k * 64 + r;
                } else if constexpr (KIND == 1) {
                    const int cv = r >> 4, ci = (r >> 2) & 3, co = r & 3;
                    kp = 8 * cv + slot; idx = co * 4 + ci;
                    dofs = (int)(cv ? d2.w_off1 : d.w_off) + k * 16 + (r & 15);
                } else {
                    const int ci = r >> 2, co = r & 3;
                    kp = slot; idx = (co * 2 + (ci >> 2)) * 4 + (ci & 3);
                    dofs = slot < ntaps ? (int)d.w_off + k * 32 + r : (int)d2.w_off1 + r;
                }
                float t = 0.0f;
#pragma unroll
                for (int w = 0; w < FB_WAVES; ++w)
#pragma unroll
                    for (int qq = 0; qq < sets; ++qq) t += sacc[(w * 64 + width * qq + kp) * 33 + idx];
                dst[dofs] = t;
            }
        });
    }
}

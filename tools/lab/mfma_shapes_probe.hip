// Probe (not part of the product): the rate of the f32 matrix instructions by shape on gfx950 - v_mfma_f32_4x4x1 (16 blocks, 256 MAC,
// 2 passes), 16x16x4 (1024 MAC, 8 passes), 32x32x2 (2048 MAC, 16 passes) - on independent accumulators, registers only.  All three have
// the same nominal 32 MAC per cycle per SIMD (157 TFLOP/s at 2.4 GHz x 1024 SIMDs).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float w0) {
    float x = threadIdx.x * 0.001f, w = w0;
    float s = 0;
    if constexpr (SHAPE == 0 || SHAPE == 1) {
        f32x4 acc[NACC];
        for (int a = 0; a < NACC; ++a) acc[a] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if constexpr (SHAPE == 0) acc[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, acc[u % NACC], 4, 0, 0);
                else acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, acc[u % NACC], 0, 0, 0);
            }
        }
        for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    } else {
        f32x16 acc[NACC];
        for (int a = 0; a < NACC; ++a) for (int j = 0; j < 16; ++j) acc[a][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, acc[u % NACC], 0, 0, 0);
        }
        for (int a = 0; a < NACC; ++a) for (int j = 0; j < 16; ++j) s += acc[a][j];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int NACC>
static void run(float* out, int blocks, const char* tag) {
    const int iters = SHAPE == 0 ? 400 : SHAPE == 1 ? 100 : 50;
    const double mac = SHAPE == 0 ? 256.0 : SHAPE == 1 ? 1024.0 : 2048.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 30; ++i) k<SHAPE, NACC><<<blocks, 256>>>(out, iters, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) k<SHAPE, NACC><<<blocks, 256>>>(out, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 20;
    const double fl = blocks * 4.0 * iters * 16 * mac * 2;
    printf("%-28s blocks %5d: %8.1f us   %6.1f TFLOP/s\n", tag, blocks, us, fl / us / 1e6);
}

int main() {
    float* out; hipMalloc(&out, 16384 * 256 * 4);
    for (int blocks : {1024, 2048, 4096}) {
        run<0, 8>(out, blocks, "4x4x1   8 accumulators");
        run<0, 16>(out, blocks, "4x4x1  16 accumulators");
        run<1, 4>(out, blocks, "16x16x4  4 accumulators");
        run<1, 8>(out, blocks, "16x16x4  8 accumulators");
        run<2, 2>(out, blocks, "32x32x2  2 accumulators");
        run<2, 4>(out, blocks, "32x32x2  4 accumulators");
    }
    return 0;
}

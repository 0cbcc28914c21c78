// Probe (not part of the product): can v_pk_fma_f32 (VALU) run beside v_mfma_f32_4x4x1_16b_f32 (matrix core) on the same SIMD?
// Per loop iteration: NM MFMAs and NV packed FMAs, all on independent accumulators.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NM, int NV>
__global__ __launch_bounds__(256) void mix(float* out, int iters, float w0) {
    f32x4 acc[8];
    f32x2 vac[16];
    for (int a = 0; a < 8; ++a) acc[a] = (f32x4){0, 0, 0, 0};
    for (int a = 0; a < 16; ++a) vac[a] = (f32x2){0, 0};
    float x = threadIdx.x * 0.001f, w = w0;
    f32x2 g = {w0, w0 * 1.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u < NM) acc[u % 8] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, acc[u % 8], 4, 0, 0);
            if (u < NV) vac[u] = __builtin_elementwise_fma((f32x2){x, x}, g, vac[u]);
        }
    }
    float s = 0;
    for (int a = 0; a < 8; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    for (int a = 0; a < 16; ++a) s += vac[a][0] + vac[a][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NM, int NV>
static void run(float* out, int blocks, const char* tag) {
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix<NM, NV><<<blocks, 256>>>(out, iters, 0.5f); mix<NM, NV><<<blocks, 256>>>(out, iters, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) mix<NM, NV><<<blocks, 256>>>(out, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 10;
    const double waves = blocks * 4.0;
    const double fl_m = waves * iters * NM * 512.0, fl_v = waves * iters * NV * 256.0;     // MFMA: 256 MAC; pk_fma: 128 MAC
    printf("%-22s blocks %5d: %8.1f us   MFMA %6.1f TF  VALU %6.1f TF  total %6.1f TF\n", tag, blocks, us, fl_m / us / 1e6, fl_v / us / 1e6,
           (fl_m + fl_v) / us / 1e6);
}

int main() {
    float* out; hipMalloc(&out, 16384 * 256 * 4);
    for (int blocks : {1024, 4096}) {
        run<16, 0>(out, blocks, "16 MFMA");
        run<0, 16>(out, blocks, "16 pk_fma");
        run<16, 16>(out, blocks, "16 MFMA + 16 pk_fma");
        run<16, 8>(out, blocks, "16 MFMA + 8 pk_fma");
        run<8, 16>(out, blocks, "8 MFMA + 16 pk_fma");
    }
    return 0;
}

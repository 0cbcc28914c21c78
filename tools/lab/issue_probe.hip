// Probe (not part of the product): does ONE wave per SIMD overlap its own v_mfma_f32_4x4x1 with its own VALU / LDS instructions, or
// only with another wave's?  Per loop iteration NM MFMAs + NV plain v_fma + NL ds_read_b128, all independent.  Occupancy is forced
// through the dynamic LDS size: 100 KB -> one 256-thread block per CU (one wave per SIMD), 60 KB -> two (two waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -w -o tools/_lab/issue_probe tools/lab/issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int NL>
__global__ __launch_bounds__(256) void mix(float* out, int iters, float w0, long long* clk) {
    extern __shared__ float4 lds[];
    f32x4 acc[8];
    float vac[16];
    float4 lacc = make_float4(0, 0, 0, 0);
    for (int a = 0; a < 8; ++a) acc[a] = (f32x4){0, 0, 0, 0};
    for (int a = 0; a < 16; ++a) vac[a] = 0.0f;
    lds[threadIdx.x] = make_float4(w0, w0, w0, w0);
    __syncthreads();
    float x = threadIdx.x * 0.001f, w = w0;
    const float4* lp = lds + (threadIdx.x & 63) + (threadIdx.x >> 6) * 64;
    const long long c0 = clock64(), w0c = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u < NM) acc[u % 8] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, acc[u % 8], 4, 0, 0);
            if (u < NV) vac[u] = fmaf(x, w, vac[u]);
            if (u < NL) { const float4 t = lp[(u * 4 + it) & 3]; lacc.x += t.x; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0c; }
    float s = lacc.x;
    for (int a = 0; a < 8; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    for (int a = 0; a < 16; ++a) s += vac[a];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NM, int NV, int NL>
static void run(float* out, int per_cu, const char* tag) {
    static long long* clk = nullptr;
    if (!clk) hipMallocManaged(&clk, 16);
    const int iters = 2000, blocks = 256 * per_cu;
    const size_t lds = per_cu == 1 ? 100 * 1024 : 60 * 1024;
    hipFuncSetAttribute((const void*)mix<NM, NV, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix<NM, NV, NL><<<blocks, 256, lds>>>(out, iters, 0.5f, clk); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) mix<NM, NV, NL><<<blocks, 256, lds>>>(out, iters, 0.5f, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 5;
    // cycles per loop iteration of one wave's stream, per SIMD (two waves share the SIMD when per_cu == 2), at 2.4 GHz
    hipDeviceSynchronize();
    // clock64 = shader cycles of wave 0, wall_clock64 = constant 100 MHz
    printf("%-34s %d wave(s)/SIMD: %8.1f us  = %6.1f ns per iteration per SIMD-slot; wave 0: %.1f shader cycles per iteration, clock %.2f GHz\n", tag,
           per_cu, us, us * 1e3 / iters / per_cu, (double)clk[0] / iters, (double)clk[0] / ((double)clk[1] * 10.0));
}

int main() {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    for (int per_cu : {1, 2}) {
        run<16, 0, 0>(out, per_cu, "16 MFMA");
        run<0, 16, 0>(out, per_cu, "16 v_fma");
        run<16, 16, 0>(out, per_cu, "16 MFMA + 16 v_fma");
        run<16, 8, 0>(out, per_cu, "16 MFMA + 8 v_fma");
        run<0, 0, 8>(out, per_cu, "8 ds_read_b128");
        run<16, 0, 8>(out, per_cu, "16 MFMA + 8 ds_read_b128");
        run<16, 8, 8>(out, per_cu, "16 MFMA + 8 v_fma + 8 ds_read");
    }
    return 0;
}

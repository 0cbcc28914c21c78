// Probe (not part of the product): SPECIALISED waves on one SIMD - wave A issues only v_mfma_f32_4x4x1, wave B only VALU (or only
// ds_read_b128): do they overlap?  512-thread workgroups, one per CU (100 KB LDS): waves 0..3 = A, 4..7 = B, pairs share a SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o tools/_lab/issue_probe2 tools/lab/issue_probe2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode bit 0: A waves run MFMAs; bit 1: B waves run VALU; bit 2: B waves run LDS reads
__global__ __launch_bounds__(512) void spec(float* out, int iters, float w0, int mode, long long* clk) {
    extern __shared__ float4 lds[];
    const int wave = threadIdx.x >> 6;
    lds[threadIdx.x] = make_float4(w0, w0, w0, w0);
    __syncthreads();
    float x = threadIdx.x * 0.001f, w = w0, s = 0.0f;
    const long long c0 = clock64();
    if (wave < 4) {
        if (mode & 1) {
            f32x4 acc[8];
            for (int a = 0; a < 8; ++a) acc[a] = (f32x4){0, 0, 0, 0};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u % 8] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, acc[u % 8], 4, 0, 0);
            }
            for (int a = 0; a < 8; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
        }
    } else {
        if (mode & 2) {
            float vac[16];
            for (int a = 0; a < 16; ++a) vac[a] = 0.0f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) vac[u] = fmaf(x, w, vac[u]);
            }
            for (int a = 0; a < 16; ++a) s += vac[a];
        }
        if (mode & 4) {
            const float4* lp = lds + (threadIdx.x & 63);
            float4 l = make_float4(0, 0, 0, 0);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { const float4 t = lp[((u * 4 + it) & 3) * 64]; l.x += t.x; }
            }
            s += l.x;
        }
    }
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) clk[threadIdx.x ? 1 : 0] = clock64() - c0;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    long long* clk; hipMallocManaged(&clk, 16);
    hipFuncSetAttribute((const void*)spec, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const int iters = 2000;
    const char* nm[8] = {"-", "A: 16 MFMA alone", "B: 16 v_fma alone", "A: 16 MFMA | B: 16 v_fma", "B: 8 ds_read_b128 alone", "A: 16 MFMA | B: 8 ds_read_b128", "B: v_fma + ds_read", "A: 16 MFMA | B: 16 v_fma + 8 ds_read"};
    for (int mode = 1; mode < 8; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        spec<<<256, 512, 100 * 1024>>>(out, iters, 0.5f, mode, clk); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) spec<<<256, 512, 100 * 1024>>>(out, iters, 0.5f, mode, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipDeviceSynchronize();
        printf("%-40s kernel %7.1f us; per iteration: wave A %6.1f cycles, wave B %6.1f cycles\n", nm[mode], ms * 1e3 / 5, (double)clk[0] / iters, (double)clk[1] / iters);
    }
    return 0;
}

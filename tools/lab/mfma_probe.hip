// Probe (not part of the product): layout and issue rate of v_mfma_f32_4x4x1_16b_f32 with A broadcast (cbsz/abid).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    f32x4 d0 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 100.0f + l, c, 4, 0, 0);
    f32x4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 100.0f + l, c, 4, 1, 0);
    f32x4 d2 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 100.0f + l, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) { out[l * 12 + i] = d0[i]; out[l * 12 + 4 + i] = d1[i]; out[l * 12 + 8 + i] = d2[i]; }
}

template <int NACC>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = (f32x4){0, 0, 0, 0};
    float x = threadIdx.x * 0.001f, w = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, acc[u % NACC], 4, 0, 0);
            acc[(u + 1) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, acc[(u + 1) % NACC], 4, 1, 0);
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 1315 * 8 * 256 * 4 + 64 * 12 * 4);
    layout<<<1, 64>>>(out);
    std::vector<float> h(64 * 12);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    int bad0 = 0, bad1 = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        if (h[l * 12 + i] != (float)(0 * 4 + i) * (100.0f + l)) bad0++;
        if (h[l * 12 + 4 + i] != (float)(1 * 4 + i) * (100.0f + l)) bad1++;
        if (h[l * 12 + 8 + i] != (float)((l / 4) * 4 + i) * (100.0f + l)) bad2++;
    }
    printf("layout check: cbsz4/abid0 mismatches %d, cbsz4/abid1 %d, no-broadcast %d  (lane5: %g %g %g %g | %g %g %g %g | %g %g %g %g)\n",
           bad0, bad1, bad2, h[60], h[61], h[62], h[63], h[64], h[65], h[66], h[67], h[68], h[69], h[70], h[71]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 27;   // 27 x 16 MFMAs = the 8->8 conv's MFMA count per wave
    for (int nacc = 1; nacc <= 4; nacc *= 2) for (int mult = 1; mult <= 8; mult *= 8) {
        auto go = [&]() { if (nacc == 1) rate<1><<<1315 * mult, 256>>>(out, iters); else if (nacc == 2) rate<2><<<1315 * mult, 256>>>(out, iters); else rate<4><<<1315 * mult, 256>>>(out, iters); };
        go(); go(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) go();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double us = ms * 1e3 / 20, flop = 1315.0 * mult * 4 * iters * 16 * 256 * 2;
        printf("accumulators %d blocks %6d: %.2f us  %.1f TFLOP/s\n", nacc, 1315 * mult, us, flop / us / 1e6);
    }
    return 0;
}

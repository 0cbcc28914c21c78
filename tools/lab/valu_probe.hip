// Timing probe (not part of the product): how fast is the conv's pure FMA stream (27 x 64 FMAs per lane, weights in
// SGPRs) with no memory traffic at all?   hipcc --offload-arch=gfx950 -O3 tools/lab/valu_probe.hip -o /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void valu_only(float* __restrict__ out, const float* __restrict__ W, float seed) {
    float acc[8], x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = 0.f; x[i] = seed + threadIdx.x * 0.001f + i; }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const float* wk = W + (MODE == 1 ? 0 : k * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int o = 0; o < 8; ++o) acc[o] = fmaf(x[i], wk[i * 8 + o], acc[o]);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] * 1.0001f;     // keep the rows "different" per offset (8 extra VALU)
    }
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o) s += acc[o];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int blocks = 1315;
    float *out, *W;
    hipMalloc(&out, (size_t)blocks * 8 * 256 * 4);      // sized for the largest launch below (mult = 8)
    hipMalloc(&W, 27 * 64 * 4);
    std::vector<float> h(27 * 64, 0.01f);
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int mult = 1; mult <= 8; mult *= 2) {
            for (int w = 0; w < 3; ++w) {
                if (mode == 0) valu_only<0><<<blocks * mult, 256>>>(out, W, 1.f); else valu_only<1><<<blocks * mult, 256>>>(out, W, 1.f);
            }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int it = 20;
            for (int i = 0; i < it; ++i) {
                if (mode == 0) valu_only<0><<<blocks * mult, 256>>>(out, W, 1.f); else valu_only<1><<<blocks * mult, 256>>>(out, W, 1.f);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / it;
            const double flop = (double)blocks * mult * 256 * 27 * 64 * 2;
            printf("mode %d (%s) blocks %6d  %.2f us  %.1f TFLOP/s\n", mode, mode ? "weights hoisted" : "weights per offset",
                   blocks * mult, us, flop / us / 1e6);
        }
    }
    return 0;
}

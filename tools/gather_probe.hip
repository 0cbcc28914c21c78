// Gather-rate probe for the 3x3x3 convolutions' neighbour reads (one 32-byte row per tap and output row, 27 taps):
//   A  lane = row, two 16-byte loads per tap at a 32-byte lane stride (what cconv_mfma_k does)
//   B  lane = (row of 32, half): ONE instruction fetches 32 whole rows = 1 KB contiguous when the neighbours are consecutive; two
//      instructions per tap and 64-row tile (the data then sits as (row, half) pairs and needs a lane swap before the MFMAs)
//   C  lane = (tap of 4, row of 8, half): the transposing weight-gradient kernel's layout
// Synthetic kernel map with the locality of an x-major sorted surface: tap (dx,dy,dz) -> row + 700 dx + 27 dy + dz.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_lab/gather_probe tools/gather_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int ORD, int XCD = 0>
__global__ __launch_bounds__(256) void gather_a(const float* __restrict__ x, const int* __restrict__ nbr, long ld, long n, float* out) {
    // XCD 1: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2): give every XCD one contiguous range of tiles
    const long nb = gridDim.x, per = (nb + 7) / 8;
    const long blk = XCD ? (long)(blockIdx.x % 8) * per + blockIdx.x / 8 : (long)blockIdx.x;
    if (blk >= nb) return;
    const long row = blk * 256 + threadIdx.x;
    const long r = row < n ? row : n - 1;
    const char* base = (const char*)x;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int kk = 0; kk < 27; ++kk) {
        // ORD 1: the three dz taps of a (dx,dy) column back to back; ORD 2: additionally the three dy columns of an x-slab back to back
        const int k = ORD == 2 ? (kk / 9) + 3 * ((kk / 3) % 3) + 9 * (kk % 3) : ORD ? (kk / 3) + 9 * (kk % 3) : kk;
        const unsigned off = (unsigned)(nbr[k * ld + r] + 1) << 5;
        const float4 a = *(const float4*)(base + off), b = *(const float4*)(base + off + 16);
        acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
    }
    if (row < n) out[row] = acc.x + acc.y + acc.z + acc.w;
}

template <int ORD>
__global__ __launch_bounds__(256) void gather_b(const float* __restrict__ x, const int* __restrict__ nbr, long ld, long n, float* out) {
    const int lane = threadIdx.x & 63;
    const long tile = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const char* base = (const char*)x;
    float4 acc = make_float4(0, 0, 0, 0);
    long rows[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { long r = tile * 64 + 32 * j + (lane >> 1); rows[j] = r < n ? r : n - 1; }
    const unsigned hoff = 16u * (lane & 1);
#pragma unroll
    for (int kk = 0; kk < 27; ++kk) {
        const int k = ORD ? (kk / 3) + 9 * (kk % 3) : kk;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned off = ((unsigned)(nbr[k * ld + rows[j]] + 1) << 5) + hoff;
            const float4 a = *(const float4*)(base + off);
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
    }
    const long row = tile * 64 + lane;
    if (row < n) out[row] = acc.x + acc.y + acc.z + acc.w;
}

__global__ __launch_bounds__(256) void gather_c(const float* __restrict__ x, const int* __restrict__ nbr, long ld, long n, float* out) {
    const int lane = threadIdx.x & 63;
    const long tile = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int t = lane >> 4, u = (lane >> 1) & 7;
    const unsigned hoff = 16u * (lane & 1);
    const char* base = (const char*)x;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int g = 0; g < 8; ++g) {                       // 8 groups of 8 rows = the 64 rows of this wave
        long r = tile * 64 + 8 * g + u;
        r = r < n ? r : n - 1;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int k = 4 * j + t;
            const unsigned off = ((unsigned)((k < 27 ? nbr[k * ld + r] : -1) + 1) << 5) + hoff;
            const float4 a = *(const float4*)(base + off);
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
    }
    const long row = tile * 64 + lane;
    if (row < n) out[row] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    const long n = 336529, ld = (n + 63) / 64 * 64;
    std::vector<int> h(27 * ld, -1);
    for (int k = 0; k < 27; ++k) {
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;          // k = column q + 9 (dz + 1), like the executor's compressed map
        for (long r = 0; r < n; ++r) {
            long t = r + 700L * dx + 27L * dy + dz;
            const bool present = ((r * 2654435761u + k * 40503u) >> 7) % 27 < 14 || k == 13;      // ~half of the taps present
            h[k * ld + r] = (present && t >= 0 && t < n) ? (int)t : -1;
        }
    }
    int* nbr; float *x, *out;
    CK(hipMalloc(&nbr, h.size() * 4)); CK(hipMalloc(&x, (n + 1) * 32)); CK(hipMalloc(&out, n * 4));
    CK(hipMemcpy(nbr, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, (n + 1) * 32));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = (int)((n + 255) / 256), iters = 50;
    for (int v = 0; v < 7; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) {
                if (v == 0) gather_a<0><<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 1) gather_b<0><<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 2) gather_c<<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 3) gather_a<1><<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 4) gather_b<1><<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 5) gather_a<1, 1><<<(blocks + 7) / 8 * 8, 256>>>(x, nbr, ld, n, out);
                else gather_a<2><<<blocks, 256>>>(x, nbr, ld, n, out);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const char* nm[7] = {"A  lane=row, taps dz-major", "B  lane=(row,half), taps dz-major", "C  transposing layout", "A' lane=row, taps column-major", "B' lane=(row,half), taps column-major", "A'' = A' + XCD-contiguous tile ranges", "A3 lane=row, taps x-slab-major (dx, dy, dz)"};
            if (rep) printf("%-40s %.2f us per pass (%ld rows, 27 taps x 32 B)\n", nm[v], ms * 1e3 / iters, n);
        }
    }
    return 0;
}

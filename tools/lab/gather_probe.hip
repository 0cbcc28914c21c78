// Gather-rate probe for the 3x3x3 convolutions' neighbour reads (one 32-byte row per tap and output row, 27 taps):
//   A  lane = row, two 16-byte loads per tap at a 32-byte lane stride (what cconv_mfma_k does)
//   B  lane = (row of 32, half): ONE instruction fetches 32 whole rows = 1 KB contiguous when the neighbours are consecutive; two
//      instructions per tap and 64-row tile (the data then sits as (row, half) pairs and needs a lane swap before the MFMAs)
//   C  lane = (tap of 4, row of 8, half): the transposing weight-gradient kernel's layout
// Synthetic kernel map with the locality of an x-major sorted surface: tap (dx,dy,dz) -> row + 700 dx + 27 dy + dz.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_lab/gather_probe tools/lab/gather_probe.hip ; run on the GPU box.
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

// E  layout A3 with the loads of absent taps masked off (exec mask) instead of reading the zero row; MODE 1: index loads only
template <int MODE>
__global__ __launch_bounds__(256) void gather_e(const float* __restrict__ x, const int* __restrict__ nbr, long ld, long n, float* out) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    const long r = row < n ? row : n - 1;
    const char* base = (const char*)x;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int kk = 0; kk < 27; ++kk) {
        const int k = (kk / 9) + 3 * ((kk / 3) % 3) + 9 * (kk % 3);
        const int v = nbr[k * ld + r];
        if (MODE == 1) { acc.x += __int_as_float(v); continue; }
        float4 a = make_float4(0, 0, 0, 0), b = a;
        if (v >= 0) {
            const unsigned off = (unsigned)(v + 1) << 5;
            a = *(const float4*)(base + off); b = *(const float4*)(base + off + 16);
        }
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

// D  band staging: the neighbours of a tile of 256 consecutive rows lie, per (dx,dy) column, in one contiguous band of the matrix
//    (x-major sorted rows: a shift by (dx,dy) preserves the order).  The block copies the 9 bands into LDS with coalesced 16-byte
//    loads (288 B per row instead of 864 B through the L1's tag lookup) and gathers the 27 taps from LDS (two 16-byte planes per band:
//    consecutive rows 16 B apart = conflict-free ds_read_b128).  Synthetic map: band q of tile t starts at 256 t + 700 dx + 27 dy - 1.
constexpr int D_R = 264;          // band rows held per tile (256 + 2, padded)
template <int T>
__global__ __launch_bounds__(256) void gather_d(const float* __restrict__ x, const int* __restrict__ nbr, long ld, long n, float* out) {
    extern __shared__ float4 lds[];             // [9][2][D_R] + one zero entry at the end
    const long i0 = (long)blockIdx.x * 256;
    const long row = i0 + threadIdx.x;
    const long r = row < n ? row : n - 1;
    const float4* xb = (const float4*)x;         // row j (-1 = pad) at xb[2 (j + 1)]
    int bstart[9];
    float4 v0[9], v1[9], v2[9];
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 9; ++q) {                 // all band loads in flight before the first LDS store
        const int dy = q / 3 - 1, dx = q % 3 - 1;
        const long b = i0 + 700L * dx + 27L * dy - 1;
        bstart[q] = (int)b;
        auto piece = [&](int pc) {
            long j = b + (pc >> 1);
            j = (j < -1 || j >= n) ? -1 : j;
            return xb[2 * (j + 1) + (pc & 1)];
        };
        if (!(T & 2)) { v0[q] = piece(t); v1[q] = piece(t + 256); if (t < 4) v2[q] = piece(t + 512); }
        else { v0[q] = v1[q] = v2[q] = make_float4(q, t, 0, 0); }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        lds[(q * 2 + (t & 1)) * D_R + (t >> 1)] = v0[q];
        lds[(q * 2 + (t & 1)) * D_R + 128 + (t >> 1)] = v1[q];
        if (t < 4) lds[(q * 2 + (t & 1)) * D_R + 256 + (t >> 1)] = v2[q];
    }
    if (threadIdx.x == 0) lds[18 * D_R] = make_float4(0, 0, 0, 0);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int kk = 0; kk < 27; ++kk) {
        const int q = kk / 3, dz = kk % 3, k = q + 9 * dz;
        const int dyy = q / 3 - 1, dxx = q % 3 - 1;
        const long tt = r + 700L * dxx + 27L * dyy + (dz - 1);
        const bool present = ((r * 2654435761u + k * 40503u) >> 7) % 27 < 14 || k == 13;
        const int v = (T & 1) ? ((present && tt >= 0 && tt < n) ? (int)tt : -1) : nbr[k * ld + r];
        const int rel = v - bstart[q];
        const int o0 = v < 0 ? 18 * D_R : (q * 2) * D_R + rel, o1 = v < 0 ? 18 * D_R : (q * 2 + 1) * D_R + rel;
        const float4 a = lds[o0], b = lds[o1];
        acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
    }
    if (row < n) out[row] = acc.x + acc.y + acc.z + acc.w;
}

// D2 = D as long-lived blocks with a software pipeline: the bands and indices of the next tile are loaded into registers while the
//      current tile's taps are read from LDS (single LDS buffer, two blocks per CU).
__global__ __launch_bounds__(256) void gather_d2(const float* __restrict__ x, const int* __restrict__ nbr, long ld, long n, float* out) {
    extern __shared__ float4 lds[];
    const float4* xb = (const float4*)x;
    const int t = threadIdx.x;
    const long ntiles = (n + 255) / 256;
    float4 v0[9], v1[9], v2[9];
    int idx[27], nidx[27];
    auto load_tile = [&](long tile) {
        const long i0 = tile * 256;
        const long r = i0 + t < n ? i0 + t : n - 1;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int dy = q / 3 - 1, dx = q % 3 - 1;
            const long b = i0 + 700L * dx + 27L * dy - 1;
            auto piece = [&](int pc) {
                long j = b + (pc >> 1);
                j = (j < -1 || j >= n) ? -1 : j;
                return xb[2 * (j + 1) + (pc & 1)];
            };
            v0[q] = piece(t); v1[q] = piece(t + 256);
            if (t < 4) v2[q] = piece(t + 512);
        }
#pragma unroll
        for (int kk = 0; kk < 27; ++kk) {
            const int q = kk / 3, dz = kk % 3, k = q + 9 * dz;
            const int dy = q / 3 - 1, dx = q % 3 - 1;
            const int v = nbr[k * ld + r];
            const int b = (int)(i0 + 700L * dx + 27L * dy - 1);
            nidx[kk] = v < 0 ? 18 * D_R : (q * 2) * D_R + (v - b);
        }
    };
    if (t == 0) lds[18 * D_R] = make_float4(0, 0, 0, 0);
    long tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                          // everyone has finished reading the previous tile
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            lds[(q * 2 + (t & 1)) * D_R + (t >> 1)] = v0[q];
            lds[(q * 2 + (t & 1)) * D_R + 128 + (t >> 1)] = v1[q];
            if (t < 4) lds[(q * 2 + (t & 1)) * D_R + 256 + (t >> 1)] = v2[q];
        }
#pragma unroll
        for (int kk = 0; kk < 27; ++kk) idx[kk] = nidx[kk];
        __syncthreads();
        if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x);
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 27; ++kk) {
            const int o0 = idx[kk], o1 = idx[kk] == 18 * D_R ? 18 * D_R : idx[kk] + D_R;
            const float4 a = lds[o0], b = lds[o1];
            acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
        }
        const long row = tile * 256 + t;
        if (row < n) out[row] = acc.x + acc.y + acc.z + acc.w;
    }
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
    std::vector<int> h_all(27 * ld), h_none(27 * ld, -1);
    for (int k = 0; k < 27; ++k) { const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1; for (long r = 0; r < n; ++r) { long t = r + 700L * dx + 27L * dy + dz; h_all[k * ld + r] = (int)(t < 0 ? 0 : t >= n ? n - 1 : t); } }
    int *nbr, *nbr_all, *nbr_none; float *x, *out;
    CK(hipMalloc(&nbr_all, h.size() * 4)); CK(hipMalloc(&nbr_none, h.size() * 4));
    CK(hipMemcpy(nbr_all, h_all.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(nbr_none, h_none.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&nbr, h.size() * 4)); CK(hipMalloc(&x, (n + 1) * 32)); CK(hipMalloc(&out, n * 4));
    CK(hipMemcpy(nbr, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, (n + 1) * 32));
    CK(hipFuncSetAttribute((const void*)gather_d<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (18 * D_R + 1) * 16));
    CK(hipFuncSetAttribute((const void*)gather_d<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (18 * D_R + 1) * 16));
    CK(hipFuncSetAttribute((const void*)gather_d<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (18 * D_R + 1) * 16));
    CK(hipFuncSetAttribute((const void*)gather_d<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (18 * D_R + 1) * 16));
    CK(hipFuncSetAttribute((const void*)gather_d2, hipFuncAttributeMaxDynamicSharedMemorySize, (18 * D_R + 1) * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = (int)((n + 255) / 256), iters = 50;
    for (int v = 0; v < 18; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) {
                if (v == 0) gather_a<0><<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 1) gather_b<0><<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 2) gather_c<<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 3) gather_a<1><<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 4) gather_b<1><<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 5) gather_a<1, 1><<<(blocks + 7) / 8 * 8, 256>>>(x, nbr, ld, n, out);
                else if (v == 6) gather_a<2><<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 7) gather_d<0><<<blocks, 256, (18 * D_R + 1) * 16>>>(x, nbr, ld, n, out);
                else if (v == 8) gather_d2<<<512, 256, (18 * D_R + 1) * 16>>>(x, nbr, ld, n, out);
                else if (v == 9) gather_d<1><<<blocks, 256, (18 * D_R + 1) * 16>>>(x, nbr, ld, n, out);
                else if (v == 10) gather_d<2><<<blocks, 256, (18 * D_R + 1) * 16>>>(x, nbr, ld, n, out);
                else if (v == 11) gather_d<3><<<blocks, 256, (18 * D_R + 1) * 16>>>(x, nbr, ld, n, out);
                else if (v == 12) gather_e<0><<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 13) gather_e<1><<<blocks, 256>>>(x, nbr, ld, n, out);
                else if (v == 14) gather_a<2><<<blocks, 256>>>(x, nbr_all, ld, n, out);
                else if (v == 15) gather_a<2><<<blocks, 256>>>(x, nbr_none, ld, n, out);
                else if (v == 16) gather_e<0><<<blocks, 256>>>(x, nbr_all, ld, n, out);
                else gather_e<0><<<blocks, 256>>>(x, nbr_none, ld, n, out);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const char* nm[18] = {"A  lane=row, taps dz-major", "B  lane=(row,half), taps dz-major", "C  transposing layout", "A' lane=row, taps column-major", "B' lane=(row,half), taps column-major", "A'' = A' + XCD-contiguous tile ranges", "A3 lane=row, taps x-slab-major (dx, dy, dz)", "D  9 bands staged in LDS, taps from LDS", "D2 = D, long-lived blocks, next tile prefetched", "D without index loads (computed)", "D without band loads", "D without either (LDS + VALU only)", "E  = A3, absent taps masked off (no load)", "E index loads only", "A3, every tap present", "A3, every tap absent (all read the zero row)", "E, every tap present", "E, every tap absent (no gathers issued)"};
            if (rep) printf("%-40s %.2f us per pass (%ld rows, 27 taps x 32 B)\n", nm[v], ms * 1e3 / iters, n);
        }
    }
    return 0;
}

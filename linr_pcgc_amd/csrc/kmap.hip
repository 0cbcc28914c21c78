// Kernel-map builder: for every voxel of a sorted coordinate list, the row index of each of its 27 neighbours.
// Replaces MinkowskiEngine's coordinate hash map + kernel-map generation (see include/linr_hip.h).
// The list is already sorted by the x-major key, so a neighbour lookup is a binary search, and the three
// dz = -1,0,+1 neighbours of one (dx,dy) column are adjacent in the list: 9 searches per voxel, not 27.
#include "common.h"

__device__ __forceinline__ long long linr_key(int x, int y, int z) {
    return ((long long)(x + 1) << 42) | ((long long)(y + 1) << 21) | (long long)(z + 1);
}

__global__ __launch_bounds__(LINR_BLOCK) void kmap_keys_k(const int32_t* __restrict__ coords, int64_t n,
                                                          long long* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i < n) keys[i] = linr_key(coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]);
}

__global__ __launch_bounds__(LINR_BLOCK) void kmap_search_k(const long long* __restrict__ keys,
                                                            const int32_t* __restrict__ coords, int64_t n,
                                                            int32_t* __restrict__ nbr, int64_t ld, int64_t row_base) {
    int64_t idx = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (idx >= 9 * n) return;
    const int q = (int)(idx / n);            // (dx,dy) column: consecutive threads -> consecutive rows
    const int64_t j = idx - (int64_t)q * n;
    const int dx = q % 3 - 1, dy = q / 3 - 1;
    const int x = coords[3 * j] + dx, y = coords[3 * j + 1] + dy, z = coords[3 * j + 2];
    const long long key0 = linr_key(x, y, z - 1);
    int64_t lo = 0, hi = n;                  // lower_bound(key0)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key0) lo = mid + 1; else hi = mid;
    }
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
        const long long target = key0 + dz;  // z+1 < 2^21: never carries into y
        if (lo < n && keys[lo] < target) ++lo;   // keys are unique: at most one step per dz
        const bool hit = lo < n && keys[lo] == target;
        const int k = q + 9 * dz;
        nbr[(int64_t)k * ld + row_base + j] = hit ? (int32_t)(row_base + lo) : -1;
    }
}

__global__ __launch_bounds__(LINR_BLOCK) void kmap_validate_k(const int32_t* __restrict__ coords, int64_t n,
                                                              int32_t* __restrict__ bad) {
    int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2];
    bool ok = x >= 0 && y >= 0 && z >= 0 && x < (1 << 20) && y < (1 << 20) && z < (1 << 20);
    if (ok && i > 0) ok = linr_key(coords[3 * i - 3], coords[3 * i - 2], coords[3 * i - 1]) < linr_key(x, y, z);
    if (!ok) atomicOr(bad, 1);
}

extern "C" size_t linr_kmap_workspace_bytes(int64_t n) { return (size_t)(n < 0 ? 0 : n) * sizeof(long long); }

extern "C" int linr_kmap_build(const int32_t* coords, int64_t n, int32_t* nbr, int64_t ld, int64_t row_base,
                               void* ws, size_t ws_bytes, void* stream) {
    if (n < 0 || ld < n || row_base < 0 || row_base + n > ld) return LINR_EINVAL;
    if (row_base + n > INT32_MAX) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!coords || !nbr || !ws) return LINR_EINVAL;
    if (ws_bytes < linr_kmap_workspace_bytes(n)) return LINR_ENOSPC;
    if (((uintptr_t)ws) & 7u) return LINR_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    long long* keys = (long long*)ws;
    kmap_keys_k<<<linr_grid(n, LINR_BLOCK), LINR_BLOCK, 0, s>>>(coords, n, keys);
    kmap_search_k<<<linr_grid(9 * n, LINR_BLOCK), LINR_BLOCK, 0, s>>>(keys, coords, n, nbr, ld, row_base);
    return linr_launch_rc();
}

extern "C" int linr_kmap_validate(const int32_t* coords, int64_t n, int32_t* bad, void* stream) {
    if (n < 0 || !bad) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!coords) return LINR_EINVAL;
    kmap_validate_k<<<linr_grid(n, LINR_BLOCK), LINR_BLOCK, 0, (hipStream_t)stream>>>(coords, n, bad);
    return linr_launch_rc();
}

// 7-neighbour occupancy of every voxel (qscTensor.set_offset_tensor, models/module_utils.py:201-224; offsets in the order of
// glob_params.py:3: self, -x, +x, -y, +y, -z, +z) read off the kernel map it is a subset of: tap k = (dx+1)+3(dy+1)+9(dz+1).
__global__ __launch_bounds__(LINR_BLOCK) void kmap_offset_feat_k(const int32_t* __restrict__ nbr, int64_t ld, int64_t row_base,
                                                                 int64_t n, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (e >= n * 7) return;
    const int64_t r = e / 7;
    const int j = (int)(e % 7);
    const int taps[7] = {13, 12, 14, 10, 16, 4, 22};
    out[e] = nbr[(int64_t)taps[j] * ld + row_base + r] >= 0 ? 1.0f : 0.0f;
}

extern "C" int linr_kmap_offset_feat(const int32_t* nbr, int64_t ld, int64_t row_base, int64_t n, float* out, void* stream) {
    if (n < 0 || row_base < 0 || ld < row_base + n) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!nbr || !out) return LINR_EINVAL;
    kmap_offset_feat_k<<<linr_grid(n * 7, LINR_BLOCK), LINR_BLOCK, 0, (hipStream_t)stream>>>(nbr, ld, row_base, n, out);
    return linr_launch_rc();
}

// Child occupancy of every parent voxel (octree_level.forward, models/module_utils.py:86-110): occ[j][4dx+2dy+dz] = 1 if
// child 2*parent[j] + (dx,dy,dz) is in the sorted child list.  The two dz children of a (dx,dy) pair are adjacent in the
// x-major list: 4 searches per parent, the same primitive as the kernel map.
__global__ __launch_bounds__(LINR_BLOCK) void octree_occ_k(const long long* __restrict__ keys, int64_t m,
                                                           const int32_t* __restrict__ parent, int64_t n,
                                                           float* __restrict__ occ) {
    const int64_t idx = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (idx >= 4 * n) return;
    const int q = (int)(idx / n);            // (dx,dy) pair: consecutive threads -> consecutive parents
    const int64_t j = idx - (int64_t)q * n;
    const int dx = q >> 1, dy = q & 1;
    const long long key0 = linr_key(2 * parent[3 * j] + dx, 2 * parent[3 * j + 1] + dy, 2 * parent[3 * j + 2]);
    int64_t lo = 0, hi = m;                  // lower_bound(key0)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key0) lo = mid + 1; else hi = mid;
    }
    const bool h0 = lo < m && keys[lo] == key0;
    if (h0) ++lo;
    const bool h1 = lo < m && keys[lo] == key0 + 1;
    occ[j * 8 + 4 * dx + 2 * dy] = h0 ? 1.0f : 0.0f;
    occ[j * 8 + 4 * dx + 2 * dy + 1] = h1 ? 1.0f : 0.0f;
}

extern "C" int linr_octree_occupancy(const int32_t* child, int64_t m, const int32_t* parent, int64_t n, float* occ, void* ws,
                                     size_t ws_bytes, void* stream) {
    if (m < 0 || n < 0) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!parent || !occ || (m > 0 && (!child || !ws))) return LINR_EINVAL;
    if (ws_bytes < linr_kmap_workspace_bytes(m)) return LINR_ENOSPC;
    if (((uintptr_t)ws) & 7u) return LINR_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    long long* keys = (long long*)ws;
    if (m > 0) kmap_keys_k<<<linr_grid(m, LINR_BLOCK), LINR_BLOCK, 0, s>>>(child, m, keys);
    octree_occ_k<<<linr_grid(4 * n, LINR_BLOCK), LINR_BLOCK, 0, s>>>(keys, m, parent, n, occ);
    return linr_launch_rc();
}

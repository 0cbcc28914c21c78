// One scale of the staged decoder as ONE call (decoder.decode_one_frame's loop body, decoder.py:153-176): the kernel map of the
// coarser level's coordinates, the 7-neighbour features read off it, the 8 decode stages (stage forward, D2H, range decoder, H2D:
// linr_net_decode_stages) and the next level's coordinates (octree_level.upper_layer, models/module_utils.py:117-127: the children
// 2 p + (dx, dy, dz) of every occupied octant, sorted x-major).  Between two scales the caller only allocates the next
// workspace, so a frame's decode holds the Python GIL for a few hundred microseconds instead of ~7 ms.
#include "common.h"
#include "layout.h"
#include <hipcub/hipcub.hpp>

namespace {

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

struct DecodeWs {              // byte offsets into the caller's workspace
    size_t nbr, lo, mask, feat, occ, probs, sdev, kws, cnt, pos, keys0, keys1, cub, arena, total;
    int64_t ld;
    size_t arena_bytes, cub_bytes;
};

size_t cub_bytes_for(int64_t n) {
    size_t a = 0, b = 0;
    // the scan runs over n + 1 items (linr_decode_scale), the sort over at most 8 n keys: reserve for exactly those calls
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, a, (const int32_t*)nullptr, (int32_t*)nullptr, (int)((n > 0 ? n : 1) + 1));
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)(8 * (n > 0 ? n : 1)), 0, 63);
    return a > b ? a : b;
}

DecodeWs layout(int64_t n, int block_layers, int bf16) {
    DecodeWs w;
    w.ld = (n + 63) / 64 * 64;
    size_t cur = 0;
    auto take = [&](size_t bytes) { size_t o = cur; cur += up256(bytes); return o; };
    w.nbr = take((size_t)27 * w.ld * 4);
    w.lo = take((size_t)9 * w.ld * 4);
    w.mask = take((size_t)w.ld * 4);
    w.feat = take((size_t)n * 7 * 4);
    w.occ = take((size_t)(n + 1) * 8 * 4);
    w.probs = take((size_t)8 * n * 4);
    w.sdev = take((size_t)n);
    w.kws = take(linr_kmap_workspace_bytes(n));
    w.cnt = take((size_t)(n + 1) * 4);
    w.pos = take((size_t)(n + 1) * 4);
    w.keys0 = take((size_t)8 * n * 8);
    w.keys1 = take((size_t)8 * n * 8);
    w.cub_bytes = cub_bytes_for(n);
    w.cub = take(w.cub_bytes);
    w.arena_bytes = bf16 ? linr_net_bf16_arena_bytes(n, block_layers) : linr_net_arena_bytes(n, block_layers);
    w.arena = take(w.arena_bytes);
    w.total = cur;
    return w;
}

// cnt[r] = occupied octants of row r (cnt[n] = 0 closes the scan)
__global__ __launch_bounds__(LINR_BLOCK) void child_count_k(const float* __restrict__ occ, int64_t n, int32_t* __restrict__ cnt) {
    const int64_t r = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (r > n) return;
    int c = 0;
    if (r < n) {
        const float4 a = *reinterpret_cast<const float4*>(occ + r * 8);
        const float4 b = *reinterpret_cast<const float4*>(occ + r * 8 + 4);
        c = (a.x != 0.f) + (a.y != 0.f) + (a.z != 0.f) + (a.w != 0.f) + (b.x != 0.f) + (b.y != 0.f) + (b.z != 0.f) + (b.w != 0.f);
    }
    cnt[r] = c;
}

// key of child 2 p + (dx, dy, dz), octant index 4 dx + 2 dy + dz (module_utils.py:90-91), x-major with B bits per axis
__global__ __launch_bounds__(LINR_BLOCK) void child_keys_k(const int32_t* __restrict__ coord, const float* __restrict__ occ,
                                                           const int32_t* __restrict__ pos, int64_t n, int bits,
                                                           uint64_t* __restrict__ keys) {
    const int64_t r = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (r >= n) return;
    const uint64_t x = 2u * (uint32_t)coord[3 * r], y = 2u * (uint32_t)coord[3 * r + 1], z = 2u * (uint32_t)coord[3 * r + 2];
    int o = pos[r];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (occ[r * 8 + k] != 0.f)
            keys[o++] = ((x + (uint64_t)(k >> 2)) << (2 * bits)) | ((y + (uint64_t)((k >> 1) & 1)) << bits) | (z + (uint64_t)(k & 1));
}

__global__ __launch_bounds__(LINR_BLOCK) void keys_to_coord_k(const uint64_t* __restrict__ keys, const int32_t* __restrict__ total,
                                                              int bits, int32_t* __restrict__ xyz, int64_t cap) {
    const int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    const int64_t m = *total < cap ? *total : cap;
    if (i >= m) return;
    const uint64_t k = keys[i], msk = ((uint64_t)1 << bits) - 1;
    xyz[3 * i] = (int32_t)(k >> (2 * bits));
    xyz[3 * i + 1] = (int32_t)((k >> bits) & msk);
    xyz[3 * i + 2] = (int32_t)(k & msk);
}

}  // namespace

extern "C" size_t linr_decode_scale_ws_bytes(int64_t n, int32_t block_layers, int32_t bf16) {
    if (n < 0 || block_layers < 1) return 0;
    return layout(n, block_layers, bf16 ? 1 : 0).total + 256;
}

extern "C" int linr_decode_scale(const int32_t* coord, int64_t n, int32_t scale_idx, int32_t model_scale_num, int32_t block_layers,
                                 int32_t child_bits, const float* params, const uint8_t* codes, float min_param, float max_param,
                                 const uint8_t* const* streams_h, const int64_t* stream_len_h, void* ws, size_t ws_bytes,
                                 float* p_pinned, uint8_t* s_pinned, int32_t* child_xyz, int64_t child_cap, int64_t* child_n_h,
                                 void* stream) {
    if (n < 0 || !child_n_h || child_bits < 1 || child_bits > 21 || block_layers < 1) return LINR_EINVAL;
    *child_n_h = 0;
    if (n == 0) return 0;
    if (!coord || (!params && !codes) || !streams_h || !stream_len_h || !ws || !p_pinned || !s_pinned || !child_xyz) return LINR_EINVAL;
    if (child_cap < 0 || n >= ((int64_t)1 << 27)) return LINR_EINVAL;
    if (((uintptr_t)ws) & 255u) return LINR_EALIGN;
    const DecodeWs w = layout(n, block_layers, codes ? 1 : 0);
    if (ws_bytes < w.total) return LINR_ENOSPC;
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)ws;
    int32_t* nbr = (int32_t*)(base + w.nbr);
    int32_t* lo = (int32_t*)(base + w.lo);
    uint32_t* mask = (uint32_t*)(base + w.mask);
    float* feat = (float*)(base + w.feat);
    float* occ_buf = (float*)(base + w.occ);
    float* occ = occ_buf + 8;                                   // zero row in front (LINR_FRAME_OCC_PADDED)
    float* probs = (float*)(base + w.probs);
    uint8_t* s_dev = (uint8_t*)(base + w.sdev);
    int32_t* cnt = (int32_t*)(base + w.cnt);
    int32_t* pos = (int32_t*)(base + w.pos);
    uint64_t* keys0 = (uint64_t*)(base + w.keys0);
    uint64_t* keys1 = (uint64_t*)(base + w.keys1);
    // kernel map of this level (padding columns of nbr / lo / mask: no neighbour), its compressed form, the scale context's features
    int rc = linr_hip_rc(hipMemsetAsync(nbr, 0xFF, (size_t)27 * w.ld * 4, s));
    if (rc) return rc;
    rc = linr_hip_rc(hipMemsetAsync(lo, 0, w.mask + (size_t)w.ld * 4 - w.lo, s));            // lo and mask are adjacent
    if (rc) return rc;
    linr_poison_hook(s, 15);
    rc = linr_kmap_build(coord, n, nbr, w.ld, 0, base + w.kws, linr_kmap_workspace_bytes(n), stream);
    if (rc) return rc;
    rc = linr_kmap_compress(nbr, w.ld, n, lo, mask, w.ld, stream);
    if (rc) return rc;
    rc = linr_kmap_offset_feat(nbr, w.ld, 0, n, feat, stream);
    if (rc) return rc;
    rc = linr_hip_rc(hipMemsetAsync(occ_buf, 0, (size_t)(n + 1) * 8 * 4, s));
    if (rc) return rc;
    int64_t row_off[2] = {0, n};
    int32_t sidx[1] = {scale_idx};
    linr_frame f;
    f.rows = n; f.n_scales = 1; f.model_scale_num = model_scale_num; f.block_layers = block_layers; f.flags = LINR_FRAME_OCC_PADDED;
    f.row_off_h = row_off; f.scale_idx_h = sidx; f.nbr = nbr; f.nbr_ld = w.ld; f.nbr_lo = lo; f.nbr_mask = mask;
    f.offset_feat = feat; f.occ = occ; f.nbr8t = nullptr;
    rc = linr_net_decode_stages(&f, params, codes, min_param, max_param, base + w.arena, w.arena_bytes, streams_h, stream_len_h, probs,
                                p_pinned, s_pinned, s_dev, stream);
    if (rc) return rc;
    // upper_layer: children of the occupied octants, sorted x-major
    linr_poison_hook(s, 15);
    child_count_k<<<linr_grid(n + 1, LINR_BLOCK), LINR_BLOCK, 0, s>>>(occ, n, cnt);
    size_t cb = w.cub_bytes;
    rc = linr_hip_rc(hipcub::DeviceScan::ExclusiveSum(base + w.cub, cb, cnt, pos, (int)(n + 1), s));
    if (rc) return rc;
    int32_t total = 0;
    rc = linr_hip_rc(hipMemcpyAsync(&total, pos + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (rc) return rc;
    child_keys_k<<<linr_grid(n, LINR_BLOCK), LINR_BLOCK, 0, s>>>(coord, occ, pos, n, child_bits, keys0);
    rc = linr_hip_rc(hipStreamSynchronize(s));                  // `total` is on the host now
    if (rc) return rc;
    if (total > child_cap) return LINR_ENOSPC;
    if (total > 0) {
        cb = w.cub_bytes;
        rc = linr_hip_rc(hipcub::DeviceRadixSort::SortKeys(base + w.cub, cb, keys0, keys1, (int)total, 0, 3 * child_bits, s));
        if (rc) return rc;
        keys_to_coord_k<<<linr_grid(total, LINR_BLOCK), LINR_BLOCK, 0, s>>>(keys1, pos + n, child_bits, child_xyz, child_cap);
    }
    *child_n_h = total;
    return linr_launch_rc();
}

// ---- sorted unique coordinate list, optionally of the parents (coords >> shift); one octree level as one call ------------------------
// torch.unique(dim=0) of custom_dataset.py:271-282 (the input cloud, shift 0) and of octree_level.forward (models/module_utils.py:
// 92,103: parent = unique(floor(child / 2)), shift 1): compact x-major keys (x << 2b | y << b | z with b = the bits a coordinate
// needs, so the radix sort runs over 3 b bits, not 63), radix sort, unique, decode.  Before these entries the host mirror spent ~30
// small torch launches per octree level on it (tools/stage_split.py: 2.5 ms per loot10 frame).
namespace {
// K = uint32_t when the compact key fits (3 b <= 32: every level of a 10-bit cloud, the parent levels of an 11-bit one), else uint64_t
template <typename K>
__global__ __launch_bounds__(LINR_BLOCK) void su_keys_k(const int32_t* __restrict__ c, int64_t n, const int32_t* __restrict__ origin, int shift,
                                                        int b, K* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int ox = origin ? origin[0] : 0, oy = origin ? origin[1] : 0, oz = origin ? origin[2] : 0;
    const uint64_t x = (uint64_t)((c[3 * i] - ox) >> shift), y = (uint64_t)((c[3 * i + 1] - oy) >> shift), z = (uint64_t)((c[3 * i + 2] - oz) >> shift);
    keys[i] = (K)((x << (2 * b)) | (y << b) | z);
}
template <typename K>
__global__ __launch_bounds__(LINR_BLOCK) void su_decode_k(const K* __restrict__ keys, const int* __restrict__ num, int64_t cap, int b,
                                                          int32_t* __restrict__ out, int64_t* __restrict__ count) {
    const int64_t m = *num < cap ? *num : cap;
    const int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i == 0) *count = m;
    if (i >= m) return;
    const uint64_t k = keys[i], mk = ((uint64_t)1 << b) - 1;
    out[3 * i] = (int32_t)(k >> (2 * b));
    out[3 * i + 1] = (int32_t)((k >> b) & mk);
    out[3 * i + 2] = (int32_t)(k & mk);
}
// child occupancy from the sorted compact CHILD keys (b bits per coordinate) and the sorted unique compact PARENT keys (b - 1 bits):
// octree_occ_k of kmap.hip with the parent count read on the device
template <typename KC, typename KP>
__global__ __launch_bounds__(LINR_BLOCK) void su_occ_k(const KC* __restrict__ ck, int64_t m, const KP* __restrict__ pk,
                                                       const int* __restrict__ num, int b, float* __restrict__ occ) {
    const int64_t n = *num;
    const int64_t idx = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (idx >= 4 * n) return;
    const int q = (int)(idx & 3);
    const int64_t j = idx >> 2;
    const int dx = q >> 1, dy = q & 1, pb = b - 1;
    const uint64_t k = pk[j], mk = ((uint64_t)1 << pb) - 1;
    const uint64_t px = k >> (2 * pb), py = (k >> pb) & mk, pz = k & mk;
    const uint64_t key0 = ((2 * px + dx) << (2 * b)) | ((2 * py + dy) << b) | (2 * pz);
    int64_t lo = 0, hi = m;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((uint64_t)ck[mid] < key0) lo = mid + 1; else hi = mid;
    }
    const bool h0 = lo < m && (uint64_t)ck[lo] == key0;
    if (h0) ++lo;
    const bool h1 = lo < m && (uint64_t)ck[lo] == key0 + 1;
    occ[j * 8 + 4 * dx + 2 * dy] = h0 ? 1.0f : 0.0f;
    occ[j * 8 + 4 * dx + 2 * dy + 1] = h1 ? 1.0f : 0.0f;
}
size_t su_cub_bytes(int64_t n) {
    size_t a = 0, b = 0, c = 0, d = 0;
    const int m = (int)(n > 0 ? n : 1);
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, m, 0, 60);
    (void)hipcub::DeviceSelect::Unique(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int*)nullptr, m);
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, c, (const uint32_t*)nullptr, (uint32_t*)nullptr, m, 0, 32);
    (void)hipcub::DeviceSelect::Unique(nullptr, d, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int*)nullptr, m);
    a = a > b ? a : b;
    c = c > d ? c : d;
    return a > c ? a : c;
}
// keys of (coords >> shift) into k0, sorted into k1, unique back into k0, *num = how many
template <typename K>
int su_sort_unique(const int32_t* coords, int64_t n, const int32_t* origin, int shift, int b, K* k0, K* k1, int* num, void* cub, size_t cb,
                   hipStream_t s) {
    su_keys_k<K><<<linr_grid(n, LINR_BLOCK), LINR_BLOCK, 0, s>>>(coords, n, origin, shift, b, k0);
    int rc = linr_hip_rc(hipcub::DeviceRadixSort::SortKeys(cub, cb, k0, k1, (int)n, 0, 3 * b, s));
    if (rc) return rc;
    return linr_hip_rc(hipcub::DeviceSelect::Unique(cub, cb, k1, k0, num, (int)n, s));
}
template <typename K>
int su_unique_coords(const int32_t* coords, int64_t n, const int32_t* origin, int shift, int b, char* base, size_t kb, int32_t* out,
                     int64_t* count, hipStream_t s) {
    K* k0 = (K*)base;
    K* k1 = (K*)(base + kb);
    int* num = (int*)(base + 3 * kb);
    int rc = su_sort_unique<K>(coords, n, origin, shift, b, k0, k1, num, base + 3 * kb + 256, su_cub_bytes(n), s);
    if (rc) return rc;
    su_decode_k<K><<<linr_grid(n, LINR_BLOCK), LINR_BLOCK, 0, s>>>(k0, num, n, b, out, count);
    return linr_launch_rc();
}
template <typename KC, typename KP>
int su_level(const int32_t* child, int64_t m, int b, char* base, size_t kb, int32_t* parent, float* occ, int64_t* count, hipStream_t s) {
    KP* k0 = (KP*)base;
    KP* k1 = (KP*)(base + kb);
    KC* ck = (KC*)(base + 2 * kb);
    int* num = (int*)(base + 3 * kb);
    su_keys_k<KC><<<linr_grid(m, LINR_BLOCK), LINR_BLOCK, 0, s>>>(child, m, nullptr, 0, b, ck);
    int rc = su_sort_unique<KP>(child, m, nullptr, 1, b - 1, k0, k1, num, base + 3 * kb + 256, su_cub_bytes(m), s);
    if (rc) return rc;
    su_decode_k<KP><<<linr_grid(m, LINR_BLOCK), LINR_BLOCK, 0, s>>>(k0, num, m, b - 1, parent, count);
    su_occ_k<KC, KP><<<linr_grid(4 * m, LINR_BLOCK), LINR_BLOCK, 0, s>>>(ck, m, k0, num, b, occ);
    return linr_launch_rc();
}
}  // namespace

extern "C" size_t linr_sort_unique_workspace_bytes(int64_t n) {
    if (n < 0) n = 0;
    return 3 * up256((size_t)n * 8) + 256 + up256(su_cub_bytes(n));
}

// coords: int32 [n,3], every (coordinate - origin) in [0, 2^coord_bits), coord_bits <= 20, any order, duplicates allowed; origin:
// device int32 [3] subtracted from every row first (the frame's coord_data_min, custom_dataset.py:276-279) or nullptr.  out: int32 [n,3]
// (room for n rows); *count (device int64) = number of distinct rows of (coords >> shift), written to out in x-major order.
extern "C" int linr_coords_sort_unique(const int32_t* coords, int64_t n, const int32_t* origin, int32_t shift, int32_t coord_bits, int32_t* out,
                                       int64_t* count, void* ws, size_t ws_bytes, void* stream) {
    if (n < 0 || shift < 0 || shift > 19 || coord_bits < 1 || coord_bits > 20 || n > INT32_MAX) return LINR_EINVAL;
    if (!count) return LINR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return linr_hip_rc(hipMemsetAsync(count, 0, sizeof(int64_t), s));
    if (!coords || !out || !ws) return LINR_EINVAL;
    if (ws_bytes < linr_sort_unique_workspace_bytes(n)) return LINR_ENOSPC;
    if (((uintptr_t)ws) & 255u) return LINR_EALIGN;
    const int b = coord_bits - shift > 1 ? coord_bits - shift : 1;
    const size_t kb = up256((size_t)n * 8);
    if (3 * b <= 32) return su_unique_coords<uint32_t>(coords, n, origin, shift, b, (char*)ws, kb, out, count, s);
    return su_unique_coords<uint64_t>(coords, n, origin, shift, b, (char*)ws, kb, out, count, s);
}

// One octree level (octree_level.forward, models/module_utils.py:86-110) as one call: child int32 [m,3] sorted x-major and unique,
// coordinates in [0, 2^coord_bits); parent [m,3] / occ [m,8] have room for m rows; *count (device int64) = the number of parents.
extern "C" int linr_octree_level(const int32_t* child, int64_t m, int32_t coord_bits, int32_t* parent, float* occ, int64_t* count, void* ws,
                                 size_t ws_bytes, void* stream) {
    if (m < 0 || coord_bits < 1 || coord_bits > 20 || m > INT32_MAX) return LINR_EINVAL;
    if (!count) return LINR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (m == 0) return linr_hip_rc(hipMemsetAsync(count, 0, sizeof(int64_t), s));
    if (!child || !parent || !occ || !ws) return LINR_EINVAL;
    if (ws_bytes < linr_sort_unique_workspace_bytes(m)) return LINR_ENOSPC;
    if (((uintptr_t)ws) & 255u) return LINR_EALIGN;
    const int b = coord_bits > 1 ? coord_bits : 2;
    const size_t kb = up256((size_t)m * 8);
    if (3 * b <= 32) return su_level<uint32_t, uint32_t>(child, m, b, (char*)ws, kb, parent, occ, count, s);
    if (3 * (b - 1) <= 32) return su_level<uint64_t, uint32_t>(child, m, b, (char*)ws, kb, parent, occ, count, s);
    return su_level<uint64_t, uint64_t>(child, m, b, (char*)ws, kb, parent, occ, count, s);
}

// Per-axis minimum and maximum of a coordinate list: out[0..2] = min, out[3..5] = max (device int32 [6]); what custom_dataset.py:
// 276-279 computes with tensor reductions (torch's int64 column reductions cost 0.32 ms each on a 784 k-point frame).
namespace {
__global__ __launch_bounds__(LINR_BLOCK) void minmax_init_k(int32_t* out) {
    if (threadIdx.x < 3) out[threadIdx.x] = INT32_MAX;
    else if (threadIdx.x < 6) out[threadIdx.x] = INT32_MIN;
}
// The flat int32 stream read coalesced: with a grid whose thread count is a multiple of 3 every thread stays on one axis.  Per block
// one LDS reduction, then 6 global atomics (integer min / max: the order of the atomics does not matter).
__global__ __launch_bounds__(LINR_BLOCK) void minmax_k(const int32_t* __restrict__ c, int64_t n3, int32_t* __restrict__ out) {
    __shared__ int s_lo[3], s_hi[3];
    if (threadIdx.x < 3) { s_lo[threadIdx.x] = INT32_MAX; s_hi[threadIdx.x] = INT32_MIN; }
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x, stride = (int64_t)gridDim.x * LINR_BLOCK;      // stride % 3 == 0
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int64_t i = t0; i < n3; i += stride) { const int v = c[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    const int a = (int)(t0 % 3);
    if (lo <= hi) { atomicMin(&s_lo[a], lo); atomicMax(&s_hi[a], hi); }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(out + threadIdx.x, s_lo[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(out + threadIdx.x, s_hi[threadIdx.x - 3]);
}
}  // namespace

extern "C" int linr_coords_minmax(const int32_t* coords, int64_t n, int32_t* out, void* stream) {
    if (n < 1 || !coords || !out) return LINR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    minmax_init_k<<<1, LINR_BLOCK, 0, s>>>(out);
    int64_t blocks = (3 * n + LINR_BLOCK * 16 - 1) / (LINR_BLOCK * 16);
    blocks = (blocks < 768 ? blocks : 768);
    blocks = (blocks + 2) / 3 * 3;          // thread count a multiple of 3: one axis per thread
    minmax_k<<<(unsigned)blocks, LINR_BLOCK, 0, s>>>(coords, 3 * n, out);
    return linr_launch_rc();
}

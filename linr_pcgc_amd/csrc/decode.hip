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

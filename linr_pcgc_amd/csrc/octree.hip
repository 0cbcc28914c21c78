// All octree levels of a frame as ONE call, without a sort (include/linr_hip.h: linr_octree_levels).
//
// MyDataset.handle_data (datautils/custom_dataset.py:289-344) walks the levels with octree_level.forward (models/module_utils.py:86-110):
// parent = unique(floor(child / 2)), occupancy[parent][4 dx + 2 dy + dz] = child 2 parent + (dx, dy, dz) present.  The per-level entry
// (csrc/decode.hip: linr_octree_level) re-SORTS every level's parent keys with a radix / merge sort (~20 launches and one host read of
// the count per level: 1.24 ms of a 784 k-point frame's 1.8 ms of staging, 19 k merge-sort launches in a sequence's kernel statistics).
// But the parents of a sorted unique child list need no sort: their compact x-major keys (X << 2 pb | Y << pb | Z, pb bits per
// coordinate) index a BITMAP of 2^(3 pb) bits - 16 MB for the finest parent level of a 10-bit cloud, an eighth of it per level above -
// and the set bits of that bitmap, read in word order, ARE the sorted unique parent list:
//   mark   every child ORs its parent's bit (integer atomics on bits: the result does not depend on their order)
//   count  popcount per word, one exclusive scan (hipcub) -> every word's first output row and the level's row count
//   emit   every word writes its parents' coordinates and compact keys (the child keys of the next level)
//   occ    every parent looks its 8 children up in the sorted child keys (4 binary searches: the dz pair is adjacent)
// Counts and output offsets stay on the device from level to level (grids are sized by upper bounds, threads beyond the live count
// leave), so the host reads all counts ONCE behind the last level.  Everything is integer work; the output is bit-identical to the
// sort-based entry (tests/test_gpu_ops.py).
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace {

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

struct LvState {           // device-resident chain state
    int64_t n_child;       // rows of the current child level
    int64_t child_off;     // row offset of the current child level inside `parents` (level 0: the caller's child list)
    int64_t out_off;       // row offset at which the current level's parents are written
};

__global__ void lv_init_k(LvState* st, const int64_t* m_dev, int64_t m) {
    st->n_child = m_dev ? (*m_dev < m ? *m_dev : m) : m;
    st->child_off = 0;
    st->out_off = 0;
}

// compact x-major keys of the level-0 children (b bits per coordinate)
__global__ __launch_bounds__(LINR_BLOCK) void lv_keys0_k(const int32_t* __restrict__ c, const LvState* __restrict__ st, int b,
                                                         uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i >= st->n_child) return;
    keys[i] = ((uint64_t)c[3 * i] << (2 * b)) | ((uint64_t)c[3 * i + 1] << b) | (uint64_t)c[3 * i + 2];
}

// every child sets the bit of its parent: key (b bits per coordinate) -> parent key (b - 1 bits per coordinate)
__global__ __launch_bounds__(LINR_BLOCK) void lv_mark_k(const uint64_t* __restrict__ ck, const LvState* __restrict__ st, int b,
                                                        uint32_t* __restrict__ bitmap) {
    const int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i >= st->n_child) return;
    const uint64_t k = ck[i], mk = ((uint64_t)1 << b) - 1;
    const uint64_t x = k >> (2 * b), y = (k >> b) & mk, z = k & mk;
    const int pb = b - 1;
    const uint64_t pk = ((x >> 1) << (2 * pb)) | ((y >> 1) << pb) | (z >> 1);
    atomicOr(bitmap + (pk >> 5), 1u << (pk & 31));
}

__global__ __launch_bounds__(LINR_BLOCK) void lv_count_k(const uint32_t* __restrict__ bitmap, int64_t words, int32_t* __restrict__ cnt) {
    const int64_t w = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (w > words) return;
    cnt[w] = w < words ? __popc(bitmap[w]) : 0;              // cnt[words] = 0 closes the scan: pos[words] = the level's row count
}

// word w writes its parents (ascending bit = ascending key): coordinates, compact keys; the thread of the closing element publishes the
// level's count
__global__ __launch_bounds__(LINR_BLOCK) void lv_emit_k(const uint32_t* __restrict__ bitmap, const int32_t* __restrict__ pos, int64_t words,
                                                        int pb, const LvState* __restrict__ st, int32_t* __restrict__ parents,
                                                        uint64_t* __restrict__ pkeys, int64_t* __restrict__ count_out, int64_t cap_rows) {
    const int64_t w = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (w > words) return;
    if (w == words) { *count_out = pos[words]; return; }
    uint32_t bits = bitmap[w];
    if (!bits) return;
    int64_t r = pos[w];
    const int64_t base = st->out_off;
    const uint64_t mk = ((uint64_t)1 << pb) - 1;
    while (bits) {
        const int j = __ffs(bits) - 1;
        bits &= bits - 1;
        const uint64_t k = ((uint64_t)w << 5) | (uint64_t)j;
        if (base + r < cap_rows) {                        // (cannot fail: cap_rows is the sum of the levels' upper bounds)
            int32_t* o = parents + 3 * (base + r);
            o[0] = (int32_t)(k >> (2 * pb)); o[1] = (int32_t)((k >> pb) & mk); o[2] = (int32_t)(k & mk);
            pkeys[r] = k;
        }
        ++r;
    }
}

// child occupancy: 4 threads per parent, one per (dx, dy); the two dz children are neighbours in the sorted child keys
__global__ __launch_bounds__(LINR_BLOCK) void lv_occ_k(const uint64_t* __restrict__ ck, const uint64_t* __restrict__ pk,
                                                       const int64_t* __restrict__ n_parent, const LvState* __restrict__ st, int b,
                                                       float* __restrict__ occ, int64_t cap_rows) {
    const int64_t n = *n_parent, m = st->n_child;
    const int64_t idx = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (idx >= 4 * n) return;
    const int q = (int)(idx & 3);
    const int64_t j = idx >> 2;
    if (st->out_off + j >= cap_rows) return;
    const int dx = q >> 1, dy = q & 1, pb = b - 1;
    const uint64_t k = pk[j], mk = ((uint64_t)1 << pb) - 1;
    const uint64_t px = k >> (2 * pb), py = (k >> pb) & mk, pz = k & mk;
    const uint64_t key0 = ((2 * px + dx) << (2 * b)) | ((2 * py + dy) << b) | (2 * pz);
    int64_t lo = 0, hi = m;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ck[mid] < key0) lo = mid + 1; else hi = mid;
    }
    const bool h0 = lo < m && ck[lo] == key0;
    if (h0) ++lo;
    const bool h1 = lo < m && ck[lo] == key0 + 1;
    float* o = occ + (st->out_off + j) * 8 + 4 * dx + 2 * dy;
    o[0] = h0 ? 1.0f : 0.0f;
    o[1] = h1 ? 1.0f : 0.0f;
}

// the level just written becomes the child level of the next one
__global__ void lv_next_k(LvState* st, const int64_t* count) {
    st->child_off = st->out_off;
    st->n_child = *count;
    st->out_off += *count;
}

struct LvPlan {
    int levels;
    int64_t cap[24];           // upper bound of level l's parent count
    int64_t words[24];         // bitmap words of level l
    int64_t cap_rows;          // sum of the caps
    size_t off_state, off_keys0, off_keys1, off_cnt, off_pos, off_cub, off_bitmap, total;
    size_t cub_bytes, bitmap_bytes;
};

bool lv_plan(int64_t m, int coord_bits, int max_levels, LvPlan& p) {
    if (m < 0 || coord_bits < 2 || coord_bits > 11 || max_levels < 1) return false;
    p.levels = max_levels < coord_bits - 1 ? max_levels : coord_bits - 1;         // parent coordinates keep >= 1 bit
    int64_t prev = m;
    p.cap_rows = 0;
    int64_t max_words = 1;
    size_t bm = 0;
    for (int l = 0; l < p.levels; ++l) {
        const int pb = coord_bits - l - 1;
        const int64_t cells = (int64_t)1 << (3 * pb);
        p.cap[l] = prev < cells ? prev : cells;
        p.words[l] = (cells + 31) >> 5;
        prev = p.cap[l];
        p.cap_rows += p.cap[l];
        if (p.words[l] > max_words) max_words = p.words[l];
        bm += up256((size_t)p.words[l] * 4);
    }
    p.bitmap_bytes = bm;
    size_t cb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, cb, (const int32_t*)nullptr, (int32_t*)nullptr, (int)(max_words + 1));
    p.cub_bytes = cb;
    size_t cur = 0;
    auto take = [&](size_t bytes) { size_t o = cur; cur += up256(bytes); return o; };
    p.off_state = take(sizeof(LvState));
    p.off_keys0 = take((size_t)(m > 0 ? m : 1) * 8);
    p.off_keys1 = take((size_t)(m > 0 ? m : 1) * 8);
    p.off_cnt = take((size_t)(max_words + 1) * 4);
    p.off_pos = take((size_t)(max_words + 1) * 4);
    p.off_cub = take(cb);
    p.off_bitmap = take(bm);
    p.total = cur;
    return true;
}

}  // namespace

extern "C" int64_t linr_octree_levels_rows(int64_t m, int32_t coord_bits, int32_t max_levels) {
    LvPlan p;
    if (!lv_plan(m, coord_bits, max_levels, p)) return -1;
    return p.cap_rows;
}

extern "C" size_t linr_octree_levels_workspace_bytes(int64_t m, int32_t coord_bits, int32_t max_levels) {
    LvPlan p;
    if (!lv_plan(m, coord_bits, max_levels, p)) return 0;
    return p.total;
}

extern "C" int32_t linr_octree_levels_count(int32_t coord_bits, int32_t max_levels) {
    LvPlan p;
    if (!lv_plan(0, coord_bits, max_levels, p)) return -1;
    return p.levels;
}

extern "C" int linr_octree_levels(const int32_t* child, int64_t m, const int64_t* m_dev, int32_t coord_bits, int32_t max_levels,
                                  int32_t* parents, float* occ, int64_t* counts, void* ws, size_t ws_bytes, void* stream) {
    LvPlan p;
    if (!lv_plan(m, coord_bits, max_levels, p) || m > INT32_MAX) return LINR_EINVAL;
    if (!counts) return LINR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (m == 0) return linr_hip_rc(hipMemsetAsync(counts, 0, sizeof(int64_t) * p.levels, s));
    if (!child || !parents || !occ || !ws) return LINR_EINVAL;
    if (ws_bytes < p.total) return LINR_ENOSPC;
    if (((uintptr_t)ws) & 255u) return LINR_EALIGN;
    char* base = (char*)ws;
    LvState* st = (LvState*)(base + p.off_state);
    uint64_t* keys[2] = {(uint64_t*)(base + p.off_keys0), (uint64_t*)(base + p.off_keys1)};
    int32_t* cnt = (int32_t*)(base + p.off_cnt);
    int32_t* pos = (int32_t*)(base + p.off_pos);
    int rc = linr_hip_rc(hipMemsetAsync(base + p.off_bitmap, 0, p.bitmap_bytes, s));          // every level's bitmap in one fill
    if (rc) return rc;
    lv_init_k<<<1, 1, 0, s>>>(st, m_dev, m);
    lv_keys0_k<<<linr_grid(m, LINR_BLOCK), LINR_BLOCK, 0, s>>>(child, st, coord_bits, keys[0]);
    size_t bm_off = p.off_bitmap;
    int64_t child_cap = m;
    for (int l = 0; l < p.levels; ++l) {
        const int b = coord_bits - l, pb = b - 1;
        uint32_t* bitmap = (uint32_t*)(base + bm_off);
        bm_off += up256((size_t)p.words[l] * 4);
        const uint64_t* ck = keys[l & 1];
        uint64_t* pk = keys[(l + 1) & 1];
        const int64_t W = p.words[l];
        lv_mark_k<<<linr_grid(child_cap, LINR_BLOCK), LINR_BLOCK, 0, s>>>(ck, st, b, bitmap);
        lv_count_k<<<linr_grid(W + 1, LINR_BLOCK), LINR_BLOCK, 0, s>>>(bitmap, W, cnt);
        size_t cb = p.cub_bytes;
        rc = linr_hip_rc(hipcub::DeviceScan::ExclusiveSum(base + p.off_cub, cb, cnt, pos, (int)(W + 1), s));
        if (rc) return rc;
        lv_emit_k<<<linr_grid(W + 1, LINR_BLOCK), LINR_BLOCK, 0, s>>>(bitmap, pos, W, pb, st, parents, pk, counts + l, p.cap_rows);
        lv_occ_k<<<linr_grid(4 * p.cap[l], LINR_BLOCK), LINR_BLOCK, 0, s>>>(ck, pk, counts + l, st, b, occ, p.cap_rows);
        lv_next_k<<<1, 1, 0, s>>>(st, counts + l);
        child_cap = p.cap[l];
    }
    return linr_launch_rc();
}

// Occupancy loss (sigmoid + BCE in bits) and the fused Adam step.  Reference call sites: include/linr_hip.h.
#include "common.h"

#define BCE_ROWS_PER_BLOCK 1024   // 4 rows per lane

// p = sigmoid(z); bits = -(t*max(log p,-100) + (1-t)*max(log(1-p),-100)) / ln 2  (torch BCELoss semantics).
// Per-block partial sums in double (fixed intra-block tree) -> slab; a single-wave second pass adds the slab into
// *bits_acc in ascending block order.
__global__ __launch_bounds__(LINR_BLOCK) void bce_bits_fwd_k(const float* __restrict__ z, const float* __restrict__ target,
                                                             int target_ld, int64_t n, float* __restrict__ p_out,
                                                             double* __restrict__ partial) {
    __shared__ double sred[LINR_BLOCK];
    double local = 0.0;
    const int64_t base = (int64_t)blockIdx.x * BCE_ROWS_PER_BLOCK;
#pragma unroll
    for (int q = 0; q < BCE_ROWS_PER_BLOCK / LINR_BLOCK; ++q) {
        const int64_t row = base + q * LINR_BLOCK + threadIdx.x;
        if (row < n) {
            const float zz = z[row];
            const float p = 1.0f / (1.0f + expf(-zz));
            const float t = target[row * target_ld];
            const float lp = fmaxf(logf(p), -100.0f);
            const float lq = fmaxf(logf(1.0f - p), -100.0f);
            const float nats = (t - 1.0f) * lq - t * lp;
            local += (double)nats;
            p_out[row] = p;
        }
    }
    sred[threadIdx.x] = local;
    __syncthreads();
    for (int s = LINR_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sred[threadIdx.x] += sred[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sred[0];
}

// one block of 1024 threads; thread t sums partials t, t+1024, ..., then a fixed shuffle tree per wave and the 16 waves in
// order: the association is fixed => bit-reproducible
#define BITS_FINISH_THREADS 1024
__global__ __launch_bounds__(BITS_FINISH_THREADS) void bce_bits_finish_k(const double* __restrict__ partial, int nblocks,
                                                                        double* __restrict__ bits_acc) {
    __shared__ double sred[BITS_FINISH_THREADS / 64];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += BITS_FINISH_THREADS) s += partial[b];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sred[0];
        for (int w = 1; w < BITS_FINISH_THREADS / 64; ++w) t += sred[w];
        *bits_acc += t * 1.4426950408889634;   // 1 / ln 2
    }
}

// torch: binary_cross_entropy_backward  g_p = g * (p - t) / max((1-p)*p, 1e-12), then sigmoid backward * p*(1-p)
__global__ __launch_bounds__(LINR_BLOCK) void bce_bits_bwd_k(const float* __restrict__ p, const float* __restrict__ target,
                                                             int target_ld, int64_t n, float gscale,
                                                             float* __restrict__ gz) {
    const int64_t row = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (row >= n) return;
    const float pp = p[row], t = target[row * target_ld];
    const float gp = gscale * (pp - t) / fmaxf((1.0f - pp) * pp, 1e-12f);
    gz[row] = gp * ((1.0f - pp) * pp);
}

int linr_bits_finish_launch(const double* partial, int count, double* bits_acc, hipStream_t s) {
    bce_bits_finish_k<<<1, BITS_FINISH_THREADS, 0, s>>>(partial, count, bits_acc);
    return linr_launch_rc();
}

extern "C" size_t linr_bce_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    return (size_t)linr_grid(n, BCE_ROWS_PER_BLOCK) * sizeof(double);
}

extern "C" int linr_bce_bits_fwd(const float* z, const float* target, int32_t target_ld, int64_t n, float* p,
                                 double* bits_acc, void* ws, size_t ws_bytes, void* stream) {
    if (n < 0 || target_ld < 1) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!z || !target || !p || !bits_acc || !ws) return LINR_EINVAL;
    if (ws_bytes < linr_bce_workspace_bytes(n)) return LINR_ENOSPC;
    if (((uintptr_t)ws) & 7u) return LINR_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)linr_grid(n, BCE_ROWS_PER_BLOCK);
    bce_bits_fwd_k<<<nb, LINR_BLOCK, 0, s>>>(z, target, target_ld, n, p, (double*)ws);
    bce_bits_finish_k<<<1, BITS_FINISH_THREADS, 0, s>>>((const double*)ws, nb, bits_acc);
    return linr_launch_rc();
}

extern "C" int linr_bce_bits_bwd(const float* p, const float* target, int32_t target_ld, int64_t n, float gscale,
                                 float* gz, void* stream) {
    if (n < 0 || target_ld < 1) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!p || !target || !gz) return LINR_EINVAL;
    bce_bits_bwd_k<<<linr_grid(n, LINR_BLOCK), LINR_BLOCK, 0, (hipStream_t)stream>>>(p, target, target_ld, n, gscale, gz);
    return linr_launch_rc();
}

// torch.optim.Adam single-tensor update (amsgrad=False, maximize=False), L2 weight decay folded into the gradient.
// Optional per-range schedule (LinrAdamRanges): torch.optim.Adam skips a parameter whose .grad is None and keeps a step
// counter per parameter.  The scale-context MLP of a scale that no frame so far has contained (custom_dataset.py:325 stops
// early on min_point_num) has no gradient yet: its 392 parameters are left untouched - no weight decay, no moment decay.
// Once it has had one, the pinned torch 1.13.1 keeps a zero tensor in .grad (zero_grad() of main.py:320), so it is updated on
// every step with its own step count.  Range r covers [begin + r*len, begin + (r+1)*len); everything outside uses the global
// scalars; `active` and the per-range scalars come from the caller's counters (linr_net_train_step).
__global__ __launch_bounds__(LINR_BLOCK) void adam_k(float* __restrict__ params, const float* __restrict__ grads,
                                                     float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                     float step_size, float bc2_sqrt, float beta1, float omb1, float beta2,
                                                     float omb2, float eps, float wd, LinrAdamRanges rg) {
    const int64_t i = (int64_t)blockIdx.x * LINR_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (rg.count > 0 && i >= rg.begin && i < rg.begin + (int64_t)rg.count * rg.len) {
        const int r = (int)((i - rg.begin) / rg.len);
        if (!rg.active[r]) return;
        step_size = rg.step_size[r];
        bc2_sqrt = rg.bc2_sqrt[r];
    }
    float mi = m[i], vi = v[i];
    params[i] = linr_adam_update(params[i], grads[i], mi, vi, step_size, bc2_sqrt, beta1, omb1, beta2, omb2, eps, wd);
    m[i] = mi;
    v[i] = vi;
}

int linr_adam_launch(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double step_size,
                     double bc2_sqrt, double beta1, double beta2, double eps, double weight_decay,
                     const LinrAdamRanges* rg, hipStream_t s) {
    if (n < 0) return LINR_EINVAL;
    if (n == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq) return LINR_EINVAL;
    LinrAdamRanges none;
    none.count = 0; none.begin = 0; none.len = 1;
    adam_k<<<linr_grid(n, LINR_BLOCK), LINR_BLOCK, 0, s>>>(
        params, grads, exp_avg, exp_avg_sq, n, (float)step_size, (float)bc2_sqrt, (float)beta1, (float)(1.0 - beta1),
        (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, rg ? *rg : none);
    return linr_launch_rc();
}

extern "C" int linr_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                              double step_size, double bc2_sqrt, double beta1, double beta2, double eps,
                              double weight_decay, void* stream) {
    return linr_adam_launch(params, grads, exp_avg, exp_avg_sq, n, step_size, bc2_sqrt, beta1, beta2, eps, weight_decay,
                            nullptr, (hipStream_t)stream);
}

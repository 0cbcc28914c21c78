// One question: do the neighbour gathers of the 3x3x3 convolutions get cheaper when the loads of ABSENT taps
// are switched off per lane with the EXEC mask - branch-free, destination pre-zeroed - instead of reading the all-zero pad row?
// (Round 3 measured the BRANCHED form: 33 vs 17 us per pass, a branch per tap serialises the gathers.)
//   ROWB 32: fp32 rows, two global_load_dwordx4 per tap (cconv_mfma_k);  ROWB 16: bf16 rows, one per tap (bconv_k)
//   C  compiler-scheduled loop, absent taps read the zero row (what the kernels do)
//   U  inline-asm loads, 9 taps in flight, absent taps read the zero row (the asm form's own baseline)
//   M  the same with   s_and_b64 exec, exec, (v >= 0)  around the tap's loads   (absent lanes issue no address)
// Kernel map: argv[1] = file of int32 [27][ld] (the real map of a frame: tools/lab/gather_exec_probe.sh dumps loot10's), rows = argv[2];
// without arguments a synthetic x-major surface with random half-present taps.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_lab/gather_exec_probe tools/lab/gather_exec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

#define TAP(kk) (((kk) / 9) + 3 * (((kk) / 3) % 3) + 9 * ((kk) % 3))

template <int ROWB>
__global__ __launch_bounds__(256) void gather_c(const char* __restrict__ base, const int* __restrict__ nbr, long ld, long n, float* out) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    const long r = row < n ? row : n - 1;
    f4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 27; ++kk) {
        const unsigned off = (unsigned)(nbr[TAP(kk) * ld + r] + 1) * ROWB;
        acc += *(const f4*)(base + off);
        if (ROWB == 32) acc += *(const f4*)(base + off + 16);
    }
    if (row < n) out[row] = acc.x + acc.y + acc.z + acc.w;
}

template <int ROWB, int MASKED>
__global__ __launch_bounds__(256) void gather_asm(const char* __restrict__ base, const int* __restrict__ nbr, long ld, long n, float* out) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    const long r = row < n ? row : n - 1;
    f4 acc = {0, 0, 0, 0};
    int v[27];
#pragma unroll
    for (int kk = 0; kk < 27; ++kk) v[kk] = nbr[TAP(kk) * ld + r];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        f4 a[9], b[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            a[j] = (f4){0, 0, 0, 0};
            b[j] = (f4){0, 0, 0, 0};
            const int vi = v[9 * s + j];
            const unsigned off = (unsigned)(vi + 1) * ROWB;
            if (MASKED) {
                if (ROWB == 32)
                    asm volatile("s_mov_b64 s[70:71], exec\n v_cmp_le_i32 vcc, 0, %3\n s_and_b64 exec, exec, vcc\n"
                                 "global_load_dwordx4 %0, %2, %4\n global_load_dwordx4 %1, %2, %4 offset:16\n s_mov_b64 exec, s[70:71]"
                                 : "+v"(a[j]), "+v"(b[j]) : "v"(off), "v"(vi), "s"(base) : "vcc", "s70", "s71", "memory");
                else
                    asm volatile("s_mov_b64 s[70:71], exec\n v_cmp_le_i32 vcc, 0, %2\n s_and_b64 exec, exec, vcc\n"
                                 "global_load_dwordx4 %0, %1, %3\n s_mov_b64 exec, s[70:71]"
                                 : "+v"(a[j]) : "v"(off), "v"(vi), "s"(base) : "vcc", "s70", "s71", "memory");
            } else {
                if (ROWB == 32)
                    asm volatile("global_load_dwordx4 %0, %2, %3\n global_load_dwordx4 %1, %2, %3 offset:16"
                                 : "+v"(a[j]), "+v"(b[j]) : "v"(off), "s"(base) : "memory");
                else
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(a[j]) : "v"(off), "s"(base) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            asm volatile("" : "+v"(a[j]), "+v"(b[j]));
            acc += a[j];
            if (ROWB == 32) acc += b[j];
        }
    }
    if (row < n) out[row] = acc.x + acc.y + acc.z + acc.w;
}

int main(int argc, char** argv) {
    long n = 336529;
    std::vector<int> h;
    long ld;
    if (argc >= 3) {
        n = atol(argv[2]);
        ld = (n + 63) / 64 * 64;
        h.resize(27 * ld);
        FILE* f = fopen(argv[1], "rb");
        if (!f || fread(h.data(), 4, h.size(), f) != h.size()) { printf("cannot read %s (%ld x 27 ints)\n", argv[1], ld); return 1; }
        fclose(f);
    } else {
        ld = (n + 63) / 64 * 64;
        h.assign(27 * ld, -1);
        for (int k = 0; k < 27; ++k) {
            const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
            for (long r = 0; r < n; ++r) {
                long t = r + 700L * dx + 27L * dy + dz;
                const bool present = ((r * 2654435761u + k * 40503u) >> 7) % 27 < 14 || k == 13;
                h[k * ld + r] = (present && t >= 0 && t < n) ? (int)t : -1;
            }
        }
    }
    long present = 0;
    for (int k = 0; k < 27; ++k) for (long r = 0; r < n; ++r) present += h[k * ld + r] >= 0;
    printf("rows %ld, taps present per row %.2f of 27 (%s)\n", n, (double)present / n, argc >= 3 ? argv[1] : "synthetic map");
    int* nbr; char* x; float *out, *ref;
    CK(hipMalloc(&nbr, h.size() * 4)); CK(hipMalloc(&x, (n + 1) * 32)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&ref, n * 4));
    CK(hipMemcpy(nbr, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    {   // rows hold their own index (fp32), the pad row zeros: the masked form must reproduce the zero-row form's sums exactly
        std::vector<float> hx((n + 1) * 8, 0.0f);
        for (long r = 0; r < n; ++r) for (int c = 0; c < 8; ++c) hx[(r + 1) * 8 + c] = (float)((r * 7 + c) % 1021);
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = (int)((n + 255) / 256);
    const char* nm[6] = {"32 B rows  C  compiler loop, zero row", "32 B rows  U  asm, 9 taps in flight, zero row", "32 B rows  M  asm, EXEC-masked loads",
                         "16 B rows  C  compiler loop, zero row", "16 B rows  U  asm, 9 taps in flight, zero row", "16 B rows  M  asm, EXEC-masked loads"};
    std::vector<float> h_ref(n), h_out(n);
    for (int v = 0; v < 6; ++v) {
        auto launch = [&]() {
            if (v == 0) gather_c<32><<<blocks, 256>>>(x, nbr, ld, n, out);
            else if (v == 1) gather_asm<32, 0><<<blocks, 256>>>(x, nbr, ld, n, out);
            else if (v == 2) gather_asm<32, 1><<<blocks, 256>>>(x, nbr, ld, n, out);
            else if (v == 3) gather_c<16><<<blocks, 256>>>(x, nbr, ld, n, out);
            else if (v == 4) gather_asm<16, 0><<<blocks, 256>>>(x, nbr, ld, n, out);
            else gather_asm<16, 1><<<blocks, 256>>>(x, nbr, ld, n, out);
        };
        for (int i = 0; i < 2000; ++i) launch();              // clock ramp
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 30; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        CK(hipMemcpy(h_out.data(), out, n * 4, hipMemcpyDeviceToHost));
        if (v % 3 == 0) h_ref = h_out;
        long bad = 0;
        for (long r = 0; r < n; ++r) bad += h_out[r] != h_ref[r];
        printf("%-50s %7.2f us per pass   %s\n", nm[v], best * 1e3 / 30, bad ? "MISMATCH vs C" : "sums identical to C");
    }
    return 0;
}

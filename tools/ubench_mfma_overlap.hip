// Do MFMAs of a given input type overlap with VALU work on gfx950?  (development tool)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_overlap.hip -o /tmp/ub && /tmp/ub
// For each MFMA flavour: cycles per loop trip of (a) NM MFMAs alone, (b) NV independent v_fma_f64 alone, (c) both
// interleaved in one wave, at 1 and 4 waves per SIMD.  (c) ~ max(a, b): separate pipes;  (c) ~ a + b: shared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define ITERS 1024
#define NV 16

enum { F64, F32, BF16 };

template <int KIND, bool DO_M, bool DO_V>
__global__ __launch_bounds__(1024) void k(double* out, long long* cyc, int seed) {
    double a[NV];
    for (int c = 0; c < NV; ++c) a[c] = 1.0 + (threadIdx.x + c + seed) * 1e-9;
    f64x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
    f32x16 e0 = {}, e1 = {};
    const double x = 1.0 + seed * 1e-6, y = 0.5;
    const float xf = (float)x, yf = 0.5f;
    bf16x8 bx, by;
    for (int i = 0; i < 8; ++i) { bx[i] = (__bf16)(1.0f + i * seed); by[i] = (__bf16)0.5f; }
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        if (DO_M) {
            if (KIND == F64) {
                d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, d1, 0, 0, 0);
            } else if (KIND == F32) {
                e0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, yf, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_32x32x2f32(yf, xf, e1, 0, 0, 0);
            } else {
                e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by, bx, e1, 0, 0, 0);
                e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by, bx, e1, 0, 0, 0);
            }
        }
        if (DO_V) {
#pragma unroll
            for (int c = 0; c < NV; ++c) a[c] = __builtin_fma(a[c], 1.0000001, 1e-9);
        }
    }
    long long t1 = clock64();
    double s = d0[0] + d1[1] + e0[0] + e1[3];
    for (int c = 0; c < NV; ++c) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND, bool M, bool V>
static double run(int threads) {
    double* out; long long* cyc;
    hipMalloc(&out, 1024 * 304 * 8); hipMalloc(&cyc, 16 * 304 * 8);
    hipLaunchKernelGGL((k<KIND, M, V>), dim3(256), dim3(threads), 0, 0, out, cyc, 1);
    hipLaunchKernelGGL((k<KIND, M, V>), dim3(256), dim3(threads), 0, 0, out, cyc, 2);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    hipFree(out); hipFree(cyc);
    return s / h.size() / ITERS;
}

int main() {
    const char* nm[] = {"v_mfma_f64_16x16x4_f64 x4", "v_mfma_f32_32x32x2_f32 x2", "v_mfma_f32_32x32x16_bf16 x4"};
    for (int threads : {256, 1024}) {
        printf("waves per SIMD = %d (clock64 ticks per loop trip; 16 v_fma_f64 per trip)\n", threads / 256);
        double v = run<F64, false, true>(threads);
        printf("  VALU alone: %.1f\n", v);
        double m0 = run<F64, true, false>(threads), b0 = run<F64, true, true>(threads);
        printf("  %-30s alone %.1f  with VALU %.1f  (sum %.1f, max %.1f)\n", nm[0], m0, b0, m0 + v, m0 > v ? m0 : v);
        double m1 = run<F32, true, false>(threads), b1 = run<F32, true, true>(threads);
        printf("  %-30s alone %.1f  with VALU %.1f  (sum %.1f, max %.1f)\n", nm[1], m1, b1, m1 + v, m1 > v ? m1 : v);
        double m2 = run<BF16, true, false>(threads), b2 = run<BF16, true, true>(threads);
        printf("  %-30s alone %.1f  with VALU %.1f  (sum %.1f, max %.1f)\n", nm[2], m2, b2, m2 + v, m2 > v ? m2 : v);
    }
    return 0;
}

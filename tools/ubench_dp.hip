// Instruction-rate micro-benchmark for the fp64 / int ops the Sinkhorn sweep is built from
// (development tool).  hipcc --offload-arch=gfx950 -O3 tools/ubench_dp.hip -o /tmp/ubench && /tmp/ubench
// Reports shader cycles per wave64 instruction at 1, 2 and 4 waves per SIMD (8 independent chains
// per lane, so the figure is issue throughput, not latency).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHAINS 8
#define ITERS 2048

enum { FMA64, ADD64, MUL64, MAX64, CVT_F64_F32, RNDNE64, CVT_I32_F64, LDEXP64, RCP64, FMA32, ADD32, MUL32, LSHL32, ADDU32,
       AND32, CNDMASK, DSREAD64, CVT_F64_I32, NOPS };
static const char* names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_max_f64", "v_cvt_f64_f32", "v_rndne_f64",
                              "v_cvt_i32_f64", "v_ldexp_f64", "v_rcp_f64", "v_fma_f32", "v_add_f32", "v_mul_f32",
                              "v_lshlrev_b32", "v_add_u32", "v_and_b32", "v_cndmask_b32", "ds_read_b64(random)",
                              "v_cvt_f64_i32"};

template <int OP>
__global__ __launch_bounds__(1024) void k(double* out, long long* cyc, int seed) {
    __shared__ double tab[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) tab[i] = 1.0 + i * 1e-3;
    __syncthreads();
    double a[CHAINS];
    float f[CHAINS];
    int n[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { a[c] = 1.0 + (threadIdx.x + c + seed) * 1e-9; f[c] = (float)a[c]; n[c] = threadIdx.x * 37 + c + seed; }
    const double k1 = 1.0000001, k2 = 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == FMA64) a[c] = __builtin_fma(a[c], k1, k2);
            if (OP == ADD64) a[c] = a[c] + k2;
            if (OP == MUL64) a[c] = a[c] * k1;
            if (OP == MAX64) a[c] = __builtin_fmax(a[c], k1 + it * 1e-12);
            if (OP == CVT_F64_F32) { a[c] = (double)f[c]; asm volatile("" : "+v"(a[c])); f[c] = __int_as_float(__float_as_int(f[c]) ^ (it & 1)); }
            if (OP == RNDNE64) { a[c] = __builtin_rint(a[c]) ; asm volatile("" : "+v"(a[c])); }
            if (OP == CVT_I32_F64) { n[c] = (int)a[c] ^ n[c]; asm volatile("" : "+v"(n[c])); }
            if (OP == LDEXP64) a[c] = __builtin_ldexp(a[c], n[c] & 1);
            if (OP == RCP64) a[c] = __builtin_amdgcn_rcp(a[c]);
            if (OP == FMA32) f[c] = __builtin_fmaf(f[c], 1.0000001f, 1e-9f);
            if (OP == ADD32) f[c] = f[c] + 1e-9f;
            if (OP == MUL32) f[c] = f[c] * 1.0000001f;
            if (OP == LSHL32) { n[c] = n[c] << 1; asm volatile("" : "+v"(n[c])); }
            if (OP == ADDU32) { n[c] = n[c] + 12345; asm volatile("" : "+v"(n[c])); }
            if (OP == AND32) { n[c] = n[c] & 0x7fffff3f; asm volatile("" : "+v"(n[c])); }
            if (OP == CNDMASK) { n[c] = (n[c] > it) ? n[c] : it; asm volatile("" : "+v"(n[c])); }
            if (OP == DSREAD64) { a[c] = tab[(n[c] * 2654435761u >> 23) & 511]; n[c] += __double2loint(a[c]); }
            if (OP == CVT_F64_I32) { a[c] = (double)n[c]; asm volatile("" : "+v"(a[c])); n[c] ^= it; }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += a[c] + f[c] + n[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(int threads) {
    const int blocks = 256;
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks * threads / 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(out, cyc, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(out, cyc, 2);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double per_wave_inst = avg / ((double)ITERS * CHAINS);
    const int wps = threads / 256;
    // wall-clock view: total wave-instructions per SIMD / time
    const double inst_per_simd = (double)ITERS * CHAINS * wps;
    printf("%-22s waves/SIMD=%d  clock64 ticks per wave-inst=%.2f  (issue interval per SIMD = %.2f ticks)  wall: %.2f ns per SIMD-inst\n",
           names[OP], wps, per_wave_inst, per_wave_inst / wps, ms * 1e6 / inst_per_simd);
    hipFree(out); hipFree(cyc);
}

template <int OP>
void all() { run<OP>(256); run<OP>(1024); }

int main() {
    all<FMA64>(); all<ADD64>(); all<MUL64>(); all<MAX64>(); all<CVT_F64_F32>(); all<RNDNE64>(); all<CVT_I32_F64>();
    all<LDEXP64>(); all<RCP64>(); all<FMA32>(); all<ADD32>(); all<MUL32>(); all<LSHL32>(); all<ADDU32>(); all<AND32>();
    all<CNDMASK>(); all<DSREAD64>(); all<CVT_F64_I32>();
    return 0;
}

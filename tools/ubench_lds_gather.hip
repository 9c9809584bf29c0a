// LDS gather rate micro-benchmark for the ADC screen (development tool).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_gather.hip -o /tmp/ulds && /tmp/ulds
// One 1024-thread block per CU with a 128 KiB table [256 codes][64 slots][8 B].  Every lane issues ds_read_b64 gathers at
// (random code) * 512 + slot * 8 with the slot pattern of the screen (32 lanes -> 32 distinct bank pairs) or with all
// lanes on random slots (conflicts), NG gathers in flight per wait, optionally followed by the screen's MFMAs.
// Prints LDS-array cycles per gather instruction per CU (2 = the conflict-free peak of 256 B/clk).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define ITERS 4096

template <int NG, bool FREE, bool MFMA>
__global__ __launch_bounds__(1024) void k(unsigned* out, long long* cyc, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 128 * 1024 / 16; i += 1024) reinterpret_cast<uint4*>(smem)[i] = make_uint4(i, i * 3, i * 5, i * 7);
    __syncthreads();
    const int l = threadIdx.x & 63;
    unsigned rnd = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    unsigned acc0 = 0, acc1 = 0;
    i32x4 macc = {0, 0, 0, 0};
    const i32x4 bsel = {1, 1 << 8, 1 << 16, 1 << 24};
    const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        uint2 e[NG];
        unsigned w[NG / 4];                                   // NG random code bytes, refreshed per iteration (0.5 VALU per gather)
#pragma unroll
        for (int j = 0; j < NG / 4; ++j) { rnd = rnd * 1664525u + 1013904223u; w[j] = rnd ^ (rnd >> 13); }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // the screen's address: v_bfe_u32 + v_lshl_add_u32 with a per-lane slot offset
            const unsigned off = base + (FREE ? ((((l & 31) + g) & 31) + 32 * (g & 1)) * 8 : (((l * 7 + g * 13) & 63) ^ (l >> 3)) * 8 * 0 + ((w[g >> 2] >> 3) & 0x1F8));
            unsigned addr;
            asm("v_bfe_u32 %0, %1, %2, 8\n\tv_lshl_add_u32 %0, %0, 9, %3" : "=&v"(addr) : "v"(w[g >> 2]), "n"(8 * (g & 3)), "v"(off));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 v = *reinterpret_cast<const u32x2 __attribute__((address_space(3)))*>(addr);
            e[g] = make_uint2(v.x, v.y);
        }
        if constexpr (MFMA) {
#pragma unroll
            for (int g = 0; g < NG; g += 2) {
                const i32x4 a = {(int)e[g].x, (int)e[g].y, (int)e[g + 1].x, (int)e[g + 1].y};
                macc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, macc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g) { acc0 ^= e[g].x; acc1 += e[g].y; }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = acc0 ^ acc1 ^ macc[0] ^ macc[1] ^ macc[2] ^ macc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NG, bool FREE, bool MFMA>
void run(const char* name) {
    const int blocks = 256;
    unsigned* out; long long* cyc;
    hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 8);
    auto kern = k<NG, FREE, MFMA>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 128 * 1024, 0, out, cyc, 1u);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 128 * 1024, 0, out, cyc, 2u);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double instr_per_cu = 16.0 * ITERS * NG;              // wave-instructions per CU
    printf("%-44s %6.2f shader cycles per gather instr per CU (clock64), %6.2f ns per gather per CU (wall %.3f ms)\n", name,
           avg / instr_per_cu, ms * 1e6 / instr_per_cu, ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<4, true, false>("conflict-free slots, 4 in flight");
    run<12, true, false>("conflict-free slots, 12 in flight");
    run<24, true, false>("conflict-free slots, 24 in flight");
    run<12, false, false>("random slots, 12 in flight");
    run<12, true, true>("conflict-free, 12 in flight + 6 MFMA 16x16x64");
    run<24, true, true>("conflict-free, 24 in flight + 12 MFMA 16x16x64");
    return 0;
}

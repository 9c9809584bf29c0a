// LDS gather rate micro-benchmark for the ADC screen (development tool), round 3: settles the clock question of the
// round-2 table (its "2 shader cycles but 2.3 ns per gather" implied a 0.8 GHz chip).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_gather.hip -o /tmp/ulds && /tmp/ulds
// Every config is run for WARM_MS of back-to-back launches BEFORE it is measured (a few-millisecond kernel launched on
// an idle chip runs at the idle clock), and every block records clock64 (s_memtime: shader cycles) next to
// wall_clock64 (s_memrealtime: constant 100 MHz), so the shader clock DURING the kernel is clock64 / wall_clock64 and
// is printed beside the sysfs sclk sampled while the kernels run.
//   KIND 0: ds_read_b64, the screen's conflict-free slot pattern (32 lanes -> 32 distinct bank pairs)
//   KIND 1: ds_read_b64, random slots (bank conflicts)
//   KIND 2: ds_read_b128, 16-byte entries, every 16-lane service group on 16 distinct 16-byte slots (conflict-free)
// ADDR 0: the screen's address arithmetic (v_bfe_u32 + v_lshl_add_u32 per gather, code words from a 24-bit LCG: 0.25-0.5
//         VALU per gather); ADDR 1: NG addresses computed ONCE per lane and reused (no address VALU at all: the LDS alone).
// The per-gather numbers come from the launch's wall time x the shader clock measured in the kernel (wave 0's own
// clock64 span understates a block: the oldest wave is served first and finishes early - the round-2 table's mistake).
// Table = 128 KiB per block ([256 codes][64 slots][8 B] or [256 codes][32 slots][16 B]); address = the screen's
// v_bfe_u32 + v_lshl_add_u32 on random code bytes refreshed with 0.5 VALU per gather.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define ITERS 2048

template <int NG, int KIND, bool MFMA, int THREADS, int ADDR>
__global__ __launch_bounds__(THREADS) void k(unsigned* out, long long* cyc, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SMEM = (KIND == 5) ? 147456 : 128 * 1024;
    for (int i = threadIdx.x; i < SMEM / 16; i += THREADS) reinterpret_cast<uint4*>(smem)[i] = make_uint4(i, i * 3, i * 5, i * 7);
    __syncthreads();
    const int l = threadIdx.x & 63;
    unsigned rnd = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    unsigned acc0 = 0, acc1 = 0;
    i32x4 macc = {0, 0, 0, 0};
    const i32x4 bsel = {1, 1 << 8, 1 << 16, 1 << 24};
    const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    // position of a lane inside its ds_read_b128 service group (MI355X_MICROARCH.md, LDS table)
    const int h = l & 31;
    const int pos128 = (h < 4) ? h : (h < 12) ? h - 4 : (h < 16) ? h - 8 : (h < 20) ? h - 8 : (h < 28) ? h - 12 : h - 16;
    unsigned fixed[ADDR ? NG : 1];
    if constexpr (ADDR == 1) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            rnd = rnd * 1664525u + 1013904223u;
            const unsigned code = (rnd >> 11) & 255u;
            fixed[g] = base + (code << 9) + (KIND == 2 ? (((pos128 + g) & 15) + 16 * (g & 1)) * 16
                                                       : KIND == 0 ? ((((l & 31) + g) & 31) + 32 * (g & 1)) * 8 : ((rnd >> 3) & 0x1F8));
        }
    }
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        unsigned w[NG / 4];
#pragma unroll
        for (int j = 0; j < NG / 4; ++j) { rnd = __umul24(rnd, 0x6255u) + 0x3c6ef35fu + j; w[j] = rnd; }   // 1-2 full-rate VALU per 4 gathers
        if constexpr (KIND == 4) {
            // round 6: ds_read_b96 at 16-BYTE ALIGNED addresses ([code][16 slots][16 B], the b128 layout, 12 of 16 bytes read):
            // the guide's table says 8 lane groups of 8 = 8 LDS cycles per wave-instruction; measured here
            typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
            u32x3 e[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const unsigned addr = __builtin_amdgcn_perm(w[g >> 2], base + (unsigned)(g & 1) * 65536u + (unsigned)((pos128 + g) & 15) * 16u,
                                                            0x03020000u | ((4u + (unsigned)(g & 3)) << 8));
                e[g] = *reinterpret_cast<const u32x3 __attribute__((address_space(3)))*>(addr);
            }
            if constexpr (MFMA) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const i32x4 a = {(int)e[g].x, (int)e[g].y, (int)e[g].z, 0};
                    macc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, macc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g) { acc0 ^= e[g].x ^ e[g].z; acc1 += e[g].y; }
            }
        } else if constexpr (KIND == 5) {
            // round 6: the 12-query ALL-RESIDENT candidate at M = 48 (48 x 256 x 12 B = 144 KiB) without ds_read_b96: two planes,
            // [code][16 slots][8 B] (queries 0-7, rows of 128 B, ds_read_b64) + [code][16 slots][4 B] (queries 8-11, rows of 64 B,
            // ds_read_b32); three blocks of 16 sub-quantisers = 3 x (32 + 16) KiB.  Rows shorter than 256 B: whether two lanes of a
            // service group collide depends on their codes.
            uint2 e8[NG]; unsigned e4[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const unsigned code = (w[g >> 2] >> (8 * (g & 3))) & 255u;
                const unsigned slot = (unsigned)((pos128 + g) & 15);
                const unsigned blk = (unsigned)(g % 3) * 49152u;
                const unsigned a8 = base + blk + code * 128u + slot * 8u;
                const unsigned a4 = base + blk + 32768u + code * 64u + slot * 4u;
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 v = *reinterpret_cast<const u32x2 __attribute__((address_space(3)))*>(a8);
                e8[g] = make_uint2(v.x, v.y);
                e4[g] = *reinterpret_cast<const unsigned __attribute__((address_space(3)))*>(a4);
            }
            if constexpr (MFMA) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const i32x4 a = {(int)e8[g].x, (int)e8[g].y, (int)e4[g], 0};
                    macc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, macc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g) { acc0 ^= e8[g].x ^ e4[g]; acc1 += e8[g].y; }
            }
        } else if constexpr (KIND == 6) {
            // round 6: the 6-bit candidate WITHOUT the matrix pipe — 16-query b128 gathers of 6-bit entries (one byte each); four
            // gathers are added as packed bytes (4 x 63 < 256: one v_add_u32 per dword), then widened to 16-bit pairs and added to
            // eight packed accumulators (v_and / v_lshr + v_and, two v_pk_add_u16 per dword): 4 + 5 VALU per gather
            uint4 e[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const unsigned addr = __builtin_amdgcn_perm(w[g >> 2], base + (unsigned)(g & 1) * 65536u + (unsigned)((pos128 + g) & 15) * 16u,
                                                            0x03020000u | ((4u + (unsigned)(g & 3)) << 8));
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = *reinterpret_cast<const u32x4 __attribute__((address_space(3)))*>(addr);
                e[g] = make_uint4(v.x & 0x3F3F3F3Fu, v.y & 0x3F3F3F3Fu, v.z & 0x3F3F3F3Fu, v.w & 0x3F3F3F3Fu);   // (the real table holds 6-bit bytes)
            }
#pragma unroll
            for (int g = 0; g < NG; g += 4) {
                unsigned sx = e[g].x + e[g + 1].x + e[g + 2].x + e[g + 3].x, sy = e[g].y + e[g + 1].y + e[g + 2].y + e[g + 3].y;
                unsigned sz = e[g].z + e[g + 1].z + e[g + 2].z + e[g + 3].z, sw = e[g].w + e[g + 1].w + e[g + 2].w + e[g + 3].w;
                // widen: even / odd bytes -> u16 pairs, packed adds (carry-free: 48 x 63 < 65536)
                acc0 += (sx & 0x00FF00FFu) + (sy & 0x00FF00FFu);
                acc1 += ((sx >> 8) & 0x00FF00FFu) + ((sy >> 8) & 0x00FF00FFu);
                macc[0] += (int)(sz & 0x00FF00FFu);
                macc[1] += (int)((sz >> 8) & 0x00FF00FFu);
                macc[2] += (int)(sw & 0x00FF00FFu);
                macc[3] += (int)((sw >> 8) & 0x00FF00FFu);
            }
        } else if constexpr (KIND == 3) {
            // ds_read_b96: dense 12-byte entries, [block of 16 sub-quantisers][code][16 slots][12 B] (192-byte rows: a code
            // shifts the banks by 48 mod 64, so lanes with different codes can collide - what does that cost?)
            typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
            u32x3 e[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const unsigned off = base + (unsigned)(g & 1) * 49152u + (unsigned)((pos128 + g) & 15) * 12u;
                unsigned addr;
                asm("v_bfe_u32 %0, %1, %2, 8\n\tv_mad_u32_u24 %0, %0, %3, %4" : "=&v"(addr) : "v"(w[g >> 2]), "n"(8 * (g & 3)), "v"(192u), "v"(off));
                e[g] = *reinterpret_cast<const u32x3 __attribute__((address_space(3)))*>(addr);
            }
            if constexpr (MFMA) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const i32x4 a = {(int)e[g].x, (int)e[g].y, (int)e[g].z, 0};
                    macc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, macc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g) { acc0 ^= e[g].x ^ e[g].z; acc1 += e[g].y; }
            }
        } else if constexpr (KIND == 2) {
            uint4 e[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const unsigned off = base + (((pos128 + g) & 15) + 16 * (g & 1)) * 16;   // [code][32 slots][16 B]
                unsigned addr;
                if constexpr (ADDR == 1) addr = fixed[g];
                else if constexpr (ADDR == 2)   // round 4's one-instruction address: [code][16 slots][16 B] rows of 256 B in 64 KiB buffers
                    addr = __builtin_amdgcn_perm(w[g >> 2], base + (unsigned)(g & 1) * 65536u + (unsigned)((pos128 + g) & 15) * 16u,
                                                 0x03020000u | ((4u + (unsigned)(g & 3)) << 8));
                else asm("v_bfe_u32 %0, %1, %2, 8\n\tv_lshl_add_u32 %0, %0, 9, %3" : "=&v"(addr) : "v"(w[g >> 2]), "n"(8 * (g & 3)), "v"(off));
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = *reinterpret_cast<const u32x4 __attribute__((address_space(3)))*>(addr);
                e[g] = make_uint4(v.x, v.y, v.z, v.w);
            }
            if constexpr (MFMA) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const i32x4 a = {(int)e[g].x, (int)e[g].y, (int)e[g].z, (int)e[g].w};
                    macc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, macc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g) { acc0 ^= e[g].x ^ e[g].z; acc1 += e[g].y + e[g].w; }
            }
        } else {
            uint2 e[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const unsigned off = base + (KIND == 0 ? ((((l & 31) + g) & 31) + 32 * (g & 1)) * 8 : ((w[g >> 2] >> 3) & 0x1F8));
                unsigned addr;
                if constexpr (ADDR == 1) addr = fixed[g];
                else if constexpr (ADDR == 2)   // [code][32 slots][8 B] rows of 256 B in 64 KiB buffers (32 sub-quantisers x 8 queries each)
                    addr = __builtin_amdgcn_perm(w[g >> 2], base + (unsigned)(g & 1) * 65536u + (unsigned)(((l & 31) + g) & 31) * 8u,
                                                 0x03020000u | ((4u + (unsigned)(g & 3)) << 8));
                else asm("v_bfe_u32 %0, %1, %2, 8\n\tv_lshl_add_u32 %0, %0, 9, %3" : "=&v"(addr) : "v"(w[g >> 2]), "n"(8 * (g & 3)), "v"(off));
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
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    out[blockIdx.x * THREADS + threadIdx.x] = acc0 ^ acc1 ^ macc[0] ^ macc[1] ^ macc[2] ^ macc[3];
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = w1 - w0; }
}

static std::string sysfs_sclk() {
    // the line marked '*' of pp_dpm_sclk of the first card that has one
    std::string best;
    for (int c = 0; c < 16 && best.empty(); ++c) {
        char path[128];
        snprintf(path, sizeof path, "/sys/class/drm/card%d/device/pp_dpm_sclk", c);
        FILE* f = fopen(path, "r");
        if (!f) continue;
        char line[128];
        while (fgets(line, sizeof line, f)) if (strchr(line, '*')) { best = line; while (!best.empty() && (best.back() == '\n' || best.back() == ' ')) best.pop_back(); }
        fclose(f);
    }
    return best.empty() ? "n/a" : best;
}

static double g_warm_ms = 300.0;

template <int NG, int KIND, bool MFMA, int THREADS, int ADDR = 0>
void run(const char* name, int blocks) {
    unsigned* out; long long* cyc;
    hipMalloc(&out, (size_t)blocks * THREADS * 4); hipMalloc(&cyc, (size_t)blocks * 16);
    auto kern = k<NG, KIND, MFMA, THREADS, ADDR>;
    constexpr int SMEM = (KIND == 5) ? 147456 : 128 * 1024;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    // one launch to size the warm-up
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(THREADS), SMEM, 0, out, cyc, 1u);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms0; hipEventElapsedTime(&ms0, a, b);
    const int nwarm = (int)(g_warm_ms / (ms0 > 0.01f ? ms0 : 0.01f)) + 1;
    for (int i = 0; i < nwarm; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(THREADS), SMEM, 0, out, cyc, 3u + i);
    const int nrep = 20;
    hipEventRecord(a, 0);
    for (int i = 0; i < nrep; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(THREADS), SMEM, 0, out, cyc, 2u);
    hipEventRecord(b, 0);
    const std::string sclk = sysfs_sclk();            // sampled while the queue above is still running
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= nrep;
    std::vector<long long> hc(2 * blocks); hipMemcpy(hc.data(), cyc, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    double cy = 0, wl = 0; for (int i = 0; i < blocks; ++i) { cy += hc[2 * i]; wl += hc[2 * i + 1]; } cy /= blocks; wl /= blocks;
    const double instr_per_cu = (THREADS / 64.0) * ITERS * NG;    // wave-instructions per CU (one block per CU)
    const double bytes = ((KIND == 2 || KIND == 6) ? 1024.0 : (KIND == 3 || KIND == 4 || KIND == 5) ? 768.0 : 512.0);
    const double ghz = cy / (wl * 10.0);                          // wall_clock64 ticks are 10 ns
    const double ns = ms * 1e6 / instr_per_cu;
    printf("%-34s %s blk %3d thr %4d | launch %.3f ms (cold %.3f) | %5.2f ns = %5.2f cyc per gather per CU (wave 0 alone: %4.2f) | clock %.2f GHz (sysfs %s) | %5.1f B/clk/CU | chip %5.1f TB/s\n",
           name, ADDR == 1 ? "fixed-addr " : ADDR == 2 ? "perm-addr  " : "screen-addr", blocks, THREADS, ms, ms0, ns, ns * ghz, cy / instr_per_cu, ghz, sclk.c_str(),
           bytes / (ns * ghz), bytes * instr_per_cu * blocks / (ms * 1e-3) / 1e12);
    fflush(stdout);
    hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
    if (argc > 1) g_warm_ms = atof(argv[1]);
    if (argc > 2 && !strcmp(argv[2], "perm")) {              // round 5: 8-query (b64) against 16-query (b128) gathers with the one-instruction address
        run<12, 2, true, 1024, 2>("b128 cf 12 + 12 MFMA", 256);
        run<24, 0, true, 1024, 2>("b64 cf 24 + 12 MFMA", 256);
        run<12, 0, true, 1024, 2>("b64 cf 12 + 6 MFMA", 256);
        run<24, 0, false, 1024, 2>("b64 cf 24, no MFMA", 256);
        run<12, 2, false, 1024, 2>("b128 cf 12, no MFMA", 256);
        run<12, 2, true, 1024>("b128 cf 12 + 12 MFMA", 256);
        run<24, 0, true, 1024>("b64 cf 24 + 12 MFMA", 256);
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "depth")) {             // round 5: gathers in flight per wave x waves per CU (one-instruction address + MFMA)
        run<4, 2, true, 1024, 2>("b128 cf 4 + 4 MFMA", 256);
        run<8, 2, true, 1024, 2>("b128 cf 8 + 8 MFMA", 256);
        run<12, 2, true, 1024, 2>("b128 cf 12 + 12 MFMA", 256);
        run<16, 2, true, 1024, 2>("b128 cf 16 + 16 MFMA", 256);
        run<4, 2, true, 768, 2>("b128 cf 4 + 4 MFMA", 256);
        run<8, 2, true, 768, 2>("b128 cf 8 + 8 MFMA", 256);
        run<12, 2, true, 768, 2>("b128 cf 12 + 12 MFMA", 256);
        run<16, 2, true, 768, 2>("b128 cf 16 + 16 MFMA", 256);
        run<4, 2, true, 512, 2>("b128 cf 4 + 4 MFMA", 256);
        run<8, 2, true, 512, 2>("b128 cf 8 + 8 MFMA", 256);
        run<12, 2, true, 512, 2>("b128 cf 12 + 12 MFMA", 256);
        run<16, 2, true, 512, 2>("b128 cf 16 + 16 MFMA", 256);
        run<8, 2, false, 1024, 2>("b128 cf 8, no MFMA", 256);
        run<4, 2, false, 1024, 2>("b128 cf 4, no MFMA", 256);
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "r6")) {                // round 6: the all-resident candidates of VERDICT r5 (12 queries; 6-bit without MFMA)
        run<12, 2, true, 1024, 2>("b128 16q (shipped loop) 12 + 12 MFMA", 256);
        run<12, 2, false, 1024, 2>("b128 16q, no MFMA", 256);
        run<12, 4, false, 1024, 2>("b96 aligned 12q, no MFMA", 256);
        run<12, 4, true, 1024, 2>("b96 aligned 12q 12 + 12 MFMA", 256);
        run<12, 3, true, 1024>("b96 dense 12-byte rows 12 + 12 MFMA", 256);
        run<12, 5, false, 1024, 2>("b64+b32 planes 12q, no MFMA", 256);
        run<12, 5, true, 1024, 2>("b64+b32 planes 12q 12 + 12 MFMA", 256);
        run<12, 6, false, 1024, 2>("b128 16q 6-bit, packed adds, no MFMA", 256);
        run<24, 6, false, 1024, 2>("b128 16q 6-bit, packed adds (24)", 256);
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "b96")) {               // the question of the 12-query single-phase screen only
        run<12, 2, true, 1024>("b128 conflict-free 12 + 12 MFMA", 256);
        run<12, 3, false, 1024>("b96 dense rows, random codes 12", 256);
        run<12, 3, true, 1024>("b96 dense rows 12 + 12 MFMA", 256);
        run<24, 3, true, 1024>("b96 dense rows 24 + 24 MFMA", 256);
        return 0;
    }
    printf("LDS gather ubench: ITERS=%d, warm-up %.0f ms per config; idle sclk now: %s\n", ITERS, g_warm_ms, sysfs_sclk().c_str());
    // the LDS alone (no address arithmetic)
    run<12, 0, false, 1024, 1>("b64 conflict-free 12", 256);
    run<24, 0, false, 1024, 1>("b64 conflict-free 24", 256);
    run<24, 0, true, 1024, 1>("b64 conflict-free 24 + 12 MFMA", 256);
    run<12, 1, false, 1024, 1>("b64 random slots 12", 256);
    run<12, 2, false, 1024, 1>("b128 conflict-free 12", 256);
    run<12, 2, true, 1024, 1>("b128 conflict-free 12 + 12 MFMA", 256);
    // with the screen's address arithmetic
    run<12, 0, false, 1024>("b64 conflict-free 12", 256);
    run<24, 0, true, 1024>("b64 conflict-free 24 + 12 MFMA", 256);
    run<12, 0, true, 1024>("b64 conflict-free 12 + 6 MFMA", 256);
    run<12, 1, true, 1024>("b64 random slots 12 + 6 MFMA", 256);
    run<12, 2, true, 1024>("b128 conflict-free 12 + 12 MFMA", 256);
    run<12, 2, true, 512>("b128 conflict-free 12 + 12 MFMA", 256);
    // active CUs and waves per CU
    run<12, 0, true, 1024>("b64 cf 12 + 6 MFMA", 64);
    run<12, 0, true, 512>("b64 cf 12 + 6 MFMA", 256);
    run<12, 0, true, 256>("b64 cf 12 + 6 MFMA", 256);
    return 0;
}

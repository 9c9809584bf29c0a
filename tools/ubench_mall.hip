// Does the 256 MB memory-side cache (MALL / Infinity Cache) keep part of a table that is streamed over and over? (development tool)
// A buffer of S MB is read `reps` times by a grid-stride kernel; the first F MB with plain loads, the rest with non-temporal loads.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mall.hip -o /tmp/ubm && /tmp/ubm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// block b reads the 16 KiB chunks b, b + grid, ... ; chunks below `plain_chunks` with plain loads
__global__ __launch_bounds__(256) void read_kernel(const f4* __restrict__ d, size_t chunks, size_t plain_chunks, float* __restrict__ out) {
    f4 acc = {0, 0, 0, 0};
    for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const f4* p = d + c * 1024 + threadIdx.x;
        f4 v[4];
        if (c < plain_chunks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = p[256 * j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load(p + 256 * j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += v[j];
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    const size_t MB = 1 << 20;
    float* out;
    CHECK(hipMalloc(&out, 4096 * 256 * 4));
    f4* d;
    CHECK(hipMalloc(&d, 2560 * MB));
    CHECK(hipMemset(d, 0, 2560 * MB));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int sizes[] = {64, 128, 192, 256, 302, 384, 512, 1024, 2416};
    for (int S : sizes) {
        const int fr[] = {0, 25, 50, 75, 100};
        for (int f : fr) {
            const size_t chunks = (size_t)S * MB / 16384, plain = chunks * f / 100;
            const int reps = 20;
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, d, chunks, plain, out);
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, d, chunks, plain, out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%5d MB, %3d %% plain loads (%4zu MB): %8.1f us per pass, %6.2f TB/s\n", S, f, plain * 16384 / MB, ms / reps * 1e3,
                   (double)S * MB / (ms / reps * 1e-3) / 1e12);
        }
    }
    return 0;
}

// Read-bandwidth micro-benchmark with the Sinkhorn sweep's access pattern (development tool): what a kernel that
// only READS the [M, B, 256] fp32 table the way sk_sweep2_kernel does can reach on this GPU — the ceiling of the
// sweep's roofline fraction.   hipcc --offload-arch=gfx950 -O3 tools/ubench_stream_read.hip -o /tmp/ubr && /tmp/ubr
//
// Pattern: grid (nbm, M) x 256 threads, a block owns one column range of one sub-quantiser, 16 lanes own a column
// (1 KiB = 4 x 16-byte loads per lane, each part of a 256-byte run), 16 columns in flight per block, the next
// column requested before the current one is consumed (depth 1 / 2 / 3), non-temporal or plain loads.  The
// "consume" is a handful of adds (so the loads cannot be dropped) or `work` dependent fp64 FMAs per entry.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void load_col(const float* p, f4 (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        v[j] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(p) + 16 * j) : reinterpret_cast<const f4*>(p)[16 * j];
}

template <bool NT, int DEPTH, int WORK>
__global__ __launch_bounds__(256, 4) void read_kernel(const float* __restrict__ d, float* __restrict__ out, unsigned B, unsigned rq,
                                                      unsigned rr) {
    const int tid = threadIdx.x, lane = tid & 15, grp = tid >> 4;
    const unsigned bi = blockIdx.x, m = blockIdx.y;
    const unsigned c0 = bi * rq + (bi < rr ? bi : rr), c1 = (bi + 1) * rq + (bi + 1 < rr ? bi + 1 : rr);
    const float* dm = d + (size_t)m * B * 256 + lane * 4;
    f4 buf[DEPTH + 1][4];
    double acc[4] = {0, 0, 0, 0};
    unsigned col = c0 + grp;
#pragma unroll
    for (int k = 0; k < DEPTH; ++k)
        if (col + 16 * k < c1) load_col<NT>(dm + (size_t)(col + 16 * k) * 256, buf[k]);
    int cur = 0;
    for (; col < c1; col += 16) {
        const unsigned nxt = col + 16 * DEPTH;
        // static indexing of the ring: unrolled by DEPTH + 1
#pragma unroll
        for (int s = 0; s <= DEPTH; ++s) {
            if (s == cur) {
                if (nxt < c1) load_col<NT>(dm + (size_t)nxt * 256, buf[(s + DEPTH) % (DEPTH + 1)]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double x0 = buf[s][j].x, x1 = buf[s][j].y, x2 = buf[s][j].z, x3 = buf[s][j].w;
#pragma unroll
                    for (int w = 0; w < WORK; ++w) {
                        x0 = __builtin_fma(x0, 1.0000001, 1e-9); x1 = __builtin_fma(x1, 1.0000001, 1e-9);
                        x2 = __builtin_fma(x2, 1.0000001, 1e-9); x3 = __builtin_fma(x3, 1.0000001, 1e-9);
                    }
                    acc[j] += (x0 + x1) + (x2 + x3);
                }
            }
        }
        cur = (cur + 1) % (DEPTH + 1);
    }
    out[((size_t)m * gridDim.x + bi) * 256 + tid] = (float)((acc[0] + acc[1]) + (acc[2] + acc[3]));
}

template <bool NT, int DEPTH, int WORK>
static void run(const char* name, const float* d, float* out, unsigned B, int M, int nbm, hipStream_t s) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid(nbm, M);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((read_kernel<NT, DEPTH, WORK>), grid, dim3(256), 0, s, d, out, B, B / nbm, B % nbm);
    const int reps = 20;
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((read_kernel<NT, DEPTH, WORK>), grid, dim3(256), 0, s, d, out, B, B / nbm, B % nbm);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)B * M * 1024.0;
    printf("%-34s nbm %4d: %8.1f us per pass, %6.2f TB/s\n", name, nbm, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
    const unsigned B = 49152; const int M = 48;
    float *d, *out;
    CHECK(hipMalloc(&d, (size_t)B * M * 1024));
    CHECK(hipMalloc(&out, (size_t)16384 * 256 * 4));
    CHECK(hipMemset(d, 0, (size_t)B * M * 1024));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    for (int nbm : {21, 42, 84, 168}) {
        run<true, 1, 0>("nt, depth 1, no work", d, out, B, M, nbm, s);
        run<false, 1, 0>("plain, depth 1, no work", d, out, B, M, nbm, s);
        run<true, 2, 0>("nt, depth 2, no work", d, out, B, M, nbm, s);
        run<true, 3, 0>("nt, depth 3, no work", d, out, B, M, nbm, s);
    }
    run<true, 1, 4>("nt, depth 1, 4 fma/entry", d, out, B, M, 21, s);
    run<true, 1, 8>("nt, depth 1, 8 fma/entry", d, out, B, M, 21, s);
    run<true, 1, 12>("nt, depth 1, 12 fma/entry", d, out, B, M, 21, s);
    run<true, 1, 16>("nt, depth 1, 16 fma/entry", d, out, B, M, 21, s);
    run<true, 2, 12>("nt, depth 2, 12 fma/entry", d, out, B, M, 21, s);
    run<true, 2, 16>("nt, depth 2, 16 fma/entry", d, out, B, M, 21, s);
    return 0;
}

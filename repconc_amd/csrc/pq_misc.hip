// Handle management plus the small byte/gather kernels of the path: decode (+ its gradient),
// centroid normalisation, code histogram, k-means sufficient statistics and centroid update.
#include "rc_common.h"
#include <string.h>

#include <math.h>

#include <new>

// ------------------------------------------------------------------------------------------ handle
extern "C" int rc_version(void) { return 101; }   // 101: + rc_pq_assign_nearest_fast, rc_index_*, RC_ESELECT

extern "C" const char* rc_error_string(int code) {
    switch (code) {
        case RC_OK: return "ok";
        case RC_EINVAL: return "invalid argument";
        case RC_ESHAPE: return "unsupported shape (K must be 256, D/M one of 8,12,16,24,32,48,64,96)";
        case RC_EHIP: return "HIP runtime error";
        case RC_EWORKSPACE: return "workspace too small";
        case RC_ECOMM: return "RCCL unavailable or collective failed";
        case RC_ESELECT: return "ADC candidate selection did not converge";
        default: return "unknown error";
    }
}

extern "C" int rc_create(rc_handle_t* out, int device) {
    if (!out) return RC_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return RC_EHIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return RC_EHIP;
    rc_handle_t h = new (std::nothrow) rc_handle_s;
    if (!h) return RC_EINVAL;
    h->device = device;
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->last_hip_error = 0;
    h->profile_on = 0;
    h->exp2_tab[0] = h->exp2_tab[1] = h->exp2_tab[2] = nullptr;
    h->comm[0] = h->comm[1] = nullptr;
    h->comm_rank = 0;
    h->comm_world = 0;
    h->side_stream = nullptr;
    h->ev_fork = h->ev_join = nullptr;
    for (auto& g : h->graphs) { g.ws = nullptr; g.graph = nullptr; g.exec = nullptr; g.stamp = 0; }
    h->graph_stamp = 0;
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    h->graph_broken = 0;
    h->capturing = 0;
    memset(&h->ipc, 0, sizeof(h->ipc));
    *out = h;
    return RC_OK;
}

extern "C" int rc_comm_destroy(rc_handle_t h);

extern "C" int rc_destroy(rc_handle_t h) {
    if (h) {
        (void)rc_comm_destroy(h);
        for (auto& v : h->prof_ev)
            for (hipEvent_t e : v) (void)hipEventDestroy(e);
        for (double* t : h->exp2_tab)
            if (t) (void)hipFree(t);
        if (h->scratch) (void)hipFree(h->scratch);
        for (auto& g : h->graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            if (g.graph) (void)hipGraphDestroy(g.graph);
        }
    }
    delete h;
    return RC_OK;
}

void* rc_scratch(rc_handle_t h, size_t bytes) {
    if (!h) return nullptr;
    if (bytes <= h->scratch_bytes) return h->scratch;
    if (h->scratch) {
        (void)hipDeviceSynchronize();                    // work queued on the old block finishes before it is freed
        (void)hipFree(h->scratch);
        h->scratch = nullptr;
        h->scratch_bytes = 0;
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    h->scratch = p;
    h->scratch_bytes = bytes;
    return p;
}

// 2^(j/N) correctly rounded to double via long-double exp2l (64-bit significand), uploaded once.
const double* rc_exp2_table(rc_handle_t h, int tb) {
    if (!h || (tb != 8 && tb != 11 && tb != 12)) return nullptr;
    const int slot = (tb == 8) ? 0 : (tb == 11) ? 1 : 2;
    if (h->exp2_tab[slot]) return h->exp2_tab[slot];
    const int N = 1 << tb;
    std::vector<double> host(N);
    for (int j = 0; j < N; ++j) host[j] = (double)exp2l((long double)j / (long double)N);
    double* dev = nullptr;
    if (hipMalloc(&dev, sizeof(double) * N) != hipSuccess) return nullptr;
    if (hipMemcpy(dev, host.data(), sizeof(double) * N, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(dev);
        return nullptr;
    }
    h->exp2_tab[slot] = dev;
    return dev;
}

// ------------------------------------------------------------------------------------------ profiling
extern "C" int rc_profile_enable(rc_handle_t h, int on) {
    rc_device_guard device_guard_(h);
    if (!h) return RC_EINVAL;
    h->profile_on = on == 2 ? 2 : (on ? 1 : 0);
    return RC_OK;
}

void rc_prof_mark(rc_handle_t h, int slot, hipStream_t s) {
    if (!h || h->profile_on != 1 || h->capturing) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    h->prof_ev[slot].push_back(e);
    if (h->prof_ev[slot].size() % 2 == 0) h->prof_n[slot].push_back(1);
}

void rc_prof_bracket(rc_handle_t h, int slot, hipStream_t s, bool open, int launches) {
    if (!h || h->profile_on != 2 || h->capturing) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    h->prof_ev[slot].push_back(e);
    if (!open) h->prof_n[slot].push_back(launches);
}

extern "C" int rc_profile_collect(rc_handle_t h, int kernel_class, int* launches, double* total_ms) {
    rc_device_guard device_guard_(h);
    if (!h || !launches || !total_ms || kernel_class < 0 || kernel_class >= RC_PROF_NSLOT) return RC_EINVAL;
    auto& v = h->prof_ev[kernel_class];
    int n = 0;
    double tot = 0.0;
    for (size_t i = 0; i + 1 < v.size(); i += 2) {
        float ms = 0.f;
        RC_HIP_CHECK(h, hipEventSynchronize(v[i + 1]));
        RC_HIP_CHECK(h, hipEventElapsedTime(&ms, v[i], v[i + 1]));
        tot += ms;
        n += (i / 2 < h->prof_n[kernel_class].size()) ? h->prof_n[kernel_class][i / 2] : 1;
    }
    for (hipEvent_t e : v) (void)hipEventDestroy(e);
    v.clear();
    h->prof_n[kernel_class].clear();
    *launches = n;
    *total_ms = tot;
    return RC_OK;
}

extern "C" int rc_last_hip_error(rc_handle_t h) { return h ? h->last_hip_error : 0; }
extern "C" int rc_num_cus(rc_handle_t h) { return h ? h->num_cus : 0; }

// ------------------------------------------------------------------------------------------ decode
template <typename CodeT>
__device__ __forceinline__ int load_code(const void* codes, int64_t i) {
    return (int)reinterpret_cast<const CodeT*>(codes)[i];
}

// out[n, m*dsub + j] = C[m, codes[n,m], j].  One thread per float4 of the output row, so the
// stores are fully coalesced; the centroid table (<= 786 KB) stays in L2.
// modeling_repconc.py:168-175.
template <typename CodeT>
__global__ __launch_bounds__(256) void decode_kernel(const void* __restrict__ codes, const float* __restrict__ C,
                                                     int64_t n, int M, int dsub, float* __restrict__ out) {
    const int D4 = (M * dsub) / 4;
    const int q4 = dsub / 4;  // float4 per sub-vector
    const int64_t total = n * D4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / D4;
        const int c4 = (int)(i - row * D4);
        const int m = c4 / q4, j4 = c4 - m * q4;
        const int code = load_code<CodeT>(codes, row * M + m) & (RC_K - 1);
        reinterpret_cast<float4*>(out)[i] =
            reinterpret_cast<const float4*>(C)[((size_t)m * RC_K + code) * q4 + j4];
    }
}

// grad_C[m, codes[n,m], j] += grad_out[n, m*dsub + j]  (fp32 atomics; the reference's
// index_put(accumulate) backward is equally order-free).
template <typename CodeT>
__global__ __launch_bounds__(256) void decode_bwd_kernel(const void* __restrict__ codes,
                                                         const float* __restrict__ go, int64_t n, int M,
                                                         int dsub, float* __restrict__ gC) {
    const int D = M * dsub;
    const int64_t total = n * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / D;
        const int c = (int)(i - row * D);
        const int m = c / dsub, j = c - m * dsub;
        const int code = load_code<CodeT>(codes, row * M + m) & (RC_K - 1);
        atomicAdd(gC + ((size_t)m * RC_K + code) * dsub + j, go[i]);
    }
}

static unsigned grid_for(rc_handle_t h, int64_t work_items) {
    int64_t g = (work_items + 255) / 256;
    const int64_t cap = (int64_t)h->num_cus * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

extern "C" int rc_pq_decode(rc_handle_t h, const void* codes, int code_dtype, const float* C, int64_t n, int M,
                            int K, int dsub, float* out, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !C || !out || n < 0 || M <= 0 || dsub <= 0) return RC_EINVAL;
    if (K != RC_K || dsub % 4 != 0) return RC_ESHAPE;
    if (n == 0) return RC_OK;
    const unsigned g = grid_for(h, n * (M * dsub / 4));
    hipStream_t s = (hipStream_t)stream;
    if (code_dtype == RC_CODE_U8)
        hipLaunchKernelGGL(decode_kernel<uint8_t>, dim3(g), dim3(256), 0, s, codes, C, n, M, dsub, out);
    else if (code_dtype == RC_CODE_I64)
        hipLaunchKernelGGL(decode_kernel<int64_t>, dim3(g), dim3(256), 0, s, codes, C, n, M, dsub, out);
    else
        return RC_EINVAL;
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

extern "C" int rc_pq_decode_bwd(rc_handle_t h, const void* codes, int code_dtype, const float* grad_out,
                                int64_t n, int M, int K, int dsub, float* grad_C, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !grad_out || !grad_C || n < 0 || M <= 0 || dsub <= 0) return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    if (n == 0) return RC_OK;
    const unsigned g = grid_for(h, n * M * dsub);
    hipStream_t s = (hipStream_t)stream;
    if (code_dtype == RC_CODE_U8)
        hipLaunchKernelGGL(decode_bwd_kernel<uint8_t>, dim3(g), dim3(256), 0, s, codes, grad_out, n, M, dsub, grad_C);
    else if (code_dtype == RC_CODE_I64)
        hipLaunchKernelGGL(decode_bwd_kernel<int64_t>, dim3(g), dim3(256), 0, s, codes, grad_out, n, M, dsub, grad_C);
    else
        return RC_EINVAL;
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// ------------------------------------------------------------------------------------------ normalise
// C[m,k,:] /= max(||C[m,k,:]||_2, 1e-12)  — F.normalize(p=2, dim=-1), modeling_repconc.py:112-116.
// Plain j-ascending sum of squares: torch's vectorised norm may round differently in the last
// ulp, so this op is specified to 1e-6 relative (METRIC_CENTROID_COS is set by no shipped recipe).
__global__ __launch_bounds__(256) void normalize_kernel(float* __restrict__ C, int64_t rows, int dsub) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float* p = C + r * dsub;
    float s = 0.f;
    for (int j = 0; j < dsub; ++j) s = s + p[j] * p[j];
    const float nrm = fmaxf(sqrtf(s), 1e-12f);
    for (int j = 0; j < dsub; ++j) p[j] = p[j] / nrm;
}

extern "C" int rc_normalize_centroids(rc_handle_t h, float* C, int M, int K, int dsub, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !C || M <= 0 || K <= 0 || dsub <= 0) return RC_EINVAL;
    const int64_t rows = (int64_t)M * K;
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, C,
                       rows, dsub);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// ------------------------------------------------------------------------------------------ histogram
// hist[m][k] = #{n : codes[n,m] == k}.  Block = (row strip, group of 4 sub-quantisers); counts are
// privatised in LDS (ds_add_u32) and flushed with one global atomic per non-empty bin.
// finetune_repconc.py:588-592.
#define HIST_MG 4
template <typename CodeT>
__global__ __launch_bounds__(256) void hist_kernel(const void* __restrict__ codes, int64_t n, int M,
                                                   int rows_per_block, int32_t* __restrict__ hist) {
    __shared__ int cnt[HIST_MG][RC_K];
    const int tid = threadIdx.x;
    for (int i = tid; i < HIST_MG * RC_K; i += 256) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const int m0 = blockIdx.y * HIST_MG;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
    for (int64_t r = r0 + tid; r < r1; r += 256) {
#pragma unroll
        for (int q = 0; q < HIST_MG; ++q)
            if (m0 + q < M) atomicAdd(&cnt[q][load_code<CodeT>(codes, r * M + m0 + q) & (RC_K - 1)], 1);
    }
    __syncthreads();
    for (int i = tid; i < HIST_MG * RC_K; i += 256) {
        const int q = i / RC_K, k = i - q * RC_K;
        const int c = cnt[q][k];
        if (c && m0 + q < M) atomicAdd(hist + (size_t)(m0 + q) * RC_K + k, c);
    }
}

extern "C" int rc_code_hist(rc_handle_t h, const void* codes, int code_dtype, int64_t n, int M, int K,
                            int32_t* hist, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !hist || n < 0 || M <= 0) return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    RC_HIP_CHECK(h, hipMemsetAsync(hist, 0, (size_t)M * RC_K * sizeof(int32_t), s));
    if (n == 0) return RC_OK;
    const int rpb = 4096;
    dim3 grid((unsigned)((n + rpb - 1) / rpb), (unsigned)((M + HIST_MG - 1) / HIST_MG));
    if (code_dtype == RC_CODE_U8)
        hipLaunchKernelGGL(hist_kernel<uint8_t>, grid, dim3(256), 0, s, codes, n, M, rpb, hist);
    else if (code_dtype == RC_CODE_I64)
        hipLaunchKernelGGL(hist_kernel<int64_t>, grid, dim3(256), 0, s, codes, n, M, rpb, hist);
    else
        return RC_EINVAL;
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// ------------------------------------------------------------------------------------------ k-means
// Lloyd sufficient statistics sums[m][k][:] (fp64) and counts[m][k], DETERMINISTIC: every sum has a fixed order, so the
// warm-up is reproducible run to run and rank to rank (round 1 merged LDS partials with fp64 atomics).
//
//  stage 1  grid (strips, M), block = 256 threads = the 256 centroids.  A block walks its strip of rows in order; the
//           row's code is block-uniform (staged through LDS in chunks of 1024), the ONE thread k == code adds the row's
//           sub-vector to its private fp64 registers — per (strip, m, k) the rows are added in ascending order, no two
//           threads ever touch the same accumulator, every x element is read exactly once.  Partials go to scratch
//           [strip][m][k][dsub].  The wave-uniform test (code >> 6 == wave) skips the three waves that do not own k.
//  stage 2  sums[m][k][j] += partials in strip order; counts likewise.
// At most 64 strips (100 MB of scratch at M = 48): 3072 blocks of ~138 k rows for the 8.84 M-row corpus.
#define KM_CHUNK 1024
#define KM_MAX_STRIPS 64

template <int JN>
__global__ __launch_bounds__(256) void kmeans_stats_det_kernel(const float* __restrict__ x, int64_t ldx,
                                                               const uint8_t* __restrict__ codes, int64_t n, int M, int dsub,
                                                               int j0, int64_t rows_per_strip, double* __restrict__ part,
                                                               unsigned* __restrict__ pcnt, const unsigned* __restrict__ gate) {
    if (gate && *gate < 0x7F800000u) return;              // finite input: the fixed-point path has done this call
    __shared__ uint8_t cs[KM_CHUNK];
    const int tid = threadIdx.x, m = blockIdx.y, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_strip;
    const int64_t r1 = (r0 + rows_per_strip < n) ? r0 + rows_per_strip : n;
    double acc[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[j] = 0.0;
    unsigned cnt = 0;
    const float* xm = x + m * dsub + j0;
    for (int64_t c0 = r0; c0 < r1; c0 += KM_CHUNK) {
        const int nc = (int)((r1 - c0 < KM_CHUNK) ? r1 - c0 : KM_CHUNK);
        __syncthreads();
        for (int i = tid; i < nc; i += 256) cs[i] = codes[(c0 + i) * M + m];
        __syncthreads();
        for (int i = 0; i < nc; ++i) {
            const int k = cs[i];                                   // block-uniform
            if ((k >> 6) == wave) {                                // wave-uniform
                if (k == tid) {
                    const float* xr = xm + (c0 + i) * ldx;
                    if constexpr (JN % 4 == 0) {
#pragma unroll
                        for (int j4 = 0; j4 < JN / 4; ++j4) {
                            const float4 v = reinterpret_cast<const float4*>(xr)[j4];
                            acc[4 * j4] += (double)v.x; acc[4 * j4 + 1] += (double)v.y;
                            acc[4 * j4 + 2] += (double)v.z; acc[4 * j4 + 3] += (double)v.w;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < JN; ++j) acc[j] += (double)xr[j];
                    }
                    ++cnt;
                }
            }
        }
    }
    double* p = part + (((size_t)blockIdx.x * M + m) * RC_K + tid) * dsub + j0;
#pragma unroll
    for (int j = 0; j < JN; ++j) p[j] = acc[j];
    if (j0 == 0) pcnt[((size_t)blockIdx.x * M + m) * RC_K + tid] = cnt;
}

__global__ __launch_bounds__(256) void kmeans_stats_reduce_kernel(const double* __restrict__ part, const unsigned* __restrict__ pcnt,
                                                                  int strips, int64_t per_strip, int dsub,
                                                                  double* __restrict__ sums, unsigned long long* __restrict__ counts,
                                                                  const unsigned* __restrict__ gate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_strip || (gate && *gate < 0x7F800000u)) return;
    double s = 0.0;
    for (int t = 0; t < strips; ++t) s += part[(size_t)t * per_strip + i];
    sums[i] += s;
    if (i % dsub == 0) {
        unsigned long long c = 0;
        for (int t = 0; t < strips; ++t) c += pcnt[(size_t)t * (per_strip / dsub) + i / dsub];
        counts[i / dsub] += c;
    }
}

// ---- exact fixed-point statistics (the default path) --------------------------------------------------------------
// The strip kernel above is deterministic because every centroid's rows are added in row order by ONE thread — 1/256 of
// the lanes at work, 1.3 ms for 65 536 rows (it was 90 % of a Lloyd iteration of the warm-up).  Integer addition does not
// care about the order: every value is split as  x S = hi + r,  hi = rint(x S),  lo = rint(r 2^38)  (S a power of two sized
// so that n values cannot overflow 63 bits; both parts exact for every x down to max|x| 2^-52) and (hi, lo) are summed with
// 64-bit integer atomics — LDS accumulators per (strip, sub-quantiser), then global.  Any order gives the same integers,
// the final conversion is one fp64 expression: bit-identical run to run, and closer to the real sum than a running fp64 sum.
// Non-finite input (max|x| = inf / NaN) takes the strip kernels, which propagate it as the reference would.
#define KM_FX_LO_BITS 38
#define KM_FX_JN 32                                     // dimensions of a sub-vector per pass: 256 x 32 x 16 B = 128 KiB of LDS

__global__ __launch_bounds__(256) void kmeans_absmax_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int D,
                                                            unsigned* __restrict__ out) {
    __shared__ unsigned s_w[4];
    unsigned mx = 0u;
    const int64_t total = n * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const unsigned u = __float_as_uint(x[(i / D) * ldx + (i % D)]) & 0x7FFFFFFFu;   // |x| as ordered bits; NaN > inf
        mx = u > mx ? u : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)mx, o);
        mx = t > mx ? t : mx;
    }
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a = s_w[0] > s_w[1] ? s_w[0] : s_w[1], b = s_w[2] > s_w[3] ? s_w[2] : s_w[3];
        atomicMax(out, a > b ? a : b);
    }
}

// grid (strips, M); dynamic LDS: hi[256][jn] | lo[256][jn] (int64) | cnt[256] (u32)
__global__ __launch_bounds__(256) void kmeans_stats_fx_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const uint8_t* __restrict__ codes, int64_t n, int M, int dsub,
                                                              int j0, int jn, int64_t rows_per_strip,
                                                              const unsigned* __restrict__ absmax, int log2n,
                                                              unsigned long long* __restrict__ ghi,
                                                              unsigned long long* __restrict__ glo,
                                                              unsigned long long* __restrict__ gcnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char km_smem[];
    unsigned long long* hi = reinterpret_cast<unsigned long long*>(km_smem);
    unsigned long long* lo = hi + RC_K * jn;
    unsigned* cnt = reinterpret_cast<unsigned*>(lo + RC_K * jn);
    const unsigned am = *absmax;
    if (am >= 0x7F800000u) return;                         // inf / NaN somewhere: the strip kernels take this call
    const int tid = threadIdx.x, m = blockIdx.y;
    for (int i = tid; i < 2 * RC_K * jn; i += 256) hi[i] = 0ull;
    cnt[tid] = 0u;
    __syncthreads();
    // S = 2^sexp with max|x| S < 2^(61 - log2n): the largest exponent of the data is (am >> 23) - 127
    const int sexp = 61 - log2n - ((int)(am >> 23) - 127 + 1);
    const double S = ldexp(1.0, sexp);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_strip;
    const int64_t r1 = (r0 + rows_per_strip < n) ? r0 + rows_per_strip : n;
    const int tpr = jn / 4, rpi = 256 / tpr;               // threads per row (one float4 each), rows per iteration
    const int q = tid % tpr;
    if (tid < rpi * tpr) {
        for (int64_t r = r0 + tid / tpr; r < r1; r += rpi) {
            const int k = codes[r * M + m];
            const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + m * dsub + j0 + 4 * q);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double d = (double)vv[e] * S;
                const double h = rint(d);
                const long long hl = (long long)h;
                const long long ll = (long long)rint((d - h) * (double)(1ull << KM_FX_LO_BITS));
                if (hl) atomicAdd(&hi[k * jn + 4 * q + e], (unsigned long long)hl);
                if (ll) atomicAdd(&lo[k * jn + 4 * q + e], (unsigned long long)ll);
            }
            if (q == 0 && j0 == 0) atomicAdd(&cnt[k], 1u);
        }
    }
    __syncthreads();
    for (int i = tid; i < RC_K * jn; i += 256) {
        const size_t g = ((size_t)m * RC_K + i / jn) * dsub + j0 + i % jn;
        if (hi[i]) atomicAdd(&ghi[g], hi[i]);
        if (lo[i]) atomicAdd(&glo[g], lo[i]);
    }
    if (j0 == 0 && cnt[tid]) atomicAdd(&gcnt[(size_t)m * RC_K + tid], (unsigned long long)cnt[tid]);
}

__global__ __launch_bounds__(256) void kmeans_stats_fx_finish_kernel(const unsigned long long* __restrict__ ghi,
                                                                     const unsigned long long* __restrict__ glo,
                                                                     const unsigned long long* __restrict__ gcnt,
                                                                     const unsigned* __restrict__ absmax, int log2n,
                                                                     int64_t total, int dsub, double* __restrict__ sums,
                                                                     unsigned long long* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned am = *absmax;
    if (i >= total || am >= 0x7F800000u) return;
    const int sexp = 61 - log2n - ((int)(am >> 23) - 127 + 1);
    const double h = (double)(long long)ghi[i], l = (double)(long long)glo[i];
    sums[i] += ldexp(h, -sexp) + ldexp(l, -sexp - KM_FX_LO_BITS);
    if (i % dsub == 0) counts[i / dsub] += gcnt[i / dsub];
}

extern "C" int rc_kmeans_stats(rc_handle_t h, const float* x, int64_t ldx, const uint8_t* codes, int64_t n, int D,
                               int M, int K, double* sums, int64_t* counts, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !x || !codes || !sums || !counts || n < 0 || M <= 0 || D <= 0 || ldx < D) return RC_EINVAL;
    if (K != RC_K || D % M != 0) return RC_ESHAPE;
    const int dsub = D / M;
    if (dsub > 256) return RC_ESHAPE;
    if (n == 0) return RC_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (ldx % 4 == 0) && (((uintptr_t)x) % 16 == 0) && (dsub % 4 == 0);
    const int64_t per_strip = (int64_t)M * RC_K * dsub;
    // exact fixed-point path (see above): float4 rows, n < 2^24 per call, finite input (checked on the device)
    const bool fx = vec && n < (1ll << 24) && !rc_env_set("RC_KMEANS_STRIPS");
    const unsigned* gate = nullptr;                                // device word: strip kernels run iff it is not finite
    if (fx) {
        int log2n = 0;
        while ((1ll << log2n) < n) ++log2n;
        const size_t lbytes = rc_align_up((size_t)per_strip * sizeof(unsigned long long), 256);
        const size_t cb = rc_align_up((size_t)M * RC_K * sizeof(unsigned long long), 256);
        // the strip fall-back below reuses the same scratch block; the absmax word sits behind its partials
        int64_t rps0 = 8192;
        int st0 = (int)((n + rps0 - 1) / rps0);
        if (st0 > KM_MAX_STRIPS) st0 = KM_MAX_STRIPS;
        const size_t strip_bytes = rc_align_up((size_t)st0 * per_strip * sizeof(double), 256) +
                                   rc_align_up((size_t)st0 * M * RC_K * sizeof(unsigned), 256);
        const size_t need = (2 * lbytes + cb + 256 > strip_bytes ? 2 * lbytes + cb + 256 : strip_bytes + 256);
        char* ws = (char*)rc_scratch(h, need);
        if (!ws) return RC_EHIP;
        unsigned long long* ghi = (unsigned long long*)ws;
        unsigned long long* glo = (unsigned long long*)(ws + lbytes);
        unsigned long long* gcnt = (unsigned long long*)(ws + 2 * lbytes);
        unsigned* absmax = (unsigned*)(ws + need - 256);
        RC_HIP_CHECK(h, hipMemsetAsync(ws, 0, 2 * lbytes + cb, s));
        RC_HIP_CHECK(h, hipMemsetAsync(absmax, 0, sizeof(unsigned), s));
        int64_t ab = (n * D + 255) / 256;
        if (ab > 4096) ab = 4096;
        hipLaunchKernelGGL(kmeans_absmax_kernel, dim3((unsigned)ab), dim3(256), 0, s, x, ldx, n, D, absmax);
        RC_LAUNCH_CHECK(h);
        int64_t rps = 2048;
        if ((n + rps - 1) / rps > 1024) rps = (n + 1023) / 1024;
        const unsigned strips = (unsigned)((n + rps - 1) / rps);
        for (int j0 = 0; j0 < dsub; j0 += KM_FX_JN) {
            const int jn = dsub - j0 < KM_FX_JN ? dsub - j0 : KM_FX_JN;
            const size_t lds = (size_t)2 * RC_K * jn * sizeof(unsigned long long) + RC_K * sizeof(unsigned);
            RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kmeans_stats_fx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kmeans_stats_fx_kernel, dim3(strips, (unsigned)M), dim3(256), lds, s, x, ldx, codes, n, M, dsub, j0, jn,
                               rps, (const unsigned*)absmax, log2n, ghi, glo, gcnt);
            RC_LAUNCH_CHECK(h);
        }
        hipLaunchKernelGGL(kmeans_stats_fx_finish_kernel, dim3((unsigned)((per_strip + 255) / 256)), dim3(256), 0, s,
                           (const unsigned long long*)ghi, (const unsigned long long*)glo, (const unsigned long long*)gcnt,
                           (const unsigned*)absmax, log2n, per_strip, dsub, sums, reinterpret_cast<unsigned long long*>(counts));
        RC_LAUNCH_CHECK(h);
        gate = absmax;            // the strip kernels below leave at once unless max|x| is inf / NaN (decided on the device)
    }
    int64_t rps = 8192;                                            // rows per strip
    int strips = (int)((n + rps - 1) / rps);
    if (strips > KM_MAX_STRIPS) {
        strips = KM_MAX_STRIPS;
        rps = (n + strips - 1) / strips;
        rps = (rps + KM_CHUNK - 1) / KM_CHUNK * KM_CHUNK;
        strips = (int)((n + rps - 1) / rps);
    }
    const size_t pbytes = rc_align_up((size_t)strips * per_strip * sizeof(double), 256);
    const size_t cbytes = rc_align_up((size_t)strips * M * RC_K * sizeof(unsigned), 256);
    char* ws = (char*)rc_scratch(h, pbytes + cbytes);
    if (!ws) return RC_EHIP;
    double* part = (double*)ws;
    unsigned* pcnt = (unsigned*)(ws + pbytes);
    dim3 grid((unsigned)strips, (unsigned)M);
    for (int j0 = 0; j0 < dsub;) {
        const int left = dsub - j0;
        if (vec && left >= 16) {
            hipLaunchKernelGGL(kmeans_stats_det_kernel<16>, grid, dim3(256), 0, s, x, ldx, codes, n, M, dsub, j0, rps, part, pcnt, gate);
            j0 += 16;
        } else if (vec && left >= 8) {
            hipLaunchKernelGGL(kmeans_stats_det_kernel<8>, grid, dim3(256), 0, s, x, ldx, codes, n, M, dsub, j0, rps, part, pcnt, gate);
            j0 += 8;
        } else if (vec && left >= 4) {
            hipLaunchKernelGGL(kmeans_stats_det_kernel<4>, grid, dim3(256), 0, s, x, ldx, codes, n, M, dsub, j0, rps, part, pcnt, gate);
            j0 += 4;
        } else {
            hipLaunchKernelGGL(kmeans_stats_det_kernel<1>, grid, dim3(256), 0, s, x, ldx, codes, n, M, dsub, j0, rps, part, pcnt, gate);
            j0 += 1;
        }
        RC_LAUNCH_CHECK(h);
    }
    hipLaunchKernelGGL(kmeans_stats_reduce_kernel, dim3((unsigned)((per_strip + 255) / 256)), dim3(256), 0, s, (const double*)part,
                       (const unsigned*)pcnt, strips, per_strip, dsub, sums, reinterpret_cast<unsigned long long*>(counts), gate);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

__global__ __launch_bounds__(256) void kmeans_update_kernel(const double* __restrict__ sums,
                                                            const long long* __restrict__ counts,
                                                            float* __restrict__ C, int64_t total, int dsub) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long c = counts[i / dsub];
    if (c > 0) C[i] = (float)(sums[i] / (double)c);
}

extern "C" int rc_kmeans_update(rc_handle_t h, const double* sums, const int64_t* counts, float* C, int M, int K,
                                int dsub, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !sums || !counts || !C || M <= 0 || dsub <= 0) return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    const int64_t total = (int64_t)M * K * dsub;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       sums, reinterpret_cast<const long long*>(counts), C, total, dsub);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// ------------------------------------------------------------------------------------------ empty clusters
// Faiss 1.7.x Clustering.cpp `split_clusters` (what `index.train`, train/run_warmup.py:113, does after every centroid
// update), restated from the published source: per clustering (= per sub-quantiser) a std::mt19937 seeded with 1234
// drives a cyclic walk cj = 0, 1, ... that accepts cluster cj as the donor of an empty cluster ci with probability
// (size_cj - 1) / (n - k); centroid ci <- centroid cj, then ci *= 1 +- 1/1024 and cj *= 1 -+ 1/1024 alternating over the
// components; the donor's (float) size is halved for the following draws.  On the device so that a Lloyd iteration has no
// host synchronisation: one block per sub-quantiser, all threads look for an empty cluster, thread 0 makes the
// (sequential, rare) walk with its own MT19937 in LDS.
namespace {
struct mt19937_lds {
    unsigned* mt;
    int idx;
    __device__ void seed(unsigned s) {
        mt[0] = s;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned)i;
        idx = 624;
    }
    __device__ unsigned next() {
        if (idx >= 624) {
            for (int i = 0; i < 624; ++i) {
                const unsigned y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        unsigned y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};
}  // namespace

__global__ __launch_bounds__(RC_K) void kmeans_split_empty_kernel(float* __restrict__ C, const long long* __restrict__ counts,
                                                                  int dsub, int* __restrict__ nsplit) {
    __shared__ unsigned s_mt[624];
    __shared__ float s_h[RC_K];
    __shared__ int s_any;
    const int m = blockIdx.x, tid = threadIdx.x;
    const long long c = counts[(size_t)m * RC_K + tid];
    s_h[tid] = (float)c;
    if (tid == 0) s_any = 0;
    __syncthreads();
    if (c == 0) s_any = 1;
    __syncthreads();
    if (!s_any || tid != 0) return;
    long long n = 0;
    float hmax = 0.f;
    for (int k = 0; k < RC_K; ++k) { n += counts[(size_t)m * RC_K + k]; hmax = fmaxf(hmax, s_h[k]); }
    const double denom = (double)(float)(n - RC_K);
    mt19937_lds rng{s_mt, 624};
    rng.seed(1234u);
    float* Cm = C + (size_t)m * RC_K * dsub;
    const float up = 1.0f + 1.0f / 1024.0f, dn = 1.0f - 1.0f / 1024.0f;
    int splits = 0;
    for (int ci = 0; ci < RC_K; ++ci) {
        if (s_h[ci] != 0.f) continue;
        int cj = 0;
        if (!(denom > 0.0) || hmax <= 1.f) {                    // Faiss would never accept: take the biggest cluster
            for (int k = 1; k < RC_K; ++k) cj = s_h[k] > s_h[cj] ? k : cj;
        } else {
            for (int draws = 0; draws < 10000000; ++draws) {
                const float p = (float)(((double)s_h[cj] - 1.0) / denom);
                const float r = (float)rng.next() / 4294967296.0f;     // float(mt()) / float(mt.max()): float(2^32 - 1) = 2^32
                if (r < p) break;
                cj = (cj + 1) % RC_K;
            }
        }
        for (int j = 0; j < dsub; ++j) {
            const float v = Cm[(size_t)cj * dsub + j];
            Cm[(size_t)ci * dsub + j] = v * ((j & 1) ? dn : up);
            Cm[(size_t)cj * dsub + j] = v * ((j & 1) ? up : dn);
        }
        s_h[ci] = s_h[cj] / 2.0f;
        s_h[cj] = s_h[cj] - s_h[ci];
        ++splits;
    }
    if (nsplit && splits) atomicAdd(nsplit, splits);
}

extern "C" int rc_kmeans_split_empty(rc_handle_t h, float* C, const int64_t* counts, int M, int K, int dsub, int* nsplit,
                                     rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !C || !counts || M <= 0 || dsub <= 0) return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    hipLaunchKernelGGL(kmeans_split_empty_kernel, dim3((unsigned)M), dim3(RC_K), 0, (hipStream_t)stream, C,
                       reinterpret_cast<const long long*>(counts), dsub, nsplit);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

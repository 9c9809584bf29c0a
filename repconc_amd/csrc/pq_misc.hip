// Handle management plus the small byte/gather kernels of the path: decode (+ its gradient),
// centroid normalisation, code histogram.  (k-means kernels: kmeans.hip)
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
        case RC_ESHAPE: return "unsupported shape (K must be 256, M a divisor of D; search: M one of 8,12,16,24,32,48,64,96)";
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
    h->km_ctl = nullptr;
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
        for (void* r : h->scratch_retired) (void)hipFree(r);
        if (h->km_ctl) (void)hipFree(h->km_ctl);
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
    // Growth RETIRES the old block instead of freeing it (released by rc_destroy): work already queued on it — and any hipGraph a
    // caller captured over rc_* calls (the warm-up's round graph bakes the scratch pointer into its kernel nodes) — keeps
    // running on valid memory, each captured call sequence consistently on the block it was captured with; nothing here
    // synchronises the device, so growing inside a stream capture is harmless too.  Geometric growth bounds what is retained.
    if (bytes < h->scratch_bytes + h->scratch_bytes / 2) bytes = h->scratch_bytes + h->scratch_bytes / 2;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    if (h->scratch) h->scratch_retired.push_back(h->scratch);
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

// widths that are not a multiple of 4 floats (MCQ_M = 128, 256, 384, 768 at hidden_size 768): one thread per float
template <typename CodeT>
__global__ __launch_bounds__(256) void decode_scalar_kernel(const void* __restrict__ codes, const float* __restrict__ C,
                                                            int64_t n, int M, int dsub, float* __restrict__ out) {
    const int D = M * dsub;
    const int64_t total = n * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / D;
        const int c = (int)(i - row * D);
        const int m = c / dsub, j = c - m * dsub;
        const int code = load_code<CodeT>(codes, row * M + m) & (RC_K - 1);
        out[i] = C[((size_t)m * RC_K + code) * dsub + j];
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
    if (K != RC_K) return RC_ESHAPE;
    if (n == 0) return RC_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dsub % 4 != 0) {
        const unsigned gs = grid_for(h, n * M * dsub);
        if (code_dtype == RC_CODE_U8)
            hipLaunchKernelGGL(decode_scalar_kernel<uint8_t>, dim3(gs), dim3(256), 0, s, codes, C, n, M, dsub, out);
        else if (code_dtype == RC_CODE_I64)
            hipLaunchKernelGGL(decode_scalar_kernel<int64_t>, dim3(gs), dim3(256), 0, s, codes, C, n, M, dsub, out);
        else
            return RC_EINVAL;
        RC_LAUNCH_CHECK(h);
        return RC_OK;
    }
    const unsigned g = grid_for(h, n * (M * dsub / 4));
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

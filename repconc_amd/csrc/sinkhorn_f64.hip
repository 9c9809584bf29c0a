// General fp64 Sinkhorn plan (a-3 at the module boundary): `sinkhorn_algorithm(out[M,K,B] fp64, eps, T, ...)` of
// models/repconc/modeling_repconc.py:137-165 accepts ANY fp64 tensor; RepCONC.quantize only ever passes the negated centred fp32
// table (:56-62), and that case runs on the streaming sweep of sinkhorn.hip.  A cost tensor that is NOT exactly representable in
// fp32 cannot use the fp32 table: this file serves it (NotImplementedError before).  Not a hot path — two
// plain, deterministic kernels per iteration on the caller's fp64 tensor, log-domain potentials:
//
//     Q_kb = exp(L_kb + f_k + g_b),  L = out / eps
//     rows:  lse_k = log sum_b exp(L_kb + g_b)                (this rank's columns; g = 0 in the first iteration)
//     cols:  f_k = -log sum_ranks exp(lse_k[rank]),   g_b = -log sum_k exp(L_kb + f_k)
//
// T iterations of the reference = T row reductions and T - 1 column reductions; its global normalisation (:148-152) and the
// 1/K, 1/B factors (:159,:163-164) are constants of a column and cancel in the plan the caller rebuilds,
// Q[:, :, b] = softmax_k(L + f_T).  The cross-rank sum of :155-157 is the all-gather of the [M, K] lse values (the Python boundary
// moves them with torch.distributed; summed here in rank order: every rank computes bit-identical potentials).
// Every log-sum-exp subtracts its maximum: no overflow for any eps (the reference's exp(out / eps) overflows beyond 709).
#include "rc_common.h"

namespace {
constexpr int SK64_THREADS = 256;

__device__ __forceinline__ double sk64_block_max(double v, double* sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int o = SK64_THREADS / 2; o > 0; o >>= 1) {
        if (tid < o) sh[tid] = fmax(sh[tid], sh[tid + o]);
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double sk64_block_sum(double v, double* sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int o = SK64_THREADS / 2; o > 0; o >>= 1) {            // fixed tree: the same sum run to run
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// block (k, m): lse[m][k] = log sum_b exp(out[m][k][b] * inv_eps + g[m][b])   (-inf for B = 0)
__global__ __launch_bounds__(SK64_THREADS) void sk64_rows_kernel(const double* __restrict__ out, const double* __restrict__ g,
                                                                 int64_t B, int M, double inv_eps, double* __restrict__ lse) {
    __shared__ double sh[SK64_THREADS];
    const int k = blockIdx.x, m = blockIdx.y, tid = threadIdx.x;
    const double* row = out + ((size_t)m * RC_K + k) * (size_t)B;
    const double* gm = g ? g + (size_t)m * (size_t)B : nullptr;
    double mx = -INFINITY;
    for (int64_t b = tid; b < B; b += SK64_THREADS) mx = fmax(mx, row[b] * inv_eps + (gm ? gm[b] : 0.0));
    mx = sk64_block_max(mx, sh);
    double s = 0.0;
    if (mx > -INFINITY && mx < INFINITY)
        for (int64_t b = tid; b < B; b += SK64_THREADS) s += exp(row[b] * inv_eps + (gm ? gm[b] : 0.0) - mx);
    s = sk64_block_sum(s, sh);
    if (tid == 0) lse[(size_t)m * RC_K + k] = (mx > -INFINITY && mx < INFINITY) ? mx + log(s) : mx;
}

// block (column tile, m); thread = centroid k in the prologue (f_k from the G ranks' lse, rank order), = column b afterwards
__global__ __launch_bounds__(SK64_THREADS) void sk64_cols_kernel(const double* __restrict__ out, const double* __restrict__ lse_g,
                                                                 int G, int64_t B, int M, double inv_eps,
                                                                 double* __restrict__ f_out, double* __restrict__ g_out) {
    __shared__ double fk[RC_K];
    const int m = blockIdx.y, tid = threadIdx.x;
    {
        double mx = -INFINITY;
        for (int r = 0; r < G; ++r) mx = fmax(mx, lse_g[((size_t)r * M + m) * RC_K + tid]);
        double s = 0.0;
        if (mx > -INFINITY && mx < INFINITY)
            for (int r = 0; r < G; ++r) s += exp(lse_g[((size_t)r * M + m) * RC_K + tid] - mx);
        const double f = (mx > -INFINITY && mx < INFINITY) ? -(mx + log(s)) : -mx;
        fk[tid] = f;
        if (blockIdx.x == 0) f_out[(size_t)m * RC_K + tid] = f;
    }
    __syncthreads();
    if (!g_out) return;
    const int64_t b = (int64_t)blockIdx.x * SK64_THREADS + tid;
    if (b >= B) return;
    const double* col = out + (size_t)m * RC_K * (size_t)B + b;    // stride B between k: coalesced over the block's columns
    double mx = -INFINITY;
    for (int k = 0; k < RC_K; ++k) mx = fmax(mx, col[(size_t)k * B] * inv_eps + fk[k]);
    double s = 0.0;
    if (mx > -INFINITY && mx < INFINITY)
        for (int k = 0; k < RC_K; ++k) s += exp(col[(size_t)k * B] * inv_eps + fk[k] - mx);
    g_out[(size_t)m * (size_t)B + b] = (mx > -INFINITY && mx < INFINITY) ? -(mx + log(s)) : -mx;
}
}  // namespace

extern "C" int rc_sk64_rows(rc_handle_t h, const double* out, const double* g, int64_t B, int M, int K, double eps, double* lse,
                            rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !lse || B < 0 || M <= 0 || (B > 0 && !out) || !(eps > 0.0)) return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    hipLaunchKernelGGL(sk64_rows_kernel, dim3(RC_K, (unsigned)M), dim3(SK64_THREADS), 0, (hipStream_t)stream, out, g, B, M,
                       1.0 / eps, lse);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

extern "C" int rc_sk64_cols(rc_handle_t h, const double* out, const double* lse_gathered, int G, int64_t B, int M, int K,
                            double eps, double* f_out, double* g_out, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !lse_gathered || !f_out || G <= 0 || B < 0 || M <= 0 || (B > 0 && !out) || !(eps > 0.0)) return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    const unsigned tiles = (unsigned)((B + SK64_THREADS - 1) / SK64_THREADS);
    hipLaunchKernelGGL(sk64_cols_kernel, dim3(tiles ? tiles : 1u, (unsigned)M), dim3(SK64_THREADS), 0, (hipStream_t)stream, out,
                       lse_gathered, G, B, M, 1.0 / eps, f_out, B > 0 ? g_out : nullptr);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

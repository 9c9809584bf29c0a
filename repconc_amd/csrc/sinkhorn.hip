// Sinkhorn-Knopp uniform-assignment solve on the centred distance table, with potentials.
//
// Reference: sinkhorn_algorithm, models/repconc/modeling_repconc.py:137-165, called from
// RepCONC.quantize (:53-66) on out = -centred.double().transpose(1,2), i.e. L[m,k,b] =
// -d[m,b,k]/eps in fp64.  The reference keeps the whole fp64 matrix Q and rescales it in place;
// here only the fp32 centred distances d[M,B,K] live in HBM (4 B/entry, read once per sweep) and
// the scalings are carried as potentials f[M,K], g[M,B]:
//
//   sweep 0        rows_k = sum_b exp(L_kb)                        f_k  = -log rows_k
//   sweep t>=1     w_kb = exp(L_kb + f_k + g_b)      (w <= 1 by construction, no max pass)
//                  c_b  = sum_k w_kb                               g_b -= log c_b
//                  rows_k = sum_b w_kb / c_b                       f_k -= log rows_k
//   last sweep     code_b = argmax_k (L_kb + f_k)    (first maximum)
//
// One exp per matrix entry per sweep.  `w <= 1`: after the previous column normalisation
// sum_k exp(L_kb+f'_k+g_b) = 1, and the row update divides entry (k,b) by rows_k >= that entry.
//
// Work split: a column (one document, 256 centroids, 1 KiB of fp32) is owned by one ROW of 16
// lanes, 16 consecutive k per lane (4 x 16-byte loads; a wave reads 4 columns = 4 KiB per sweep
// step).  The column sum is a 4-step DPP rotate-add inside the row; the row sums accumulate in
// registers (16 fp64 per lane) and are reduced once per block through LDS, then across blocks in
// a fixed order by sk_update_kernel — sums are deterministic run to run.
#include "rc_common.h"

#define SK_EPL 16                 // matrix entries (k) per lane
#define SK_GROUP (RC_K / SK_EPL)  // 16 lanes own one column
#define SK_THREADS 256
#define SK_GROUPS_PER_BLOCK (SK_THREADS / SK_GROUP)  // 16 columns in flight per block

// exp(t) for t <= ~700, |relative error| ~2e-16: n = rint(t*log2 e), r = t - n*ln2 (two-step),
// degree-12 Taylor polynomial on |r| <= 0.3466 (remainder r^13/13! < 1.7e-16), scaled by 2^n
// with v_ldexp_f64 (handles gradual underflow; t < -745 gives 0).
__device__ __forceinline__ double sk_exp(double t) {
    const double n = __builtin_rint(t * 1.4426950408889634074);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, t);
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}

__device__ __forceinline__ void sk_load_col(const float* __restrict__ p, float (&v)[SK_EPL]) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int j = 0; j < SK_EPL / 4; ++j) {
        const float4 a = q[j];
        v[4 * j] = a.x; v[4 * j + 1] = a.y; v[4 * j + 2] = a.z; v[4 * j + 3] = a.w;
    }
}

// One sweep.  grid = (blocks per m, M).  FIRST: sweep 0 (no potentials, no column
// normalisation).  part[m][blk][k] receives the block's row sums.
template <bool FIRST>
__global__ __launch_bounds__(SK_THREADS) void sk_pass_kernel(const float* __restrict__ d,
                                                             const double* __restrict__ f,
                                                             const double* __restrict__ g,
                                                             double* __restrict__ colsum,
                                                             double* __restrict__ part, int64_t B,
                                                             int cols_per_block, double ninv_eps) {
    __shared__ double red[SK_GROUPS_PER_BLOCK][RC_K];  // 32 KiB
    const int m = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & (SK_GROUP - 1);
    const int grp = tid / SK_GROUP;
    const int64_t c0 = (int64_t)blockIdx.x * cols_per_block;
    const int64_t c1 = (c0 + cols_per_block < B) ? c0 + cols_per_block : B;

    double fk[SK_EPL], R[SK_EPL];
#pragma unroll
    for (int i = 0; i < SK_EPL; ++i) {
        fk[i] = FIRST ? 0.0 : f[(size_t)m * RC_K + lane * SK_EPL + i];
        R[i] = 0.0;
    }
    const float* dm = d + (size_t)m * B * RC_K + lane * SK_EPL;
    const double* gm = g + (size_t)m * B;
    double* cm = colsum + (size_t)m * B;

    int64_t col = c0 + grp;
    float cur[SK_EPL];
    if (col < c1) sk_load_col(dm + col * RC_K, cur);
    while (col < c1) {
        const int64_t nxt = col + SK_GROUPS_PER_BLOCK;
        float pre[SK_EPL];
        if (nxt < c1) sk_load_col(dm + nxt * RC_K, pre);  // prefetch the next column
        double w[SK_EPL];
        if constexpr (FIRST) {
#pragma unroll
            for (int i = 0; i < SK_EPL; ++i) {
                w[i] = sk_exp((double)cur[i] * ninv_eps);
                R[i] += w[i];
            }
        } else {
            const double gb = gm[col];
            double c = 0.0;
#pragma unroll
            for (int i = 0; i < SK_EPL; ++i) {
                w[i] = sk_exp(__builtin_fma((double)cur[i], ninv_eps, fk[i] + gb));
                c += w[i];
            }
            c = rc_row16_allreduce_sum(c);
            const double rc = 1.0 / c;
#pragma unroll
            for (int i = 0; i < SK_EPL; ++i) R[i] = __builtin_fma(w[i], rc, R[i]);
            if (lane == 0) cm[col] = c;
        }
        if (nxt < c1) {
#pragma unroll
            for (int i = 0; i < SK_EPL; ++i) cur[i] = pre[i];
        }
        col = nxt;
    }
    // block reduction of the row sums, fixed order over the 16 column groups
#pragma unroll
    for (int i = 0; i < SK_EPL; ++i) red[grp][lane * SK_EPL + i] = R[i];
    __syncthreads();
    double s = red[0][tid];
#pragma unroll
    for (int q = 1; q < SK_GROUPS_PER_BLOCK; ++q) s += red[q][tid];
    part[((size_t)m * gridDim.x + blockIdx.x) * RC_K + tid] = s;
}

// rows[m][k] = sum over blocks of part[m][blk][k], blocks ascending.
__global__ __launch_bounds__(RC_K) void sk_reduce_part_kernel(const double* __restrict__ part, int nblk,
                                                              double* __restrict__ rows) {
    const int m = blockIdx.x, k = threadIdx.x;
    const double* p = part + (size_t)m * nblk * RC_K + k;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += p[(size_t)i * RC_K];
    rows[(size_t)m * RC_K + k] = s;
}

// Blocks [0,M): f[m][k] = (first ? 0 : f) - log(sum_r rows_all[r][m][k]), ranks ascending.
// Blocks [M, ..): g[i] -= log(colsum[i]) over the M*B columns (skipped on the first sweep).
__global__ __launch_bounds__(RC_K) void sk_update_kernel(const double* __restrict__ rows_all, int G,
                                                         double* __restrict__ f, double* __restrict__ g,
                                                         const double* __restrict__ colsum, int64_t MB,
                                                         int M, int first, int* __restrict__ flags) {
    bool bad = false;
    if ((int)blockIdx.x < M) {
        const int m = blockIdx.x, k = threadIdx.x;
        double s = 0.0;
        for (int r = 0; r < G; ++r) s += rows_all[((size_t)r * M + m) * RC_K + k];
        const double fo = first ? 0.0 : f[(size_t)m * RC_K + k];
        f[(size_t)m * RC_K + k] = fo - log(s);
        bad = !(s > 0.0) || !(s < INFINITY);
    } else if (!first) {
        const int64_t stride = (int64_t)(gridDim.x - M) * blockDim.x;
        for (int64_t i = (int64_t)(blockIdx.x - M) * blockDim.x + threadIdx.x; i < MB; i += stride) {
            const double c = colsum[i];
            g[i] -= log(c);
            bad |= !(c > 0.0) || !(c < INFINITY);
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, RC_FLAG_NONFINITE);
}

__global__ __launch_bounds__(256) void sk_zero_kernel(double* __restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = 0.0;
}

// code[b][m] = argmax_k (L_kb + f_k), first maximum.  Same column ownership as the sweeps; the
// (value, index) pair is reduced across the 16 lanes with rotations; ties keep the lower k.
__global__ __launch_bounds__(SK_THREADS) void sk_argmax_kernel(const float* __restrict__ d,
                                                               const double* __restrict__ f, int64_t B,
                                                               int M, int cols_per_block, double ninv_eps,
                                                               uint8_t* __restrict__ codes_u8,
                                                               int64_t* __restrict__ codes_i64) {
    const int m = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & (SK_GROUP - 1);
    const int grp = tid / SK_GROUP;
    const int64_t c0 = (int64_t)blockIdx.x * cols_per_block;
    const int64_t c1 = (c0 + cols_per_block < B) ? c0 + cols_per_block : B;
    double fk[SK_EPL];
#pragma unroll
    for (int i = 0; i < SK_EPL; ++i) fk[i] = f[(size_t)m * RC_K + lane * SK_EPL + i];
    const float* dm = d + (size_t)m * B * RC_K + lane * SK_EPL;
    for (int64_t col = c0 + grp; col < c1; col += SK_GROUPS_PER_BLOCK) {
        float cur[SK_EPL];
        sk_load_col(dm + col * RC_K, cur);
        double best = __builtin_fma((double)cur[0], ninv_eps, fk[0]);
        int bi = lane * SK_EPL;
#pragma unroll
        for (int i = 1; i < SK_EPL; ++i) {
            const double s = __builtin_fma((double)cur[i], ninv_eps, fk[i]);
            if (s > best) { best = s; bi = lane * SK_EPL + i; }
        }
#define SK_ARGMAX_STEP(N)                                                     \
        {                                                                     \
            const double ob = rc_dpp_row_ror<N>(best);                        \
            const int oi = rc_dpp_row_ror<N>(bi);                             \
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; } \
        }
        SK_ARGMAX_STEP(8) SK_ARGMAX_STEP(4) SK_ARGMAX_STEP(2) SK_ARGMAX_STEP(1)
#undef SK_ARGMAX_STEP
        if (lane == 0) {
            if (codes_u8) codes_u8[col * M + m] = (uint8_t)bi;
            if (codes_i64) codes_i64[col * M + m] = (int64_t)bi;
        }
    }
}

// ------------------------------------------------------------------------------------------ host
// columns per block: large enough to amortise the per-block reduction, small enough that the
// grid has >= ~16 blocks per CU to balance the tail.
static int sk_cols_per_block(int64_t B, int M) {
    int cpb = 512;
    while (cpb > 64 && ((B + cpb - 1) / cpb) * M < 4096) cpb >>= 1;
    return cpb;
}

extern "C" size_t rc_sk_pass_ws_bytes(int64_t B, int M, int K) {
    if (B <= 0 || M <= 0 || K != RC_K) return 0;
    const int cpb = sk_cols_per_block(B, M);
    const int64_t nblk = (B + cpb - 1) / cpb;
    return rc_align_up((size_t)M * nblk * RC_K * sizeof(double), 256);
}

extern "C" int rc_sk_pass(rc_handle_t h, const float* d, const double* f, const double* g, double* colsum,
                          double* rows, int64_t B, int M, int K, double eps, int first, void* ws,
                          size_t ws_bytes, rc_stream_t stream) {
    if (!h || !d || !rows || B <= 0 || M <= 0 || !(eps > 0.0)) return RC_EINVAL;
    if (!first && (!f || !g || !colsum)) return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    if (!ws || ws_bytes < rc_sk_pass_ws_bytes(B, M, K)) return RC_EWORKSPACE;
    const int cpb = sk_cols_per_block(B, M);
    const int64_t nblk = (B + cpb - 1) / cpb;
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)ws;
    const double ninv = -1.0 / eps;
    dim3 grid((unsigned)nblk, (unsigned)M);
    if (first) {
        hipLaunchKernelGGL(sk_pass_kernel<true>, grid, dim3(SK_THREADS), 0, s, d, f, g, colsum, part, B, cpb, ninv);
    } else {
        rc_prof_mark(h, RC_PROF_SK_PASS, s);
        hipLaunchKernelGGL(sk_pass_kernel<false>, grid, dim3(SK_THREADS), 0, s, d, f, g, colsum, part, B, cpb, ninv);
        rc_prof_mark(h, RC_PROF_SK_PASS, s);
    }
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(sk_reduce_part_kernel, dim3(M), dim3(RC_K), 0, s, part, (int)nblk, rows);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

extern "C" int rc_sk_update(rc_handle_t h, const double* rows_all, int G, double* f, double* g,
                            const double* colsum, int64_t B, int M, int K, int first, int* flags,
                            rc_stream_t stream) {
    if (!h || !rows_all || !f || !flags || G <= 0 || B <= 0 || M <= 0) return RC_EINVAL;
    if (!first && (!g || !colsum)) return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t MB = (int64_t)M * B;
    int64_t extra = 0;
    if (!first) {
        extra = (MB + RC_K * 4 - 1) / (RC_K * 4);
        const int64_t cap = (int64_t)h->num_cus * 8;
        if (extra > cap) extra = cap;
        if (extra < 1) extra = 1;
    }
    hipLaunchKernelGGL(sk_update_kernel, dim3((unsigned)(M + extra)), dim3(RC_K), 0, s, rows_all, G, f, g, colsum,
                       MB, M, first, flags);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

extern "C" int rc_sk_argmax(rc_handle_t h, const float* d, const double* f, int64_t B, int M, int K, double eps,
                            uint8_t* codes_u8, int64_t* codes_i64, rc_stream_t stream) {
    if (!h || !d || !f || B <= 0 || M <= 0 || !(eps > 0.0) || (!codes_u8 && !codes_i64)) return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    const int cpb = sk_cols_per_block(B, M);
    const int64_t nblk = (B + cpb - 1) / cpb;
    hipLaunchKernelGGL(sk_argmax_kernel, dim3((unsigned)nblk, (unsigned)M), dim3(SK_THREADS), 0,
                       (hipStream_t)stream, d, f, B, M, cpb, -1.0 / eps, codes_u8, codes_i64);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// ---- one-call single-rank constrained assignment -----------------------------------------
struct sk_ws_layout {
    size_t d, minmax, f, g, colsum, rows, part, dist_ws, total;
};
static sk_ws_layout sk_layout(int64_t B, int M) {
    sk_ws_layout L;
    size_t o = 0;
    L.d = o;       o += rc_align_up((size_t)M * B * RC_K * sizeof(float), 256);
    L.minmax = o;  o += rc_align_up((size_t)2 * M * sizeof(float), 256);
    L.f = o;       o += rc_align_up((size_t)M * RC_K * sizeof(double), 256);
    L.g = o;       o += rc_align_up((size_t)M * B * sizeof(double), 256);
    L.colsum = o;  o += rc_align_up((size_t)M * B * sizeof(double), 256);
    L.rows = o;    o += rc_align_up((size_t)M * RC_K * sizeof(double), 256);
    L.part = o;    o += rc_sk_pass_ws_bytes(B, M, RC_K);
    L.dist_ws = o; o += rc_pq_dist_table_ws_bytes(B, M);
    L.total = o;
    return L;
}

extern "C" size_t rc_pq_assign_sinkhorn_ws_bytes(int64_t B, int M, int K) {
    if (B <= 0 || M <= 0 || K != RC_K) return 0;
    return sk_layout(B, M).total;
}

extern "C" int rc_pq_assign_sinkhorn(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B, int D,
                                     int M, int K, double eps, int iters, uint8_t* codes_u8, int64_t* codes_i64,
                                     int* flags, void* ws, size_t ws_bytes, rc_stream_t stream) {
    if (!h || !x || !C || !flags || B < 0 || M <= 0 || iters < 1 || !(eps > 0.0) || (!codes_u8 && !codes_i64))
        return RC_EINVAL;
    if (K != RC_K || D % M != 0 || !rc_dsub_supported(D / M)) return RC_ESHAPE;
    if (B == 0) return RC_OK;
    if (B == 1) {
        // One column: the first row normalisation (:158) makes every entry Q_k/Q_k = 1 exactly, so the
        // reference's argmax is a K-way exact tie and returns index 0 for every sub-quantiser.
        hipStream_t s1 = (hipStream_t)stream;
        if (codes_u8) RC_HIP_CHECK(h, hipMemsetAsync(codes_u8, 0, (size_t)M, s1));
        if (codes_i64) RC_HIP_CHECK(h, hipMemsetAsync(codes_i64, 0, (size_t)M * sizeof(int64_t), s1));
        return RC_OK;
    }
    const sk_ws_layout L = sk_layout(B, M);
    if (!ws || ws_bytes < L.total) return RC_EWORKSPACE;
    char* w = (char*)ws;
    float* d = (float*)(w + L.d);
    float* minmax = (float*)(w + L.minmax);
    double* f = (double*)(w + L.f);
    double* g = (double*)(w + L.g);
    double* colsum = (double*)(w + L.colsum);
    double* rows = (double*)(w + L.rows);
    void* part = w + L.part;
    const size_t part_bytes = rc_sk_pass_ws_bytes(B, M, K);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = rc_pq_dist_table(h, x, ldx, C, B, D, M, K, d, minmax, w + L.dist_ws,
                               rc_pq_dist_table_ws_bytes(B, M), stream)) != RC_OK) return rc;
    if ((rc = rc_pq_centre(h, d, minmax, B, M, K, stream)) != RC_OK) return rc;
    {
        const int64_t n = (int64_t)M * B;
        int64_t gx = (n + 255) / 256;
        if (gx > (int64_t)h->num_cus * 8) gx = (int64_t)h->num_cus * 8;
        hipLaunchKernelGGL(sk_zero_kernel, dim3((unsigned)gx), dim3(256), 0, s, g, n);
        RC_LAUNCH_CHECK(h);
    }
    if ((rc = rc_sk_pass(h, d, f, g, colsum, rows, B, M, K, eps, 1, part, part_bytes, stream)) != RC_OK) return rc;
    if ((rc = rc_sk_update(h, rows, 1, f, g, colsum, B, M, K, 1, flags, stream)) != RC_OK) return rc;
    for (int t = 1; t < iters; ++t) {
        if ((rc = rc_sk_pass(h, d, f, g, colsum, rows, B, M, K, eps, 0, part, part_bytes, stream)) != RC_OK) return rc;
        if ((rc = rc_sk_update(h, rows, 1, f, g, colsum, B, M, K, 0, flags, stream)) != RC_OK) return rc;
    }
    return rc_sk_argmax(h, d, f, B, M, K, eps, codes_u8, codes_i64, stream);
}

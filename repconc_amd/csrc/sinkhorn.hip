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
// lanes, 16 k per lane in four groups of 4 (sk_kidx; 4 x 16-byte loads, each a full 256-byte run
// per row; a wave reads 4 columns = 4 KiB per sweep step).  The column sum is a 4-step DPP rotate-add inside the row; the row sums accumulate in
// registers (16 fp64 per lane) and are reduced once per block through LDS, then across blocks in
// a fixed order by sk_update_kernel — sums are deterministic run to run.
#include "rc_common.h"

#include <stdlib.h>

#define SK_EPL 16                 // matrix entries (k) per lane
#define SK_GROUP (RC_K / SK_EPL)  // 16 lanes own one column
#define SK_THREADS 256
#define SK_GROUPS_PER_BLOCK (SK_THREADS / SK_GROUP)  // 16 columns in flight per block

// ---- exp for the sweeps ---------------------------------------------------------------------------
// The exponent is produced directly in units of 1/N octave (N = 2^TB): the factor N/ln2 is folded into
// -1/eps and into the potentials, so with u = (L + f)*N/ln2 and the column potential split into an
// integer part gq and a residual (below),
//     exp(L + f + g) ~ 2^((u + gq)/N) = 2^e * 2^(j/N) * 2^(r/N),   n = rint(u), r = u - n in [-.5,.5],
//     j = (n + gq) & (N-1),  e = (n + gq) >> TB.
// 2^(j/N) comes from an N-entry table staged in LDS (rc_handle owns the device copy, built on the host
// with exp2l), 2^(r/N) - 1 from a short Taylor polynomial in z = r ln2/N: |z| <= ln2/2N, degree 3 at
// N = 2048 (remainder z^4/24 < 4e-17), degree 4 at N = 256.  Relative error ~2e-16 plus the rounding
// of u itself (|u| eps_64 N/ln2 -> <1e-13 in the exponent, the same as rounding L+f+g directly).
//
// Column potential: g_b enters only through 2^(g_b N/ln2 / N).  Its integer part gq = rint(g_b N/ln2)
// is added to n as an INTEGER; the residual factor 2^(rg/N), rg = g_b N/ln2 - gq, is common to the
// whole column, cancels in w/colsum, and is put back by sk_update_kernel:
//     log(colsum_true) = log(colsum_stored) + rg ln2/N.
template <int TB>
__device__ __forceinline__ double sk_exp2n(double u, int gq, const double* __restrict__ tab) {
    constexpr int N = 1 << TB;
    constexpr double Z = 0.69314718055994530942 / (double)N;  // ln2 / N
    const double n = __builtin_rint(u);
    const double r = u - n;
    const int ni = (int)n + gq;
    const double T = tab[ni & (N - 1)];
    double q;
    if constexpr (TB >= 11) {
        q = __builtin_fma(Z * Z * Z / 6.0, r, Z * Z / 2.0);
    } else {
        q = __builtin_fma(Z * Z * Z * Z / 24.0, r, Z * Z * Z / 6.0);
        q = __builtin_fma(q, r, Z * Z / 2.0);
    }
    q = __builtin_fma(q, r, Z);
    q = q * r;
    return __builtin_ldexp(__builtin_fma(T, q, T), ni >> TB);
}

// Column ownership.  A column (one document, 256 centroids, 1 KiB of fp32) is owned by G = 256/EPL
// consecutive lanes, EPL entries per lane in EPL/4 groups of 4:  k = 4G*(i/4) + 4*lane + (i%4).  Each
// of the lane's 16-byte loads is then part of a 16G-byte run read by the G lanes together (whole cache
// lines per instruction) — a private contiguous run per lane costs 4x the L1/TA line touches.
template <int EPL>
__device__ __forceinline__ int sk_kidx(int lane, int i) {
    return ((i >> 2) * (4 * (RC_K / EPL))) + (lane << 2) + (i & 3);
}

// p = column base + 4*lane
template <int EPL>
__device__ __forceinline__ void sk_load_col(const float* __restrict__ p, float (&v)[EPL]) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int j = 0; j < EPL / 4; ++j) {
        const float4 a = q[(RC_K / EPL) * j];
        v[4 * j] = a.x; v[4 * j + 1] = a.y; v[4 * j + 2] = a.z; v[4 * j + 3] = a.w;
    }
}

// sum over the G lanes that own a column; every lane of the group gets the same bits
template <int G>
__device__ __forceinline__ double sk_group_sum(double v) {
    v = rc_row16_allreduce_sum(v);
    if constexpr (G == 32) {
        // lanes l and l^16 (the two DPP rows of a 32-lane half wave): ds_swizzle xor 0x10
        const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F);
        const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);
        v += __hiloint2double(hi, lo);
    }
    return v;
}

// One column step of a sweep (not FIRST): exponentials, column sum, normalised row-sum update.
template <int TB, int EPL, int ABL>
__device__ __forceinline__ void sk_column(const float (&x)[EPL], const double (&fk)[EPL], double (&R)[EPL],
                                          double gscaled, double nscale_eps, const double* __restrict__ tab,
                                          double* __restrict__ csum_out, bool writer) {
    const int gq = (int)__builtin_rint(gscaled);
    double w[EPL];
    double c = 0.0;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        if (ABL == 1) w[i] = __builtin_fma((double)x[i], nscale_eps, fk[i]) + gq;
        else w[i] = sk_exp2n<TB>(__builtin_fma((double)x[i], nscale_eps, fk[i]), gq, tab);
        c += w[i];
    }
    c = sk_group_sum<RC_K / EPL>(c);
    const double rc = 1.0 / c;
#pragma unroll
    for (int i = 0; i < EPL; ++i) R[i] = __builtin_fma(w[i], rc, R[i]);
    if (writer) *csum_out = c;
}

// One sweep.  grid = (blocks per m, M).  FIRST: sweep 0 (no potentials, no column normalisation).
// part[m][blk][k] receives the block's row sums.  `scale` = N/ln2, `nscale_eps` = -scale/eps.
// ABL: development ablations (1 = no exp, 2 = no HBM loads).
template <bool FIRST, int TB, int EPL, int ABL = 0>
__global__ __launch_bounds__(SK_THREADS) void sk_pass_kernel(const float* __restrict__ d,
                                                             const double* __restrict__ f,
                                                             const double* __restrict__ g,
                                                             double* __restrict__ colsum,
                                                             double* __restrict__ part, int64_t B,
                                                             int cols_per_block, double nscale_eps, double scale,
                                                             const double* __restrict__ exp2_tab) {
    constexpr int N = 1 << TB;
    constexpr int G = RC_K / EPL;          // lanes per column
    constexpr int NG = SK_THREADS / G;     // columns in flight per block
    extern __shared__ __attribute__((aligned(16))) double sk_smem[];
    double* tab = sk_smem;                                                // [N]
    double(*red)[RC_K] = reinterpret_cast<double(*)[RC_K]>(sk_smem + N);  // [NG][256]
    const int m = blockIdx.y;
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += SK_THREADS) tab[i] = exp2_tab[i];
    __syncthreads();
    const int lane = tid & (G - 1);
    const int grp = tid / G;
    const int64_t c0 = (int64_t)blockIdx.x * cols_per_block;
    const int64_t c1 = (c0 + cols_per_block < B) ? c0 + cols_per_block : B;

    double fk[EPL], R[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        fk[i] = FIRST ? 0.0 : f[(size_t)m * RC_K + sk_kidx<EPL>(lane, i)] * scale;
        R[i] = 0.0;
    }
    const float* dm = d + (size_t)m * B * RC_K + lane * 4;
    const double* gm = g + (size_t)m * B;
    double* cm = colsum + (size_t)m * B;

    // two columns per trip, register buffers ping-pong so the prefetched data is never copied
    float xa[EPL], xb[EPL];
    int64_t col = c0 + grp;
    if (col < c1) sk_load_col<EPL>(dm + col * RC_K, xa);
    while (col < c1) {
        const int64_t colb = col + NG;
        if (ABL == 2) {
#pragma unroll
            for (int i = 0; i < EPL; ++i) xb[i] = xa[i] * 0.999f;
        } else if (colb < c1) sk_load_col<EPL>(dm + colb * RC_K, xb);
        if constexpr (FIRST) {
#pragma unroll
            for (int i = 0; i < EPL; ++i) R[i] += sk_exp2n<TB>((double)xa[i] * nscale_eps, 0, tab);
        } else {
            sk_column<TB, EPL, ABL>(xa, fk, R, gm[col] * scale, nscale_eps, tab, cm + col, lane == 0);
        }
        if (colb >= c1) break;
        const int64_t cola = colb + NG;
        if (ABL == 2) {
#pragma unroll
            for (int i = 0; i < EPL; ++i) xa[i] = xb[i] * 0.999f;
        } else if (cola < c1) sk_load_col<EPL>(dm + cola * RC_K, xa);
        if constexpr (FIRST) {
#pragma unroll
            for (int i = 0; i < EPL; ++i) R[i] += sk_exp2n<TB>((double)xb[i] * nscale_eps, 0, tab);
        } else {
            sk_column<TB, EPL, ABL>(xb, fk, R, gm[colb] * scale, nscale_eps, tab, cm + colb, lane == 0);
        }
        col = cola;
    }
    // block reduction of the row sums, fixed order over the NG column groups
#pragma unroll
    for (int i = 0; i < EPL; ++i) red[grp][sk_kidx<EPL>(lane, i)] = R[i];
    __syncthreads();
    double s = red[0][tid];
#pragma unroll
    for (int q = 1; q < NG; ++q) s += red[q][tid];
    part[((size_t)m * gridDim.x + blockIdx.x) * RC_K + tid] = s;
}

// rows[m][k] = sum over blocks of part[m][blk][k], blocks ascending.
__global__ __launch_bounds__(RC_K) void sk_reduce_part_kernel(const double* __restrict__ part, int nblk,
                                                              double* __restrict__ rows) {
    const int m = blockIdx.x, k = threadIdx.x;
    const double* p = part + (size_t)m * nblk * RC_K + k;
    double s = 0.0;
    int i = 0;
    for (; i + 16 <= nblk; i += 16) {   // 16 independent loads in flight, summed in block order
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[(size_t)(i + j) * RC_K];
#pragma unroll
        for (int j = 0; j < 16; ++j) s += v[j];
    }
    for (; i < nblk; ++i) s += p[(size_t)i * RC_K];
    rows[(size_t)m * RC_K + k] = s;
}

// Blocks [0,M): f[m][k] = (first ? 0 : f) - log(sum_j src[j*stride_j + m*stride_m + k]), j
// ascending: src is either the all-gathered per-rank row sums (j = rank) or, on a single rank,
// the block partials of the sweep (j = block) — the same fixed-order sum either way.
// Blocks [M, ..): g[i] -= log(colsum[i]) over the M*B columns (skipped on the first sweep).
__global__ __launch_bounds__(RC_K) void sk_update_kernel(const double* __restrict__ rows_all, int G,
                                                         int64_t stride_j, int64_t stride_m,
                                                         double* __restrict__ f, double* __restrict__ g,
                                                         const double* __restrict__ colsum, int64_t MB,
                                                         int M, int first, double scale,
                                                         int* __restrict__ flags) {
    bool bad = false;
    if ((int)blockIdx.x < M) {
        const int m = blockIdx.x, k = threadIdx.x;
        const double* p = rows_all + (size_t)m * stride_m + k;
        double s = 0.0;
        int r = 0;
        for (; r + 16 <= G; r += 16) {
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = p[(size_t)(r + j) * stride_j];
#pragma unroll
            for (int j = 0; j < 16; ++j) s += v[j];
        }
        for (; r < G; ++r) s += p[(size_t)r * stride_j];
        const double fo = first ? 0.0 : f[(size_t)m * RC_K + k];
        f[(size_t)m * RC_K + k] = fo - log(s);
        bad = !(s > 0.0) || !(s < INFINITY);
    } else if (!first) {
        const int64_t stride = (int64_t)(gridDim.x - M) * blockDim.x;
        for (int64_t i = (int64_t)(blockIdx.x - M) * blockDim.x + threadIdx.x; i < MB; i += stride) {
            // colsum was accumulated with the integer part of g*scale only (sk_exp2n): put the
            // residual factor 2^(rg/N) back, log(colsum_true) = log(colsum) + rg/scale
            const double c = colsum[i];
            const double go = g[i];
            const double gs = go * scale;
            const double rg = gs - __builtin_rint(gs);
            g[i] = go - (log(c) + rg / scale);
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
    for (int i = 0; i < SK_EPL; ++i) fk[i] = f[(size_t)m * RC_K + sk_kidx<SK_EPL>(lane, i)];
    const float* dm = d + (size_t)m * B * RC_K + lane * 4;
    for (int64_t col = c0 + grp; col < c1; col += SK_GROUPS_PER_BLOCK) {
        float cur[SK_EPL];
        sk_load_col<SK_EPL>(dm + col * RC_K, cur);
        double best = __builtin_fma((double)cur[0], ninv_eps, fk[0]);
        int bi = sk_kidx<SK_EPL>(lane, 0);
#pragma unroll
        for (int i = 1; i < SK_EPL; ++i) {   // k ascends with i inside a lane: strict > keeps the first maximum
            const double s = __builtin_fma((double)cur[i], ninv_eps, fk[i]);
            if (s > best) { best = s; bi = sk_kidx<SK_EPL>(lane, i); }
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

// exp table resolution: N = 2^TB entries (16 KiB of LDS at TB = 11).  RC_SK_TB=8 selects the 256-entry
// table + one more polynomial term (kept for A/B measurements).
static int sk_tb() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("RC_SK_TB");
        v = (e && atoi(e) == 8) ? 8 : 11;
    }
    return v;
}
static double sk_scale() { return (double)(1 << sk_tb()) / 0.69314718055994530942; }

// launch one sweep; block partials land in `part` [M][nblk][K]
static int sk_launch_pass(rc_handle_t h, const float* d, const double* f, const double* g, double* colsum,
                          double* part, int64_t B, int M, double eps, int first, hipStream_t s) {
    const int cpb = sk_cols_per_block(B, M);
    const int64_t nblk = (B + cpb - 1) / cpb;
    const int tb = sk_tb();
    const double* tab = rc_exp2_table(h, tb);
    if (!tab) return RC_EHIP;
    const double scale = sk_scale();
    const double nse = -scale / eps;
    dim3 grid((unsigned)nblk, (unsigned)M);
    static int epl = -1, abl = -1;
    if (epl < 0) { const char* e = getenv("RC_SK_EPL"); epl = (e && atoi(e) == 8) ? 8 : 16; }
    if (abl < 0) { const char* e = getenv("RC_SK_ABLATE"); abl = e ? atoi(e) : 0; }
    const size_t lds = ((size_t)(1 << tb) + (size_t)(SK_THREADS / (RC_K / epl)) * RC_K) * sizeof(double);
#define SK_LAUNCH(FIRST_, TB_, EPL_, ABL_)                                                                     \
    hipLaunchKernelGGL((sk_pass_kernel<FIRST_, TB_, EPL_, ABL_>), grid, dim3(SK_THREADS), lds, s, d, f, g, colsum, \
                       part, B, cpb, nse, scale, tab)
    if (first) {
        if (tb == 11 && epl == 8) SK_LAUNCH(true, 11, 8, 0);
        else if (tb == 11) SK_LAUNCH(true, 11, 16, 0);
        else if (epl == 8) SK_LAUNCH(true, 8, 8, 0);
        else SK_LAUNCH(true, 8, 16, 0);
    } else {
        rc_prof_mark(h, RC_PROF_SK_PASS, s);
        if (abl == 1 && epl == 8) SK_LAUNCH(false, 11, 8, 1);
        else if (abl == 2 && epl == 8) SK_LAUNCH(false, 11, 8, 2);
        else if (abl == 1) SK_LAUNCH(false, 11, 16, 1);
        else if (abl == 2) SK_LAUNCH(false, 11, 16, 2);
        else if (tb == 11 && epl == 8) SK_LAUNCH(false, 11, 8, 0);
        else if (tb == 11) SK_LAUNCH(false, 11, 16, 0);
        else if (epl == 8) SK_LAUNCH(false, 8, 8, 0);
        else SK_LAUNCH(false, 8, 16, 0);
        rc_prof_mark(h, RC_PROF_SK_PASS, s);
    }
#undef SK_LAUNCH
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// f / g update from `n` addends per (m,k) laid out as src[j*stride_j + m*stride_m + k]
static int sk_launch_update(rc_handle_t h, const double* src, int n, int64_t stride_j, int64_t stride_m, double* f,
                            double* g, const double* colsum, int64_t B, int M, int first, int* flags,
                            hipStream_t s) {
    const int64_t MB = (int64_t)M * B;
    int64_t extra = 0;
    if (!first) {
        extra = (MB + RC_K * 4 - 1) / (RC_K * 4);
        const int64_t cap = (int64_t)h->num_cus * 8;
        if (extra > cap) extra = cap;
        if (extra < 1) extra = 1;
    }
    hipLaunchKernelGGL(sk_update_kernel, dim3((unsigned)(M + extra)), dim3(RC_K), 0, s, src, n, stride_j, stride_m, f, g,
                       colsum, MB, M, first, sk_scale(), flags);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
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
    const int rc = sk_launch_pass(h, d, f, g, colsum, part, B, M, eps, first, s);
    if (rc != RC_OK) return rc;
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
    return sk_launch_update(h, rows_all, G, (int64_t)M * RC_K, RC_K, f, g, colsum, B, M, first, flags,
                            (hipStream_t)stream);
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
    const int cpb = sk_cols_per_block(B, M);
    const int nblk = (int)((B + cpb - 1) / cpb);
    double* partd = (double*)part;
    (void)rows;
    (void)part_bytes;
    for (int t = 0; t < iters; ++t) {
        const int first = (t == 0);
        if ((rc = sk_launch_pass(h, d, f, g, colsum, partd, B, M, eps, first, s)) != RC_OK) return rc;
        if ((rc = sk_launch_update(h, partd, nblk, RC_K, (int64_t)nblk * RC_K, f, g, colsum, B, M, first, flags,
                                   s)) != RC_OK) return rc;
    }
    return rc_sk_argmax(h, d, f, B, M, K, eps, codes_u8, codes_i64, stream);
}

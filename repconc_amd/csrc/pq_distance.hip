// Sub-vector vs centroid squared distances with the reference's exact fp32 rounding, the
// per-sub-quantiser min/max + centring, and the fused nearest-code assignment.
//
// Reference: models/repconc/modeling_repconc.py:49-52 (distance table, argmin) and :73-85
// (center_distance_for_constraint).  The oracle is torch-CPU, whose `.sum(-1)` over a
// contiguous fp32 row adds in a fixed 8-lane / 4-accumulator order;
// sqdist_exact() reproduces that order.  This file is compiled with -ffp-contract=off: every
// sub, mul and add below is individually rounded, exactly like the reference's three tensor ops.
#include "rc_common.h"

// d = sum_j (x_j - c_j)^2 in torch-CPU order.  XA / CA are anything indexable ([]), so callers
// can keep one operand in VGPRs and let the other come from scalar (wave-uniform) loads.
// The element-wise part (sub, square, the lane-wise accumulator adds) is written on pairs of floats: gfx950 executes a
// <2 x float> add / mul as ONE packed instruction (v_pk_add_f32 / v_pk_mul_f32, IEEE per element), which takes the
// 47 VALU instructions of a 16-wide distance down to 27; only the final 8-lane chain stays scalar.
typedef float rc_f32x2 __attribute__((ext_vector_type(2)));
template <int DSUB, typename XA, typename CA>
__device__ __forceinline__ float sqdist_exact(const XA& x, const CA& c) {
    static_assert(DSUB % 2 == 0, "pairs");
    constexpr int NV = DSUB / 8;    // 8-wide vectors in the row
    constexpr int TAIL = DSUB % 8;  // trailing scalars
    constexpr int FULL = NV / 4;    // rounds that feed all four accumulators
    rc_f32x2 sq[DSUB / 2];          // sq[4 v + p] = lanes (2 p, 2 p + 1) of vector v
#pragma unroll
    for (int j = 0; j < DSUB / 2; ++j) {
        // x - c as x + (-c): the negation is exact and (with c loop-invariant in the callers) hoisted, the packed add has no
        // packed subtract twin
        const rc_f32x2 xv = {x[2 * j], x[2 * j + 1]}, ncv = {-c[2 * j], -c[2 * j + 1]};
        const rc_f32x2 t = xv + ncv;
        sq[j] = t * t;
    }
    rc_f32x2 a[4];
    if constexpr (FULL == 0) {
        // fewer than four vectors: they all land in accumulator 0, in order; the other
        // three accumulators stay +0 and adding them is exact.
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            rc_f32x2 s = sq[p];
#pragma unroll
            for (int v = 1; v < NV; ++v) s = s + sq[4 * v + p];
            a[p] = s;
        }
    } else {
        rc_f32x2 acc[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                rc_f32x2 s = sq[4 * q + p];
#pragma unroll
                for (int i = 1; i < FULL; ++i) s = s + sq[4 * (4 * i + q) + p];
                acc[q][p] = s;
            }
#pragma unroll
        for (int v = 4 * FULL; v < NV; ++v)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[0][p] = acc[0][p] + sq[4 * v + p];
#pragma unroll
        for (int p = 0; p < 4; ++p) a[p] = ((acc[0][p] + acc[1][p]) + acc[2][p]) + acc[3][p];
    }
    float r;
    if constexpr (TAIL == 0) {
        r = a[0].x;  // (0 + lane0) is exact
        r = r + a[0].y;
#pragma unroll
        for (int p = 1; p < 4; ++p) { r = r + a[p].x; r = r + a[p].y; }
    } else {
        r = sq[NV * 4].x;
        r = r + sq[NV * 4].y;
#pragma unroll
        for (int j = 1; j < TAIL / 2; ++j) { r = r + sq[NV * 4 + j].x; r = r + sq[NV * 4 + j].y; }
#pragma unroll
        for (int p = 0; p < 4; ++p) { r = r + a[p].x; r = r + a[p].y; }
    }
    return r;
}

// ------------------------------------------------------------------------------------------
// Distance table d[M,B,K].  Block = one sub-quantiser m x a strip of rows; thread k keeps
// centroid C[m,k,:] in VGPRs; the row slice x[b, m*dsub..] is wave-uniform, so it arrives
// through scalar loads.  Each wave stores 64 consecutive floats of d per row (coalesced), the
// block 1 KiB.  Per-block (max,min) go to `mm_part` [M][gridDim.x][2].
#define DIST_MAX_ROWS 128
template <int DSUB>
__global__ __launch_bounds__(RC_K) void dist_table_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ C, int64_t B,
                                                          int rows_per_block, float* __restrict__ d,
                                                          float* __restrict__ mm_part) {
    const int m = blockIdx.y;
    const int k = threadIdx.x;
    const int M = gridDim.y;
    float c[DSUB];
    {
        const float4* cp = reinterpret_cast<const float4*>(C + ((size_t)m * RC_K + k) * DSUB);
#pragma unroll
        for (int j = 0; j < DSUB / 4; ++j) {
            const float4 v = cp[j];
            c[4 * j] = v.x; c[4 * j + 1] = v.y; c[4 * j + 2] = v.z; c[4 * j + 3] = v.w;
        }
    }
    const int64_t b0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t b1 = (b0 + rows_per_block < B) ? b0 + rows_per_block : B;
    float mx = -INFINITY, mn = INFINITY;
    float* drow = d + ((size_t)m * B + b0) * RC_K + k;
    // The block's strip of row slices (rows x dsub floats, <= 48 KiB) is staged in LDS once; every thread then reads
    // row b's slice with broadcast ds_read_b128 (all lanes the same address: conflict-free).  The earlier version
    // fetched each slice with a scalar load and waited for it in every iteration (VALU 37 % busy).
    __shared__ __attribute__((aligned(16))) float xs[DIST_MAX_ROWS * DSUB];
    const int rows = (int)(b1 - b0);
    for (int i = k; i < rows * (DSUB / 4); i += RC_K) {
        const int r = i / (DSUB / 4), j4 = i - r * (DSUB / 4);
        reinterpret_cast<float4*>(xs)[i] = *reinterpret_cast<const float4*>(x + (b0 + r) * ldx + m * DSUB + 4 * j4);
    }
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
        float xc[DSUB];
#pragma unroll
        for (int j4 = 0; j4 < DSUB / 4; ++j4) {
            const float4 v = reinterpret_cast<const float4*>(xs)[r * (DSUB / 4) + j4];
            xc[4 * j4] = v.x; xc[4 * j4 + 1] = v.y; xc[4 * j4 + 2] = v.z; xc[4 * j4 + 3] = v.w;
        }
        const float s = sqdist_exact<DSUB>(xc, c);
        __builtin_nontemporal_store(s, drow);              // 2.4 GB written once, next read by another kernel
        drow += RC_K;
        mx = fmaxf(mx, s);
        mn = fminf(mn, s);
    }
    if (mm_part) {
        __shared__ float smx[RC_K / 64], smn[RC_K / 64];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, o));
            mn = fminf(mn, __shfl_xor(mn, o));
        }
        if ((k & 63) == 0) { smx[k >> 6] = mx; smn[k >> 6] = mn; }
        __syncthreads();
        if (k == 0) {
#pragma unroll
            for (int w = 1; w < RC_K / 64; ++w) { mx = fmaxf(mx, smx[w]); mn = fminf(mn, smn[w]); }
            float* o = mm_part + ((size_t)m * gridDim.x + blockIdx.x) * 2;
            o[0] = mx;
            o[1] = mn;
        }
    }
    (void)M;
}

// minmax[m] = max over blocks, minmax[M+m] = min over blocks (max/min are order independent).
__global__ __launch_bounds__(256) void minmax_final_kernel(const float* __restrict__ mm_part, int nblk,
                                                           int M, float* __restrict__ minmax) {
    const int m = blockIdx.x;
    float mx = -INFINITY, mn = INFINITY;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        mx = fmaxf(mx, mm_part[((size_t)m * nblk + i) * 2]);
        mn = fminf(mn, mm_part[((size_t)m * nblk + i) * 2 + 1]);
    }
    __shared__ float smx[4], smn[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        mn = fminf(mn, __shfl_xor(mn, o));
    }
    if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = mx; smn[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mx = fmaxf(mx, smx[w]); mn = fminf(mn, smn[w]); }
        minmax[m] = mx;
        minmax[M + m] = mn;
    }
}

// (d - mid)/amp in place; mid=(mx+mn)/2, amp=(mx-mid)+1e-5f, IEEE fp32 division
// (modeling_repconc.py:81-84).  grid.y = m, grid-stride over the B*K entries of that m.
__global__ __launch_bounds__(256) void centre_kernel(float* __restrict__ d, const float* __restrict__ minmax,
                                                     int64_t per_m, int M) {
    const int m = blockIdx.y;
    const float mx = minmax[m], mn = minmax[M + m];
    const float mid = (mx + mn) / 2.0f;
    const float amp = (mx - mid) + 1e-5f;
    float4* p = reinterpret_cast<float4*>(d + (size_t)m * per_m);
    const int64_t n4 = per_m / 4;  // per_m = B*256 is a multiple of 4
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        typedef float c_f4 __attribute__((ext_vector_type(4)));
        c_f4* pv = reinterpret_cast<c_f4*>(p + i);
        c_f4 v = __builtin_nontemporal_load(pv);           // streamed: read once, written once
        v.x = (v.x - mid) / amp;
        v.y = (v.y - mid) / amp;
        v.z = (v.z - mid) / amp;
        v.w = (v.w - mid) / amp;
        __builtin_nontemporal_store(v, pv);
    }
}

// ------------------------------------------------------------------------------------------
// Nearest code (index build): thread = one row b, loop over sub-quantisers and centroids.
// The row slice lives in VGPRs; C[m,k,:] is wave-uniform (scalar loads); running (min, argmin)
// needs no cross-lane traffic.  First minimum wins (strict <), as torch.argmin does.
// grid = (row blocks, m-chunks): blockIdx.y owns the sub-quantisers [y*mc, (y+1)*mc).  For a large
// corpus there is one chunk (mc = M) and the codes go out as one contiguous, coalesced [rows, M] byte slab
// staged in LDS; small batches split M so that the grid still fills 256 CUs.
template <int DSUB>
__global__ __launch_bounds__(256) void assign_nearest_kernel(const float* __restrict__ x, int64_t ldx,
                                                             const float* __restrict__ C, int64_t B,
                                                             int M, int mc, uint8_t* __restrict__ codes_u8,
                                                             int64_t* __restrict__ codes_i64) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile[];  // [256][mc]
    const int tid = threadIdx.x;
    const int64_t b = (int64_t)blockIdx.x * 256 + tid;
    const bool live = b < B;
    const float* xrow = x + (live ? b : (B - 1)) * ldx;
    const int m0 = blockIdx.y * mc;
    for (int mi = 0; mi < mc; ++mi) {
        const int m = m0 + mi;
        float xs[DSUB];
        const float4* xp = reinterpret_cast<const float4*>(xrow + m * DSUB);
#pragma unroll
        for (int j = 0; j < DSUB / 4; ++j) {
            const float4 v = xp[j];
            xs[4 * j] = v.x; xs[4 * j + 1] = v.y; xs[4 * j + 2] = v.z; xs[4 * j + 3] = v.w;
        }
        const float* cm = C + (size_t)m * RC_K * DSUB;
        float best = INFINITY;
        int bi = 0;
#pragma unroll 2
        for (int k = 0; k < RC_K; ++k) {
            const float s = sqdist_exact<DSUB>(xs, cm + k * DSUB);
            if (s < best) { best = s; bi = k; }
        }
        tile[tid * mc + mi] = (unsigned char)bi;
    }
    __syncthreads();
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    const int64_t rows = (B - row0 < 256) ? (B - row0) : 256;
    if (mc == M) {
        const int64_t nbytes = rows * M;
        if (codes_u8) {
            unsigned char* dst = codes_u8 + row0 * M;  // row0*M is a multiple of 16 (256*M)
            const int64_t n16 = nbytes / 16;
            for (int64_t i = tid; i < n16; i += 256)
                reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(tile)[i];
            for (int64_t i = n16 * 16 + tid; i < nbytes; i += 256) dst[i] = tile[i];
        }
        if (codes_i64) {
            int64_t* dst = codes_i64 + row0 * M;
            for (int64_t i = tid; i < nbytes; i += 256) dst[i] = (int64_t)tile[i];
        }
    } else if (live) {
        for (int mi = 0; mi < mc; ++mi) {
            const unsigned char c = tile[tid * mc + mi];
            if (codes_u8) codes_u8[b * M + m0 + mi] = c;
            if (codes_i64) codes_i64[b * M + m0 + mi] = (int64_t)c;
        }
    }
}

// ------------------------------------------------------------------------------------------ any width
// The reference takes every divisor of hidden_size as MCQ_M (modeling_repconc.py:41).  The kernels above are specialised
// (registers, packed arithmetic) for the widths the recipes use; these two cover every other width with the SAME fp32
// arithmetic at run-time width: torch-CPU's `sum(-1)` order of aten/native/cpu/SumKernel.cpp —
//   dsub >= 8: 8-wide vectors, four ILP accumulators fed round by round, and torch's CASCADE: after every 16 rounds
//              (512 floats) the running sums move up one level and restart from 0, levels merged lowest first at the end
//              (dsub = 768: rounds 16..23 + rounds 0..15, not a plain running sum); left-over vectors to accumulator 0;
//              0 += 1, 2, 3; (tail scalars from 0) + lane 0 .. lane 7;
//   dsub <  8: the scalar twin: p_j = x_j (j < 4) when dsub >= 4, the remaining scalars to p_0 in order, p_0 += p_1, p_2, p_3
// — restated in oracle/pq_oracle.{py,c} and checked there against torch for all 18 divisors of 768; fixtures M = 1, 6, 128.
// Two cascade levels cover 255 rounds: dsub < 8192.
#define RC_DSUB_RT_MAX 8191
__device__ __forceinline__ float sqdist_exact_rt(const float* __restrict__ x, const float* __restrict__ c, int dsub) {
    if (dsub < 8) {
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        const int full = dsub >> 2;
        if (full) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float t = x[j] - c[j]; p[j] = p[j] + t * t; }
        }
        for (int j = 4 * full; j < dsub; ++j) { const float t = x[j] - c[j]; p[0] = p[0] + t * t; }
        float r = p[0] + p[1];
        r = r + p[2];
        return r + p[3];
    }
    const int nv = dsub >> 3, tail = dsub & 7, full = nv >> 2;
    float a0[4][8], a1[4][8];                             // cascade levels 0 and 1: [ilp][lane]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int l = 0; l < 8; ++l) { a0[j][l] = 0.f; a1[j][l] = 0.f; }
    int i = 0;
    while (i + 16 <= full) {
        for (int t16 = 0; t16 < 16; ++t16, ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int l = 0; l < 8; ++l) {
                    const int e = 32 * i + 8 * j + l;
                    const float t = x[e] - c[e];
                    a0[j][l] = a0[j][l] + t * t;
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int l = 0; l < 8; ++l) { a1[j][l] = a1[j][l] + a0[j][l]; a0[j][l] = 0.f; }
    }
    for (; i < full; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const int e = 32 * i + 8 * j + l;
                const float t = x[e] - c[e];
                a0[j][l] = a0[j][l] + t * t;
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int l = 0; l < 8; ++l) a0[j][l] = a0[j][l] + a1[j][l];       // (levels 2, 3 are +0)
    for (int v = 4 * full; v < nv; ++v) {
#pragma unroll
        for (int l = 0; l < 8; ++l) { const float t = x[8 * v + l] - c[8 * v + l]; a0[0][l] = a0[0][l] + t * t; }
    }
    float r = 0.f;
    for (int j = 0; j < tail; ++j) { const float t = x[8 * nv + j] - c[8 * nv + j]; r = r + t * t; }
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        float t = a0[0][l] + a0[1][l];
        t = t + a0[2][l];
        t = t + a0[3][l];
        r = r + t;
    }
    return r;
}

// grid (row strips, M) as dist_table_kernel (same mm_part layout); the strip's slices are staged in LDS RT_ROWS rows at a time
#define DIST_RT_LDS_FLOATS 8192
__global__ __launch_bounds__(RC_K) void dist_table_rt_kernel(const float* __restrict__ x, int64_t ldx,
                                                             const float* __restrict__ C, int64_t B, int dsub,
                                                             int rows_per_block, float* __restrict__ d,
                                                             float* __restrict__ mm_part) {
    __shared__ float xs[DIST_RT_LDS_FLOATS];
    const int m = blockIdx.y, k = threadIdx.x;
    const float* c = C + ((size_t)m * RC_K + k) * dsub;
    const int64_t b0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t b1 = (b0 + rows_per_block < B) ? b0 + rows_per_block : B;
    const int rt = DIST_RT_LDS_FLOATS / dsub;              // >= 1 for dsub <= 8192
    float mx = -INFINITY, mn = INFINITY;
    for (int64_t r0 = b0; r0 < b1; r0 += rt) {
        const int rows = (int)((b1 - r0 < rt) ? b1 - r0 : rt);
        __syncthreads();
        for (int i = k; i < rows * dsub; i += RC_K) {
            const int r = i / dsub, j = i - r * dsub;
            xs[i] = x[(r0 + r) * ldx + (int64_t)m * dsub + j];
        }
        __syncthreads();
        for (int r = 0; r < rows; ++r) {
            const float s = sqdist_exact_rt(xs + r * dsub, c, dsub);
            d[((size_t)m * B + r0 + r) * RC_K + k] = s;
            mx = fmaxf(mx, s);
            mn = fminf(mn, s);
        }
    }
    if (mm_part) {
        __shared__ float smx[RC_K / 64], smn[RC_K / 64];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, o));
            mn = fminf(mn, __shfl_xor(mn, o));
        }
        if ((k & 63) == 0) { smx[k >> 6] = mx; smn[k >> 6] = mn; }
        __syncthreads();
        if (k == 0) {
#pragma unroll
            for (int w = 1; w < RC_K / 64; ++w) { mx = fmaxf(mx, smx[w]); mn = fminf(mn, smn[w]); }
            float* o = mm_part + ((size_t)m * gridDim.x + blockIdx.x) * 2;
            o[0] = mx;
            o[1] = mn;
        }
    }
}

// Nearest code at any width: grid (row strips, M), thread = centroid k; per row the block reduces (distance, k) to the FIRST
// minimum exactly as the sequential scan `if (s < best)` from best = +inf does (NaN and +inf never win: code 0).
#define ASSIGN_RT_ROWS 8
__global__ __launch_bounds__(RC_K) void assign_nearest_rt_kernel(const float* __restrict__ x, int64_t ldx,
                                                                 const float* __restrict__ C, int64_t B, int M, int dsub,
                                                                 uint8_t* __restrict__ codes_u8, int64_t* __restrict__ codes_i64) {
    __shared__ float xs[DIST_RT_LDS_FLOATS];
    __shared__ float s_v[RC_K / 64];
    __shared__ int s_k[RC_K / 64];
    const int m = blockIdx.y, k = threadIdx.x;
    const float* c = C + ((size_t)m * RC_K + k) * dsub;
    const int64_t b0 = (int64_t)blockIdx.x * ASSIGN_RT_ROWS;
    const int64_t b1 = (b0 + ASSIGN_RT_ROWS < B) ? b0 + ASSIGN_RT_ROWS : B;
    const int rt = DIST_RT_LDS_FLOATS / dsub;
    for (int64_t r0 = b0; r0 < b1; r0 += rt) {
        const int rows = (int)((b1 - r0 < rt) ? b1 - r0 : rt);
        __syncthreads();
        for (int i = k; i < rows * dsub; i += RC_K) {
            const int r = i / dsub, j = i - r * dsub;
            xs[i] = x[(r0 + r) * ldx + (int64_t)m * dsub + j];
        }
        __syncthreads();
        for (int r = 0; r < rows; ++r) {
            const float s = sqdist_exact_rt(xs + r * dsub, c, dsub);
            float bv = (s < INFINITY) ? s : INFINITY;      // NaN / +inf: never chosen
            int bk = (s < INFINITY) ? k : RC_K;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o);
                const int ok = __shfl_xor(bk, o);
                if (ov < bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
            }
            if ((k & 63) == 0) { s_v[k >> 6] = bv; s_k[k >> 6] = bk; }
            __syncthreads();
            if (k == 0) {
#pragma unroll
                for (int w = 1; w < RC_K / 64; ++w)
                    if (s_v[w] < bv || (s_v[w] == bv && s_k[w] < bk)) { bv = s_v[w]; bk = s_k[w]; }
                const int code = bk < RC_K ? bk : 0;
                if (codes_u8) codes_u8[(r0 + r) * M + m] = (uint8_t)code;
                if (codes_i64) codes_i64[(r0 + r) * M + m] = (int64_t)code;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------ host
static int dist_rows_per_block(int64_t B) { return B >= 16384 ? DIST_MAX_ROWS : 32; }

extern "C" size_t rc_pq_dist_table_ws_bytes(int64_t B, int M) {
    if (B <= 0 || M <= 0) return 0;
    const int rpb = dist_rows_per_block(B);
    const int64_t nblk = (B + rpb - 1) / rpb;
    return rc_align_up((size_t)M * nblk * 2 * sizeof(float), 256);
}

extern "C" int rc_pq_dist_table(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B,
                                int D, int M, int K, float* d, float* minmax, void* ws, size_t ws_bytes,
                                rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !x || !C || !d || B < 0 || M <= 0 || D <= 0 || ldx < D) return RC_EINVAL;
    if (K != RC_K || D % M != 0 || D / M > RC_DSUB_RT_MAX) return RC_ESHAPE;
    const bool special = rc_dsub_supported(D / M);
    if (special && (((uintptr_t)x & 15) || (ldx % 4) != 0)) return RC_EINVAL;  // float4 row loads
    if (B == 0) return RC_OK;
    if (minmax && (!ws || ws_bytes < rc_pq_dist_table_ws_bytes(B, M))) return RC_EWORKSPACE;
    const int rpb = dist_rows_per_block(B);
    const int64_t nblk = (B + rpb - 1) / rpb;
    hipStream_t s = (hipStream_t)stream;
    float* part = minmax ? (float*)ws : nullptr;
    dim3 grid((unsigned)nblk, (unsigned)M);
    rc_prof_mark(h, RC_PROF_DIST_TABLE, s);
    if (special) {
        RC_DISPATCH_DSUB(D / M, hipLaunchKernelGGL(dist_table_kernel<DSUB>, grid, dim3(RC_K), 0, s, x, ldx, C, B, rpb, d, part));
    } else {                                              // any other width: same arithmetic at run-time width
        hipLaunchKernelGGL(dist_table_rt_kernel, grid, dim3(RC_K), 0, s, x, ldx, C, B, D / M, rpb, d, part);
    }
    rc_prof_mark(h, RC_PROF_DIST_TABLE, s);
    RC_LAUNCH_CHECK(h);
    if (minmax) {
        hipLaunchKernelGGL(minmax_final_kernel, dim3(M), dim3(256), 0, s, part, (int)nblk, M, minmax);
        RC_LAUNCH_CHECK(h);
    }
    return RC_OK;
}

extern "C" int rc_pq_centre(rc_handle_t h, float* d, const float* minmax, int64_t B, int M, int K,
                            rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !d || !minmax || B < 0 || M <= 0) return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    if (B == 0) return RC_OK;
    const int64_t per_m = B * RC_K;
    int64_t gx = (per_m / 4 + 255) / 256;
    const int64_t cap = (int64_t)h->num_cus * 8;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(centre_kernel, dim3((unsigned)gx, (unsigned)M), dim3(256), 0, (hipStream_t)stream, d,
                       minmax, per_m, M);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

extern "C" int rc_pq_assign_nearest(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B,
                                    int D, int M, int K, uint8_t* codes_u8, int64_t* codes_i64,
                                    rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !x || !C || B < 0 || M <= 0 || D <= 0 || ldx < D || (!codes_u8 && !codes_i64)) return RC_EINVAL;
    if (K != RC_K || D % M != 0 || D / M > RC_DSUB_RT_MAX) return RC_ESHAPE;
    if (!rc_dsub_supported(D / M)) {                       // any other width (the reference takes every divisor)
        if (B == 0) return RC_OK;
        hipLaunchKernelGGL(assign_nearest_rt_kernel, dim3((unsigned)((B + ASSIGN_RT_ROWS - 1) / ASSIGN_RT_ROWS), (unsigned)M),
                           dim3(RC_K), 0, (hipStream_t)stream, x, ldx, C, B, M, D / M, codes_u8, codes_i64);
        RC_LAUNCH_CHECK(h);
        return RC_OK;
    }
    if (((uintptr_t)x & 15) || (ldx % 4) != 0) return RC_EINVAL;  // float4 row loads
    if (B == 0) return RC_OK;
    const int64_t nblk = (B + 255) / 256;
    // m-chunks: the smallest split of M (a divisor) that gives the grid >= ~8 blocks per CU
    int chunks = 1;
    for (int c = 1; c <= M; ++c)
        if (M % c == 0) { chunks = c; if (nblk * c >= (int64_t)h->num_cus * 8) break; }
    const int mc = M / chunks;
    const size_t lds = (size_t)256 * mc;
    hipStream_t s = (hipStream_t)stream;
    rc_prof_mark(h, RC_PROF_ASSIGN_NEAREST, s);
    RC_DISPATCH_DSUB(D / M, hipLaunchKernelGGL(assign_nearest_kernel<DSUB>, dim3((unsigned)nblk, (unsigned)chunks), dim3(256),
                                               lds, s, x, ldx, C, B, M, mc, codes_u8, codes_i64));
    rc_prof_mark(h, RC_PROF_ASSIGN_NEAREST, s);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

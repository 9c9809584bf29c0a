// Sinkhorn-Knopp uniform-assignment solve on the centred distance table, with potentials.
//
// Reference: sinkhorn_algorithm, models/repconc/modeling_repconc.py:137-165, called from
// RepCONC.quantize (:53-66) on out = -centred.double().transpose(1,2), i.e. L[m,k,b] = -d[m,b,k]/eps in
// fp64.  The reference keeps the whole fp64 matrix Q and rescales it in place; here only the fp32 centred
// distances d[M,B,K] live in HBM (4 B/entry, read once per sweep) and the scalings are carried as
// potentials f[M,K], g[M,B]:
//
//   sweep 0        rows_k = sum_b exp(L_kb)
//   sweep t>=1     f_k  = f_k - log(sum over ranks of rows_k)        (f = 0 before sweep 1)
//                  g_b  = g_b - log c_b                              (from sweep t-1; g = 0 in sweep 1)
//                  w_kb = exp(L_kb + f_k + g_b)   (<= 1 by construction, no max pass)
//                  c_b  = sum_k w_kb ,  rows_k = sum_b w_kb / c_b
//   argmax         f_k  = f_k - log(sum rows_k) ;  code_b = argmax_k (L_kb + f_k)   (first maximum)
//
// T reference iterations (:153-163) = sweeps 0..T-1 + the argmax pass: T+1 reads of d, ONE exp per entry
// per sweep.  `w <= 1`: after the previous column normalisation sum_k exp(L+f'+g) = 1, and the row update
// divides entry (k,b) by rows_k >= that entry.  The /K, /B, global sum and final *B of :148-164 cancel.
//
// Everything an iteration needs is fused into ONE launch: the prologue of sweep t applies the row and
// column updates that follow sweep t-1 (each block recomputes the 256 potentials of its sub-quantiser from
// the all-gathered row sums and updates g for its own columns), and the last block of every
// sub-quantiser to finish reduces the block partials (agent-scope hand-off: write-through stores, drained,
// one relaxed counter increment; the reducer reads with L1-bypassing loads).  Between two sweeps only the
// cross-rank all-gather of rows[M,K] remains (multi-GPU), nothing on a single GPU.
//
// Work split: a column (one document, 256 centroids, 1 KiB of fp32) is owned by 16 lanes, 16 k per lane in
// four groups of 4 (sk_kidx): every 16-byte load is part of a 256-byte run read by the 16 lanes together.
// The column sum is a 4-step DPP rotate-add; row sums accumulate in registers, are reduced over the 16
// column groups of the block through LDS in a fixed order, over blocks in block order, over ranks in rank
// order — no floating-point atomics anywhere, so results are identical run to run and rank to rank.
#include "rc_common.h"

#include <stdlib.h>

#define SK_EPL 16                 // matrix entries (k) per lane
#define SK_GROUP (RC_K / SK_EPL)  // 16 lanes own one column
#define SK_THREADS 256
#define SK_NG (SK_THREADS / SK_GROUP)  // 16 columns in flight per block
#define SK_TB 11                       // exp table: 2^11 entries (16 KiB of LDS)
#define SK_N (1 << SK_TB)
#ifndef SK_RED_INFLIGHT
#define SK_RED_INFLIGHT 24               // partial loads in flight in the last-block reducer
#endif
// memory order of the fused exchange's flag store (system scope).  RELEASE: the flag publishes the row sums stored before the
// block barrier by the HIP memory model; RELAXED was the round-5 form, correct on gfx942 / gfx950 only because its
// data stores are write-through system-scope atomics drained with s_waitcnt vmcnt(0).  -DSK_XCHG_FLAG_ORDER=__ATOMIC_RELAXED for A/B.
#ifndef SK_XCHG_FLAG_ORDER
#define SK_XCHG_FLAG_ORDER __ATOMIC_RELEASE
#endif
#define SK_MAX_CPB 512                 // columns per block (LDS holds their integer column exponents)
#define SK_LN2 0.69314718055994530942

// ---- exp for the sweeps ---------------------------------------------------------------------------
// The exponent is produced directly in units of 1/N octave (N = 2^TB): the factor N/ln2 is folded into
// -1/eps and into the potentials, so with u = (L + f)*N/ln2 and the column potential split into an
// integer part gq and a residual (below),
//     exp(L + f + g) ~ 2^((u + gq)/N) = 2^e * 2^(j/N) * 2^(r/N),   n = rint(u), r = u - n in [-.5,.5],
//     j = (n + gq) & (N-1),  e = (n + gq) >> TB.
// 2^(j/N) comes from an N-entry table staged in LDS (rc_handle owns the device copy, built on the host
// with exp2l), 2^(r/N) - 1 from a Taylor polynomial in z = r ln2/N, |z| <= ln2/2N = 1.7e-4: degree 3
// (remainder z^4/24 < 4e-17).  Relative error ~2e-16 plus the rounding of u itself (|u| eps_64 ->
// <1e-13 in the exponent, the same as rounding L+f+g directly).  11.4 fp64 instructions per entry
// against 24.5 for a straight degree-12 polynomial exp.
//
// Column potential: g_b enters only through 2^(g_b N/ln2 / N).  Its integer part gq = rint(g_b N/ln2) is
// added to n as an INTEGER; the residual factor 2^(rg/N), rg = g_b N/ln2 - gq, is common to the whole
// column, cancels in w/colsum, and is restored when g is updated:
//     log(colsum_true) = log(colsum_stored) + rg ln2/N.
// gq8 = 8 * gq: the column exponent arrives pre-multiplied by the table's entry size, so that ONE v_lshl_add gives
// 8 (n + gq), one v_and its byte offset into the table (which sits at LDS address 0 of the dynamic segment) and one
// shift the power of two — 3 integer instructions per entry instead of 4 (add, and, lshl_add, ashr).
__device__ __forceinline__ double sk_exp2n(double u, int gq8, const double* __restrict__ tab) {
    constexpr double Z = SK_LN2 / (double)SK_N;
    const double n = __builtin_rint(u);
    const double r = u - n;
    const int ni8 = ((int)n << 3) + gq8;
    // a generic pointer into LDS is {shared aperture, byte offset}: its low 32 bits are the LDS address.  The table is the
    // first thing in the kernel's (dynamic-only) LDS segment, so that base is 0 with today's code generation and the add
    // folds away; if a compiler ever places the segment elsewhere the add stays and the result is still right.
    const unsigned tbase = static_cast<unsigned>(reinterpret_cast<uintptr_t>(tab));
    typedef const double __attribute__((address_space(3))) sk_lds_cd;
    const double T = *reinterpret_cast<sk_lds_cd*>(tbase + static_cast<unsigned>(ni8 & ((SK_N - 1) << 3)));
    double q = __builtin_fma(Z * Z * Z / 6.0, r, Z * Z / 2.0);
    q = __builtin_fma(q, r, Z);
    q = q * r;
    return __builtin_ldexp(__builtin_fma(T, q, T), ni8 >> (SK_TB + 3));
}

__device__ __forceinline__ int sk_kidx(int lane, int i) { return ((i >> 2) << 6) + (lane << 2) + (i & 3); }

// p = column base + 4*lane
__device__ __forceinline__ void sk_load_col(const float* __restrict__ p, float (&v)[SK_EPL]) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int j = 0; j < SK_EPL / 4; ++j) {
        // streamed once per sweep (2.4 GB >> every cache): non-temporal loads (global_load ... nt), -2.3 % sweep time
        typedef float sk_f4 __attribute__((ext_vector_type(4)));
        const sk_f4 a = __builtin_nontemporal_load(reinterpret_cast<const sk_f4*>(q + 16 * j));
        v[4 * j] = a.x; v[4 * j + 1] = a.y; v[4 * j + 2] = a.z; v[4 * j + 3] = a.w;
    }
}

// centre 16 entries of a column in place and write them back (same addresses as sk_load_col)
__device__ __forceinline__ void sk_centre_store(float* __restrict__ p, float (&v)[SK_EPL], float mid, float amp) {
    typedef float sk_f4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < SK_EPL / 4; ++j) {
        sk_f4 a;
        a.x = v[4 * j] = (v[4 * j] - mid) / amp;
        a.y = v[4 * j + 1] = (v[4 * j + 1] - mid) / amp;
        a.z = v[4 * j + 2] = (v[4 * j + 2] - mid) / amp;
        a.w = v[4 * j + 3] = (v[4 * j + 3] - mid) / amp;
        __builtin_nontemporal_store(a, reinterpret_cast<sk_f4*>(p) + 16 * j);
    }
}

// One column step (t >= 1): exponentials, column sum, normalised row-sum update.
// Row potentials either in registers (fk, 32 VGPRs: 152 in total, three waves per SIMD) or re-read from LDS at use
// (FKLDS: 120 VGPRs, four waves per SIMD; with the reduction scratch aliased onto the exp table the block needs 36 KiB
// of LDS, so four blocks fit a CU).  Measured at 49 152 x 48: 0.470 vs 0.480 ms per sweep; at 6 144 rows per rank the
// register variant is the (slightly) faster one, so the host picks by grid size.
template <bool FKLDS>
__device__ __forceinline__ void sk_column(const float (&x)[SK_EPL], const double (&fk)[SK_EPL],
                                          const double* __restrict__ fk_lds, int lane, double (&R)[SK_EPL], int gq8,
                                          double nscale_eps, const double* __restrict__ tab,
                                          double* __restrict__ csum_out, bool writer) {
    double w[SK_EPL];
    double c = 0.0;
    int ln = lane;
    if constexpr (FKLDS) asm volatile("" : "+v"(ln));     // re-read from LDS every column (keeps the loads un-hoisted)
#pragma unroll
    for (int i = 0; i < SK_EPL; ++i) {
        const double fki = FKLDS ? fk_lds[sk_kidx(ln, i)] : fk[i];
        w[i] = sk_exp2n(__builtin_fma((double)x[i], nscale_eps, fki), gq8, tab);
        c += w[i];
    }
    c = rc_row16_allreduce_sum(c);
    const double rc = 1.0 / c;
#pragma unroll
    for (int i = 0; i < SK_EPL; ++i) R[i] = __builtin_fma(w[i], rc, R[i]);
    if (writer) *csum_out = c;
}

// f for sub-quantiser m, centroid k after `t` row normalisations: f_prev - log(sum_r rows_prev[r][m][k]),
// ranks ascending; f_prev = 0 for t == 1.  Shared by the sweep and the argmax prologues.
__device__ __forceinline__ double sk_row_potential(const double* __restrict__ rows_prev, int G, int M, int m,
                                                   int k, const double* __restrict__ f_in, int t, bool& bad) {
    double s = 0.0;
    for (int r = 0; r < G; ++r) s += rows_prev[((size_t)r * M + m) * RC_K + k];
    bad |= !(s > 0.0) || !(s < INFINITY);
    const double fo = (t == 1) ? 0.0 : f_in[(size_t)m * RC_K + k];
    return fo - log(s);
}

// One sweep.  grid = (blocks per m, M).  Dynamic LDS: tab[N] | red[16][256] | fk[256] | gq[cpb] | flag.
// CENTRE (first sweep only): d holds the RAW distance table; every entry is centred on the fly exactly like
// centre_kernel — (d - mid)/amp, IEEE division, range from cmx/cmn (modeling_repconc.py:81-84) — stored back (the
// later sweeps read the centred table) and used at once: one pass over the table instead of centre_kernel's
// read + write followed by the first sweep's read.
template <bool FIRST, bool CENTRE = false, bool FKLDS = false>
__global__ __launch_bounds__(SK_THREADS, (FKLDS ? 4 : 3)) void sk_sweep_kernel(
    const float* __restrict__ d, const double* __restrict__ rows_prev, int G, const double* __restrict__ f_in,
    double* __restrict__ f_out, double* __restrict__ g, double* __restrict__ colsum, double* __restrict__ part,
    unsigned* __restrict__ counters, double* __restrict__ rows_out, int64_t B, int cols_per_block,
    double nscale_eps, double scale, const double* __restrict__ exp2_tab, int t, int* __restrict__ flags,
    const float* __restrict__ cmx = nullptr, const float* __restrict__ cmn = nullptr) {
    static_assert(FIRST || !CENTRE, "centring is fused into the first sweep only");
    static_assert(!(FIRST && FKLDS), "the first sweep has no row potentials");
    extern __shared__ __attribute__((aligned(16))) double sk_smem[];
    double* tab = sk_smem;                                                   // [N]
    // FKLDS: the reduction scratch reuses the table's LDS (the table is dead by then): 36 KiB per block, 4 blocks per CU
    double(*red)[RC_K] = reinterpret_cast<double(*)[RC_K]>(sk_smem + (FKLDS ? 0 : SK_N));   // [16][256]
    double* fk_lds = sk_smem + (FKLDS ? 0 : SK_N) + SK_NG * RC_K;                           // [256]
    int* gq_lds = reinterpret_cast<int*>(fk_lds + RC_K);                     // [SK_MAX_CPB]
    int* last_flag = gq_lds + SK_MAX_CPB;

    const int m = blockIdx.y, M = gridDim.y;
    const int tid = threadIdx.x;
    const int lane = tid & (SK_GROUP - 1);
    const int grp = tid / SK_GROUP;
    const int64_t c0 = (int64_t)blockIdx.x * cols_per_block;
    const int64_t c1 = (c0 + cols_per_block < B) ? c0 + cols_per_block : B;
    const int ncols = (int)(c1 - c0);

    for (int i = tid; i < SK_N; i += SK_THREADS) tab[i] = exp2_tab[i];
    if constexpr (!FIRST) {
        // ---- updates that follow sweep t-1 (modeling_repconc.py:157-158 and :162) ----
        bool bad = false;
        const double fn = sk_row_potential(rows_prev, G, M, m, tid, f_in, t, bad);
        if (blockIdx.x == 0) f_out[(size_t)m * RC_K + tid] = fn;
        fk_lds[tid] = fn * scale;
        double* gm = g + (size_t)m * B + c0;
        const double* cm = colsum + (size_t)m * B + c0;
        for (int j = tid; j < ncols; j += SK_THREADS) {
            double gn = 0.0;
            if (t > 1) {
                // colsum was accumulated with the integer part of g*scale only: put the residual factor
                // 2^(rg/N) back, log(colsum_true) = log(colsum) + rg/scale
                const double go = gm[j], c = cm[j];
                const double gs = go * scale;
                const double rg = gs - __builtin_rint(gs);
                gn = go - (log(c) + rg / scale);
                bad |= !(c > 0.0) || !(c < INFINITY);
            }
            gm[j] = gn;
            gq_lds[j] = (int)__builtin_rint(gn * scale) << 3;     // pre-multiplied by 8 (sk_exp2n)
        }
        if (__any(bad) && (tid & 63) == 0) atomicOr(flags, RC_FLAG_NONFINITE);
    }
    __syncthreads();

    double fk[SK_EPL], R[SK_EPL];
#pragma unroll
    for (int i = 0; i < SK_EPL; ++i) {
        fk[i] = FIRST ? 0.0 : fk_lds[sk_kidx(lane, i)];
        R[i] = 0.0;
    }
    const float* dm = d + (size_t)m * B * RC_K + lane * 4;
    double* cm = colsum + (size_t)m * B;
    float cmid = 0.f, camp = 1.f;
    if constexpr (CENTRE) {
        const float mx = cmx[m], mn = cmn[m];
        cmid = (mx + mn) / 2.0f;                                  // centre_kernel's arithmetic, pq_distance.hip
        camp = (mx - cmid) + 1e-5f;
    }

    // two columns per trip; the register buffers ping-pong so prefetched data is never copied
    float xa[SK_EPL], xb[SK_EPL];
    int64_t col = c0 + grp;
    if (col < c1) sk_load_col(dm + col * RC_K, xa);
    while (col < c1) {
        const int64_t colb = col + SK_NG;
        if (colb < c1) sk_load_col(dm + colb * RC_K, xb);
        if constexpr (FIRST) {
            if constexpr (CENTRE) sk_centre_store(const_cast<float*>(dm) + col * RC_K, xa, cmid, camp);
#pragma unroll
            for (int i = 0; i < SK_EPL; ++i) R[i] += sk_exp2n((double)xa[i] * nscale_eps, 0, tab);
        } else {
            sk_column<FKLDS>(xa, fk, fk_lds, lane, R, gq_lds[col - c0], nscale_eps, tab, cm + col, lane == 0);
        }
        if (colb >= c1) break;
        const int64_t cola = colb + SK_NG;
        if (cola < c1) sk_load_col(dm + cola * RC_K, xa);
        if constexpr (FIRST) {
            if constexpr (CENTRE) sk_centre_store(const_cast<float*>(dm) + colb * RC_K, xb, cmid, camp);
#pragma unroll
            for (int i = 0; i < SK_EPL; ++i) R[i] += sk_exp2n((double)xb[i] * nscale_eps, 0, tab);
        } else {
            sk_column<FKLDS>(xb, fk, fk_lds, lane, R, gq_lds[colb - c0], nscale_eps, tab, cm + colb, lane == 0);
        }
        col = cola;
    }

    // ---- block reduction of the row sums, fixed order over the 16 column groups ----
    if constexpr (FKLDS) __syncthreads();                                    // every wave is done with the table
#pragma unroll
    for (int i = 0; i < SK_EPL; ++i) red[grp][sk_kidx(lane, i)] = R[i];
    __syncthreads();
    double s = red[0][tid];
#pragma unroll
    for (int q = 1; q < SK_NG; ++q) s += red[q][tid];

    // ---- hand the partial to whichever block of this m finishes last (placement independent):
    // write-through (sc1) 8-byte stores, every wave drains them, one relaxed agent-scope counter add.
    const unsigned nblk = gridDim.x;
    double* pm = part + (size_t)m * nblk * RC_K;
    __hip_atomic_store(pm + (size_t)blockIdx.x * RC_K + tid, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(counters + m, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *last_flag = ((old + 1u) % nblk == 0u);
    }
    __syncthreads();
    if (*last_flag) {
        // reducer: the partials were stored write-through, read them with L1-bypassing (sc1) loads, summed in
        // block order; SK_RED_INFLIGHT loads in flight (the reducer is the tail of every sweep: pure L2 latency)
        double acc = 0.0;
        unsigned i = 0;
        for (; i + SK_RED_INFLIGHT <= nblk; i += SK_RED_INFLIGHT) {
            double v[SK_RED_INFLIGHT];
#pragma unroll
            for (int j = 0; j < SK_RED_INFLIGHT; ++j)
                v[j] = __hip_atomic_load(pm + (size_t)(i + j) * RC_K + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < SK_RED_INFLIGHT; ++j) acc += v[j];
        }
        for (; i < nblk; ++i)
            acc += __hip_atomic_load(pm + (size_t)i * RC_K + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rows_out[(size_t)m * RC_K + tid] = acc;
    }
}

// code[b][m] = argmax_k (L_kb + f_k), first maximum, with f brought up to date from the last sweep's row
// sums in the prologue.  Same column ownership as the sweeps; the (value, index) pair is reduced across
// the 16 lanes with rotations; ties keep the lower k.
__global__ __launch_bounds__(SK_THREADS) void sk_argmax_kernel(const float* __restrict__ d,
                                                               const double* __restrict__ rows_prev, int G,
                                                               const double* __restrict__ f_in, int t, int64_t B,
                                                               int cols_per_block, double ninv_eps,
                                                               int code_stride, int m_offset,
                                                               uint8_t* __restrict__ codes_u8,
                                                               int64_t* __restrict__ codes_i64,
                                                               int* __restrict__ flags) {
    __shared__ double fk_lds[RC_K];
    const int m = blockIdx.y, M = gridDim.y;
    const int tid = threadIdx.x;
    const int lane = tid & (SK_GROUP - 1);
    const int grp = tid / SK_GROUP;
    {
        bool bad = false;
        fk_lds[tid] = sk_row_potential(rows_prev, G, M, m, tid, f_in, t, bad);
        if (blockIdx.x == 0 && __any(bad) && (tid & 63) == 0) atomicOr(flags, RC_FLAG_NONFINITE);
    }
    __syncthreads();
    const int64_t c0 = (int64_t)blockIdx.x * cols_per_block;
    const int64_t c1 = (c0 + cols_per_block < B) ? c0 + cols_per_block : B;
    double fk[SK_EPL];
#pragma unroll
    for (int i = 0; i < SK_EPL; ++i) fk[i] = fk_lds[sk_kidx(lane, i)];
    const float* dm = d + (size_t)m * B * RC_K + lane * 4;
    for (int64_t col = c0 + grp; col < c1; col += SK_NG) {
        float cur[SK_EPL];
        sk_load_col(dm + col * RC_K, cur);
        double best = __builtin_fma((double)cur[0], ninv_eps, fk[0]);
        int bi = sk_kidx(lane, 0);
#pragma unroll
        for (int i = 1; i < SK_EPL; ++i) {   // k ascends with i inside a lane: strict > keeps the first maximum
            const double s = __builtin_fma((double)cur[i], ninv_eps, fk[i]);
            if (s > best) { best = s; bi = sk_kidx(lane, i); }
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
            // codes [B, code_stride]; this launch owns the sub-quantisers m_offset .. m_offset + M - 1
            if (codes_u8) codes_u8[col * code_stride + m_offset + m] = (uint8_t)bi;
            if (codes_i64) codes_i64[col * code_stride + m_offset + m] = (int64_t)bi;
        }
    }
}

// ================================================================================================ sweep, version 2
// Same mathematics, restructured around three observations (round 2):
//
//  (1) The column potential g_b never reaches the result: rows_k = sum_b w_kb / c_b with w_kb = exp(L_kb + f_k + g_b) and
//      c_b = sum_k w_kb is independent of g_b, and so is the final argmax_k (L_kb + f_k).  g only keeps w inside the fp64
//      range (w <= ~1).  So it is carried as an INTEGER number of 1/N octaves per column (gq8 = 8 gq, one int32 per
//      (m, b), read and written by the column's owner lanes inside the main loop) and updated with the crudest log2 there
//      is — the exponent and top mantissa bits of c_b as an integer, |error| < 0.09 octave: no fp64 g / colsum arrays, no
//      per-column fp64 log, no staging of a chunk's exponents in LDS, no limit on the columns of a block.
//  (2) exp costs 8 fp64-rate instructions instead of 10: the argument is kept non-negative by a power-of-two offset that
//      the prologue derives from the smallest row potential (u' = (L + f) N/ln2 + OFF >= 0 for every |L| <= 1/eps), so
//      v_fract_f64 / v_cvt_i32_f64 split it directly (no rndne + sub); with N = 4096 table entries a degree-2 minimax
//      polynomial on [0,1) reproduces 2^(r/N) to 2.5e-14 (the rounding of u' itself is 8e-14, as in version 1).  The column
//      normalisation multiplies by a Newton-refined v_rcp_f64 instead of dividing.
//  (3) Work partition: grid (nbm, M) with nbm = floor(resident blocks / M) equal column ranges per sub-quantiser (M = 48:
//      21 x 48 = 1008 blocks on the chip's 1024 slots at 4 per CU): every block does the same work in ONE round (version 1
//      ran 4608 blocks on 1024 slots = 4.5 rounds), loads the exp table once and pays the prologue once.  A block's row
//      sums go to slot blockIdx.x of its sub-quantiser's partial list; the last block of an m to arrive adds the slots in
//      order and re-arms the arrival counter.
//
// LDS (static): table 32 KiB (the block-reduction scratch aliases it) + 256 offset potentials + a few words: 34 KiB, four
// blocks per CU.  Order of every sum is fixed by (B, M, gridDim.x): results are identical run to run.
#define SK2_TB 12
#define SK2_N (1 << SK2_TB)
#define SK2_C0 0x1.0000000000072p+0      // minimax of 2^(r/4096) on [0,1], degree 2 (tools/exp2_minimax.py): max error 2.52e-14
#define SK2_C1 0x1.62e42fdfa7202p-13
#define SK2_C2 0x1.ec06883f312a2p-27
#define SK2_MAX_BLOCKS 2048
#define SK2_UMAX 134217728.0             // 2^27: bound on u' (n << 3 must stay inside int32)

__device__ __forceinline__ double sk2_exp(double u, int goff8, const double* __restrict__ s_tab) {
    const double rf = __builtin_amdgcn_fract(u);             // u >= 0: u = n + rf
    const int n = (int)u;                                    // v_cvt_i32_f64 truncates = floor for u >= 0
    const int ni8 = (n << 3) + goff8;
    const double T = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(s_tab) + (ni8 & ((SK2_N - 1) << 3)));
    const double p = __builtin_fma(__builtin_fma(SK2_C2, rf, SK2_C1), rf, SK2_C0);
    return __builtin_ldexp(T * p, ni8 >> (SK2_TB + 3));
}

// position of potential (lane, i) in s_fk: pairs (i, i+1) of one lane are adjacent and the 16 lanes of a column group
// read 16 consecutive 16-byte slots (one conflict-free ds_read_b128 per pair)
__device__ __forceinline__ int sk2_fkpos(int lane, int i) { return (((i >> 1) << 4) + lane) * 2 + (i & 1); }

template <bool FIRST, bool FKLDS>
__device__ __forceinline__ void sk2_column(const float (&x)[SK_EPL], const double (&fk)[SK_EPL],
                                           const double* __restrict__ s_fk, const double* __restrict__ s_tab, int lane,
                                           double (&R)[SK_EPL], int g8, int off8, double offd, double nse,
                                           int* __restrict__ gq_out, bool writer, bool& bad) {
    const int goff8 = g8 - off8;
    if constexpr (FIRST) {
#pragma unroll
        for (int i = 0; i < SK_EPL; ++i) R[i] += sk2_exp(__builtin_fma((double)x[i], nse, offd), goff8, s_tab);
    } else {
        double w[SK_EPL];
        double c = 0.0;
        int ln = lane;
        if constexpr (FKLDS) asm volatile("" : "+v"(ln));    // re-read from LDS every column (keeps the loads un-hoisted)
#pragma unroll
        for (int i = 0; i < SK_EPL; i += 2) {
            double f0, f1;
            if constexpr (FKLDS) {
                const double2 p = *reinterpret_cast<const double2*>(s_fk + sk2_fkpos(ln, i));
                f0 = p.x; f1 = p.y;
            } else {
                f0 = fk[i]; f1 = fk[i + 1];
            }
            w[i] = sk2_exp(__builtin_fma((double)x[i], nse, f0), goff8, s_tab);
            c += w[i];
            w[i + 1] = sk2_exp(__builtin_fma((double)x[i + 1], nse, f1), goff8, s_tab);
            c += w[i + 1];
        }
        c = rc_row16_allreduce_sum(c);
        double y = __builtin_amdgcn_rcp(c);                   // 2^-24 relative; two Newton steps -> rounding level
        double e = __builtin_fma(-c, y, 1.0);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-c, y, 1.0);
        y = __builtin_fma(y, e, y);
#pragma unroll
        for (int i = 0; i < SK_EPL; ++i) R[i] = __builtin_fma(w[i], y, R[i]);
        bad |= !(c > 0.0) || !(c < INFINITY);
        // new column exponent: log2(c) N ~ (high word of c - high word of 1.0) >> (20 - TB), kept pre-multiplied by 8
        if (writer) *gq_out = g8 - (((__double2hiint(c) - 0x3FF00000) >> (20 - SK2_TB - 3)) & ~7);
    }
}

// The B columns of a sub-quantiser are split into gridDim.x contiguous ranges, the first r = B % gridDim.x of them one
// column longer (q + 1 = B / gridDim.x + 1 columns).
__device__ __forceinline__ unsigned sk2_range_lo(unsigned i, unsigned q, unsigned r) { return i * q + (i < r ? i : r); }

// Wave priority for the main loop of the sweeps.  The four blocks that share a CU (one wave of each per SIMD) do not run
// at the same speed: the SIMD's arbiter prefers the OLDEST wave, so the block dispatched first streams its columns in
// ~260 us and the fourth in ~390 us (trace of a resident experimental kernel, profiles/r04e_sk_tiers.txt; a static
// s_setprio in the opposite order reverses the ranking exactly), and a launch lasts as long as its slowest block.  The
// priority therefore rotates: every 2^shift ticks of the 100 MHz clock the four dispatch rounds of a CU move on by one
// level, so each of them holds each level a quarter of the time and they finish together.  49 152 x 48: 0.421 ->
// 0.398 ms per sweep (step 43.6 -> 41.2 ms), 6 144 x 48: 59.6 -> 57.2 us; RC_SK_PRIO=0 switches it off.
__device__ __forceinline__ void sk_setprio(int p) {     // s_setprio takes an immediate
    switch (p & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
#define SK_PRIO_SHIFT 10                 // 2^10 ticks of 10 ns

// ---- fused exchange (round 5; IPC transport of comm.hip, modeling_repconc.py:155-157) ------------------------------------------
// One Sinkhorn iteration on N > 1 ranks is ONE launch: the block that reduces sub-quantiser m's partials stores the [K] row
// sums straight into every peer's receive region (slot `rank` of the dense [world][M][K] all-gather result) and, once its
// stores have drained, writes the exchange's sequence number into flag (m, rank) of every peer; the next sweep's blocks of m
// wait for the `world` flags of m in their prologue — AFTER the exp2 table and the first column have been requested, so the
// flight time of the peers' stores hides behind those loads — and read the gathered sums with system-scope loads.
// Sequence numbers never repeat (seq_base: a device word the solve sets before its first sweep, so a replayed graph reads the
// current one), hence nothing is re-armed and a late or repeated arrival cannot be mistaken for the next exchange; the
// two-parity argument of comm.hip covers the regions, per sub-quantiser: rank q's push of (exchange n + 2, m) follows its
// m-blocks' wait for MY push of (n + 1, m), which my reducer issues after all my m-blocks of sweep n + 1 have read region n.
// A wait that times out raises RC_FLAG_COMM in the result flags and in the transport's status word; a broken transport
// neither waits nor pushes (peers time out instead of reading half an exchange).
__device__ __forceinline__ bool sk_xchg_wait_flag(const unsigned long long* __restrict__ fp, unsigned long long want,
                                                  const sk_xchg& x, int* __restrict__ flags) {
    // ACQUIRE at system scope on the load that sees the flag (pairs with the pusher's RELEASE store below): the peers' row sums
    // are visible to the loads that follow the block barrier — by the HIP memory model, not only because relaxed system-scope
    // accesses carry sc0 sc1 on gfx942 / gfx950.  One cache invalidate per polling wave.
    if (__hip_atomic_load(fp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= want) return true;
    if (__hip_atomic_load(x.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & RC_FLAG_COMM) {
        atomicOr(flags, RC_FLAG_COMM);
        return false;
    }
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > x.timeout_ticks) {
            atomicOr(flags, RC_FLAG_COMM);
            atomicOr(x.status, RC_FLAG_COMM);
            return false;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);               // system scope
    return true;
}

// MODE 0: first sweep on a centred table; 1: first sweep, centring fused (d holds the raw table); 2: sweep t >= 1.
// XCHG: the row sums leave through sk_xchg (x.push) and / or the prologue waits for the previous exchange (x.wait).
template <int MODE, bool FKLDS, bool XCHG = false>
__global__ __launch_bounds__(SK_THREADS, (MODE == 2 && !FKLDS) ? 3 : 4) void sk_sweep2_kernel(
    float* __restrict__ d, const double* __restrict__ rows_prev, int G, const double* __restrict__ f_in,
    double* __restrict__ f_out, int* __restrict__ gq, double* __restrict__ part, unsigned* __restrict__ counters,
    double* __restrict__ rows_out, unsigned B, int M, unsigned rq, unsigned rr, double nse, double scale, double lmax,
    const double* __restrict__ exp2_tab, int t, int* __restrict__ flags, const float* __restrict__ cmx,
    const float* __restrict__ cmn, int prio_shift, int blocks_per_round, const sk_xchg x) {
    constexpr bool FIRST = MODE != 2;
    static_assert(!(FIRST && FKLDS), "the first sweep has no row potentials");
    __shared__ __attribute__((aligned(16))) double s_tab[SK2_N];      // red[16][256] aliases it after the main loop
    __shared__ __attribute__((aligned(16))) double s_fk[RC_K];
    __shared__ double s_mm[8];
    __shared__ int s_last;
    double(*red)[RC_K] = reinterpret_cast<double(*)[RC_K]>(s_tab);

    const int tid = threadIdx.x;
    const int lane = tid & (SK_GROUP - 1);
    const int grp = tid / SK_GROUP;
    const unsigned bi = blockIdx.x, m = blockIdx.y;
    const unsigned c0 = sk2_range_lo(bi, rq, rr), c1 = sk2_range_lo(bi + 1u, rq, rr);
    bool bad = false;
    {
        // ---- (a) everything with a long latency first: the table, the first column.  XCHG: the table goes global -> LDS
        // by DMA (global_load_lds_dwordx4, no registers: the wait below must not push the prologue into scratch)
        constexpr bool TAB_DMA = XCHG;       // (the plain kernel with the DMA table measures the same: 5.88 vs 5.87 ms per step)
        double2 tv[TAB_DMA ? 1 : SK2_N / 2 / SK_THREADS];
        if constexpr (TAB_DMA) {
#pragma unroll
            for (int j = 0; j < SK2_N / 2 / SK_THREADS; ++j) {
                const int w0 = j * SK_THREADS + (tid & ~63);              // first 16-byte chunk of this wave (wave-uniform)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(reinterpret_cast<const double2*>(exp2_tab) + w0 + (tid & 63)),
                    (__attribute__((address_space(3))) void*)(reinterpret_cast<double2*>(s_tab) + w0), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < SK2_N / 2 / SK_THREADS; ++j)
                tv[j] = reinterpret_cast<const double2*>(exp2_tab)[j * SK_THREADS + tid];
        }
        float* dm = d + (size_t)m * B * RC_K + lane * 4;
        int* gm = gq + (size_t)m * B;
        float xa[SK_EPL], xb[SK_EPL];
        int ga = 0, gb = 0;
        unsigned col = c0 + grp;
        if (col < c1) {
            sk_load_col(dm + (size_t)col * RC_K, xa);
            if (!FIRST && t > 1) ga = gm[col];
        }
        // ---- (b) row potentials of this m after the update that follows sweep t-1 (modeling_repconc.py:157-158):
        // f = f_prev - log(sum over ranks of rows_prev), ranks ascending; the loads go out before the table is parked in
        // LDS, the logarithm comes after (the table registers are dead by then)
        double rs = 1.0, fo = 0.0;
        if constexpr (!FIRST) {
            rs = 0.0;
            if constexpr (XCHG) {
                if (x.wait) {
                    // lanes r < world of the first wave watch flag (m, r) of the previous exchange (number seq_base + t - 1,
                    // written as seq_base + t) while the other waves sit at the barrier (no issue slots, one poller per
                    // block); the gathered sums are then read past every cache
                    if (tid < G)
                        (void)sk_xchg_wait_flag(x.wait_flags + (size_t)m * RC_IPC_MAX_WORLD + tid,
                                                *x.seq_base + (unsigned long long)t, x, flags);
                    __syncthreads();
                    for (int r = 0; r < G; ++r)
                        rs += __hip_atomic_load(rows_prev + ((size_t)r * M + m) * RC_K + tid, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    for (int r = 0; r < G; ++r) rs += rows_prev[((size_t)r * M + m) * RC_K + tid];
                }
            } else {
                for (int r = 0; r < G; ++r) rs += rows_prev[((size_t)r * M + m) * RC_K + tid];
            }
            bad |= !(rs > 0.0) || !(rs < INFINITY);
            if (t > 1) fo = f_in[(size_t)m * RC_K + tid];
        }
        if constexpr (TAB_DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of the table have landed
        } else {
#pragma unroll
            for (int j = 0; j < SK2_N / 2 / SK_THREADS; ++j)
                reinterpret_cast<double2*>(s_tab)[j * SK_THREADS + tid] = tv[j];
        }
        double fn = 0.0;
        if constexpr (!FIRST) {
            fn = fo - log(rs);
            if (bi == 0) f_out[(size_t)m * RC_K + tid] = fn;
        }
        double flo = fn, fhi = fn;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            flo = fmin(flo, __shfl_xor(flo, o));
            fhi = fmax(fhi, __shfl_xor(fhi, o));
        }
        if ((tid & 63) == 0) { s_mm[tid >> 6] = flo; s_mm[4 + (tid >> 6)] = fhi; }
        __syncthreads();
        flo = fmin(fmin(s_mm[0], s_mm[1]), fmin(s_mm[2], s_mm[3]));
        fhi = fmax(fmax(s_mm[4], s_mm[5]), fmax(s_mm[6], s_mm[7]));
        // offset: smallest power of two that makes (L + f) scale + OFF >= 0 for every |L| <= lmax
        const double need = (lmax - flo) * scale * (1.0 + 1e-9) + 4.0;
        int ex = 0;
        (void)frexp(need, &ex);
        const double offd = ldexp(1.0, ex);
        // reported at once (a value kept for the end of the kernel would sit in scratch for the whole main loop)
        if ((!(need < SK2_UMAX / 2) || !((lmax + fhi) * scale + offd < SK2_UMAX)) && tid == 0) atomicOr(flags, RC_FLAG_RANGE);
        const int off8 = (int)offd << 3;
        {
            const int ki = ((tid >> 6) << 2) | (tid & 3), kl = (tid >> 2) & 15;     // tid = sk_kidx(kl, ki)
            s_fk[sk2_fkpos(kl, ki)] = __builtin_fma(fn, scale, offd);
        }
        __syncthreads();

        double fk[SK_EPL], R[SK_EPL];
#pragma unroll
        for (int i = 0; i < SK_EPL; ++i) {
            fk[i] = (FIRST || FKLDS) ? 0.0 : s_fk[sk2_fkpos(lane, i)];
            R[i] = 0.0;
        }
        float cmid = 0.f, camp = 1.f;
        if constexpr (MODE == 1) {
            const float mx = cmx[m], mn = cmn[m];
            cmid = (mx + mn) / 2.0f;                              // centre_kernel's arithmetic, pq_distance.hip
            camp = (mx - cmid) + 1e-5f;
        }
        // ---- (c) main loop, two columns per trip (ping-pong register buffers, next column always in flight)
        const int round = (int)((blockIdx.y * gridDim.x + blockIdx.x) / (unsigned)blocks_per_round);   // dispatch round of the CU
        while (col < c1) {
            if (prio_shift >= 0) sk_setprio(round + (int)(wall_clock64() >> prio_shift));
            const unsigned colb = col + SK_NG;
            if (colb < c1) {
                sk_load_col(dm + (size_t)colb * RC_K, xb);
                if (!FIRST && t > 1) gb = gm[colb];
            }
            if constexpr (MODE == 1) sk_centre_store(dm + (size_t)col * RC_K, xa, cmid, camp);
            sk2_column<FIRST, FKLDS>(xa, fk, s_fk, s_tab, lane, R, ga, off8, offd, nse, gm + col, lane == 0, bad);
            if (colb >= c1) break;
            const unsigned cola = colb + SK_NG;
            if (cola < c1) {
                sk_load_col(dm + (size_t)cola * RC_K, xa);
                if (!FIRST && t > 1) ga = gm[cola];
            }
            if constexpr (MODE == 1) sk_centre_store(dm + (size_t)colb * RC_K, xb, cmid, camp);
            sk2_column<FIRST, FKLDS>(xb, fk, s_fk, s_tab, lane, R, gb, off8, offd, nse, gm + colb, lane == 0, bad);
            col = cola;
        }
        // ---- (d) block reduction of the row sums, fixed order over the 16 column groups (scratch = the dead table)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SK_EPL; ++i) red[grp][sk_kidx(lane, i)] = R[i];
        __syncthreads();
        // an opaque copy of the thread index: the epilogue's addresses are formed here, not before the main loop (where
        // the compiler would park them in scratch for the duration — the kernel sits exactly at its 128-VGPR budget)
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        double s = red[0][tid_e];
#pragma unroll
        for (int q = 1; q < SK_NG; ++q) s += red[q][tid_e];
        // ---- (e) hand the partial to whichever block of this m arrives last: write-through (sc1) 8-byte stores, every
        // wave drains them, one relaxed agent-scope counter add; the reducer reads with L1-bypassing loads.
        const unsigned cnt = gridDim.x, slot = bi;
        double* pm = part + (size_t)m * cnt * RC_K;
        __hip_atomic_store(pm + (size_t)slot * RC_K + tid_e, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid_e == 0) {
            const unsigned old = __hip_atomic_fetch_add(counters + m, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int last = (old + 1u == cnt);
            // the last arriver re-arms the counter: the next sweep may use another grid (first sweep: 4 blocks per CU,
            // register-potential sweeps: 3), so nothing may depend on the count left behind
            if (last) __hip_atomic_store(counters + m, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if constexpr (XCHG)      // 2: the reducer also pushes (decided by one thread: the branch holds a barrier)
                if (last && x.push && !(__hip_atomic_load(x.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & RC_FLAG_COMM))
                    last = 2;
            s_last = last;
        }
        __syncthreads();
        if (s_last) {
            // slots in order, 8 loads in flight (the reducer is the tail of the sweep: pure L2 latency)
            double acc = 0.0;
            unsigned i = 0;
            for (; i + 8 <= cnt; i += 8) {
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = __hip_atomic_load(pm + (size_t)(i + j) * RC_K + tid_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += v[j];
            }
            for (; i < cnt; ++i)
                acc += __hip_atomic_load(pm + (size_t)i * RC_K + tid_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rows_out) rows_out[(size_t)m * RC_K + tid_e] = acc;
            if constexpr (XCHG) {
                if (s_last == 2) {
                    // slot `rank` of every peer's region of this exchange: write-through system-scope stores, drained by
                    // every wave, then the sequence number into flag (m, rank) of each peer (one lane per peer)
                    const size_t off = x.push_data_off + (((size_t)x.rank * M + m) * RC_K + tid_e) * sizeof(double);
                    for (int p = 0; p < x.world; ++p)
                        __hip_atomic_store(reinterpret_cast<double*>(x.peers[p] + off), acc, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid_e < x.world)
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(x.peers[tid_e] + x.push_flag_off) +
                                               (size_t)m * RC_IPC_MAX_WORLD + x.rank,
                                           *x.seq_base + (unsigned long long)t + 1ull, SK_XCHG_FLAG_ORDER,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
    if (__any(bad) && (tid & 63) == 0) atomicOr(flags, RC_FLAG_NONFINITE);
}

// ------------------------------------------------------------------------------------------ host
// columns per block: every block pays a fixed prologue (16 KiB table, 256 potentials, its columns' g) and
// the partial hand-off, so blocks should be as long as the grid allows while still filling the chip once:
// 3 blocks are resident per CU (VGPR-limited), 256 CUs -> the longest block length that still yields >= 768 blocks.
// Measured on MI355X (us per sweep; blocks in brackets):  B = 6144, M = 48: 96 -> 93 [3072], 192 -> 80 [1536],
// 384 -> 75 [768], 512 -> 87 [576];  B = 6144, M = 24 (one chain of the two-chain multi-rank solve): 96 -> 48 [1536],
// 192 -> 39 [768], 256 -> 46 [576];  B = 12288, M = 48: 192 -> 163, 384 -> 155 [1536], 512 -> 158 [1152];
// B >= 24576 -> 512.
static int sk_cols_per_block(int64_t B, int M) {
    const int forced = rc_env_int("RC_SK_CPB", 0);   // development / test override, read per call
    if (forced >= 16 && forced <= SK_MAX_CPB) return forced;
    static const int cand[] = {512, 384, 256, 192, 128, 96, 64};
    for (int c : cand)
        if (((B + c - 1) / c) * M >= 768) return c;
    return 64;
}
static double sk_scale() { return (double)SK_N / SK_LN2; }

// version 2 (default) unless RC_SK_V1=1 (the round-1 kernel, kept for A/B runs and as a cross-check in the tests)
static bool sk_use_v2() { return rc_env_int("RC_SK_V1", 0) == 0; }
// blocks per sub-quantiser of the version-2 sweep.  Large launches: the chip's resident slots (`per_cu` per CU) divided by M
// — every block does the same work in one round.  Small launches pay per BLOCK (32 KiB table, 256 logarithms, the hand-over of
// its partial sums), so they take fewer and longer blocks — measured in round 4 (ms per 100-iteration solve, default of rounds
// 2-3 -> this rule; B x M): 6144 x 24, one chain of the 8-GPU recipe: 3.99 -> 3.62; 6144 x 12: 3.51 -> 2.66; 3072 x 24: 2.82 ->
// 2.40; 3072 x 48: 3.62 -> 3.44; 1536 x 24: 2.30 -> 1.86; 1024 x 48: 2.13 -> 1.87; 8192 x 8: 3.47 -> 2.55; 2048 x 8: 1.70 -> 1.35;
// 6144 x 48, 12288 x 24 and everything larger: unchanged —
//   * three blocks per CU instead of four while a block would hold fewer than 384 columns (768 blocks = exactly three per CU
//     beat 1008 = "almost four" by 9 % at 6144 x 24, and 912 or 648 blocks, which load the CUs unevenly, lose it again);
//   * never fewer than 128 columns per block,
//   * but one block per CU while a block still gets 64 columns (batches of a few hundred rows).
// RC_SK_NB overrides the TOTAL number of blocks (tests).
static int sk2_blocks_per_m(rc_handle_t h, int64_t B, int M, int per_cu) {
    const int64_t cus = (h && h->num_cus > 0) ? h->num_cus : 256;
    int64_t nb = rc_env_int("RC_SK_NB", 0);
    const bool forced = nb > 0;
    if (!forced) {
        nb = (int64_t)per_cu * cus;
        if (per_cu > 3 && B / (nb / M > 0 ? nb / M : 1) < 384) nb = 3 * cus;
    }
    if (nb > SK2_MAX_BLOCKS) nb = SK2_MAX_BLOCKS;
    int64_t nbm = nb / M;
    if (!forced) {
        if (nbm > B / 128) nbm = B / 128;
        if (nbm * M < cus) {                                  // not even one block per CU: shorter blocks, down to 64 columns
            nbm = (cus + M - 1) / M;
            if (nbm > B / 64) nbm = B / 64;
        }
    } else if (nbm > B / 32) {
        nbm = B / 32;
    }
    if (nbm < 1) nbm = 1;
    return (int)nbm;
}

struct sk_sweep_ws {
    size_t part, counters, total;
};
static sk_sweep_ws sk_ws(int64_t B, int M) {
    sk_sweep_ws w;
    const int cpb = sk_cols_per_block(B, M);
    const int64_t nblk = (B + cpb - 1) / cpb;
    size_t slots = (size_t)M * nblk;                                               // version 1
    const size_t slots2 = (size_t)SK2_MAX_BLOCKS + M;                              // version 2, any grid (M * nbm <= max blocks)
    if (slots2 > slots) slots = slots2;
    w.part = 0;
    w.counters = rc_align_up(slots * RC_K * sizeof(double), 256);
    w.total = w.counters + rc_align_up((size_t)M * sizeof(unsigned), 256);
    return w;
}

// launch of the version-2 sweep.  mode 0 / 1 / 2 as in sk_sweep2_kernel; g is the caller's [M,B] fp64 scratch, used as
// int32 [M*B] column exponents.
static int sk2_launch(rc_handle_t h, int mode, float* d, const double* rows_prev, int G, double* f2, double* g,
                      double* rows_out, int64_t B, int M, double eps, int t, int* flags, void* ws, const float* mx,
                      const float* mn, hipStream_t s, const sk_xchg* xc = nullptr) {
    const sk_sweep_ws W = sk_ws(B, M);
    const double* tab = rc_exp2_table(h, SK2_TB);
    if (!tab) return RC_EHIP;
    double* part = (double*)((char*)ws + W.part);
    unsigned* counters = (unsigned*)((char*)ws + W.counters);
    const double scale = (double)SK2_N / SK_LN2;
    const double nse = -scale / eps;
    const double lmax = (1.0 + 1e-6) / eps;                // |centred distance| < 1 (amp = half range + 1e-5)
    int* gq = reinterpret_cast<int*>(g);
    if (B >= (1ll << 31)) return RC_ESHAPE;
    const unsigned Bu = (unsigned)B;
    const double* f_in = (mode == 2) ? f2 + (size_t)((t - 1) & 1) * M * RC_K : nullptr;
    double* f_out = (mode == 2) ? f2 + (size_t)(t & 1) * M * RC_K : nullptr;
    // default: potentials re-read from LDS (120 VGPRs, four blocks per CU); RC_SK_FKLDS=0: potentials in registers, three
    // blocks per CU.  The first sweep uses the same grid.
    const bool fklds = rc_env_int("RC_SK_FKLDS", 1) != 0;
    const unsigned nbm = (unsigned)sk2_blocks_per_m(h, B, M, fklds ? 4 : 3);
    const dim3 grid(nbm, (unsigned)M);
    // rotating wave priority (sk_setprio); RC_SK_PRIO=0: off, n >= 1: rotate every 2^(n-1) ticks instead of the default
    const int prio_env = rc_env_int("RC_SK_PRIO", -1);
    const int prio_shift = prio_env == 0 ? -1 : (prio_env > 0 ? prio_env - 1 : SK_PRIO_SHIFT);
    const int per_round = (h && h->num_cus > 0) ? h->num_cus : 256;
    const sk_xchg x0 = {};
    const sk_xchg xv = xc ? *xc : x0;
#define SK2_GO(MODE, FK, XC)                                                                                              \
    hipLaunchKernelGGL((sk_sweep2_kernel<MODE, FK, XC>), grid, dim3(SK_THREADS), 0, s, d, rows_prev, G, f_in, f_out, gq, part, \
                       counters, rows_out, Bu, M, Bu / nbm, Bu % nbm, nse, scale, lmax, tab, t, flags, mx, mn, prio_shift,  \
                       per_round, xv)
    if (mode != 2) {
        RC_HIP_CHECK(h, hipMemsetAsync(counters, 0, (size_t)M * sizeof(unsigned), s));
        if (mode == 0) { if (xc) SK2_GO(0, false, true); else SK2_GO(0, false, false); }
        else           { if (xc) SK2_GO(1, false, true); else SK2_GO(1, false, false); }
    } else {
        rc_prof_mark(h, RC_PROF_SK_PASS, s);
        if (fklds) { if (xc) SK2_GO(2, true, true); else SK2_GO(2, true, false); }
        else       { if (xc) SK2_GO(2, false, true); else SK2_GO(2, false, false); }
        rc_prof_mark(h, RC_PROF_SK_PASS, s);
    }
#undef SK2_GO
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

extern "C" size_t rc_sk_ws_bytes(int64_t B, int M, int K) {
    if (B <= 0 || M <= 0 || K != RC_K) return 0;
    return sk_ws(B, M).total;
}

bool rc_sk_xchg_capable() { return sk_use_v2(); }

int rc_sk_sweep_x(rc_handle_t h, const float* d, const double* rows_prev, int G, double* f2, double* g, double* colsum,
                  double* rows_out, int64_t B, int M, double eps, int t, int* flags, void* ws, size_t ws_bytes,
                  hipStream_t s, const sk_xchg* xc) {
    if (!xc) return rc_sk_sweep(h, d, rows_prev, G, f2, g, colsum, rows_out, B, M, RC_K, eps, t, flags, ws, ws_bytes, (rc_stream_t)s);
    if (!sk_use_v2() || !h || !d || !flags || B <= 0 || M <= 0 || t < 0 || !(eps > 0.0)) return RC_EINVAL;
    if (t > 0 && (!rows_prev || G <= 0 || !f2 || !g)) return RC_EINVAL;
    if (!ws || ws_bytes < sk_ws(B, M).total) return RC_EWORKSPACE;
    return sk2_launch(h, t == 0 ? 0 : 2, const_cast<float*>(d), rows_prev, G, f2, g, rows_out, B, M, eps, t, flags, ws, nullptr,
                      nullptr, s, xc);
}

extern "C" int rc_sk_sweep(rc_handle_t h, const float* d, const double* rows_prev, int G, double* f2, double* g,
                           double* colsum, double* rows_out, int64_t B, int M, int K, double eps, int t,
                           int* flags, void* ws, size_t ws_bytes, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !d || !rows_out || !flags || B <= 0 || M <= 0 || t < 0 || !(eps > 0.0)) return RC_EINVAL;
    if (t > 0 && (!rows_prev || G <= 0 || !f2 || !g || !colsum)) return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    const sk_sweep_ws W = sk_ws(B, M);
    if (!ws || ws_bytes < W.total) return RC_EWORKSPACE;
    if (sk_use_v2())
        return sk2_launch(h, t == 0 ? 0 : 2, const_cast<float*>(d), rows_prev, G, f2, g, rows_out, B, M, eps, t, flags, ws,
                          nullptr, nullptr, (hipStream_t)stream);
    const double* tab = rc_exp2_table(h, SK_TB);
    if (!tab) return RC_EHIP;
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)((char*)ws + W.part);
    unsigned* counters = (unsigned*)((char*)ws + W.counters);
    const int cpb = sk_cols_per_block(B, M);
    const int64_t nblk = (B + cpb - 1) / cpb;
    const double scale = sk_scale();
    const double nse = -scale / eps;
    const size_t lds = ((size_t)SK_N + SK_NG * RC_K + RC_K) * sizeof(double) + (SK_MAX_CPB + 4) * sizeof(int);
    dim3 grid((unsigned)nblk, (unsigned)M);
    if (t == 0) {
        RC_HIP_CHECK(h, hipMemsetAsync(counters, 0, (size_t)M * sizeof(unsigned), s));
        hipLaunchKernelGGL(sk_sweep_kernel<true>, grid, dim3(SK_THREADS), lds, s, d, rows_prev, G,
                           (const double*)nullptr, (double*)nullptr, g, colsum, part, counters, rows_out, B, cpb, nse,
                           scale, tab, t, flags);
    } else {
        const double* f_in = f2 + (size_t)((t - 1) & 1) * M * RC_K;
        double* f_out = f2 + (size_t)(t & 1) * M * RC_K;
        // grids of >= 4 resident rounds' worth of blocks take the four-waves-per-SIMD variant (potentials from LDS)
        const int force_fklds = rc_env_int("RC_SK_FKLDS", -1);
        const bool fklds = force_fklds >= 0 ? force_fklds != 0 : (nblk * M >= 4096);
        const size_t lds4 = ((size_t)SK_NG * RC_K + RC_K) * sizeof(double) + (SK_MAX_CPB + 4) * sizeof(int);
        rc_prof_mark(h, RC_PROF_SK_PASS, s);
        if (fklds)
            hipLaunchKernelGGL((sk_sweep_kernel<false, false, true>), grid, dim3(SK_THREADS), lds4, s, d, rows_prev, G, f_in,
                               f_out, g, colsum, part, counters, rows_out, B, cpb, nse, scale, tab, t, flags);
        else
            hipLaunchKernelGGL((sk_sweep_kernel<false, false, false>), grid, dim3(SK_THREADS), lds, s, d, rows_prev, G, f_in,
                               f_out, g, colsum, part, counters, rows_out, B, cpb, nse, scale, tab, t, flags);
        rc_prof_mark(h, RC_PROF_SK_PASS, s);
    }
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// First sweep with the centring fused in: d is the RAW table of M sub-quantisers, mx / mn their (global) maxima and
// minima.  Equivalent to rc_pq_centre followed by rc_sk_sweep(t = 0).
int rc_sk_sweep0_centre(rc_handle_t h, float* d, const float* mx, const float* mn, double* g, double* colsum,
                        double* rows_out, int64_t B, int M, double eps, int* flags, void* ws, size_t ws_bytes,
                        hipStream_t s, const sk_xchg* xc) {
    const sk_sweep_ws W = sk_ws(B, M);
    if (!ws || ws_bytes < W.total) return RC_EWORKSPACE;
    if (sk_use_v2())
        return sk2_launch(h, 1, d, nullptr, 1, nullptr, g, rows_out, B, M, eps, 0, flags, ws, mx, mn, s, xc);
    if (xc) return RC_EINVAL;
    const double* tab = rc_exp2_table(h, SK_TB);
    if (!tab) return RC_EHIP;
    double* part = (double*)((char*)ws + W.part);
    unsigned* counters = (unsigned*)((char*)ws + W.counters);
    const int cpb = sk_cols_per_block(B, M);
    const int64_t nblk = (B + cpb - 1) / cpb;
    const double scale = sk_scale();
    const size_t lds = ((size_t)SK_N + SK_NG * RC_K + RC_K) * sizeof(double) + (SK_MAX_CPB + 4) * sizeof(int);
    RC_HIP_CHECK(h, hipMemsetAsync(counters, 0, (size_t)M * sizeof(unsigned), s));
    hipLaunchKernelGGL((sk_sweep_kernel<true, true>), dim3((unsigned)nblk, (unsigned)M), dim3(SK_THREADS), lds, s,
                       (const float*)d, (const double*)nullptr, 1, (const double*)nullptr, (double*)nullptr, g, colsum,
                       part, counters, rows_out, B, cpb, -scale / eps, scale, tab, 0, flags, mx, mn);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// argmax over the sub-quantisers of `d` (M of them), written into codes [B, code_stride] at column m_offset
int rc_sk_argmax_strided(rc_handle_t h, const float* d, const double* rows_prev, int G, const double* f2, int64_t B,
                         int M, double eps, int t, int code_stride, int m_offset, uint8_t* codes_u8,
                         int64_t* codes_i64, int* flags, hipStream_t s) {
    const int cpb = sk_cols_per_block(B, M);
    const int64_t nblk = (B + cpb - 1) / cpb;
    const double* f_in = f2 + (size_t)((t - 1) & 1) * M * RC_K;
    hipLaunchKernelGGL(sk_argmax_kernel, dim3((unsigned)nblk, (unsigned)M), dim3(SK_THREADS), 0, s, d, rows_prev, G,
                       f_in, t, B, cpb, -1.0 / eps, code_stride, m_offset, codes_u8, codes_i64, flags);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

extern "C" int rc_sk_argmax(rc_handle_t h, const float* d, const double* rows_prev, int G, const double* f2,
                            int64_t B, int M, int K, double eps, int t, uint8_t* codes_u8, int64_t* codes_i64,
                            int* flags, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !d || !rows_prev || !f2 || !flags || G <= 0 || B <= 0 || M <= 0 || t < 1 || !(eps > 0.0) ||
        (!codes_u8 && !codes_i64))
        return RC_EINVAL;
    if (K != RC_K) return RC_ESHAPE;
    return rc_sk_argmax_strided(h, d, rows_prev, G, f2, B, M, eps, t, M, 0, codes_u8, codes_i64, flags,
                                (hipStream_t)stream);
}

// ---- one-call single-rank constrained assignment -----------------------------------------
extern "C" size_t rc_pq_assign_sinkhorn_ws_bytes(int64_t B, int M, int K) {
    if (B <= 0 || M <= 0 || K != RC_K) return 0;
    return rc_solve_ws_bytes(B, M, 1);
}

extern "C" int rc_pq_assign_sinkhorn(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B, int D,
                                     int M, int K, double eps, int iters, uint8_t* codes_u8, int64_t* codes_i64,
                                     int* flags, void* ws, size_t ws_bytes, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !x || !C || !flags || B < 0 || M <= 0 || iters < 1 || !(eps > 0.0) || (!codes_u8 && !codes_i64))
        return RC_EINVAL;
    if (K != RC_K || D % M != 0) return RC_ESHAPE;             // any width: rc_pq_dist_table picks the kernel
    if (B == 0) return RC_OK;
    return rc_solve_chains(h, x, ldx, C, B, D, M, eps, iters, 1, codes_u8, codes_i64, flags, ws, ws_bytes,
                           (hipStream_t)stream);
}

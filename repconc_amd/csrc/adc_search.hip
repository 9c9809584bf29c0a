// PQ asymmetric-distance (inner-product) top-k search over raw uint8 codes.
//
// Reference: `index.search(query_embeds, topk)` on a faiss.IndexPQ(D, M, 8, METRIC_INNER_PRODUCT)
// or its 1-list IVFPQ clone — models/repconc/evaluate_repconc.py:78-135,180-185 and
// models/jpq/finetune_jpq.py:176.  Faiss is not vendored in the reference; the arithmetic restated
// here is Faiss 1.7.x's published IndexPQ behaviour: per query an
// inner-product table LUT[m][k] = <q_m, C[m,k]>, score(n) = sum_m LUT[m][code[n,m]] accumulated
// m-ascending in fp32, the k largest returned in decreasing order (ties: lower id first).
//
// Pipeline for one batch of queries (all on one stream, no host round trip):
//   1. adc_lut_kernel        LUT[nq][M][256] (fp32; 48 KiB per query at M=48)
//   2. adc_scan_kernel<SAMPLE> scores of S <= 32768 evenly spread rows -> sample[nq][S]
//   3. adc_threshold_kernel  per query: the r-th largest sample score (LDS radix select) = tau_q;
//                            r is chosen so that ~ (r/S)*N >> k rows pass, i.e. the true top-k are
//                            all >= tau_q with overwhelming probability (host checks the count)
//   4. full scan, rows with score >= tau_q are appended (wave-aggregated atomics) to a per-query candidate
//      list of 64-bit keys.  Two implementations with IDENTICAL output:
//        a. adc_scan_kernel<FILTER>: exact fp32 scores for every row (small indexes);
//        b. integer screening (N >= 2^18, k <= 2048): adc_qlut_kernel quantises each query's tables to 8 bits with a
//           common step Delta_q (l = round((LUT - min_m)/Delta_q), nearest since round 3); the screen kernel sums the bytes of 8 queries per
//           LDS gather (one ds_read_b64 serves 8 queries instead of 2) — adc_screen_mfma_kernel on the matrix cores
//           (v_mfma_i32_32x32x32_i8 against a selection matrix; two table phases for M > 64), adc_screen_kernel on
//           the VALU for M % 8 != 0 — and keeps every row with S_int >= T_q, where
//           T_q = ceil((tau_q - sum_m min_m)/Delta_q - M/2) - 2 is a RIGOROUS lower bound (each of the M entries is off by
//           at most half a step — rounds 1/2 truncated: a whole step, slack M + 2, ~1.4x more survivors —, plus float
//           rounding), so no row with exact score >= tau_q is ever lost; adc_rescore_kernel then
//           computes the exact fp32 score of the survivors (~1.4x the final candidates) and applies the exact test.
//           The candidate set, hence the result, is the same as (a).
//   5. adc_select_kernel     per query: radix-select cut to the k best scores (+ties), bitonic sort in LDS, emit top-k
//
// The scan is the hot kernel.  A block keeps the LUTs of QT queries in LDS, interleaved
// [m][k][QT] so ONE ds_read_b64 / b128 gather serves QT queries, and streams a tile of codes
// (consecutive lanes = consecutive rows, 16-byte loads).  Blocks that share a code tile are
// adjacent in the grid, so a tile is fetched from HBM about once per XCD and re-read from L2.
#include "rc_common.h"
#include <stdio.h>

#include <limits.h>
#include <string.h>

#include <type_traits>

#define ADC_THREADS 1024
#define ADC_SAMPLE_MAX 32768
#define ADC_KTH_LIST 4096            // members of the selected value bin kept in LDS by adc_kth_largest_v
#define ADC_CAND_CAP 16384
#define ADC_TILE_DOCS 32768
#ifndef RC_ADC_IMG16
#define RC_ADC_IMG16 0         // 1: the permuted code image holds 16-bit codes (one v_mad_u32_u16 per gather address instead of bfe + lshl_add)
#endif
#define ADC_IMG_ES (RC_ADC_IMG16 ? 2 : 1)   // bytes per code in the image
// Rows per block of the conflict-free screen.  Consecutive query groups re-read the same tile of the code image, which
// therefore has to stay in the XCD's 4 MiB L2: 3 MiB of image per tile.  (Round 2 first used 65536 rows for every M: at
// M = 96 the 6 MiB tile was re-fetched from HBM by every group - FETCH_SIZE 64 GB per 1200-query launch against 2.3 GB at
// M = 48.)  Multiples of 2048 rows (16 waves x 8 chunks x 16 rows).
#ifndef ADC_T96
#define ADC_T96 32768
#endif
__host__ __device__ constexpr int adc_cf_tile_rows(int M) { return M > 64 ? ADC_T96 : (M > 48 ? 49152 : 65536); }

__device__ __forceinline__ unsigned adc_order_key(float s) {
    const unsigned u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float adc_unorder_key(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// One step of the 8-bit radix select: from the 256-bin histogram of the keys that match `prefix`, the bin that holds the
// need-th largest key, i.e. the largest b with sum_{j >= b} hist[j] >= need — computed by 256 threads with a wave scan.
// (One thread walking down from bin 255 is a chain of dependent LDS reads: ~10 us per pass, 40 of the 46 us a threshold
// block took.)  Called by every thread of a block of >= 256 threads; `need` must have been read before; ends in a barrier.
__device__ __forceinline__ void adc_pick_bin(const unsigned* hist, unsigned need, unsigned prefix, int shift, unsigned* s_scan,
                                             unsigned* sel_prefix, unsigned* sel_rank) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned v = 0u, incl = 0u;
    if (tid < 256) {
        v = hist[255 - tid];                                  // thread t owns bin 255 - t: prefix over t = suffix over bins
        incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = (unsigned)__shfl_up((int)incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_scan[wv] = incl;
    }
    __syncthreads();
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < 3; ++w) incl += (w < wv) ? s_scan[w] : 0u;
        const unsigned excl = incl - v;
        if (incl >= need && excl < need) {
            *sel_prefix = prefix | ((unsigned)(255 - tid) << shift);
            *sel_rank = need - excl;
        } else if (tid == 255 && incl < need) {               // fewer matching keys than asked for: what the walk did
            *sel_prefix = prefix;
            *sel_rank = need - incl;
        }
    }
    __syncthreads();
}

// rank-th largest of n 32-bit keys (key_at(i), i < n; rank in [1, n]) by radix select, 8 bits per pass — but only over the
// bits in which the keys DIFFER: a block min / max first, the common leading bits are the result's.  Scores of one query's
// candidates share their sign / exponent byte (often the next one too): a pass over such a byte sends every key to ONE
// histogram bin, i.e. n LDS atomics on one address, one after the other (round 3: two of the four passes of the 32 768-key
// threshold kernel, ~100 of its 130 us per 1200 queries).  Called by every thread of a block of >= 256 threads; `hist`
// [256], `s_scan` [4], `s_sel` [2], `s_mm` [2] in LDS.
template <typename KeyAt>
__device__ __forceinline__ unsigned adc_kth_largest(KeyAt key_at, int64_t n, unsigned rank, unsigned* hist, unsigned* s_scan,
                                                    unsigned* s_sel, unsigned* s_mm) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (tid == 0) { s_mm[0] = 0xFFFFFFFFu; s_mm[1] = 0u; }
    __syncthreads();
    unsigned mn = 0xFFFFFFFFu, mx = 0u;
    for (int64_t i = tid; i < n; i += nthr) {
        const unsigned k = key_at(i);
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned a = (unsigned)__shfl_xor((int)mn, o), b = (unsigned)__shfl_xor((int)mx, o);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((tid & 63) == 0) { atomicMin(&s_mm[0], mn); atomicMax(&s_mm[1], mx); }
    __syncthreads();
    const unsigned lo = s_mm[0], hi_key = s_mm[1];
    if (lo == hi_key) return hi_key;                          // all keys equal (block-uniform)
    const int top = 31 - __clz((int)(lo ^ hi_key));          // highest bit in which two keys differ
    int undecided = top + 1;                                  // bits [0, undecided)
    if (tid == 0) { s_sel[0] = hi_key & ~((2u << top) - 1u); s_sel[1] = rank; }
    __syncthreads();
    while (undecided > 0) {
        const int width = undecided < 8 ? undecided : 8, shift = undecided - width;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = s_sel[0], need = s_sel[1];
        const unsigned himask = undecided >= 32 ? 0u : (0xFFFFFFFFu << undecided), dmask = (1u << width) - 1u;
        for (int64_t i = tid; i < n; i += nthr) {
            const unsigned k = key_at(i);
            if ((k & himask) == prefix) atomicAdd(&hist[(k >> shift) & dmask], 1u);
        }
        __syncthreads();
        adc_pick_bin(hist, need, prefix, shift, s_scan, &s_sel[0], &s_sel[1]);
        undecided = shift;
    }
    return s_sel[0];
}

// The same answer, faster on real score distributions: bit-radix passes see a float's sign / exponent structure — a
// near-Gaussian sample puts half of its keys into one or two bins of the first pass whatever window of bits it uses
// (measured: skipping the common leading bits alone made the kernels SLOWER, the min / max pass cost more than it saved).
// So the first cut is made in VALUE space: 256 equal bins over [min, max] of the scores (a monotone function of the key:
// bin(s) = min(255, int((s - smin) scale)), so "the bin that holds the rank-th largest" is well defined) — the fullest bin of
// a Gaussian sample holds ~1.3 % of it — then the members of that one bin (a few dozen in the tail where the thresholds
// live) are collected into `list` and the bit-radix select above runs on them.  Non-finite extremes, a degenerate range or
// a bin longer than list_cap: the plain bit-radix select over everything.  `s_aux`: 8 words of LDS.
// MM_READY: the caller has already reduced the keys' minimum / maximum into s_aux[2] / s_aux[3] (e.g. while loading them),
// zeroed hist and s_aux[4], and synchronised.
template <bool MM_READY = false, typename KeyAt>
__device__ __forceinline__ unsigned adc_kth_largest_v(KeyAt key_at, int64_t n, unsigned rank, unsigned* hist, unsigned* s_scan,
                                                      unsigned* s_aux, unsigned* list, int list_cap) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    unsigned* s_sel = s_aux, *s_mm = s_aux + 2, *s_cnt = s_aux + 4;
    if constexpr (!MM_READY) {
        if (tid == 0) { s_mm[0] = 0xFFFFFFFFu; s_mm[1] = 0u; *s_cnt = 0u; }
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        unsigned mn = 0xFFFFFFFFu, mx = 0u;
        for (int64_t i = tid; i < n; i += nthr) {
            const unsigned k = key_at(i);
            mn = k < mn ? k : mn;
            mx = k > mx ? k : mx;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned a = (unsigned)__shfl_xor((int)mn, o), b = (unsigned)__shfl_xor((int)mx, o);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        if ((tid & 63) == 0) { atomicMin(&s_mm[0], mn); atomicMax(&s_mm[1], mx); }
        __syncthreads();
    }
    const unsigned lo = s_mm[0], hi_key = s_mm[1];
    if (lo == hi_key) return hi_key;
    const float smin = adc_unorder_key(lo), smax = adc_unorder_key(hi_key);
    const float scale = 256.0f / (smax - smin);
    const bool linear = (smin - smin == 0.f) && (smax - smax == 0.f) && (scale - scale == 0.f);     // all finite (block-uniform)
    if (!linear) {
        __syncthreads();
        return adc_kth_largest(key_at, n, rank, hist, s_scan, s_sel, s_mm);
    }
    auto bin_of = [&](unsigned k) {
        const int b = (int)((adc_unorder_key(k) - smin) * scale);
        return b > 255 ? 255 : b;
    };
    for (int64_t i = tid; i < n; i += nthr) atomicAdd(&hist[bin_of(key_at(i))], 1u);
    __syncthreads();
    adc_pick_bin(hist, rank, 0u, 0, s_scan, &s_sel[0], &s_sel[1]);    // s_sel[0] = bin, s_sel[1] = rank inside it (ends in a barrier)
    const int b = (int)s_sel[0];
    const unsigned inside = s_sel[1], members = hist[b];
    __syncthreads();
    if ((int)members > list_cap)
        return adc_kth_largest(key_at, n, rank, hist, s_scan, s_sel, s_mm);
    for (int64_t i = tid; i < n; i += nthr) {
        const unsigned k = key_at(i);
        if (bin_of(k) == b) list[atomicAdd(s_cnt, 1u)] = k;
    }
    __syncthreads();
    return adc_kth_largest([&](int64_t i) { return list[i]; }, (int64_t)members, inside, hist, s_scan, s_sel, s_mm);
}

// ------------------------------------------------------------------------------------------ 1. LUT
// grid (nq, M), block 256 (= k).  j-ascending multiply then add, each rounded (no FMA).
__global__ __launch_bounds__(RC_K) void adc_lut_kernel(const float* __restrict__ C, const float* __restrict__ q,
                                                       int D, int M, float* __restrict__ lut) {
    const int qi = blockIdx.x, m = blockIdx.y, k = threadIdx.x;
    const int dsub = D / M;
    const float* qs = q + (size_t)qi * D + m * dsub;  // wave-uniform
    const float* c = C + ((size_t)m * RC_K + k) * dsub;
    float s = 0.f;
    for (int j = 0; j < dsub; ++j) s = s + qs[j] * c[j];
    lut[((size_t)qi * M + m) * RC_K + k] = s;
}

// Same arithmetic, supported dsub: thread k keeps its centroid row in registers and walks a chunk of queries (the
// query slice is block-uniform -> scalar loads), so the 16 KiB centroid table is read once per 32 queries instead of
// once per query (0.40 ms -> per 1200 queries at M = 48 for the kernel above).
#define ADC_LUT_QCHUNK 32
template <int DSUB>
__global__ __launch_bounds__(RC_K) void adc_lut_rows_kernel(const float* __restrict__ C, const float* __restrict__ q,
                                                            int nq, int D, int M, float* __restrict__ lut) {
    const int m = blockIdx.y, k = threadIdx.x;
    float c[DSUB];
    const float4* cp = reinterpret_cast<const float4*>(C + ((size_t)m * RC_K + k) * DSUB);
#pragma unroll
    for (int j = 0; j < DSUB / 4; ++j) {
        const float4 v = cp[j];
        c[4 * j] = v.x; c[4 * j + 1] = v.y; c[4 * j + 2] = v.z; c[4 * j + 3] = v.w;
    }
    const int q0 = blockIdx.x * ADC_LUT_QCHUNK;
    const int q1 = (q0 + ADC_LUT_QCHUNK < nq) ? q0 + ADC_LUT_QCHUNK : nq;
    for (int qi = q0; qi < q1; ++qi) {
        const float* qs = q + (size_t)qi * D + m * DSUB;   // block-uniform
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < DSUB; ++j) s = s + qs[j] * c[j];
        lut[((size_t)qi * M + m) * RC_K + k] = s;
    }
}

// ------------------------------------------------------------------------------------------ 2/4. scan
template <int QT> struct adc_vec;
template <> struct adc_vec<1> { using type = float; };
template <> struct adc_vec<2> { using type = float2; };
template <> struct adc_vec<4> { using type = float4; };

template <int QT>
__device__ __forceinline__ void adc_acc(float (&s)[QT], const typename adc_vec<QT>::type& v) {
    if constexpr (QT == 1) { s[0] = s[0] + v; }
    if constexpr (QT == 2) { s[0] = s[0] + v.x; s[1] = s[1] + v.y; }
    if constexpr (QT == 4) { s[0] = s[0] + v.x; s[1] = s[1] + v.y; s[2] = s[2] + v.z; s[3] = s[3] + v.w; }
}

enum { ADC_SAMPLE = 0, ADC_FILTER = 1 };

// grid (query groups, doc tiles).  SAMPLE: row n_i = floor(i*N/S), i in the tile, dense output.
// FILTER: rows of the tile, candidates with score >= thr[q] appended as keys
// (ordered(score) << 32 | ~row), so a descending key sort is (score desc, row asc).
template <int M, int QT, int MODE>
__global__ __launch_bounds__(ADC_THREADS) void adc_scan_kernel(const uint8_t* __restrict__ codes, int64_t N,
                                                               const float* __restrict__ lut, int nq, int64_t S,
                                                               float* __restrict__ sample,
                                                               const float* __restrict__ thr,
                                                               unsigned* __restrict__ cand_count,
                                                               unsigned long long* __restrict__ cand) {
    using V = typename adc_vec<QT>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    V* tab = reinterpret_cast<V*>(smem);  // [M*256]
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * QT;
    // stage the QT tables interleaved; queries past nq replicate the last valid one
    for (int i = tid; i < M * RC_K; i += ADC_THREADS) {
        float v[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int qi = (q0 + t < nq) ? q0 + t : nq - 1;
            v[t] = lut[(size_t)qi * M * RC_K + i];
        }
        if constexpr (QT == 1) tab[i] = v[0];
        if constexpr (QT == 2) tab[i] = make_float2(v[0], v[1]);
        if constexpr (QT == 4) tab[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
    float tq[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) tq[t] = (MODE == ADC_FILTER && q0 + t < nq) ? thr[q0 + t] : INFINITY;
    __syncthreads();

    const int64_t total = (MODE == ADC_SAMPLE) ? S : N;
    const int64_t t0 = (int64_t)blockIdx.y * ADC_TILE_DOCS;
    const int64_t t1 = (t0 + ADC_TILE_DOCS < total) ? t0 + ADC_TILE_DOCS : total;
    constexpr int W = (M % 16 == 0) ? 16 : (M % 8 == 0) ? 8 : 4;  // load width in bytes
    constexpr int NW = M / W;
    for (int64_t i0 = t0; i0 < t1; i0 += ADC_THREADS) {   // wave-uniform trip count
        const int64_t i = i0 + tid;
        const bool live = i < t1;
        const int64_t ii = live ? i : (t1 - 1);
        int64_t n = ii;
        if constexpr (MODE == ADC_SAMPLE) n = (int64_t)(((uint64_t)ii * (uint64_t)N) / (uint64_t)S);  // ii < 2^15, N < 2^32
        const uint8_t* cp = codes + n * M;
        unsigned w[M / 4];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            if constexpr (W == 16) {
                const uint4 v = reinterpret_cast<const uint4*>(cp)[j];
                w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
            } else if constexpr (W == 8) {
                const uint2 v = reinterpret_cast<const uint2*>(cp)[j];
                w[2 * j] = v.x; w[2 * j + 1] = v.y;
            } else {
                w[j] = reinterpret_cast<const unsigned*>(cp)[j];
            }
        }
        float s[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) s[t] = 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const unsigned c = (w[m >> 2] >> (8 * (m & 3))) & 0xFFu;
            adc_acc<QT>(s, tab[m * RC_K + c]);
        }
        if constexpr (MODE == ADC_SAMPLE) {
            if (live) {
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    if (q0 + t < nq) sample[(size_t)(q0 + t) * S + i] = s[t];
            }
        } else {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const bool pass = live && (s[t] >= tq[t]);
                const unsigned long long mask = __ballot(pass);
                if (mask) {  // wave-uniform
                    const int lane = tid & 63;
                    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                    unsigned base = 0;
                    if (lane == (int)__builtin_ctzll(mask)) base = atomicAdd(cand_count + q0 + t, (unsigned)__popcll(mask));
                    base = __shfl(base, (int)__builtin_ctzll(mask));
                    const unsigned slot = base + rank;
                    if (pass && slot < ADC_CAND_CAP)
                        cand[(size_t)(q0 + t) * ADC_CAND_CAP + slot] =
                            ((unsigned long long)adc_order_key(s[t]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)n);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ 3. threshold
// One block per query: r-th largest of S sample scores by an 8-bit-per-pass radix select on the
// order-preserving key, everything in LDS.  r <= 0 or r > S: tau = -inf (keep every row).
__global__ __launch_bounds__(1024) void adc_threshold_kernel(const float* __restrict__ sample, int64_t S, int r,
                                                             float* __restrict__ thr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* keys = reinterpret_cast<unsigned*>(smem);  // [S]
    __shared__ unsigned hist[256];
    __shared__ unsigned s_aux[8];
    __shared__ unsigned s_scan[4];
    __shared__ unsigned s_list[ADC_KTH_LIST];
    const int qi = blockIdx.x, tid = threadIdx.x;
    if (r <= 0 || r > S) {
        if (tid == 0) thr[qi] = -INFINITY;
        return;
    }
    if (tid == 0) { s_aux[2] = 0xFFFFFFFFu; s_aux[3] = 0u; s_aux[4] = 0u; }
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    unsigned mn = 0xFFFFFFFFu, mx = 0u;                       // minimum / maximum on the way into the LDS
    for (int64_t i = tid; i < S; i += 1024) {
        const unsigned k = adc_order_key(sample[(size_t)qi * S + i]);
        keys[i] = k;
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned a = (unsigned)__shfl_xor((int)mn, o), b = (unsigned)__shfl_xor((int)mx, o);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((tid & 63) == 0) { atomicMin(&s_aux[2], mn); atomicMax(&s_aux[3], mx); }
    __syncthreads();
    const unsigned kth = adc_kth_largest_v<true>([&](int64_t i) { return keys[i]; }, S, (unsigned)r, hist, s_scan, s_aux, s_list,
                                                 ADC_KTH_LIST);
    if (tid == 0) thr[qi] = adc_unorder_key(kth);
}

// ---- bitonic sort of P keys (descending) in LDS, register-blocked ----------------------------------------------------
// The plain network makes one LDS round trip (read 2, write 2 keys per pair) and one barrier per (size, stride) stage: 66
// stages for 2048 keys = 4.2 MB of LDS traffic per query — at four blocks per CU the LDS pipe, not latency, was the
// kernel's whole time (round 4 measurement: 138 us per 1200 queries with one block per CU, 150 us with four).  Here a
// work item takes the 2^NB keys that differ in NB consecutive index bits, runs the NB stages of those strides in
// registers and writes the keys back: ceil(c / 3) round trips for the c strides of a merge, and the merges of sizes 2, 4, 8
// in ONE pass: 24 round trips for 2048 keys.  Keys live at padded positions i + i / 32 so that the stride-1 / 2 / 4
// passes (a lane's keys 8, 16, 32 apart from its neighbour's) do not fall on the same banks.
__device__ __forceinline__ int adc_sp(int i) { return i + (i >> 5); }
__device__ __forceinline__ void adc_cmpx(unsigned long long& a, unsigned long long& b, bool desc) {
    const unsigned long long lo = a < b ? a : b, hi = a < b ? b : a;
    a = desc ? hi : lo;
    b = desc ? lo : hi;
}
template <int NB>
__device__ __forceinline__ void adc_bitonic_pass(unsigned long long* keys, int P, int size, int L, int tid, int nthr) {
    constexpr int NK = 1 << NB;
    const int lsh = 31 - __clz(L);
    for (int t = tid; t < (P >> NB); t += nthr) {
        const int base = ((t >> lsh) << (lsh + NB)) | (t & (L - 1));
        const bool desc = (base & size) == 0;
        unsigned long long v[NK];
#pragma unroll
        for (int j = 0; j < NK; ++j) v[j] = keys[adc_sp(base + j * L)];
#pragma unroll
        for (int b = NB - 1; b >= 0; --b)
#pragma unroll
            for (int j = 0; j < NK; ++j)
                if (!(j & (1 << b))) adc_cmpx(v[j], v[j | (1 << b)], desc);
#pragma unroll
        for (int j = 0; j < NK; ++j) keys[adc_sp(base + j * L)] = v[j];
    }
    __syncthreads();
}
// keys[adc_sp(0 .. P)) sorted descending; P a power of two >= 8; called by every thread of the block, ends in a barrier
__device__ __forceinline__ void adc_bitonic_sort_lds(unsigned long long* keys, int P, int tid, int nthr) {
    // sizes 2, 4, 8 on 8 consecutive keys
    for (int t = tid; t < (P >> 3); t += nthr) {
        const int base = t << 3;
        unsigned long long v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = keys[adc_sp(base + j)];
#pragma unroll
        for (int sz = 2; sz <= 8; sz <<= 1)
#pragma unroll
            for (int st = sz >> 1; st > 0; st >>= 1)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (!(j & st)) adc_cmpx(v[j], v[j | st], ((base + j) & sz) == 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) keys[adc_sp(base + j)] = v[j];
    }
    __syncthreads();
    for (int size = 16; size <= P; size <<= 1) {
        int c = 31 - __clz(size);                             // strides size/2 .. 1: c of them, the first chunk takes c mod 3
        int s = size >> 1;
        while (c > 0) {
            const int nb = (c % 3) ? (c % 3) : 3;
            const int L = s >> (nb - 1);
            if (nb == 3) adc_bitonic_pass<3>(keys, P, size, L, tid, nthr);
            else if (nb == 2) adc_bitonic_pass<2>(keys, P, size, L, tid, nthr);
            else adc_bitonic_pass<1>(keys, P, size, L, tid, nthr);
            c -= nb;
            s = L >> 1;
        }
    }
}

// ------------------------------------------------------------------------------------------ 5. select
// One block per query.  Sort the candidate keys descending (bitonic, LDS), emit the first k.
// status |= 1 if fewer than min(k,N) candidates were collected, |= 2 if the list overflowed.
// Round 4: the block's LDS holds `cap` keys, cap = the power of two >= max(4096, 2 k) (host, adc_select_cap): lists longer
// than max(2048, 2 k) are first cut down to the k best scores (+ every tie at the k-th score) by a radix select over the
// list in global memory (L2), so what is sorted always fits — 32 KiB and 512 threads per block at k = 1000, four blocks
// per CU, where round 3 reserved 128 KiB (one 1024-thread block per CU: 27 us of barrier-to-barrier latency per query with
// nothing to overlap it; 138 -> 60 us per 1200 queries).  Only a tie group at the k-th score that does not fit takes the
// sort in global memory (same network, same result).
__global__ __launch_bounds__(1024) void adc_select_kernel(unsigned long long* __restrict__ cand,
                                                          const unsigned* __restrict__ cand_count, int64_t N, int k,
                                                          int64_t id_offset, float* __restrict__ scores,
                                                          int64_t* __restrict__ ids, int* __restrict__ status,
                                                          int* __restrict__ qstatus, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int qi = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const unsigned raw = cand_count[qi];
    const int cnt = raw > ADC_CAND_CAP ? ADC_CAND_CAP : (int)raw;
    unsigned long long* gk = cand + (size_t)qi * ADC_CAND_CAP;
    unsigned long long* lk = reinterpret_cast<unsigned long long*>(smem);
    const int64_t want = (k < N) ? k : N;
    if (tid == 0) {
        int st = 0;
        if ((int64_t)cnt < want) st |= 1;
        if (raw > ADC_CAND_CAP) st |= 2;
        if (st) {
            atomicOr(status, st);
            if (qstatus) atomicOr(qstatus + qi, st);          // which query: the caller repeats only those
        }
    }
    __shared__ unsigned hist[256];
    __shared__ unsigned s_aux[8], survivors;
    __shared__ unsigned s_scan[4];
    int n = cnt;
    bool in_lds = cnt <= cap;                                 // block-uniform
    if ((cnt > 2048 && cnt > 2 * k) || !in_lds) {
        if (tid == 0) survivors = 0u;
        // k-th largest score key (barriers inside); the bin list borrows the (still unused) key buffer
        const unsigned kth = adc_kth_largest_v([&](int64_t i) { return (unsigned)(gk[i] >> 32); }, cnt, (unsigned)(k < cnt ? k : cnt),
                                               hist, s_scan, s_aux, reinterpret_cast<unsigned*>(lk), 2 * cap);
        __syncthreads();
        for (int i = tid; i < cnt; i += nthr) {
            const unsigned long long key = gk[i];
            if ((unsigned)(key >> 32) >= kth) {
                const unsigned slot = atomicAdd(&survivors, 1u);
                if ((int)slot < cap) lk[adc_sp((int)slot)] = key;
            }
        }
        __syncthreads();
        in_lds = (int)survivors <= cap;
        if (in_lds) n = (int)survivors;                       // >= min(k, cnt)
    }
    int P = 1024;
    while (P < n) P <<= 1;
    if (!in_lds) {
        for (int i = cnt + tid; i < P; i += nthr) gk[i] = 0ull;           // P <= ADC_CAND_CAP
        __syncthreads();
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < (P >> 1); t += nthr) {
                    const int lo = ((t / stride) * (stride << 1)) + (t % stride);
                    const int hi = lo + stride;
                    const bool desc = ((lo & size) == 0);
                    const unsigned long long a = gk[lo], b = gk[hi];
                    if ((a < b) == desc) { gk[lo] = b; gk[hi] = a; }
                }
                __syncthreads();
            }
        }
    } else {
        if (n != cnt) {
            for (int i = n + tid; i < P; i += nthr) lk[adc_sp(i)] = 0ull;
        } else {
            for (int i = tid; i < P; i += nthr) lk[adc_sp(i)] = (i < cnt) ? gk[i] : 0ull;
        }
        __syncthreads();
        adc_bitonic_sort_lds(lk, P, tid, nthr);
    }
    for (int j = tid; j < k; j += nthr) {
        float sc = -INFINITY;
        int64_t id = -1;
        if (j < n) {
            const unsigned long long key = in_lds ? lk[adc_sp(j)] : gk[j];
            sc = adc_unorder_key((unsigned)(key >> 32));
            id = (int64_t)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) + id_offset;
        }
        scores[(size_t)qi * k + j] = sc;
        ids[(size_t)qi * k + j] = id;
    }
}

// ------------------------------------------------------------------------------------------ 4b. screening
#define ADC_SCREEN_MIN_N (1 << 18)
#define ADC_ID_CAP 32768

// One block per query: per-m minimum, the common step Delta = max_m(range_m)/255, the integer threshold and
// the byte tables, written interleaved [group][m][c][QS] (group = query / QS) so the screen kernel copies one
// contiguous slab per block.
__global__ __launch_bounds__(RC_K) void adc_qlut_kernel(const float* __restrict__ lut, const float* __restrict__ thr,
                                                        int M, int QS, uint8_t* __restrict__ qlut,
                                                        int* __restrict__ tint) {
    __shared__ float lo_m[128];
    __shared__ float red_lo[4], red_hi[4];
    __shared__ float s_delta;
    const int qi = blockIdx.x, c = threadIdx.x;
    const float* lq = lut + (size_t)qi * M * RC_K;
    float maxrange = 0.f;
    for (int m = 0; m < M; ++m) {
        const float v = lq[m * RC_K + c];
        float lo = v, hi = v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o));
            hi = fmaxf(hi, __shfl_xor(hi, o));
        }
        if ((c & 63) == 0) { red_lo[c >> 6] = lo; red_hi[c >> 6] = hi; }
        __syncthreads();
        lo = fminf(fminf(red_lo[0], red_lo[1]), fminf(red_lo[2], red_lo[3]));
        hi = fmaxf(fmaxf(red_hi[0], red_hi[1]), fmaxf(red_hi[2], red_hi[3]));
        if (c == 0) lo_m[m] = lo;
        maxrange = fmaxf(maxrange, hi - lo);
        __syncthreads();
    }
    if (c == 0) {
        float delta = maxrange / 255.0f;
        if (!(delta > 0.f)) delta = 1.0f;
        s_delta = delta;
        double A = 0.0;
        for (int m = 0; m < M; ++m) A += (double)lo_m[m];
        const float t = thr[qi];
        int T;
        if (t == -INFINITY) {
            T = INT_MIN;
        } else {
            const double v = ceil(((double)t - A) / (double)delta - 0.5 * (double)M) - 2.0;   // entries rounded to NEAREST: |error| <= 1/2 each
            T = v < -2.0e9 ? INT_MIN : (v > 2.0e9 ? INT_MAX : (int)v);
        }
        tint[qi] = T;
    }
    __syncthreads();
    const float delta = s_delta;
    uint8_t* dst = qlut + (size_t)(qi / QS) * M * RC_K * QS + (qi % QS);
    for (int m = 0; m < M; ++m) {
        const float v = (lq[m * RC_K + c] - lo_m[m]) / delta;
        int l = (int)floorf(v + 0.5f);
        l = l < 0 ? 0 : (l > 255 ? 255 : l);
        dst[((size_t)m * RC_K + c) * QS] = (uint8_t)l;
    }
}

// grid (query groups of QS, doc tiles).  LDS: [M][256] entries of QS bytes (uint2 for QS = 8, uint for 4).
template <int M, int QS>
__global__ __launch_bounds__(ADC_THREADS) void adc_screen_kernel(const uint8_t* __restrict__ codes, int64_t N,
                                                                 const uint8_t* __restrict__ qlut,
                                                                 const int* __restrict__ tint, int nq,
                                                                 unsigned* __restrict__ id_count,
                                                                 unsigned* __restrict__ ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using E = typename std::conditional<QS == 8, uint2, unsigned>::type;
    E* tab = reinterpret_cast<E*>(smem);
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * QS;
    {
        const uint4* src = reinterpret_cast<const uint4*>(qlut + (size_t)blockIdx.x * M * RC_K * QS);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < M * RC_K * QS / 16; i += ADC_THREADS) dst[i] = src[i];
    }
    int tq[QS];
#pragma unroll
    for (int t = 0; t < QS; ++t) tq[t] = (q0 + t < nq) ? tint[q0 + t] : INT_MAX;
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.y * ADC_TILE_DOCS;
    const int64_t t1 = (t0 + ADC_TILE_DOCS < N) ? t0 + ADC_TILE_DOCS : N;
    constexpr int W = (M % 16 == 0) ? 16 : (M % 8 == 0) ? 8 : 4;
    constexpr int NW = M / W;
    // the row of the NEXT trip is loaded before the current one is scored (register double buffer), so the
    // code loads (L2 latency) overlap the 48 gathers of the current row
    auto load_row = [&](int64_t n, unsigned (&dst)[M / 4]) {
        const uint8_t* cp = codes + (n < t1 ? n : (t1 - 1)) * M;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            if constexpr (W == 16) {
                const uint4 v = reinterpret_cast<const uint4*>(cp)[j];
                dst[4 * j] = v.x; dst[4 * j + 1] = v.y; dst[4 * j + 2] = v.z; dst[4 * j + 3] = v.w;
            } else if constexpr (W == 8) {
                const uint2 v = reinterpret_cast<const uint2*>(cp)[j];
                dst[2 * j] = v.x; dst[2 * j + 1] = v.y;
            } else {
                dst[j] = reinterpret_cast<const unsigned*>(cp)[j];
            }
        }
    };
    unsigned w[M / 4], wn[M / 4];
    load_row(t0 + tid, w);
    for (int64_t i0 = t0; i0 < t1; i0 += ADC_THREADS) {
        const int64_t n = i0 + tid;
        const bool live = n < t1;
        load_row(n + ADC_THREADS, wn);
        // Byte accumulation in packed 16-bit fields: one v_perm_b32 spreads bytes (0,2) of a gathered word into
        // the two halves of a dword, another bytes (1,3); plain 32-bit adds then accumulate two queries at
        // once (a field never exceeds 96*255 < 2^16, so no carry crosses).  4 full-rate VALU ops per 4 queries —
        // v_dot4_u32_u8 with a one-hot mask does one query per instruction at half rate (PMC: VALU-bound at
        // 11.5 instr/gather, 4 cycles each).
        unsigned pe[QS / 4], po[QS / 4];
#pragma unroll
        for (int u = 0; u < QS / 4; ++u) pe[u] = po[u] = 0u;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const unsigned c = (w[m >> 2] >> (8 * (m & 3))) & 0xFFu;
            const E v = tab[m * RC_K + c];
            if constexpr (QS == 8) {
                pe[0] += __builtin_amdgcn_perm(0u, v.x, 0x0C020C00u);
                po[0] += __builtin_amdgcn_perm(0u, v.x, 0x0C030C01u);
                pe[1] += __builtin_amdgcn_perm(0u, v.y, 0x0C020C00u);
                po[1] += __builtin_amdgcn_perm(0u, v.y, 0x0C030C01u);
            } else {
                pe[0] += __builtin_amdgcn_perm(0u, v, 0x0C020C00u);
                po[0] += __builtin_amdgcn_perm(0u, v, 0x0C030C01u);
            }
        }
        int acc[QS];   // query t = 4u + j: byte j of word u -> (j even ? pe : po)[u], field j/2
#pragma unroll
        for (int u = 0; u < QS / 4; ++u) {
            acc[4 * u + 0] = (int)(pe[u] & 0xFFFFu);
            acc[4 * u + 1] = (int)(po[u] & 0xFFFFu);
            acc[4 * u + 2] = (int)(pe[u] >> 16);
            acc[4 * u + 3] = (int)(po[u] >> 16);
        }
#pragma unroll
        for (int t = 0; t < QS; ++t) {
            const bool pass = live && (acc[t] >= tq[t]);
            const unsigned long long mask = __ballot(pass);
            if (mask) {  // wave-uniform
                const int lane = tid & 63;
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                unsigned base = 0;
                if (lane == (int)__builtin_ctzll(mask)) base = atomicAdd(id_count + q0 + t, (unsigned)__popcll(mask));
                base = __shfl(base, (int)__builtin_ctzll(mask));
                const unsigned slot = base + rank;
                if (pass && slot < ADC_ID_CAP) ids[(size_t)(q0 + t) * ADC_ID_CAP + slot] = (unsigned)n;
            }
        }
#pragma unroll
        for (int j = 0; j < M / 4; ++j) w[j] = wn[j];
    }
}

// XCD-aware block remap (guide T1).  The dispatcher places workgroup b on XCD b % 8 and every XCD has a private L2; the
// grid is (query groups, row tiles) with the group index fastest, so by default the ~150 blocks that scan the same
// row tile land on all 8 XCDs and the tile is fetched from HBM once per XCD.  The bijective remap hands each XCD a
// contiguous range of (tile, group) pairs: all groups of a tile run on ONE XCD, whose L2 (4 MiB) keeps the 1.5 MiB tile.
__device__ __forceinline__ void adc_xcd_remap(unsigned& group, unsigned& tile) {
    const unsigned gx = gridDim.x, total = gx * gridDim.y;
    const unsigned lin = blockIdx.y * gx + blockIdx.x;
    const unsigned q = total / 8u, r = total % 8u, xcd = lin % 8u;
    const unsigned virt = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + lin / 8u;
    group = virt % gx;
    tile = virt / gx;
}

// The same screen with the byte accumulation on the matrix cores (M % 8 == 0; 8 queries per group, 4 when M > 64).
// PMC on adc_screen_kernel<48,8>: 9.3e9 VALU instructions per 1200-query launch = 95 % of the kernel's VALU cycles,
// the LDS gathers active 12 of its 18 ms — the v_perm/v_add accumulation is the limiter.  Here a wave takes 32 rows;
// lanes l and l+32 share row (l & 31) and gather the two halves of its M codes.  Two gathered 8-byte entries (two
// sub-quantisers x 8 queries) ARE the 16-byte A operand of v_mfma_i32_32x32x32_i8 in lane-natural layout
// (A[row][t], t = 8 g + query); B is the constant selection matrix B[t][j] = [t % 8 == j], so
//     D[row][j] += sum_{g, half} entry_{g,half}[j]      — one MFMA folds 4 sub-quantisers of 32 rows x 8 queries
// (QS = 4: four 4-byte entries per lane, 8 sub-quantisers x 4 queries).
// The pairing of A and B bytes is by (half-wave, byte position), so it does not depend on how the hardware numbers k.
// VALU per gather drops from ~10 instructions to the address computation; results arrive as D: lane j (< 8) of each
// half-wave holds query j's sums for 16 rows (row = (r&3) + 8 (r>>2) + 4 (lane>>5)).  The MFMA is signed: bytes are
// staged as l - 128 and the integer threshold is lowered by 128 M.
typedef int adc_i32x4 __attribute__((ext_vector_type(4)));
typedef int adc_i32x16 __attribute__((ext_vector_type(16)));

template <int M, int QS>
__global__ __launch_bounds__(ADC_THREADS) void adc_screen_mfma_kernel(const uint8_t* __restrict__ codes, int64_t N,
                                                                      const uint8_t* __restrict__ qlut,
                                                                      const int* __restrict__ tint, int nq,
                                                                      unsigned* __restrict__ id_count,
                                                                      unsigned* __restrict__ ids) {
    constexpr int HM = M / 2, NW = HM / 4;                  // codes per half-wave, dwords of codes per lane
    constexpr int G = 16 / QS;                              // gathered entries per A operand (QS bytes each)
    static_assert((QS == 8 || QS == 4) && HM % G == 0 && HM % 4 == 0, "unsupported (M, QS)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    unsigned bgroup, btile;
    adc_xcd_remap(bgroup, btile);
    const int q0 = (int)bgroup * QS;
    {
        const uint4* src = reinterpret_cast<const uint4*>(qlut + (size_t)bgroup * M * RC_K * QS);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < M * RC_K * QS / 16; i += ADC_THREADS) {
            uint4 v = src[i];
            v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;   // l -> l - 128 (signed)
            dst[i] = v;
        }
    }
    const int l = tid & 63, wv = tid >> 6;
    const int d = l & 31, hh = l >> 5;
    int tq = INT_MAX;                                        // this lane's query (as the D column j = d)
    if (d < QS && q0 + d < nq) {
        const int t = tint[q0 + d];
        tq = (t == INT_MIN) ? INT_MIN : t - 128 * M;
    }
    adc_i32x4 bsel = {0, 0, 0, 0};                           // B[t][j = d] = [t % QS == d]
    if (d < QS) {
        const int one = 1 << (8 * (d & 3));
        if constexpr (QS == 8) {
            bsel[d >> 2] = one;
            bsel[2 + (d >> 2)] = one;
        } else {
            bsel[0] = bsel[1] = bsel[2] = bsel[3] = one;
        }
    }
    __syncthreads();
    const int64_t t0 = (int64_t)btile * ADC_TILE_DOCS;
    const int64_t t1 = (t0 + ADC_TILE_DOCS < N) ? t0 + ADC_TILE_DOCS : N;
    constexpr int NWAVES = ADC_THREADS / 64;
    const unsigned char* tabh = smem + (size_t)hh * HM * RC_K * QS;      // this half-wave's sub-quantisers
    auto load_row = [&](int64_t n, unsigned (&dst)[NW]) {
        const uint8_t* cp = codes + (n < t1 ? n : (t1 - 1)) * M + hh * HM;
        if constexpr (HM % 16 == 0) {
#pragma unroll
            for (int j = 0; j < HM / 16; ++j) {
                const uint4 v = reinterpret_cast<const uint4*>(cp)[j];
                dst[4 * j] = v.x; dst[4 * j + 1] = v.y; dst[4 * j + 2] = v.z; dst[4 * j + 3] = v.w;
            }
        } else if constexpr (HM % 8 == 0) {
#pragma unroll
            for (int j = 0; j < HM / 8; ++j) {
                const uint2 v = reinterpret_cast<const uint2*>(cp)[j];
                dst[2 * j] = v.x; dst[2 * j + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NW; ++j) dst[j] = reinterpret_cast<const unsigned*>(cp)[j];
        }
    };
    unsigned w[NW], wn[NW];
    load_row(t0 + wv * 32 + d, w);
    for (int64_t i0 = t0 + wv * 32; i0 < t1; i0 += NWAVES * 32) {   // wave-uniform
        load_row(i0 + NWAVES * 32 + d, wn);
        adc_i32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int n = 0; n < HM / G; ++n) {
            adc_i32x4 a;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int mm = G * n + g;
                const unsigned c = (w[mm >> 2] >> (8 * (mm & 3))) & 0xFFu;
                const unsigned char* e = tabh + ((size_t)mm * RC_K + c) * QS;
                if constexpr (QS == 8) {
                    const uint2 v = *reinterpret_cast<const uint2*>(e);
                    a[2 * g] = (int)v.x;
                    a[2 * g + 1] = (int)v.y;
                } else {
                    a[g] = (int)*reinterpret_cast<const unsigned*>(e);
                }
            }
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bsel, acc, 0, 0, 0);
        }
        bool any = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) any |= (acc[r] >= tq);
        if (__ballot(any)) {                                  // rare: ~2e-4 of the (row, query) pairs pass
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t n = i0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (acc[r] >= tq && n < t1) {
                    const unsigned slot = atomicAdd(id_count + q0 + d, 1u);
                    if (slot < ADC_ID_CAP) ids[(size_t)(q0 + d) * ADC_ID_CAP + slot] = (unsigned)n;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) w[j] = wn[j];
    }
}

// Phased variant: more queries per gather than the LDS can hold tables for.  LDS keeps the byte tables of M / NP
// sub-quantisers at a time; a wave keeps the partial sums of R = 4 chunks of 32 rows in accumulator registers across the
// table swaps, and consecutive rounds visit the phases in alternating direction (0..NP-1 | NP-1..0 | ...) so the table
// already in LDS is reused: NP - 1 refills per round of 16 waves x R x 32 = 2048 rows.
//   QS = 16 (one ds_read_b128 = 16 queries = the whole A operand; B[t][j] = [t == j]): M <= 64 with NP = 2 — half the LDS
//            instructions per query of the one-pass 8-query kernel;
//   QS = 8, NP = 2: M = 96 (its 8-query tables are 196 KiB; the one-pass kernel had to fall back to 4 queries per gather).
// Same arithmetic, threshold and candidate list as adc_screen_mfma_kernel.
template <int M, int QS, int NP>
__global__ __launch_bounds__(ADC_THREADS) void adc_screen_mfma2_kernel(const uint8_t* __restrict__ codes, int64_t N,
                                                                       const uint8_t* __restrict__ qlut,
                                                                       const int* __restrict__ tint, int nq,
                                                                       unsigned* __restrict__ id_count,
                                                                       unsigned* __restrict__ ids) {
    constexpr int PM = M / NP, HM = PM / 2, R = 4;           // sub-quantisers per phase, per half-wave; chunks per wave
    constexpr int G = 16 / QS;                               // gathers per A operand
    static_assert((QS == 8 || QS == 16) && M % NP == 0 && PM % 2 == 0 && HM % G == 0 && HM % 4 == 0, "unsupported (M, QS, NP)");
    constexpr int NWAVES = ADC_THREADS / 64;
    constexpr int ROUND = NWAVES * R * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    unsigned bgroup, btile;
    adc_xcd_remap(bgroup, btile);
    const int q0 = (int)bgroup * QS;
    const uint8_t* qsrc = qlut + (size_t)bgroup * M * RC_K * QS;
    auto fill = [&](int phase) {
        const uint4* src = reinterpret_cast<const uint4*>(qsrc + (size_t)phase * PM * RC_K * QS);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < PM * RC_K * QS / 16; i += ADC_THREADS) {
            uint4 v = src[i];
            v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;   // l -> l - 128 (signed)
            dst[i] = v;
        }
    };
    const int l = tid & 63, wv = tid >> 6;
    const int d = l & 31, hh = l >> 5;
    int tq = INT_MAX;
    if (d < QS && q0 + d < nq) {
        const int t = tint[q0 + d];
        tq = (t == INT_MIN) ? INT_MIN : t - 128 * M;
    }
    adc_i32x4 bsel = {0, 0, 0, 0};                           // B[t][j = d] = [t % QS == d]
    if (d < QS) {
        const int one = 1 << (8 * (d & 3));
        bsel[d >> 2] = one;
        if constexpr (QS == 8) bsel[2 + (d >> 2)] = one;
    }
    const int64_t t0 = (int64_t)btile * ADC_TILE_DOCS;
    const int64_t t1 = (t0 + ADC_TILE_DOCS < N) ? t0 + ADC_TILE_DOCS : N;
    const unsigned char* tabh = smem + (size_t)hh * HM * RC_K * QS;
    int in_lds = -1;
    bool forward = true;
    for (int64_t r0 = t0; r0 < t1; r0 += ROUND) {            // block-uniform
        adc_i32x16 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = adc_i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int step = 0; step < NP; ++step) {
            const int phase = forward ? step : NP - 1 - step;
            if (in_lds != phase) {
                __syncthreads();                             // every wave is done gathering from the old tables
                fill(phase);
                __syncthreads();
                in_lds = phase;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t i0 = r0 + (int64_t)(wv * R + r) * 32;
                if (i0 < t1) {                                // wave-uniform
                    const int64_t n = i0 + d;
                    const uint8_t* cp = codes + (n < t1 ? n : (t1 - 1)) * M + phase * PM + hh * HM;
                    unsigned char cb[HM];                     // this lane's codes of the phase (HM bytes, 4-byte aligned)
                    if constexpr (HM % 8 == 0) {
#pragma unroll
                        for (int j = 0; j < HM / 8; ++j) {
                            const uint2 v = reinterpret_cast<const uint2*>(cp)[j];
#pragma unroll
                            for (int b = 0; b < 4; ++b) { cb[8 * j + b] = (v.x >> (8 * b)) & 0xFFu; cb[8 * j + 4 + b] = (v.y >> (8 * b)) & 0xFFu; }
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < HM / 4; ++j) {
                            const unsigned v = reinterpret_cast<const unsigned*>(cp)[j];
#pragma unroll
                            for (int b = 0; b < 4; ++b) cb[4 * j + b] = (v >> (8 * b)) & 0xFFu;
                        }
                    }
#pragma unroll
                    for (int g2 = 0; g2 < HM / G; ++g2) {
                        adc_i32x4 a;
                        if constexpr (QS == 16) {
                            const uint4 e = *reinterpret_cast<const uint4*>(tabh + ((size_t)g2 * RC_K + cb[g2]) * QS);
                            a = adc_i32x4{(int)e.x, (int)e.y, (int)e.z, (int)e.w};
                        } else {
                            const int ma = 2 * g2, mb = 2 * g2 + 1;
                            const uint2 ea = *reinterpret_cast<const uint2*>(tabh + ((size_t)ma * RC_K + cb[ma]) * QS);
                            const uint2 eb = *reinterpret_cast<const uint2*>(tabh + ((size_t)mb * RC_K + cb[mb]) * QS);
                            a = adc_i32x4{(int)ea.x, (int)ea.y, (int)eb.x, (int)eb.y};
                        }
                        acc[r] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bsel, acc[r], 0, 0, 0);
                    }
                }
            }
        }
        forward = !forward;                                   // the tables now in LDS go first next round
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i0 = r0 + (int64_t)(wv * R + r) * 32;
            if (i0 < t1) {
                bool any = false;
#pragma unroll
                for (int e = 0; e < 16; ++e) any |= (acc[r][e] >= tq);
                if (__ballot(any)) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int64_t n = i0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        if (acc[r][e] >= tq && n < t1) {
                            const unsigned slot = atomicAdd(id_count + q0 + d, 1u);
                            if (slot < ADC_ID_CAP) ids[(size_t)(q0 + d) * ADC_ID_CAP + slot] = (unsigned)n;
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------- 4b'. conflict-free screen
// Round-1 PMC on adc_screen_mfma_kernel<48,8>: 68 % of the LDS cycles of its random 8-byte gathers were bank conflicts
// (6.3 cycles per ds_read_b64 instead of 2) and the selection-matrix MFMA (32x32x32, 8 useful columns of 32) kept the
// matrix pipe busy half of the time.  Both go away with two observations:
//
//  * the sum over sub-quantisers is commutative (integer adds), so the lanes of a wave need not visit the sub-quantisers
//    in the same order.  The byte tables are laid out [code][slot][8 queries] with one 8-byte SLOT per sub-quantiser —
//    a slot's LDS bank pair is slot mod 32 whatever the code — and in every step the 32 lanes that the LDS services
//    together (a ds_read_b64 is processed as lanes 0-31, then 32-63) read 32 DIFFERENT slots mod 32: lane (r, g) of a
//    16-row chunk (r = row, g = lane quarter) walks block-relative sub-quantiser (r + (S/4) tau(g) + j) mod S in step j,
//    S = 32 or 16 = size of the block of sub-quantisers, tau(g) = 2 (g & 1) + (g >> 1); a 16-block is stored twice
//    (slots 16 apart) and lanes 16-31 of the group use the second copy.  The gathers are conflict-free BY CONSTRUCTION,
//    for any codes.  The lane's codes must then arrive in its own visiting order: the index keeps a second, permuted
//    image of the code matrix (rc_adc_scan_image; the permutation of row n depends on n mod 16 only), 1 byte per code.
//  * v_mfma_i32_16x16x64_i8 takes 16 rows x 64 k-bytes (8 gathered entries per row) per 16 cycles instead of 32 rows x 32
//    k-bytes (4 entries) per 32: twice the entries per matrix-pipe cycle for the same selection-matrix trick.
//
// Supported: sub-quantisers per table phase PM = 16, 32, 48, 64 (M = 96 runs two phases of 48, accumulators of 8 chunks
// kept in registers across the table swap, phases visited in alternating order).  Everything downstream (rigorous integer
// threshold, exact fp32 rescoring of the survivors from the canonical codes) is unchanged, so results stay bit-identical.
template <int PM>
struct adc_cf {
    static_assert(PM % 16 == 0 && PM >= 16 && PM <= 64, "table phase of 16/32/48/64 sub-quantisers");
    static constexpr int N32 = PM / 32, HAS16 = (PM % 32) / 16;
    static constexpr int SLOTS = 32 * (N32 + HAS16);       // 8-byte slots per code: LDS row of SLOTS * 8 bytes
    static constexpr int STEPS = PM / 4;                   // gathers per lane per 16-row chunk
    static constexpr int TABLE_BYTES = RC_K * SLOTS * 8;
};
// step s (0 .. PM/4-1) of a lane -> size of the block of sub-quantisers it falls in, the block's first sub-quantiser
// (= its first slot) and the step index inside the block.  32-blocks first, then the 16-block.
__host__ __device__ constexpr int adc_cf_bsize(int PM, int s) { return s < 8 * (PM / 32) ? 32 : 16; }
__host__ __device__ constexpr int adc_cf_bbase(int PM, int s) { return s < 8 * (PM / 32) ? 32 * (s / 8) : 32 * (PM / 32); }
__host__ __device__ constexpr int adc_cf_bstep(int PM, int s) { return s < 8 * (PM / 32) ? s % 8 : s - 8 * (PM / 32); }
// block-relative sub-quantiser that lane (r, g) reads in step j of a block of size S
__host__ __device__ inline int adc_cf_mloc(int S, int j, int r, int g) {
    return (r + (S >> 2) * (2 * (g & 1) + (g >> 1)) + j) & (S - 1);
}
// slot (within the phase's table) and sub-quantiser (within the phase) of step s for lane (r, g)
__host__ __device__ inline void adc_cf_step(int PM, int s, int r, int g, int& slot, int& m) {
    const int S = adc_cf_bsize(PM, s), base = adc_cf_bbase(PM, s), j = adc_cf_bstep(PM, s);
    const int ml = adc_cf_mloc(S, j, r, g);
    const int lam = r + 16 * (g & 1);                      // lane index inside the 32 lanes the LDS services together
    m = base + ml;
    slot = base + ml + S * (lam / S);                      // S = 16: second copy for lanes 16-31
}

// image[n][phase][g][s] = codes[n][phase * PM + m(s; n mod 16, g)] for rows n0 <= n < n0 + cnt.
// tile_rows > 0 (flat-search image of a two-phase M, round 3): the image is stored tile by tile, PHASE-MAJOR inside a tile of
// tile_rows rows — [n / T][phase][n % T][PM] — so that a pass over one phase streams dense PM-byte rows (with 96-byte
// rows a wave's 16-row code load touches twelve half-used cache lines instead of six full ones).  The layout does not
// depend on the capacity of the buffer, so rows can still be appended; the buffer holds whole tiles.
__global__ __launch_bounds__(256) void adc_scan_image_kernel(const uint8_t* __restrict__ codes, int64_t n0, int64_t cnt,
                                                             int M, int PM, uint8_t* __restrict__ image, int64_t tile_rows) {
    const int64_t total = cnt * M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = n0 + i / M;
        const int pos = (int)(i % M);
        const int phase = pos / PM, rem = pos % PM;
        const int g = rem / (PM / 4), st = rem % (PM / 4);
        int slot, m;
        adc_cf_step(PM, st, (int)(n & 15), g, slot, m);
        const uint8_t c = codes[n * M + phase * PM + m];
        const int64_t at = tile_rows > 0 ? ((n / tile_rows) * (M / PM) + phase) * tile_rows * PM + (n % tile_rows) * PM + rem
                                         : n * M + pos;
        if (ADC_IMG_ES == 2) reinterpret_cast<uint16_t*>(image)[at] = c;
        else image[at] = c;
    }
}

// Quantisation of a query's tables to 8 bits, split in two kernels (round 2; the one-kernel form spent 0.31 ms per
// 1200 queries in 96 block barriers and 19.6 M single-byte stores):
//   adc_qstats_kernel        per query: lo[m] = min_k LUT[m][k], delta = max_m range / 255, integer threshold  (3 barriers)
//   adc_qlut_cf_write_kernel byte tables of the conflict-free screen, [group of 8 queries][phase][code][slot][8], every
//                            copy of a 16-block filled; thread = code, 16-byte stores (two slots x 8 queries)
//   adc_qbyte_write_kernel   the IVF form: one table per QUERY, [phase][code][slot] one byte per entry
// The arithmetic per entry is exactly adc_qlut_kernel's: floor((v - lo_m) / delta + 0.5) clamped to [0, 255].
#define ADC_QSTAT_STRIDE 128          // floats per query: lo[0..M), delta at [127]
__global__ __launch_bounds__(RC_K) void adc_qstats_kernel(const float* __restrict__ lut, const float* __restrict__ thr,
                                                          int M, float* __restrict__ qstat, int* __restrict__ tint) {
    __shared__ float s_lo[ADC_QSTAT_STRIDE];
    __shared__ float s_rng[ADC_QSTAT_STRIDE];
    const int qi = blockIdx.x, c = threadIdx.x, lane = c & 63, wv = c >> 6;
    const float* lq = lut + (size_t)qi * M * RC_K;
    // a wave owns sub-quantisers wv, wv + 4, ...: four codes per lane, one wave reduction per sub-quantiser
    for (int m = wv; m < M; m += 4) {
        const float4 v = reinterpret_cast<const float4*>(lq + (size_t)m * RC_K)[lane];
        float lo = fminf(fminf(v.x, v.y), fminf(v.z, v.w)), hi = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o));
            hi = fmaxf(hi, __shfl_xor(hi, o));
        }
        if (lane == 0) {
            qstat[(size_t)qi * ADC_QSTAT_STRIDE + m] = lo;
            s_lo[m] = lo;
            s_rng[m] = hi - lo;
        }
    }
    __syncthreads();
    if (c == 0) {
        float maxrange = 0.f;
        double A = 0.0;
        for (int m = 0; m < M; ++m) {
            maxrange = fmaxf(maxrange, s_rng[m]);
            A += (double)s_lo[m];
        }
        float delta = maxrange / 255.0f;
        if (!(delta > 0.f)) delta = 1.0f;
        qstat[(size_t)qi * ADC_QSTAT_STRIDE + ADC_QSTAT_STRIDE - 1] = delta;
        const float t = thr[qi];
        int T;
        if (t == -INFINITY) {
            T = INT_MIN;
        } else {
            const double v = ceil(((double)t - A) / (double)delta - 0.5 * (double)M) - 2.0;   // entries rounded to NEAREST: |error| <= 1/2 each
            T = v < -2.0e9 ? INT_MIN : (v > 2.0e9 ? INT_MAX : (int)v);
        }
        tint[qi] = T;
    }
}

__device__ __forceinline__ unsigned adc_quant8(float v, float lo, float delta) {
    int l = (int)floorf((v - lo) / delta + 0.5f);           // nearest: the screen's one-sided slack is M / 2 + 2 steps, not M + 2
    l = l < 0 ? 0 : (l > 255 ? 255 : l);
    return (unsigned)l;
}

template <int PM>
__global__ __launch_bounds__(64) void adc_qlut_cf_write_kernel(const float* __restrict__ lut, const float* __restrict__ qstat,
                                                               int M, int nq, uint8_t* __restrict__ qlut) {
    constexpr int SLOTS = adc_cf<PM>::SLOTS;
    const int g = blockIdx.x, c = blockIdx.y * 64 + threadIdx.x;
    const int NP = M / PM;
    const int nv = (nq - 8 * g) < 8 ? (nq - 8 * g) : 8;       // valid queries of the group (block-uniform)
    float delta[8];
#pragma unroll
    for (int qq = 0; qq < 8; ++qq)
        delta[qq] = qq < nv ? qstat[(size_t)(8 * g + qq) * ADC_QSTAT_STRIDE + ADC_QSTAT_STRIDE - 1] : 1.0f;
    for (int phase = 0; phase < NP; ++phase) {
        uint4* row = reinterpret_cast<uint4*>(qlut + (((size_t)g * NP + phase) * RC_K + c) * SLOTS * 8);
        for (int s2 = blockIdx.z; s2 < SLOTS / 2; s2 += gridDim.z) {     // slot pairs are dealt over grid.z
            unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int sl = 2 * s2 + h2;
                const int mp = sl < PM ? sl : sl - 16;           // second copy of the 16-block
                const int m = phase * PM + mp;
#pragma unroll
                for (int qq = 0; qq < 8; ++qq) {
                    if (qq < nv) {
                        const int q = 8 * g + qq;
                        const unsigned l = adc_quant8(lut[((size_t)q * M + m) * RC_K + c],
                                                      qstat[(size_t)q * ADC_QSTAT_STRIDE + m], delta[qq]);
                        w[2 * h2 + (qq >> 2)] |= l << (8 * (qq & 3));
                    }
                }
            }
            row[s2] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

template <int PM>
__global__ __launch_bounds__(RC_K) void adc_qbyte_write_kernel(const float* __restrict__ lut, const float* __restrict__ qstat,
                                                               int M, uint8_t* __restrict__ qbyte) {
    // compact rows: [phase][code][PM] bytes, one per sub-quantiser (the second copy of a 16-block exists only in LDS)
    const int qi = blockIdx.x, c = threadIdx.x;
    const int NP = M / PM;
    const float* lq = lut + (size_t)qi * M * RC_K;
    const float* st = qstat + (size_t)qi * ADC_QSTAT_STRIDE;
    const float delta = st[ADC_QSTAT_STRIDE - 1];
    for (int phase = 0; phase < NP; ++phase) {
        uint4* row = reinterpret_cast<uint4*>(qbyte + (((size_t)qi * NP + phase) * RC_K + c) * PM);
#pragma unroll
        for (int s16 = 0; s16 < PM / 16; ++s16) {
            unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int m = phase * PM + 16 * s16 + j;
                w[j >> 2] |= adc_quant8(lq[m * RC_K + c], st[m], delta) << (8 * (j & 3));
            }
            row[s16] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

typedef int adc_i32x4v __attribute__((ext_vector_type(4)));
typedef unsigned adc_u32x2v __attribute__((ext_vector_type(2)));

// grid (groups of 8 queries, row tiles of adc_cf_tile_rows(M) rows), XCD-remapped like the other screens.
// A wave owns R chunks of 16 rows per round; lanes (r = l & 15, g = l >> 4).
// IVF mode (list-centric scan of csrc/ivf_search.hip's index): a block is a TASK = (coarse cell, up to 8 of the queries that
// probe it); its rows are the cell's row range, its byte tables are transposed on the fly from the per-query tables
// (adc_qstats_kernel + adc_qbyte_write_kernel), its thresholds are those of its queries.
struct adc_ivf_tasks {
    const int* task_list;        // [tasks] cell of the task
    const int* task_qstart;      // [tasks] first entry of the task's queries in sorted_q
    const int* task_qcnt;        // [tasks] 1 .. 8 queries
    const int* sorted_q;         // query ids ordered by probed cell
    const int64_t* list_off;     // [nlist + 1] row ranges of the cells
    const uint8_t* qbyte;        // [nq][NP][256][PM] per-query byte tables, one byte per sub-quantiser
    const int* ntasks;           // device-side task count when the list is padded (rc_ivf_search_probes), else NULL
};

// Two-pass form of a two-phase screen (M = 96; round 3).  The one-launch form keeps a round's accumulators in registers
// across the table swap and pays two block-wide barriers + two synchronous 128 KiB refills per 2048-row round: 27 ms per
// 1200 queries against 2 x 10 ms of gathers.  PART = 1 / 2 run ONE phase each over the whole index with the tables resident
// (no barrier, no refill — the M = 48 kernel's schedule): pass 1 writes every (row, query) partial sum as an int16
// (|sum of 48 biased bytes| <= 6144) to HBM, pass 2 adds it to its own sum before the threshold test.  The partial sums are
// a pure stream (written once, read once, non-temporal): 2 x 2 bytes per (row, query) = 42 GB per 1200-query batch over
// ~20 ms, ~2 TB/s of an otherwise idle HBM.  Layout: [group][chunk of 16 rows][lane quarter g][column r < 8][4 rows] int16,
// i.e. the accumulator registers as they are: 512 contiguous bytes per wave and chunk, 8-byte stores.
struct adc_part_args {
    short* buf;            // partial sums of this launch's groups
    unsigned group0;       // first 8-query group of this launch (tables / thresholds are indexed by group0 + block group)
    unsigned nchunks;      // 16-row chunks per group in `buf` (whole tiles)
};

template <int M, int NP, int R, bool IVF = false, int THREADS = ADC_THREADS, int PART = 0>
__global__ __launch_bounds__(THREADS, 4) void adc_screen_cf_kernel(const uint8_t* __restrict__ image, int64_t N,
                                                                    const uint8_t* __restrict__ qlut,
                                                                    const int* __restrict__ tint, int nq,
                                                                    unsigned* __restrict__ id_count,
                                                                    unsigned* __restrict__ ids, adc_ivf_tasks T,
                                                                    adc_part_args PA) {
    static_assert(PART == 0 || (NP == 2 && !IVF), "two-pass form: flat search with two table phases");
    constexpr int NPE = PART ? 1 : NP;                     // table phases visited per round by THIS launch
    constexpr int PM = M / NP;
    using L = adc_cf<PM>;
    constexpr int STEPS = L::STEPS, NW = STEPS * ADC_IMG_ES / 4;   // code dwords per lane per chunk and phase
    static_assert(STEPS % 4 == 0, "whole dwords of codes per lane");
    constexpr int NWAVES = THREADS / 64;
    constexpr int ROUND = NWAVES * R * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    unsigned bgroup = 0, btile = 0;
    if constexpr (!IVF) adc_xcd_remap(bgroup, btile);
    const unsigned lgroup = bgroup;                          // group inside this launch (partial-sum buffer)
    if constexpr (PART != 0) bgroup += PA.group0;
    const int q0 = (int)bgroup * 8;
    const uint8_t* qsrc = qlut + (size_t)bgroup * NP * L::TABLE_BYTES;
    // IVF: blocks are dealt to the XCDs round-robin; give every XCD a CONTIGUOUS range of the (cell-ordered) task list so
    // that the tasks of one cell run on one XCD, close in time, and share its rows in that L2
    unsigned task = blockIdx.x;
    if constexpr (IVF) {
        const unsigned total = T.ntasks ? (unsigned)*T.ntasks : gridDim.x;   // a padded list is split by its real length
        const unsigned q = total / 8u, rr = total % 8u, xcd = blockIdx.x % 8u, j = blockIdx.x / 8u;
        if (j >= q + (xcd < rr ? 1u : 0u)) return;                           // padding (block-uniform)
        task = (xcd < rr ? xcd * (q + 1u) : rr * (q + 1u) + (xcd - rr) * q) + j;
    }
    // the task's queries (block-uniform scalars), -1 = empty slot
    int tqid[8];
    if constexpr (IVF) {
        const int qs = T.task_qstart[task], qc = T.task_qcnt[task];
        if (qc <= 0) return;                                  // padding of a device-planned task list (block-uniform)
#pragma unroll
        for (int j = 0; j < 8; ++j) tqid[j] = (j < qc) ? T.sorted_q[qs + j] : -1;
    }
    auto fill = [&](int phase) {
        if constexpr (!IVF) {
            const uint4* src = reinterpret_cast<const uint4*>(qsrc + (size_t)phase * L::TABLE_BYTES);
            uint4* dst = reinterpret_cast<uint4*>(smem);
            for (int i = tid; i < L::TABLE_BYTES / 16; i += THREADS) {
                uint4 v = src[i];
                v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;   // l -> l - 128 (signed)
                dst[i] = v;
            }
        } else {
            // byte transpose: dword i of a query's compact table ([code][PM] bytes) holds the bytes of sub-quantisers
            // 4 u .. 4 u + 3 of one code; the LDS entry of one (code, slot) is the 8 queries' bytes side by side (two 4 x 4
            // byte transposes by v_perm).  Slots of a 16-block are written twice (second copy 16 slots further).
            constexpr int QB = RC_K * PM;                       // bytes of one query's table phase
            constexpr int DPC = PM / 4;                         // dwords per code
            // all QB / 4 / THREADS x 8 loads of the thread are issued before the first transpose: the per-query tables
            // (nq x M x 256 bytes) live in the memory-side cache at best, and a task is short (one cell)
            constexpr int FI = QB / 4 / THREADS;
            static_assert(QB / 4 % THREADS == 0, "whole iterations");
            unsigned dd[FI][8];
#pragma unroll
            for (int f = 0; f < FI; ++f)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    dd[f][j] = tqid[j] >= 0 ? reinterpret_cast<const unsigned*>(T.qbyte + ((size_t)tqid[j] * NP + phase) * QB)[tid + f * THREADS] : 0u;
#pragma unroll
            for (int f = 0; f < FI; ++f) {
                const int i = tid + f * THREADS;
                const int code = i / DPC, u = i % DPC;           // constant divisor
                const unsigned (&d)[8] = dd[f];
                unsigned o[8];                                   // o[2 t] = queries 0-3 of entry t, o[2 t + 1] = queries 4-7
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const unsigned a0 = d[4 * hq], a1 = d[4 * hq + 1], a2 = d[4 * hq + 2], a3 = d[4 * hq + 3];
                    const unsigned t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
                    const unsigned u0 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), u1 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
                    o[0 + hq] = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
                    o[2 + hq] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
                    o[4 + hq] = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
                    o[6 + hq] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
                }
                const uint4 lo4 = make_uint4(o[0] ^ 0x80808080u, o[1] ^ 0x80808080u, o[2] ^ 0x80808080u, o[3] ^ 0x80808080u);
                const uint4 hi4 = make_uint4(o[4] ^ 0x80808080u, o[5] ^ 0x80808080u, o[6] ^ 0x80808080u, o[7] ^ 0x80808080u);
                uint4* e = reinterpret_cast<uint4*>(smem + ((size_t)code * L::SLOTS + 4 * u) * 8);   // slot 4 u of the code's row
                e[0] = lo4;
                e[1] = hi4;
                if (L::HAS16 && 4 * u >= 32 * L::N32) {          // 16-block: second copy
                    e[8] = lo4;
                    e[9] = hi4;
                }
            }
        }
    };
    const int l = tid & 63, wv = tid >> 6;
    const int r = l & 15, g = l >> 4;
    int tq = INT_MAX;                                        // this lane's query = D column (l & 15)
    int myq = -1;                                            // ... and its id
    if constexpr (IVF) {
#pragma unroll
        for (int j = 0; j < 8; ++j) myq = (r == j) ? tqid[j] : myq;
    } else if (r < 8 && q0 + r < nq) {
        myq = q0 + r;
    }
    if (myq >= 0) {
        const int t = tint[myq];
        tq = (t == INT_MIN) ? INT_MIN : t - 128 * M;
    }
    adc_i32x4v bsel = {0, 0, 0, 0};                          // B[k][j = r] = [k % 8 == r], same bytes in every lane quarter
    if (r < 8) {
        const int one = 1 << (8 * (r & 3));
        bsel[r >> 2] = one;
        bsel[2 + (r >> 2)] = one;
    }
    unsigned off[STEPS];                                     // byte offset of this lane's slot in step s
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        int slot, m;
        adc_cf_step(PM, s, r, g, slot, m);
        // absolute LDS address of the slot in table row 0 (a generic pointer into LDS is {aperture, byte offset}: the low
        // 32 bits are the LDS address), so the gather address below needs no further base add
        off[s] = (unsigned)slot * 8u + static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    }
    constexpr int TILE = adc_cf_tile_rows(M);
    int64_t t0 = (int64_t)btile * TILE;
    int64_t t1 = (t0 + TILE < N) ? t0 + TILE : N;
    unsigned row_lo = 0;                                      // rows of the tile before this are not the task's
    if constexpr (IVF) {
        const int cell = T.task_list[task];
        const int64_t a = T.list_off[cell];
        t1 = T.list_off[cell + 1];
        t0 = a & ~(int64_t)15;                                // chunks start on multiples of 16 rows: the image's permutation
        row_lo = (unsigned)(a - t0);                          // depends on the ABSOLUTE row index mod 16
        if (t1 <= a) return;                                  // empty cell (block-uniform)
    }
    // Flat sequence of steps it = round * NP + i; a round covers ROUND rows and visits the NP table phases, odd rounds in
    // reverse order, so the tables already in LDS are used first (NP - 1 refills per round).  The codes of step it + 1 are
    // loaded while step it is gathered (NP == 1, two buffers); with two phases the eight chunks' codes are loaded at the start
    // of the step (requesting them right after the previous step's last gather instead was tried: it keeps them live across
    // the epilogue, 24 bytes of scratch per lane, and the M = 96 screen got 3 % slower).  All row arithmetic is 32-bit and relative to the tile (<= 32768 rows x M bytes).
    const unsigned nrows = (unsigned)(t1 - t0);
    const int nrounds = (int)((nrows + ROUND - 1) / ROUND);
    const int nsteps = nrounds * NPE;
    const uint8_t* __restrict__ tile = image + t0 * M * ADC_IMG_ES;
    auto phase_of = [&](int it) {
        if constexpr (PART != 0) return PART - 1;
        const int rd = it / NP, i = it % NP;
        return (rd & 1) ? NP - 1 - i : i;
    };
    // partial sums of chunk c of the round that starts at tile row r0: 8 bytes per lane with r < 8
    auto part_ptr = [&](unsigned r0, int c) {
        const size_t chunk = (size_t)((t0 + r0) >> 4) + (size_t)c;
        return reinterpret_cast<adc_u32x2v*>(PA.buf + (((size_t)lgroup * PA.nchunks + chunk) * 32 + (size_t)(g * 8 + (r & 7))) * 4);
    };
    const unsigned lane_off = (unsigned)(g * STEPS * ADC_IMG_ES), lane_row = (unsigned)(wv * R * 16 + r);
    // flat search with two table phases: the image is tile-blocked and phase-major (adc_scan_image_kernel), a block's tile
    // is one storage tile: dense PM-byte rows per phase.  The IVF index keeps the row-major image (cells start anywhere).
    constexpr bool BLOCKED = !IVF && NP > 1;
    constexpr unsigned ROWB = (unsigned)((BLOCKED ? PM : M) * ADC_IMG_ES);
    auto load_step = [&](int it, unsigned (&dst)[R][NW]) {
        const unsigned base = (unsigned)(it / NPE) * ROUND + lane_row;
        const unsigned col = (unsigned)phase_of(it) * (BLOCKED ? (unsigned)(TILE * PM * ADC_IMG_ES) : (unsigned)(PM * ADC_IMG_ES)) + lane_off;
#pragma unroll
        for (int c = 0; c < R; ++c) {
            unsigned n = base + 16u * c;
            n = n < nrows ? n : nrows - 1u;                    // rows past the end of the tile: the last row again
            const unsigned* cp = reinterpret_cast<const unsigned*>(tile + (n * ROWB + col));
#pragma unroll
            for (int j = 0; j < NW; ++j) dst[c][j] = cp[j];
        }
    };
    // ping-pong code buffers only where the registers allow: the IVF variant at 48 sub-quantisers per phase (table transpose +
    // aggregated survivor slots on top of the 12-step gather pipeline) spilled 100 bytes per lane with them
    constexpr bool PREFETCH = (NPE == 1) && !(IVF && PM == 48);
    adc_i32x4v acc[R];
    // pass 2: the partial sums stream from HBM (written once by pass 1, never cached): they are requested one round (R = 4
    // chunks, ~2 us) ahead, ping-pong like the code buffers.  Only the lanes with r < 8 own a column of D, so the lanes with
    // r >= 8 fetch the sums of chunk c + R/2 for their neighbour r - 8 (half the registers; a DPP row rotation by 8 hands
    // them over when they are added).
    constexpr int PR = PART == 2 ? (R + 1) / 2 : 1;
    auto load_part = [&](int it, adc_u32x2v (&dst)[PR]) {
        if constexpr (PART == 2) {
            if (it < nsteps) {
                const unsigned r0p = (unsigned)(it / NPE) * ROUND + (unsigned)(wv * R * 16);
#pragma unroll
                for (int c = 0; c < PR; ++c) {
                    const int cc = (r < 8) ? c : c + PR;
                    dst[c] = (cc < R) ? __builtin_nontemporal_load(part_ptr(r0p, cc)) : adc_u32x2v{0u, 0u};
                }
            }
        }
    };
    // one step: gather + fold the R chunks of step `it` from the codes in `w`; the next step's codes go to `wn`
    auto run_step = [&](int it, unsigned (&w)[R][NW], unsigned (&wn)[PREFETCH ? R : 1][NW], int& in_lds,
                        const adc_u32x2v (&part)[PR]) {
        const int phase = phase_of(it);
        if (it % NPE == 0) {
#pragma unroll
            for (int c = 0; c < R; ++c) acc[c] = adc_i32x4v{0, 0, 0, 0};
        }
        if (in_lds != phase) {
            if (in_lds >= 0) __syncthreads();                 // every wave is done gathering from the old tables
            fill(phase);
            __syncthreads();
            in_lds = phase;
        }
        if constexpr (PREFETCH) {
            if (it + 1 < nsteps) load_step(it + 1, wn);
        } else {
            load_step(it, w);
        }
        // Software pipeline over the R chunks of the step: all STEPS gathers of chunk c + 1 are issued before the MFMAs
        // of chunk c, so a wave keeps a whole chunk of LDS reads in flight.  Address of a gather: two VALU instructions,
        // v_bfe_u32 (the code byte) + v_lshl_add_u32 (code * row bytes + this lane's slot offset).
        uint2 ea[STEPS], eb[STEPS];
        const unsigned rowbytes = L::SLOTS * 8;
        (void)rowbytes;
        auto gather = [&](int c, uint2 (&e)[STEPS]) {
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                unsigned addr;
#if RC_ADC_IMG16
                if (s & 1)
                    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(addr) : "v"(w[c][s >> 1]), "s"(rowbytes), "v"(off[s]));
                else
                    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(addr) : "v"(w[c][s >> 1]), "s"(rowbytes), "v"(off[s]));
#else
                asm("v_bfe_u32 %0, %1, %2, 8\n\tv_lshl_add_u32 %0, %0, %3, %4"
                    : "=&v"(addr)
                    : "v"(w[c][s >> 2]), "n"(8 * (s & 3)), "n"(__builtin_ctz(L::SLOTS * 8)), "v"(off[s]));
#endif
                typedef unsigned adc_u32x2 __attribute__((ext_vector_type(2)));
                const adc_u32x2 v = *reinterpret_cast<const adc_u32x2 __attribute__((address_space(3)))*>(addr);
                e[s] = make_uint2(v.x, v.y);
            }
        };
        auto fold = [&](int c, const uint2 (&e)[STEPS]) {
#pragma unroll
            for (int s2 = 0; s2 < STEPS / 2; ++s2) {
                const adc_i32x4v a = {(int)e[2 * s2].x, (int)e[2 * s2].y, (int)e[2 * s2 + 1].x, (int)e[2 * s2 + 1].y};
                acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, acc[c], 0, 0, 0);
            }
        };
        gather(0, ea);
#pragma unroll
        for (int c = 0; c < R; c += 2) {
            if (c + 1 < R) gather(c + 1, eb);
            fold(c, ea);
            if (c + 2 < R) gather(c + 2, ea);
            if (c + 1 < R) fold(c + 1, eb);
        }
        if constexpr (PART == 1) {
            // pass 1: the accumulators go to HBM as they are (int16 pairs), nothing is tested
            const unsigned r0p = (unsigned)(it / NPE) * ROUND + (unsigned)(wv * R * 16);
            if (r < 8) {
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const adc_u32x2v v = {__builtin_amdgcn_perm((unsigned)acc[c][1], (unsigned)acc[c][0], 0x05040100u),
                                          __builtin_amdgcn_perm((unsigned)acc[c][3], (unsigned)acc[c][2], 0x05040100u)};
                    __builtin_nontemporal_store(v, part_ptr(r0p, c));
                }
            }
        }
        if constexpr (PART == 2) {
#pragma unroll
            for (int c = 0; c < R; ++c) {
                // chunk c < PR: this lane's own registers; chunk c >= PR: held by lane r + 8 (meaningful for r < 8 only)
                unsigned px = part[c % PR].x, py = part[c % PR].y;
                if (c >= PR) { px = (unsigned)rc_dpp_row_ror<8>((int)px); py = (unsigned)rc_dpp_row_ror<8>((int)py); }
                acc[c][0] += ((int)(px << 16)) >> 16;
                acc[c][1] += ((int)px) >> 16;
                acc[c][2] += ((int)(py << 16)) >> 16;
                acc[c][3] += ((int)py) >> 16;
            }
        }
        if (PART != 1 && it % NPE == NPE - 1) {
            // survivors are rare (~2e-4 of the (row, query) pairs): one max over the round's accumulators decides
            int top = INT_MIN;
#pragma unroll
            for (int c = 0; c < R; ++c) top = max(top, max(max(acc[c][0], acc[c][1]), max(acc[c][2], acc[c][3])));
            if (__ballot(top >= tq)) {
                // Flat search: survivors are rare (~2e-4), one atomic each.  IVF: a query keeps a few per cent of the rows it
                // probes, and one atomic per survivor on 1200 counters was half of the screen's time (nprobe 32) - there the
                // four lanes (r, g = 0..3) of a query reserve their slots with ONE atomic per wave and round.
                const unsigned r0 = (unsigned)(it / NPE) * ROUND + (unsigned)(wv * R * 16);
                if constexpr (!IVF) {
#pragma unroll
                    for (int c = 0; c < R; ++c) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned n = r0 + 16u * c + 4u * g + e;   // D[row = 4 g + e][column = r]
                            if (acc[c][e] >= tq && n < nrows && n >= row_lo) {
                                const unsigned slot = atomicAdd(id_count + myq, 1u);
                                if (slot < ADC_ID_CAP) ids[(size_t)myq * ADC_ID_CAP + slot] = (unsigned)(t0 + n);
                            }
                        }
                    }
                } else {
                unsigned mine = 0;
#pragma unroll
                for (int c = 0; c < R; ++c) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned n = r0 + 16u * c + 4u * g + e;   // D[row = 4 g + e][column = r]
                        mine += (acc[c][e] >= tq && n < nrows && n >= row_lo) ? 1u : 0u;
                    }
                }
                const unsigned c0 = __shfl(mine, r), c1 = __shfl(mine, r + 16), c2 = __shfl(mine, r + 32), c3 = __shfl(mine, r + 48);
                const unsigned total = c0 + c1 + c2 + c3;
                unsigned base = 0;
                if (g == 0 && total) base = atomicAdd(id_count + myq, total);      // total > 0 implies a live query
                base = __shfl(base, r);
                unsigned slot = base + (g > 0 ? c0 : 0u) + (g > 1 ? c1 : 0u) + (g > 2 ? c2 : 0u);
                if (mine) {
#pragma unroll
                    for (int c = 0; c < R; ++c) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned n = r0 + 16u * c + 4u * g + e;
                            if (acc[c][e] >= tq && n < nrows && n >= row_lo) {
                                if (slot < ADC_ID_CAP) ids[(size_t)myq * ADC_ID_CAP + slot] = (unsigned)(t0 + n);
                                ++slot;
                            }
                        }
                    }
                }
                }
            }
        }
    };
    unsigned wa[R][NW], wb[PREFETCH ? R : 1][NW];
    adc_u32x2v pa[PR], pb[PR];
    int in_lds = -1;
    if constexpr (PREFETCH) {
        // ping-pong over the two code buffers (and, pass 2, the two partial-sum buffers): no register copies between steps
        load_step(0, wa);
        load_part(0, pa);
        for (int it = 0; it < nsteps; it += 2) {              // block-uniform
            load_part(it + 1, pb);
            run_step(it, wa, wb, in_lds, pa);
            load_part(it + 2, pa);
            if (it + 1 < nsteps) run_step(it + 1, wb, wa, in_lds, pb);
        }
    } else {
        for (int it = 0; it < nsteps; ++it) run_step(it, wa, wb, in_lds, pa);
    }
}


// ------------------------------------------------------------------------------------ 4b''. 16 queries per gather
// Round 3.  PMC and the round-3 micro-benchmark (tools/ubench_lds_gather.hip, profiles/r03a_ubench_lds_gather.txt) agree
// that the conflict-free 8-query screen above is not bound by the LDS array (2.0 of ~6 cycles per gather and CU) but by
// instruction issue: two address instructions + half an i8 MFMA (~3 VALU-equivalents on the shared issue port) per
// 8-byte gather.  A ds_read_b128 gather serves 16 queries for the same address arithmetic and one MFMA: 5.55 cycles per
// 16 queries against 2 x 4.0 in the micro-benchmark.  The price is table size — 16 queries x M x 256 bytes no longer
// fit the LDS — so the sub-quantisers are visited in PHASES of 16 (64 KiB of tables: [code][slot 0..15][16 queries], a
// code's row = 256 bytes = all 64 banks once) with TWO buffers: while a phase is gathered, the next one is copied into
// the other buffer by global_load_lds_dwordx4 (asynchronous, no registers), one barrier per phase change, no refill on
// the critical path.  What round 2 measured against phases (synchronous 128 KiB refills from the memory-side cache: the
// two-phase M = 96 screen pays 43 % for them) does not apply: the refill is prefetched, and the block -> (group, tile)
// map gives every XCD a fixed set of ~10 query groups whose tables (192 KiB each at M = 48) stay in its L2.
//
// Conflict freedom for ds_read_b128: the LDS serves a wave in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,
// 28-31} and the same +32; MI355X_MICROARCH.md), 16 lanes x 16 bytes = one pass over the 64 banks if they read 16
// different 16-byte slots.  Lane l (row r = l & 15, quarter g = l >> 4) has position p(l & 31) in its service group and
// reads slot (p + j + 4 (l >> 5)) mod 16 in step j = 0..3 — distinct inside every group, and the four lanes of a row
// (positions a, a + 8 in both halves) cover the 16 slots exactly once.  As in the 8-query screen a lane must receive its
// codes in its own visiting order: the flat-search image of these M is [tile][phase][row][quarter g][step j] (tiles of
// ADC_Q16_TILE rows, phase-major inside a tile: a wave's code load for 16 rows is 256 contiguous bytes).
// Accumulation: v_mfma_i32_16x16x64_i8 with A = the lane's 16 gathered bytes (one sub-quantiser x 16 queries), B[k][n] =
// [k mod 16 == n]: D[row][query] += sum over the row's four lanes.  Everything downstream is unchanged.
#ifndef ADC_Q16_TILE
#define ADC_Q16_TILE 32768
#endif
#ifndef ADC_Q16_R
#define ADC_Q16_R 8                // chunks of 16 rows per wave and round (16: accumulators kept as int16 pairs between phases)
#endif
#ifndef ADC_Q16_WAVES
#define ADC_Q16_WAVES 16           // waves per block: a round = WAVES x R x 16 rows
#endif
#ifndef ADC_Q16_PACK
#define ADC_Q16_PACK 0             // 1: accumulators kept as int16 pairs between phases (more chunks per wave in 128 VGPRs)
#endif
#define ADC_Q16_SCAP 256           // survivor entries per wave held in LDS between flushes
__host__ __device__ constexpr int adc_q16_pos(int h32) {
    return (h32 < 4) ? h32 : (h32 < 12) ? h32 - 4 : (h32 < 16) ? h32 - 8 : (h32 < 20) ? h32 - 8 : (h32 < 28) ? h32 - 12 : h32 - 16;
}
// sub-quantiser (within its phase of 16) = LDS slot that lane `lane` of a wave reads in step j
__host__ __device__ constexpr int adc_q16_slot(int lane, int j) { return (adc_q16_pos(lane & 31) + j + 4 * (lane >> 5)) & 15; }

// Image of rows n0 <= n < n0 + cnt.  Inside a (tile, phase) block of T x 16 bytes the bytes are ordered the way the
// kernel's waves consume them: [round of 2048 rows][wave][lane = r + 16 g][chunk c][step j], row = 2048 round + 128 wave +
// 16 c + r — a lane's codes for the 8 chunks of a step are 32 contiguous bytes (two 16-byte loads), a wave's 2 KiB.
// byte (n, phase, g, j) = codes[n][16 phase + slot(lane = (n & 15) + 16 g, j)]
__host__ __device__ inline int64_t adc_q16_image_at(int64_t n, int NPH, int phase, int g, int j) {
    constexpr int64_t T = ADC_Q16_TILE;
    constexpr int RW = ADC_Q16_R * 16;                       // rows per wave and round
    const int64_t nt = n % T;
    const int64_t round = nt / (RW * ADC_Q16_WAVES), nr = nt % (RW * ADC_Q16_WAVES);
    const int wv = (int)(nr / RW), c = (int)((nr % RW) / 16), r = (int)(nr % 16);
    return ((n / T) * NPH + phase) * T * 16 + round * (int64_t)(RW * ADC_Q16_WAVES * 16) + (int64_t)wv * (RW * 16) +
           (int64_t)(r + 16 * g) * (ADC_Q16_R * 4) + c * 4 + j;
}
__global__ __launch_bounds__(256) void adc_q16_image_kernel(const uint8_t* __restrict__ codes, int64_t n0, int64_t cnt, int M,
                                                            uint8_t* __restrict__ image) {
    const int64_t total = cnt * M;
    const int NPH = M / 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = n0 + i / M;
        const int pos = (int)(i % M);
        const int phase = pos >> 4, g = (pos >> 2) & 3, j = pos & 3;
        const int m = 16 * phase + adc_q16_slot((int)(n & 15) + 16 * g, j);
        image[adc_q16_image_at(n, NPH, phase, g, j)] = codes[n * M + m];
    }
}

// byte tables [group of 16 queries][phase][code][slot][16 queries], biased by -128 (the MFMA is signed): thread = (code,
// slot) — 64 x 16 threads per block, one 16-byte store each (round 3: thread = code walked the 16 slots, 900 waves in all
// for 14.7 M entries: 91 us per 1200 queries at M = 48)
__global__ __launch_bounds__(1024) void adc_qlut16_write_kernel(const float* __restrict__ lut, const float* __restrict__ qstat,
                                                                int M, int nq, uint8_t* __restrict__ qlut) {
    const int G = blockIdx.x, c = blockIdx.y * 64 + threadIdx.x, phase = blockIdx.z, sl = threadIdx.y;
    const int NPH = M / 16;
    const int nv = (nq - 16 * G) < 16 ? (nq - 16 * G) : 16;
    uint4* row = reinterpret_cast<uint4*>(qlut + (((size_t)G * NPH + phase) * RC_K + c) * 256);
    const int m = 16 * phase + sl;
    unsigned w[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};    // absent queries: byte 0 -> -128
    float v[16];
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) v[qq] = (qq < nv) ? lut[((size_t)(16 * G + qq) * M + m) * RC_K + c] : 0.f;
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) {
        if (qq < nv) {
            const int q = 16 * G + qq;
            const unsigned l = adc_quant8(v[qq], qstat[(size_t)q * ADC_QSTAT_STRIDE + m],
                                          qstat[(size_t)q * ADC_QSTAT_STRIDE + ADC_QSTAT_STRIDE - 1]);
            w[qq >> 2] = (w[qq >> 2] & ~(0xFFu << (8 * (qq & 3)))) | ((l ^ 0x80u) << (8 * (qq & 3)));
        }
    }
    row[sl] = make_uint4(w[0], w[1], w[2], w[3]);
}

typedef unsigned adc_u32x4v __attribute__((ext_vector_type(4)));

// grid: 8 x ceil(groups / 8) x tiles blocks, dealt so that XCD x (= block id mod 8) owns the groups == x (mod 8).
template <int M>
__global__ __launch_bounds__(ADC_Q16_WAVES * 64) void adc_screen_q16_kernel(const uint8_t* __restrict__ image, int64_t N,
                                                                            const uint8_t* __restrict__ qlut,
                                                                            const int* __restrict__ tint, int nq, int groups,
                                                                            unsigned* __restrict__ id_count,
                                                                            unsigned* __restrict__ ids, int flags) {
    constexpr int R = ADC_Q16_R, NWAVES = ADC_Q16_WAVES;
    constexpr int NPH = M / 16, ROUND = NWAVES * R * 16, TILE = ADC_Q16_TILE;
    constexpr int BUF = RC_K * 256;                           // 64 KiB: one phase of one group
    static_assert(TILE % ROUND == 0 && R % 4 == 0, "whole rounds per tile; a lane's codes of a step = R / 4 16-byte loads");
    constexpr int NV = R / 4;                                 // 16-byte code loads per lane and step
    constexpr bool PACK = ADC_Q16_PACK != 0;                  // accumulators as int16 pairs between phases (|sum| <= 128 M)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
    const int r = l & 15, g = l >> 4;
    const unsigned xcd = blockIdx.x & 7u, jx = blockIdx.x >> 3;
    unsigned group, btile;
    if (groups >= 8) {                                        // XCD x owns the groups == x (mod 8), every tile
        const unsigned gpx = (unsigned)(groups + 7) / 8u;     // groups per XCD
        group = (jx % gpx) * 8u + xcd;
        btile = jx / gpx;
        if (group >= (unsigned)groups) return;                // block-uniform
    } else {                                                  // few queries (JPQ steps, validation): every XCD takes all the
        group = jx % (unsigned)groups;                        // groups (their tables fit any L2) and the tiles == x (mod 8)
        btile = (jx / (unsigned)groups) * 8u + xcd;
        if ((int64_t)btile * TILE >= N) return;
    }
    const int q0 = (int)group * 16;
    const uint8_t* qsrc = qlut + (size_t)group * NPH * BUF;
    // asynchronous copy of one phase's tables into an LDS buffer: 16 waves x 4 pieces of 1 KiB
    auto stage = [&](int phase, int buf) {
        const uint8_t* src = qsrc + (size_t)phase * BUF;
#pragma unroll
        for (int i = 0; i < (BUF / 1024 + NWAVES - 1) / NWAVES; ++i) {
            const int piece = i * NWAVES + wv;               // wave-uniform
            if (piece < BUF / 1024)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)piece * 1024 + l * 16),
                                                 (__attribute__((address_space(3))) void*)(smem + (size_t)buf * BUF + (size_t)piece * 1024),
                                                 16, 0, 0);
        }
    };
    int tq = INT_MAX, myq = -1;
    if (q0 + r < nq) {
        myq = q0 + r;
        const int t = tint[myq];
        tq = (t == INT_MIN) ? INT_MIN : t - 128 * M;
    }
    adc_i32x4v bsel = {0, 0, 0, 0};                           // B[k][n = r] = [k mod 16 == r]
    bsel[r >> 2] = 1 << (8 * (r & 3));
    const bool rc_q16_setprio = (flags & 1) != 0;
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    if (lds0 & 0xFFFFu) __builtin_trap();                     // the one-instruction gather address needs 64 KiB-aligned table buffers
    unsigned off[4], offb[4];                                 // this lane's slot offsets (absolute LDS address): buffer 0 / current buffer
#pragma unroll
    for (int j = 0; j < 4; ++j) off[j] = lds0 + (unsigned)adc_q16_slot(l, j) * 16u;
    const int64_t t0 = (int64_t)btile * TILE;
    const int64_t t1 = (t0 + TILE < N) ? t0 + TILE : N;
    const unsigned nrows = (unsigned)(t1 - t0);
    const int nrounds = (int)((nrows + ROUND - 1) / ROUND);
    const int nsteps = nrounds * NPH;
    const uint8_t* __restrict__ tile = image + t0 * M;       // the tile's storage: [phase][round][wave][lane][c][j]
    auto phase_of = [&](int it) { const int rd = it / NPH, i = it % NPH; return (rd & 1) ? NPH - 1 - i : i; };
    // next step after `it` whose phase differs from phase_of(it) (nsteps if none)
    auto next_change = [&](int it) {
        int k = it + 1;
        while (k < nsteps && phase_of(k) == phase_of(it)) ++k;
        return k;
    };
    // this lane's 32 bytes of codes of step `it`: rows past the end of the index read the (allocated, unspecified) padding of
    // the last tile — any byte is a valid code for the gathers, and those rows are masked at the survivor test
    const unsigned lane_at = (unsigned)(wv * (R * 16 * 16) + l * (R * 4));
    auto load_step = [&](int it, adc_u32x4v (&dst)[NV]) {
        const adc_u32x4v* cp = reinterpret_cast<const adc_u32x4v*>(tile + ((size_t)phase_of(it) * (TILE * 16) +
                                                                          (size_t)(it / NPH) * (ROUND * 16) + lane_at));
#pragma unroll
        for (int v = 0; v < NV; ++v) dst[v] = cp[v];
    };
    adc_i32x4v acc[PACK ? 1 : R];
    unsigned accp[PACK ? R : 1][2];                           // PACK: (acc0 | acc1 << 16), (acc2 | acc3 << 16)
    int buf = 0;
    // per-wave survivor list in LDS: entries (row in tile << 4 | query column); flushed to the per-query id lists (one
    // global atomic per entry, all of a flush in flight together) when 64 more might not fit, and at the end
    constexpr int SCAP = ADC_Q16_SCAP;
    unsigned* sbuf = reinterpret_cast<unsigned*>(smem + 2 * BUF) + wv * SCAP;
    int scount = 0;                                          // wave-uniform
    auto flush_survivors = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's list writes are in the LDS (in-order queue)
        for (int i = l; i < scount; i += 64) {
            const unsigned e = sbuf[i];
            const int q = q0 + (int)(e & 15u);
            const unsigned slot = atomicAdd(id_count + q, 1u);
            if (slot < ADC_ID_CAP) ids[(size_t)q * ADC_ID_CAP + slot] = (unsigned)(t0 + (e >> 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // list reads done before it is overwritten
        scount = 0;
    };
    // survivor test of one chunk's sums (D[row = 4 g + e][column = r]); survivors go to the wave's LDS list
    auto test_chunk = [&](const adc_i32x4v& v, unsigned rbase) {
        const int top = max(max(v[0], v[1]), max(v[2], v[3]));
        if (__ballot(top >= tq)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned n = rbase + 4u * g + e;
                const bool hit = v[e] >= tq && n < nrows;
                const unsigned long long mask = __ballot(hit);
                if (mask) {                                       // wave-uniform
                    if (scount + 64 > SCAP) flush_survivors();
                    if (hit) sbuf[scount + __popcll(mask & ((1ull << l) - 1ull))] = (n << 4) | (unsigned)r;
                    scount += (int)__popcll(mask);
                }
            }
        }
    };
    auto run_step = [&](int it, const adc_u32x4v (&w)[NV], adc_u32x4v (&wn)[NV]) {
        if (it > 0 && phase_of(it) != phase_of(it - 1)) {
            if constexpr (NPH <= 2) {
                // both phases' tables stay in the two buffers (M = 32): a phase change is an address offset, nothing else
                buf ^= 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) offb[j] = off[j] + (unsigned)buf * (unsigned)BUF;
            } else {
            // phase change: this phase's tables were requested into the other buffer one segment ago
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces have landed
            __syncthreads();                                 // everybody's have, and everybody is done with the old buffer
            buf ^= 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) offb[j] = off[j] + (unsigned)buf * (unsigned)BUF;
            const int nx = next_change(it);
            if (nx < nsteps) stage(phase_of(nx), buf ^ 1);   // the buffer just vacated
            }
        }
        if (it + 1 < nsteps) load_step(it + 1, wn);
        // Software pipeline over the chunks: the 4 gathers of chunk c + 1 are ISSUED before the 4 MFMAs of chunk c (the
        // scheduler, left alone, reuses one register quad and waits for every gather: one LDS round trip per MFMA).
        adc_u32x4v ea[4], eb[4];
        auto gather = [&](int c, adc_u32x4v (&e)[4]) {
            const unsigned wc = w[c >> 2][c & 3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // LDS address = buffer base (a multiple of 64 KiB: bytes 2-3) | code << 8 (byte 1) | slot offset (byte 0), built by
                // ONE v_perm_b32 from the code word and the lane's slot constant (round 3: v_bfe_u32 + v_lshl_add_u32).  The
                // dynamic LDS of this kernel starts at address 0 (no static __shared__), checked once per block below.
                const unsigned addr = __builtin_amdgcn_perm(wc, offb[j], 0x03020000u | ((4u + (unsigned)j) << 8));
                e[j] = *reinterpret_cast<const adc_u32x4v __attribute__((address_space(3)))*>(addr);
            }
        };
        const bool first = (it % NPH == 0);                   // block-uniform: the round's first step starts from zero
        const bool last = (it % NPH == NPH - 1);
        const unsigned r0 = (unsigned)(it / NPH) * ROUND + (unsigned)(wv * R * 16);
        auto fold = [&](int c, const adc_u32x4v (&e)[4]) {
            if constexpr (!PACK) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const adc_i32x4v a = {(int)e[j][0], (int)e[j][1], (int)e[j][2], (int)e[j][3]};
                    if (j == 0 && first) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, adc_i32x4v{0, 0, 0, 0}, 0, 0, 0);
                    else acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, acc[c], 0, 0, 0);
                }
            } else {
                // the chunk's sums live as two int16 pairs between phases: unpacked into the first MFMA's C, packed again after
                // the fourth (v_perm), tested right here in the round's last phase
                adc_i32x4v v = {0, 0, 0, 0};
                if (!first)
                    v = adc_i32x4v{((int)(accp[c][0] << 16)) >> 16, ((int)accp[c][0]) >> 16, ((int)(accp[c][1] << 16)) >> 16,
                                   ((int)accp[c][1]) >> 16};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const adc_i32x4v a = {(int)e[j][0], (int)e[j][1], (int)e[j][2], (int)e[j][3]};
                    v = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, v, 0, 0, 0);
                }
                if (last) {
                    test_chunk(v, r0 + 16u * c);
                } else {
                    accp[c][0] = __builtin_amdgcn_perm((unsigned)v[1], (unsigned)v[0], 0x05040100u);
                    accp[c][1] = __builtin_amdgcn_perm((unsigned)v[3], (unsigned)v[2], 0x05040100u);
                }
            }
        };
        gather(0, ea);
#pragma unroll
        for (int c = 0; c < R; c += 2) {
            // Progress-proportional priority: the arbiter serves the OLDEST ready wave first, so without this wave 0 finishes
            // a segment in a third of the time the block needs and the last waves run alone, latencies exposed, while
            // the others wait at the phase change.  A wave that is behind in its segment outranks one that is ahead.
            if (rc_q16_setprio) {
                if (c == 0) __builtin_amdgcn_s_setprio(3);
                else if (c == R / 4) __builtin_amdgcn_s_setprio(2);
                else if (c == R / 2) __builtin_amdgcn_s_setprio(1);
                else if (c == 3 * R / 4) __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            gather(c + 1, eb);
            __builtin_amdgcn_sched_barrier(0);
            fold(c, ea);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < R) gather(c + 2, ea);
            __builtin_amdgcn_sched_barrier(0);
            fold(c + 1, eb);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!PACK) {
            if (last) {
                // A wave's round holds 128 rows x 16 queries: about every third round has a survivor (2e-4 per pair), so the
                // test is made per CHUNK (one max3 pair + compare + ballot each) and only a chunk that has one is scanned.
                // Survivors go to the wave's LDS list (see flush_survivors): no global atomic — a ~2 us round trip — inside
                // the loop, where one waiting wave holds up the other fifteen at the next phase change (9.7 -> 9.1 ms).
#pragma unroll
                for (int c = 0; c < R; ++c) test_chunk(acc[c], r0 + 16u * c);
            }
        }
    };
    // prologue: first phase into buffer 0, the next distinct phase into buffer 1
    stage(phase_of(0), 0);
    if constexpr (NPH == 2) stage(1, 1);                     // phase p lives in buffer p for the whole block (phase_of(0) = 0)
    adc_u32x4v wa[NV], wb[NV];
    load_step(0, wa);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (NPH > 2) {
        const int nx = next_change(0);
        if (nx < nsteps) stage(phase_of(nx), 1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) offb[j] = off[j];
    for (int it = 0; it < nsteps; it += 2) {                 // block-uniform
        run_step(it, wa, wb);
        if (it + 1 < nsteps) run_step(it + 1, wb, wa);
    }
    flush_survivors();
}

// One block per query: exact fp32 score (m ascending, from 0) of every screened row; rows with score >= tau go
// to the key list exactly as adc_scan_kernel<FILTER> would have put them.
// A block's table is 4 M x 256 bytes of LDS and every survivor costs one dependent M-byte read from HBM, so the kernel lives
// on rows in flight: the block is as large as the LDS lets the CU hold 16+ waves (adc_rescore_threads), the table and the
// codes move in 16-byte pieces, and every thread has two rows in flight (round 4; 512 threads and 4-byte loads before:
// M = 96 ran 8 waves per CU).
template <int M>
__device__ __forceinline__ float adc_rescore_row(const uint8_t* __restrict__ cp, const float* __restrict__ tab) {
    constexpr int W = (M % 16 == 0) ? 16 : (M % 8 == 0) ? 8 : 4;  // load width in bytes
    unsigned w[M / 4];
#pragma unroll
    for (int j = 0; j < M / W; ++j) {
        if constexpr (W == 16) {
            const uint4 v = reinterpret_cast<const uint4*>(cp)[j];
            w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
        } else if constexpr (W == 8) {
            const uint2 v = reinterpret_cast<const uint2*>(cp)[j];
            w[2 * j] = v.x; w[2 * j + 1] = v.y;
        } else {
            w[j] = reinterpret_cast<const unsigned*>(cp)[j];
        }
    }
    float s = 0.f;
#if defined(RC_ABL_RESCORE) && (RC_ABL_RESCORE & 2)
#pragma unroll
    for (int m = 0; m < M / 4; ++m) s = s + __uint_as_float(w[m]);
#else
#pragma unroll
    for (int m = 0; m < M; ++m) s = s + tab[m * RC_K + ((w[m >> 2] >> (8 * (m & 3))) & 0xFFu)];
#endif
    return s;
}

static int adc_rescore_threads(int M) { return M * RC_K * 4 > 80 * 1024 ? 1024 : 512; }

template <int M>
__global__ __launch_bounds__(1024) void adc_rescore_kernel(const uint8_t* __restrict__ codes,
                                                          const float* __restrict__ lut,
                                                          const float* __restrict__ thr,
                                                          const unsigned* __restrict__ id_count,
                                                          const unsigned* __restrict__ ids,
                                                          unsigned* __restrict__ cand_count,
                                                          unsigned long long* __restrict__ cand,
                                                          int* __restrict__ status,
                                                          const int64_t* __restrict__ rowmap,
                                                          int* __restrict__ qstatus = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tab = reinterpret_cast<float*>(smem);  // [M][256]
    const int qi = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const unsigned raw = id_count[qi];
    const unsigned cnt = raw > ADC_ID_CAP ? ADC_ID_CAP : raw;
    const unsigned* qids = ids + (size_t)qi * ADC_ID_CAP;
    // the first rows' ids and codes are requested before the table: their latency hides behind the staging
    unsigned n0 = 0, n1 = 0;
    if (tid < (int)cnt) n0 = qids[tid];
    if (tid + nthr < (int)cnt) n1 = qids[tid + nthr];
    {
        const float4* l4 = reinterpret_cast<const float4*>(lut + (size_t)qi * M * RC_K);
        float4* t4 = reinterpret_cast<float4*>(tab);
        for (int i = tid; i < M * RC_K / 4; i += nthr) t4[i] = l4[i];
    }
    if (tid == 0 && raw > ADC_ID_CAP) {
        atomicOr(status, 2);
        if (qstatus) atomicOr(qstatus + qi, 2);
    }
    const float tau = thr[qi];
    // this block is the only writer of the query's key list: slots come from an LDS counter, the global count is written
    // once at the end (round 3: one returning global atomic per wave and iteration, all on ONE address — 27 of 150 us)
    __shared__ unsigned s_slots;
    if (tid == 0) s_slots = 0u;
    const unsigned base0 = cand_count[qi];
    __syncthreads();
    for (unsigned i0 = 0; i0 < cnt; i0 += 2 * nthr) {
        const unsigned ia = i0 + tid, ib = ia + nthr;
        const bool la = ia < cnt, lb = ib < cnt;
        const unsigned na = n0, nb = n1;
        // next pair of ids (dependent chain: id -> codes), requested before this pair is scored
        n0 = (ia + 2 * nthr < cnt) ? qids[ia + 2 * nthr] : 0u;
        n1 = (ib + 2 * nthr < cnt) ? qids[ib + 2 * nthr] : 0u;
        // a wave whose 64 slots are all past the end of the list does nothing (the last iteration of a 2100-row list has
        // 96 live slots of 2048: without the test the kernel did 1.9 x the lookups the list needs)
        float sa = 0.f, sb = 0.f;
        if (__ballot(la)) sa = adc_rescore_row<M>(codes + (size_t)(la ? na : 0u) * M, tab);
        if (__ballot(lb)) sb = adc_rescore_row<M>(codes + (size_t)(lb ? nb : 0u) * M, tab);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool live = h ? lb : la;
            const float sc = h ? sb : sa;
            const unsigned n = h ? nb : na;
            const bool pass = live && (sc >= tau);
#if defined(RC_ABL_RESCORE) && (RC_ABL_RESCORE & 1)
            const unsigned long long mask = __ballot(pass && sc == 12345.678f);
#else
            const unsigned long long mask = __ballot(pass);
#endif
            if (mask) {
                const int lane = tid & 63;
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                unsigned base = 0;
                if (lane == (int)__builtin_ctzll(mask)) base = atomicAdd(&s_slots, (unsigned)__popcll(mask));
                base = __shfl(base, (int)__builtin_ctzll(mask));
                const unsigned slot = base0 + base + rank;
                if (pass && slot < ADC_CAND_CAP) {
                    // IVF: rows are stored cell-major; the key carries the row's corpus position so ties order by corpus id
                    const unsigned id = rowmap ? (unsigned)rowmap[n] : n;
                    cand[(size_t)qi * ADC_CAND_CAP + slot] =
                        ((unsigned long long)adc_order_key(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - id);
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0 && s_slots) cand_count[qi] = base0 + s_slots;
}

// ------------------------------------------------------------------------------------------ host
extern "C" size_t rc_adc_scan_image_bytes(int64_t N, int M);
struct adc_ws_layout {
    size_t lut, sample, thr, cnt, cand, qlut, tint, qstat, idcnt, ids, image, partial, partial_bytes, total;
    int64_t S;
};
// partial sums of the two-pass M = 96 screen: 256 bytes per (group of 8 queries, chunk of 16 rows); at most ADC_PART_CAP
// bytes are kept, the groups are processed in as many launches as that takes
#define ADC_PART_CAP (8ull << 30)
struct adc_part_plan { unsigned nchunks, groups_per_pass; size_t bytes; };
static adc_part_plan adc_part_plan_for(int64_t N, int M, int nq);
static int adc_qs_for(int M) { (void)M; return 16; }   // table groups are sized for 16 queries (covers the 8- and 4-query kernels)
// conflict-free screen (adc_screen_cf_kernel): M = 16, 32, 48, 64 in one table phase, 96 in two
static bool adc_cf_supported(int M) { return M == 16 || M == 32 || M == 48 || M == 64 || M == 96; }
// M = 96: two phases of 48 sub-quantisers (128 KiB of tables, one 1024-thread block per CU).  Tried and rejected in round 2:
// three phases of 32 (64 KiB of tables, TWO 512-thread blocks per CU so that one block's refill hides behind the other's
// gathers; 126 VGPRs, no spills, both blocks resident): 39 ms instead of 27 ms per 1200 queries flat, no change for the IVF
// tasks — twice the refills per row and a third exposed code load per round cost more than the overlap returns.  Also without
// effect on the 27 ms (or worse): three 32-wide phases with the NEXT phase's table copied into a second LDS buffer by
// global_load_lds_dwordx4 while the current one is gathered (one barrier per step, no refill on the critical path: 36 ms —
// 64 gathers per wave between barriers do not amortise the pipeline ramp), 8 waves per CU with 16 / 12 chunks each
// (192 / 144 gathers per wave between barriers, 242 / 168 VGPRs: 29.4 / 32.7 ms), requesting the next step's codes before the refill, and an XCD mapping of 8 groups x 4 tiles that
// keeps the tables L2-resident.  What the two-phase screen pays over 2 x the one-phase time (20 ms) is the pipeline
// drain and ramp-up of 16 waves around the two barriers of every 2048-row round.
static int adc_cf_phase_m(int M) { return M == 96 ? 48 : M; }
static int adc_ivf_phase_m(int M) { return adc_cf_phase_m(M); }   // table phase of the round-2 IVF screen (RC_IVF_PIPE=0)
static size_t adc_cf_table_bytes(int M) {                  // per group of 8 queries, all phases
    const int PM = adc_cf_phase_m(M);
    return (size_t)(M / PM) * RC_K * (32 * (PM / 32 + (PM % 32) / 16)) * 8;
}
static bool adc_use_cf(int64_t N, int M) {
    return N >= (1 << 18) && adc_cf_supported(M) && !rc_env_set("RC_ADC_VALU_SCREEN") && !rc_env_set("RC_ADC_OLD_SCREEN");
}
static adc_ws_layout adc_layout(int64_t N, int M, int nq, bool own_image = true) {
    adc_ws_layout L;
    L.S = N < ADC_SAMPLE_MAX ? N : ADC_SAMPLE_MAX;
    size_t o = 0;
    L.lut = o;    o += rc_align_up((size_t)nq * M * RC_K * sizeof(float), 256);
    L.sample = o; o += rc_align_up((size_t)nq * (size_t)L.S * sizeof(float), 256);
    L.thr = o;    o += rc_align_up((size_t)nq * sizeof(float), 256);
    L.cnt = o;    o += rc_align_up((size_t)nq * sizeof(unsigned), 256);
    L.cand = o;   o += rc_align_up((size_t)nq * ADC_CAND_CAP * sizeof(unsigned long long), 256);
    L.qlut = L.tint = L.qstat = L.idcnt = L.ids = o;
    if (N >= ADC_SCREEN_MIN_N) {
        const int QS = adc_qs_for(M);
        size_t qb = (size_t)((nq + QS - 1) / QS) * M * RC_K * QS;
        if (adc_cf_supported(M)) {
            const size_t cb = (size_t)((nq + 7) / 8) * adc_cf_table_bytes(M);
            if (cb > qb) qb = cb;
        }
        L.qlut = o;  o += rc_align_up(qb, 256);
        L.tint = o;  o += rc_align_up((size_t)nq * sizeof(int), 256);
        L.qstat = o; o += rc_align_up((size_t)nq * ADC_QSTAT_STRIDE * sizeof(float), 256);
        L.idcnt = o; o += rc_align_up((size_t)nq * sizeof(unsigned), 256);
        L.ids = o;   o += rc_align_up((size_t)nq * ADC_ID_CAP * sizeof(unsigned), 256);
    }
    L.image = o;
    if (own_image && N >= ADC_SCREEN_MIN_N && adc_cf_supported(M)) o += rc_align_up(rc_adc_scan_image_bytes(N, M), 256);
    L.partial = o;
    L.partial_bytes = adc_part_plan_for(N, M, nq).bytes;
    o += rc_align_up(L.partial_bytes, 256);
    L.total = o;
    return L;
}
static adc_part_plan adc_part_plan_for(int64_t N, int M, int nq) {
    adc_part_plan P = {0u, 0u, 0};
    if (M != 96 || N < ADC_SCREEN_MIN_N || nq <= 0 || rc_env_int("RC_ADC_TWO_PASS", 0) == 0) return P;
    const int64_t tile = adc_cf_tile_rows(M);
    P.nchunks = (unsigned)(((N + tile - 1) / tile) * (tile / 16));
    const size_t per_group = (size_t)P.nchunks * 256;
    const unsigned groups = (unsigned)((nq + 7) / 8);
    size_t gp = ADC_PART_CAP / per_group;
    if (gp < 1) gp = 1;
    P.groups_per_pass = gp < groups ? (unsigned)gp : groups;
    P.bytes = (size_t)P.groups_per_pass * per_group;
    return P;
}

extern "C" size_t rc_adc_search_ws_bytes(int64_t N, int M, int K, int nq, int k) {
    if (N <= 0 || M <= 0 || K != RC_K || nq <= 0 || k <= 0) return 0;
    return adc_layout(N, M, nq, true).total;
}
// workspace when the caller keeps the permuted code image itself (rc_adc_search_img with image != NULL)
extern "C" size_t rc_adc_search_img_ws_bytes(int64_t N, int M, int K, int nq, int k) {
    if (N <= 0 || M <= 0 || K != RC_K || nq <= 0 || k <= 0) return 0;
    return adc_layout(N, M, nq, false).total;
}
// Which M run the 16-query screen (adc_screen_q16_kernel) in the flat search: all that have an image (M = 16/32/48/64/96:
// 3.3 / 6.2 / 9.1 / 12.0 / 17.1 ms per 1200 queries x 8.84 M rows against 5.1 / 7.0 / 10.1 / 12.8 / 26.7 ms for the
// 8-query screen).  The choice fixes the layout of the index's flat-search image, so it is read ONCE per process:
// RC_ADC_Q16=0 selects the 8-query screens (development A/B).
static bool adc_q16_for(int M) {
    static int mode = -1;                                    // 0 none, 1 default set, 2 all
    if (mode < 0) {
        const char* e = getenv("RC_ADC_Q16");
        mode = (!e || !*e) ? 1 : (!strcmp(e, "0") ? 0 : (!strcmp(e, "all") ? 2 : 1));
    }
    if (mode == 0 || !adc_cf_supported(M)) return false;
    (void)mode;
    return true;                                             // every M with an image: faster than the 8-query screen at all of them
}
// the flat-search image of a two-phase M is tile-blocked (whole tiles of adc_cf_tile_rows(M) rows); so is every image of
// the 16-query screen (tiles of ADC_Q16_TILE rows)
static int64_t adc_img_tile(int M) { return adc_q16_for(M) ? ADC_Q16_TILE : (adc_cf_phase_m(M) != M ? adc_cf_tile_rows(M) : 0); }
// bytes of the permuted code image of an N-row index (0: this M has no conflict-free screen, no image is used)
extern "C" size_t rc_adc_scan_image_bytes(int64_t N, int M) {
    if (N < 0 || !adc_cf_supported(M)) return 0;
    const int64_t T = adc_img_tile(M);
    const int64_t rows = T > 0 ? (N + T - 1) / T * T : N;
    return (size_t)rows * M * ADC_IMG_ES;
}
// Host-side description of the conflict-free layout (no GPU involved; what tests/test_abi.py checks): for lane `lane`
// (0..63) of a wave and gather step `step` (0 .. steps_per_lane-1) of one table phase: the 8-byte LDS slot it reads and the
// phase-relative sub-quantiser that slot belongs to.  Returns the number of steps per lane, or RC_ESHAPE.
extern "C" int rc_adc_cf_describe(int M, int lane, int step, int* slot, int* m, int* slots_per_code, int* phases) {
    if (!adc_cf_supported(M)) return RC_ESHAPE;
    const int PM = adc_cf_phase_m(M);
    if (lane < 0 || lane > 63 || step < 0 || step >= PM / 4 || !slot || !m) return RC_EINVAL;
    adc_cf_step(PM, step, lane & 15, lane >> 4, *slot, *m);
    if (slots_per_code) *slots_per_code = 32 * (PM / 32 + (PM % 32) / 16);
    if (phases) *phases = M / PM;
    return PM / 4;
}

// (Re)build rows [n0, n0 + n) of the image from the canonical codes [N, M] (both pointers = row 0 of the index).
static int adc_scan_image_impl(rc_handle_t h, const uint8_t* codes, int64_t n0, int64_t n, int M, uint8_t* image, int64_t tile,
                               rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !image || n0 < 0 || n < 0) return RC_EINVAL;
    if (!adc_cf_supported(M)) return RC_ESHAPE;
    if (n == 0) return RC_OK;
    int64_t blocks = (n * M + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    // rows layout (tile == 0) = the IVF search's image: its table phases may differ from the flat search's
    hipLaunchKernelGGL(adc_scan_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, codes, n0, n, M,
                       tile == 0 ? adc_ivf_phase_m(M) : adc_cf_phase_m(M), image, tile);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}
// flat-search image (what rc_adc_search_img / rc_adc_search_q take): rc_adc_scan_image_bytes(N, M) bytes
extern "C" int rc_adc_scan_image(rc_handle_t h, const uint8_t* codes, int64_t n0, int64_t n, int M, uint8_t* image,
                                 rc_stream_t stream) {
    if (adc_q16_for(M)) {
        rc_device_guard device_guard_(h);
        if (!h || !codes || !image || n0 < 0 || n < 0) return RC_EINVAL;
        if (n == 0) return RC_OK;
        int64_t blocks = (n * M + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(adc_q16_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, codes, n0, n, M, image);
        RC_LAUNCH_CHECK(h);
        return RC_OK;
    }
    return adc_scan_image_impl(h, codes, n0, n, M, image, adc_img_tile(M), stream);
}
// host-side description of the 16-query screen's layout (tests): slot read by `lane` in step j, or RC_ESHAPE
extern "C" int rc_adc_q16_describe(int M, int lane, int step, int* slot) {
    if (!adc_cf_supported(M)) return RC_ESHAPE;
    if (lane < 0 || lane > 63 || step < 0 || step > 3 || !slot) return RC_EINVAL;
    *slot = adc_q16_slot(lane, step);
    return adc_q16_for(M) ? 1 : 0;                           // 1: this M's flat search uses the layout
}
// row-major image [N][M] (what the list-centric IVF search takes: its cells start at arbitrary rows)
static bool ivf_pipe() {                                    // RC_IVF_PIPE=0: the round-2 IVF screen (one block per task)
    static int v = -1;
    if (v < 0) { const char* e = getenv("RC_IVF_PIPE"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
__global__ void ivfs_image_kernel(const uint8_t* __restrict__ codes, int64_t n0, int64_t cnt, int M, uint8_t* __restrict__ image);
// bytes of the IVF image of N rows (whole chunks of 16 rows)
extern "C" size_t rc_adc_scan_image_rows_bytes(int64_t N, int M) {
    if (!adc_cf_supported(M) || N < 0) return 0;
    return (size_t)((N + 15) / 16 * 16) * M * ADC_IMG_ES;
}
// host-side description of that image (no GPU involved): byte offset of codes[n][m], or -1
extern "C" int64_t rc_adc_scan_image_rows_at(int M, int64_t n, int m) {
    if (!adc_cf_supported(M) || n < 0 || m < 0 || m >= M) return -1;
    if (!ivf_pipe()) {                                      // round-2 layout: row-major, adc_cf_step order inside the row
        const int PM = adc_ivf_phase_m(M), ph = m / PM;
        for (int g = 0; g < 4; ++g)
            for (int st = 0; st < PM / 4; ++st) {
                int slot, mm;
                adc_cf_step(PM, st, (int)(n & 15), g, slot, mm);
                if (mm == m - ph * PM) return n * M + ph * PM + g * (PM / 4) + st;
            }
        return -1;
    }
    const int p = m / 32, PM = (M - 32 * p) >= 32 ? 32 : 16;
    for (int g = 0; g < 4; ++g)
        for (int st = 0; st < PM / 4; ++st) {
            int slot, mm;
            adc_cf_step(PM, st, (int)(n & 15), g, slot, mm);
            if (mm == m - 32 * p) return (n >> 4) * (int64_t)(16 * M) + 16 * 32 * p + (g * 16 + (int)(n & 15)) * (PM / 4) + st;
        }
    return -1;
}
extern "C" int rc_adc_scan_image_rows(rc_handle_t h, const uint8_t* codes, int64_t n0, int64_t n, int M, uint8_t* image,
                                      rc_stream_t stream) {
    if (!ivf_pipe()) return adc_scan_image_impl(h, codes, n0, n, M, image, 0, stream);
    rc_device_guard device_guard_(h);
    if (!h || !codes || !image || n0 < 0 || n < 0) return RC_EINVAL;
    if (!adc_cf_supported(M)) return RC_ESHAPE;
    if (n == 0) return RC_OK;
    int64_t blocks = (n * M + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(ivfs_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, codes, n0, n, M, image);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

static int adc_qt_for(int M) {
    if (M <= 32) return 4;   // <= 128 KiB of tables
    if (M <= 64) return 2;   // M=48: 96 KiB, M=64: 128 KiB
    return 1;                // M=96: 96 KiB
}

struct adc_bufs {
    float* lut; float* sample; float* thr; unsigned* cnt; unsigned long long* cand;
    uint8_t* qlut; int* tint; unsigned* idcnt; unsigned* ids; float* qstat; short* partial;
};

template <int M, int QT>
static int adc_launch_scans(rc_handle_t h, const uint8_t* codes, const uint8_t* image, int64_t N, int nq, int64_t S,
                            const adc_bufs& b, int r, int k, int* status, int* qstatus, hipStream_t s) {
    const size_t lds = (size_t)M * RC_K * QT * sizeof(float);
    const unsigned qg = (unsigned)((nq + QT - 1) / QT);
    auto ksample = adc_scan_kernel<M, QT, ADC_SAMPLE>;
    auto kfilter = adc_scan_kernel<M, QT, ADC_FILTER>;
    RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)ksample, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ksample, dim3(qg, (unsigned)((S + ADC_TILE_DOCS - 1) / ADC_TILE_DOCS)), dim3(ADC_THREADS), lds, s,
                       codes, N, b.lut, nq, S, b.sample, b.thr, b.cnt, b.cand);
    RC_LAUNCH_CHECK(h);
    const size_t tl = (size_t)S * sizeof(unsigned);
    RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)adc_threshold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)tl));
    hipLaunchKernelGGL(adc_threshold_kernel, dim3((unsigned)nq), dim3(1024), tl, s, b.sample, S, r, b.thr);
    RC_LAUNCH_CHECK(h);
    const unsigned tiles = (unsigned)((N + ADC_TILE_DOCS - 1) / ADC_TILE_DOCS);
    // small indexes: exact scan (the screen's fixed costs do not pay)
    if (N < ADC_SCREEN_MIN_N) {
        RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kfilter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
        hipLaunchKernelGGL(kfilter, dim3(qg, tiles), dim3(ADC_THREADS), lds, s, codes, N, b.lut, nq, S, b.sample, b.thr,
                           b.cnt, b.cand);
        rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
        RC_LAUNCH_CHECK(h);
        return RC_OK;
    }
    RC_HIP_CHECK(h, hipMemsetAsync(b.idcnt, 0, (size_t)nq * sizeof(unsigned), s));
    // Screen variant: 8 queries per gather on the matrix cores (tables in LDS: one pass for M <= 64, two half-table
    // phases above); M % 8 != 0 and the A/B switches RC_ADC_VALU_SCREEN / RC_ADC_ONE_PHASE use the older kernels.
    const bool valu_screen = rc_env_set("RC_ADC_VALU_SCREEN");
    const bool one_phase = rc_env_set("RC_ADC_ONE_PHASE");
    auto screen = [&](auto kern, int QS, size_t sl) -> int {
        hipLaunchKernelGGL(adc_qlut_kernel, dim3((unsigned)nq), dim3(RC_K), 0, s, b.lut, b.thr, M, QS, b.qlut, b.tint);
        RC_LAUNCH_CHECK(h);
        RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sl));
        rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
        hipLaunchKernelGGL(kern, dim3((unsigned)((nq + QS - 1) / QS), tiles), dim3(ADC_THREADS), sl, s, codes, N, b.qlut,
                           b.tint, nq, b.idcnt, b.ids);
        rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
        RC_LAUNCH_CHECK(h);
        return RC_OK;
    };
    constexpr int QS1 = (M <= 64) ? 8 : 4;                  // one-pass kernels: M * 256 * QS bytes of LDS
    int src = RC_OK;
    constexpr bool CF = (M == 16 || M == 32 || M == 48 || M == 64 || M == 96);
    if (CF && image != nullptr && adc_q16_for(M)) {
        if constexpr (CF) {
            // 16 queries per ds_read_b128 gather, phases of 16 sub-quantisers, double-buffered tables (adc_screen_q16_kernel)
            auto kern = adc_screen_q16_kernel<M>;
            constexpr int TH = ADC_Q16_WAVES * 64;
            constexpr int sl = 2 * RC_K * 256 + ADC_Q16_WAVES * ADC_Q16_SCAP * 4;      // two table buffers + the survivor lists
            const int groups = (nq + 15) / 16;
            hipLaunchKernelGGL(adc_qstats_kernel, dim3((unsigned)nq), dim3(RC_K), 0, s, b.lut, b.thr, M, b.qstat, b.tint);
            RC_LAUNCH_CHECK(h);
            hipLaunchKernelGGL(adc_qlut16_write_kernel, dim3((unsigned)groups, RC_K / 64, M / 16), dim3(64, 16), 0, s, b.lut,
                               (const float*)b.qstat, M, nq, b.qlut);
            RC_LAUNCH_CHECK(h);
            RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, sl));
            const unsigned tiles16 = (unsigned)((N + ADC_Q16_TILE - 1) / ADC_Q16_TILE);
            const unsigned gpx = (unsigned)(groups + 7) / 8u;
            const unsigned nblocks = groups >= 8 ? 8u * gpx * tiles16 : 8u * (unsigned)groups * ((tiles16 + 7u) / 8u);
            rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
            hipLaunchKernelGGL(kern, dim3(nblocks), dim3(TH), sl, s, image, N, b.qlut, b.tint, nq, groups, b.idcnt, b.ids,
                               rc_env_int("RC_ADC_Q16_PRIO", 1));
            rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
            RC_LAUNCH_CHECK(h);
        }
    } else if (CF && image != nullptr) {
        if constexpr (CF) {
            constexpr int NP = (M == 96) ? 2 : 1, PM = M / NP;
            constexpr int R = (NP > 1) ? 8 : ((M == 64 || (ADC_IMG_ES == 2 && M == 48)) ? 2 : 4);   // register budget
            constexpr int TH = ADC_THREADS;
            constexpr int sl = adc_cf<PM>::TABLE_BYTES;
            hipLaunchKernelGGL(adc_qstats_kernel, dim3((unsigned)nq), dim3(RC_K), 0, s, b.lut, b.thr, M, b.qstat, b.tint);
            RC_LAUNCH_CHECK(h);
            hipLaunchKernelGGL(adc_qlut_cf_write_kernel<PM>, dim3((unsigned)((nq + 7) / 8), RC_K / 64, 4), dim3(64), 0, s, b.lut,
                               (const float*)b.qstat, M, nq, b.qlut);
            RC_LAUNCH_CHECK(h);
            const unsigned cf_tiles = (unsigned)((N + adc_cf_tile_rows(M) - 1) / adc_cf_tile_rows(M));
            const unsigned groups = (unsigned)((nq + 7) / 8);
            bool done = false;
            if constexpr (NP == 2) {
                // two passes with resident tables and the partial sums through HBM (see adc_part_args): opt-in with
                // RC_ADC_TWO_PASS=1 (and a workspace sized with it).  [MI355X, round 3] 1200 queries x 8.84 M rows: pass 1
                // 3 x 3.8 ms + pass 2 3 x 4.7 ms = 25.5 ms against 26.7 ms for the one-launch form below — the barriers
                // were NOT what the two-phase screen pays over 2 x 9.8 ms (the M = 48 kernel): each pass, with the M = 48
                // kernel's schedule, resident tables and a dense 48-byte-row image, is still 16 % / 44 % slower than that
                // kernel.  1.5 % for 8 GB of workspace: not the default.
                const adc_part_plan pp = adc_part_plan_for(N, M, nq);
                if (b.partial && pp.groups_per_pass > 0) {
                    // chunks per wave and round: pass 2 carries the round's partial sums besides the M = 48 kernel's registers
                    // (R = 4: 128 VGPRs + 36 bytes of scratch; R = 2: no spill); the layout of the partial sums does not
                    // depend on R, so the passes may differ.  RC_ADC_PART_R1 / _R2 = 2 | 4 for A/B runs.
                    auto k1 = rc_env_int("RC_ADC_PART_R1", 4) == 2 ? adc_screen_cf_kernel<M, NP, 2, false, TH, 1>
                                                                   : adc_screen_cf_kernel<M, NP, 4, false, TH, 1>;
                    auto k2 = rc_env_int("RC_ADC_PART_R2", 4) == 4 ? adc_screen_cf_kernel<M, NP, 4, false, TH, 2>
                                                                   : adc_screen_cf_kernel<M, NP, 2, false, TH, 2>;
                    RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, sl));
                    RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, sl));
                    rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
                    for (unsigned g0 = 0; g0 < groups; g0 += pp.groups_per_pass) {
                        const unsigned ng = groups - g0 < pp.groups_per_pass ? groups - g0 : pp.groups_per_pass;
                        const adc_part_args pa = {b.partial, g0, pp.nchunks};
                        hipLaunchKernelGGL(k1, dim3(ng, cf_tiles), dim3(TH), sl, s, image, N, b.qlut, b.tint, nq, b.idcnt, b.ids,
                                           adc_ivf_tasks{}, pa);
                        hipLaunchKernelGGL(k2, dim3(ng, cf_tiles), dim3(TH), sl, s, image, N, b.qlut, b.tint, nq, b.idcnt, b.ids,
                                           adc_ivf_tasks{}, pa);
                    }
                    rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
                    RC_LAUNCH_CHECK(h);
                    done = true;
                }
            }
            if (!done) {
                auto kern = adc_screen_cf_kernel<M, NP, R, false, TH>;
                RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, sl));
                rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
                hipLaunchKernelGGL(kern, dim3(groups, cf_tiles), dim3(TH), sl, s, image, N, b.qlut,
                                   b.tint, nq, b.idcnt, b.ids, adc_ivf_tasks{}, adc_part_args{});
                rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
                RC_LAUNCH_CHECK(h);
            }
        }
    } else if constexpr (M % 8 == 0 && M > 64) {
        if (!valu_screen && !one_phase) src = screen(adc_screen_mfma2_kernel<M, 8, 2>, 8, (size_t)(M / 2) * RC_K * 8);
        else if (!valu_screen) src = screen(adc_screen_mfma_kernel<M, QS1>, QS1, (size_t)M * RC_K * QS1);
        else src = screen(adc_screen_kernel<M, QS1>, QS1, (size_t)M * RC_K * QS1);
    } else if constexpr (M % 16 == 0) {
        // 16 queries per ds_read_b128 in two phases: measured SLOWER than the one-pass 8-query kernel at M = 48
        // (66-70 k vs 75-78 k queries/s: a b128 gather costs as many LDS cycles per query as a b64 one, and the
        // table refills come on top); kept behind RC_ADC_Q16 for experiments.
        const bool q16 = rc_env_set("RC_ADC_Q16");
        if (!valu_screen && q16) src = screen(adc_screen_mfma2_kernel<M, 16, 2>, 16, (size_t)(M / 2) * RC_K * 16);
        else if (!valu_screen) src = screen(adc_screen_mfma_kernel<M, QS1>, QS1, (size_t)M * RC_K * QS1);
        else src = screen(adc_screen_kernel<M, QS1>, QS1, (size_t)M * RC_K * QS1);
    } else if constexpr (M % 8 == 0) {
        if (!valu_screen) src = screen(adc_screen_mfma_kernel<M, QS1>, QS1, (size_t)M * RC_K * QS1);
        else src = screen(adc_screen_kernel<M, QS1>, QS1, (size_t)M * RC_K * QS1);
    } else {
        src = screen(adc_screen_kernel<M, QS1>, QS1, (size_t)M * RC_K * QS1);
    }
    if (src != RC_OK) return src;
    auto krescore = adc_rescore_kernel<M>;
    const size_t rl = (size_t)M * RC_K * sizeof(float);
    RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)krescore, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl));
    hipLaunchKernelGGL(krescore, dim3((unsigned)nq), dim3(adc_rescore_threads(M)), rl, s, codes, b.lut, b.thr, b.idcnt, b.ids, b.cnt, b.cand,
                       status, (const int64_t*)nullptr, qstatus);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

#define ADC_CASE(MM, QQ) \
    case MM: rc = adc_launch_scans<MM, QQ>(h, codes, image, N, nq, L.S, bufs, r, k, status, qstatus, s); break;

extern "C" int rc_adc_lut(rc_handle_t h, const float* C, const float* q, int nq, int D, int M, int K, float* lut,
                          rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !C || !q || !lut || nq < 0 || M <= 0 || D <= 0) return RC_EINVAL;
    if (K != RC_K || D % M != 0) return RC_ESHAPE;
    if (nq == 0) return RC_OK;
    const dim3 cg((unsigned)((nq + ADC_LUT_QCHUNK - 1) / ADC_LUT_QCHUNK), (unsigned)M);
    switch (D / M) {
#define ADC_LUT_CASE(DS)                                                                                              \
        case DS:                                                                                                      \
            hipLaunchKernelGGL(adc_lut_rows_kernel<DS>, cg, dim3(RC_K), 0, (hipStream_t)stream, C, q, nq, D, M, lut); \
            break;
        ADC_LUT_CASE(8) ADC_LUT_CASE(12) ADC_LUT_CASE(16) ADC_LUT_CASE(24) ADC_LUT_CASE(32) ADC_LUT_CASE(48)
        ADC_LUT_CASE(64) ADC_LUT_CASE(96)
#undef ADC_LUT_CASE
        default:
            hipLaunchKernelGGL(adc_lut_kernel, dim3((unsigned)nq, (unsigned)M), dim3(RC_K), 0, (hipStream_t)stream, C, q, D,
                               M, lut);
    }
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// sort + emit stage, shared with the IVF path (ivf_search.hip)
int rc_adc_launch_select(rc_handle_t h, unsigned long long* cand, const unsigned* cnt, int nq, int64_t N, int k,
                         int64_t id_offset, float* scores, int64_t* ids, int* status, hipStream_t s, int* qstatus = nullptr) {
    int cap = 4096;                                            // keys held in LDS: >= max(2048, 2 k), see adc_select_kernel
    while (cap < 2 * k && cap < ADC_CAND_CAP) cap <<= 1;
    if (const int e = rc_env_int("RC_ADC_SELECT_CAP", 0)) cap = e;        // tests: 1024 forces the global-memory sort
    const size_t ss = (size_t)(cap + cap / 32) * sizeof(unsigned long long);       // padded positions, adc_sp
    const int nthr = cap <= 8192 ? 512 : 1024;
    RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)adc_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)((ADC_CAND_CAP + ADC_CAND_CAP / 32) * sizeof(unsigned long long))));
    hipLaunchKernelGGL(adc_select_kernel, dim3((unsigned)nq), dim3(nthr), ss, s, cand, cnt, N, k, id_offset, scores, ids,
                       status, qstatus, cap);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

extern "C" int rc_adc_search(rc_handle_t h, const uint8_t* codes, int64_t N, int M, int K, const float* C, int D,
                             const float* q, int nq, int k, int64_t id_offset, double sel_slack, float* scores,
                             int64_t* ids, int* status, void* ws, size_t ws_bytes, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    return rc_adc_search_img(h, codes, nullptr, N, M, K, C, D, q, nq, k, id_offset, sel_slack, scores, ids, status, ws,
                             ws_bytes, stream);
}

// The search proper.  scan_image: the index's permuted code image (rc_adc_scan_image) or NULL — then, where the
// conflict-free screen applies, the image is rebuilt in the workspace on every call (one extra pass over the codes).
extern "C" int rc_adc_search_q(rc_handle_t h, const uint8_t* codes, const uint8_t* scan_image, int64_t N, int M, int K,
                               const float* C, int D, const float* q, int nq, int k, int64_t id_offset, double sel_slack,
                               float* scores, int64_t* ids, int* status, int* qstatus, void* ws, size_t ws_bytes,
                               rc_stream_t stream);

extern "C" int rc_adc_search_img(rc_handle_t h, const uint8_t* codes, const uint8_t* scan_image, int64_t N, int M, int K,
                                 const float* C, int D, const float* q, int nq, int k, int64_t id_offset,
                                 double sel_slack, float* scores, int64_t* ids, int* status, void* ws, size_t ws_bytes,
                                 rc_stream_t stream) {
    return rc_adc_search_q(h, codes, scan_image, N, M, K, C, D, q, nq, k, id_offset, sel_slack, scores, ids, status, nullptr,
                           ws, ws_bytes, stream);
}

// qstatus: NULL, or nq ints (zeroed by the caller) that receive the status bits PER QUERY (bit0 too few candidates, bit1 a
// list overflowed), so that a caller repeats or re-routes only the queries concerned (rc_adc_search_exact never fails).
extern "C" int rc_adc_search_q(rc_handle_t h, const uint8_t* codes, const uint8_t* scan_image, int64_t N, int M, int K,
                               const float* C, int D, const float* q, int nq, int k, int64_t id_offset, double sel_slack,
                               float* scores, int64_t* ids, int* status, int* qstatus, void* ws, size_t ws_bytes,
                               rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !C || !q || !scores || !ids || !status || N <= 0 || nq < 0 || k <= 0 || M <= 0 || D <= 0)
        return RC_EINVAL;
    if (K != RC_K || D % M != 0 || N > 0xFFFFFFFFll || k > ADC_CAND_CAP / 2) return RC_ESHAPE;
    if (nq == 0) return RC_OK;
    const adc_ws_layout L = adc_layout(N, M, nq, scan_image == nullptr);
    if (!ws || ws_bytes < L.total) return RC_EWORKSPACE;
    char* w = (char*)ws;
    const uint8_t* image = nullptr;
    if (adc_use_cf(N, M)) {
        image = scan_image;
        if (!image) {
            const int irc = rc_adc_scan_image(h, codes, 0, N, M, (uint8_t*)(w + L.image), stream);
            if (irc != RC_OK) return irc;
            image = (const uint8_t*)(w + L.image);
        }
    }
    float* lut = (float*)(w + L.lut);
    unsigned* cnt = (unsigned*)(w + L.cnt);
    unsigned long long* cand = (unsigned long long*)(w + L.cand);
    const adc_bufs bufs = {lut, (float*)(w + L.sample), (float*)(w + L.thr), cnt, cand, (uint8_t*)(w + L.qlut),
                           (int*)(w + L.tint), (unsigned*)(w + L.idcnt), (unsigned*)(w + L.ids), (float*)(w + L.qstat),
                           L.partial_bytes ? (short*)(w + L.partial) : nullptr};
    hipStream_t s = (hipStream_t)stream;
    int rc = rc_adc_lut(h, C, q, nq, D, M, K, lut, stream);
    if (rc != RC_OK) return rc;
    RC_HIP_CHECK(h, hipMemsetAsync(cnt, 0, (size_t)nq * sizeof(unsigned), s));
    // rank of the sample score used as the filter threshold
    int r;
    if (N <= ADC_CAND_CAP) {
        r = 0;  // tau = -inf: every row is a candidate, the select kernel sorts them all
    } else if (L.S == N) {
        r = k;  // the sample is the whole index: tau is the exact k-th score
    } else {
        const double mu = (double)k * (double)L.S / (double)N;
        r = (int)(mu + sel_slack * sqrt(mu + 1.0) + 4.0) + 1;
        // large k: keep the expected candidate count (r N / S) below ~80 % of the list capacity as long as that still
        // leaves 2.5 sigma of head-room over k
        const double r_cap = 0.8 * (double)ADC_CAND_CAP * (double)L.S / (double)N;
        if ((double)r > r_cap && r_cap >= mu + 2.5 * sqrt(mu + 1.0) + 2.0) r = (int)r_cap;
        if (r > L.S) r = (int)L.S;
        if (r < 1) r = 1;       // a (hugely) negative slack: the best sample score
    }
    switch (M) {
        ADC_CASE(8, 4) ADC_CASE(12, 4) ADC_CASE(16, 4) ADC_CASE(24, 4) ADC_CASE(32, 4)
        ADC_CASE(48, 2) ADC_CASE(64, 2) ADC_CASE(96, 1)
        default: return RC_ESHAPE;
    }
    if (rc != RC_OK) return rc;
    (void)adc_qt_for;
    return rc_adc_launch_select(h, cand, cnt, nq, N, k, id_offset, scores, ids, status, s, qstatus);
}

// ------------------------------------------------------------------------------------------ exact search (never fails)
// evaluate_repconc.py:180-185 relies on Faiss's IndexPQ.search, which returns for ANY index content.  The fast path above
// places a candidate threshold from a sample and can, on degenerate data (thousands of rows with identical codes, all rows
// tied), keep too few or too many candidates however the slack is set.  This path has no such failure mode: exact fp32
// scores of every row (adc_scan_kernel<SAMPLE> with the sample = the whole index), then the k-th largest 64-bit key
// (ordered(score) << 32 | ~row: distinct for distinct rows, so "the k best in (score desc, id asc) order" is a unique set)
// by an 8-pass byte-wise radix select over all rows, then a compaction of the keys >= that key (exactly min(k, N) of them)
// and the ordinary sort + emit.  Cost: N x 4 bytes of scores per query and ~10 passes over them — for the handful of
// queries the fast path hands over, not for whole batches.
#define ADC_EXACT_QX 8             // queries per round (scores [QX][N] fp32 in the workspace)
__device__ __forceinline__ unsigned long long adc_exact_key(float s, int64_t i) {
    return ((unsigned long long)adc_order_key(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
}
// grid (slices, queries of the round): histogram of byte `pass` (0 = most significant) over the keys whose higher bytes
// equal prefix[q]
__global__ __launch_bounds__(256) void adc_exact_hist_kernel(const float* __restrict__ sc, int64_t N,
                                                             const unsigned long long* __restrict__ prefix, int pass,
                                                             unsigned* __restrict__ hist) {
    __shared__ unsigned h[256];
    const int qx = blockIdx.y, tid = threadIdx.x;
    h[tid] = 0u;
    __syncthreads();
    const unsigned long long pf = prefix[qx];
    const int shift = 56 - 8 * pass;
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    const float* row = sc + (size_t)qx * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < N; i += (int64_t)gridDim.x * 256) {
        const unsigned long long key = adc_exact_key(row[i], i);
        if ((key & himask) == pf) atomicAdd(&h[(unsigned)(key >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    if (h[tid]) atomicAdd(hist + (size_t)qx * 256 + tid, h[tid]);
}
// one block of 256 threads per query: the bin that holds the rank-th largest key; prefix and rank move on, hist is zeroed
__global__ __launch_bounds__(256) void adc_exact_pick_kernel(unsigned* __restrict__ hist, unsigned long long* __restrict__ prefix,
                                                             unsigned* __restrict__ rank, int pass) {
    __shared__ unsigned s_scan[4];
    __shared__ unsigned sel_prefix, sel_rank;
    const int qx = blockIdx.x, tid = threadIdx.x;
    unsigned* hq = hist + (size_t)qx * 256;
    const unsigned need = rank[qx];
    if (tid == 0) { sel_prefix = 0u; sel_rank = need; }
    __syncthreads();
    adc_pick_bin(hq, need, 0u, 0, s_scan, &sel_prefix, &sel_rank);     // bin index lands in sel_prefix (shift 0, prefix 0)
    if (tid == 0) {
        prefix[qx] |= (unsigned long long)(sel_prefix & 0xFFu) << (56 - 8 * pass);
        rank[qx] = sel_rank;
    }
    hq[tid] = 0u;
}
__global__ __launch_bounds__(256) void adc_exact_init_kernel(unsigned* __restrict__ hist, unsigned long long* __restrict__ prefix,
                                                             unsigned* __restrict__ rank, unsigned* __restrict__ cnt, unsigned want) {
    const int qx = blockIdx.x, tid = threadIdx.x;
    hist[(size_t)qx * 256 + tid] = 0u;
    if (tid == 0) { prefix[qx] = 0ull; rank[qx] = want; cnt[qx] = 0u; }
}
// keys >= the selected key (= the min(k, N) best rows) go to the candidate list
__global__ __launch_bounds__(256) void adc_exact_collect_kernel(const float* __restrict__ sc, int64_t N,
                                                                const unsigned long long* __restrict__ prefix,
                                                                unsigned* __restrict__ cand_count,
                                                                unsigned long long* __restrict__ cand) {
    const int qx = blockIdx.y, tid = threadIdx.x;
    const unsigned long long kth = prefix[qx];
    const float* row = sc + (size_t)qx * N;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < N; i0 += (int64_t)gridDim.x * 256) {      // wave-uniform trip count
        const int64_t i = i0 + tid;
        const unsigned long long key = i < N ? adc_exact_key(row[i], i) : 0ull;
        const bool pass = i < N && key >= kth;
        const unsigned long long mask = __ballot(pass);
        if (mask) {
            const int lane = tid & 63;
            unsigned base = 0;
            if (lane == (int)__builtin_ctzll(mask)) base = atomicAdd(cand_count + qx, (unsigned)__popcll(mask));
            base = __shfl(base, (int)__builtin_ctzll(mask));
            const unsigned slot = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
            if (pass && slot < ADC_CAND_CAP) cand[(size_t)qx * ADC_CAND_CAP + slot] = key;
        }
    }
}

struct adc_exact_layout { size_t lut, sc, hist, prefix, rank, cnt, cand, status, total; };
static adc_exact_layout adc_exact_ws(int64_t N, int M, int nq) {
    adc_exact_layout L;
    size_t o = 0;
    const int qx = nq < ADC_EXACT_QX ? nq : ADC_EXACT_QX;
    L.lut = o;    o += rc_align_up((size_t)nq * M * RC_K * sizeof(float), 256);
    L.sc = o;     o += rc_align_up((size_t)qx * (size_t)N * sizeof(float), 256);
    L.hist = o;   o += rc_align_up((size_t)qx * 256 * sizeof(unsigned), 256);
    L.prefix = o; o += rc_align_up((size_t)qx * sizeof(unsigned long long), 256);
    L.rank = o;   o += rc_align_up((size_t)qx * sizeof(unsigned), 256);
    L.cnt = o;    o += rc_align_up((size_t)qx * sizeof(unsigned), 256);
    L.cand = o;   o += rc_align_up((size_t)qx * ADC_CAND_CAP * sizeof(unsigned long long), 256);
    L.status = o; o += 256;
    L.total = o;
    return L;
}
extern "C" size_t rc_adc_search_exact_ws_bytes(int64_t N, int M, int K, int nq, int k) {
    if (N <= 0 || M <= 0 || K != RC_K || nq <= 0 || k <= 0) return 0;
    return adc_exact_ws(N, M, nq).total;
}

template <int M, int QT>
static int adc_exact_scores(rc_handle_t h, const uint8_t* codes, int64_t N, const float* lut, int nq, float* sc, hipStream_t s) {
    const size_t lds = (size_t)M * RC_K * QT * sizeof(float);
    auto kern = adc_scan_kernel<M, QT, ADC_SAMPLE>;
    RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((nq + QT - 1) / QT), (unsigned)((N + ADC_TILE_DOCS - 1) / ADC_TILE_DOCS)),
                       dim3(ADC_THREADS), lds, s, codes, N, lut, nq, N, sc, (const float*)nullptr, (unsigned*)nullptr,
                       (unsigned long long*)nullptr);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

extern "C" int rc_adc_search_exact(rc_handle_t h, const uint8_t* codes, int64_t N, int M, int K, const float* C, int D,
                                   const float* q, int nq, int k, int64_t id_offset, float* scores, int64_t* ids, void* ws,
                                   size_t ws_bytes, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !C || !q || !scores || !ids || N <= 0 || nq < 0 || k <= 0 || M <= 0 || D <= 0) return RC_EINVAL;
    if (K != RC_K || D % M != 0 || N > 0xFFFFFFFFll || k > ADC_CAND_CAP / 2) return RC_ESHAPE;
    if (nq == 0) return RC_OK;
    const adc_exact_layout L = adc_exact_ws(N, M, nq);
    if (!ws || ws_bytes < L.total) return RC_EWORKSPACE;
    char* w = (char*)ws;
    hipStream_t s = (hipStream_t)stream;
    float* lut = (float*)(w + L.lut);
    float* sc = (float*)(w + L.sc);
    unsigned* hist = (unsigned*)(w + L.hist);
    unsigned long long* prefix = (unsigned long long*)(w + L.prefix);
    unsigned* rank = (unsigned*)(w + L.rank);
    unsigned* cnt = (unsigned*)(w + L.cnt);
    unsigned long long* cand = (unsigned long long*)(w + L.cand);
    int* status = (int*)(w + L.status);
    int rc = rc_adc_lut(h, C, q, nq, D, M, K, lut, stream);
    if (rc != RC_OK) return rc;
    const unsigned want = (unsigned)((int64_t)k < N ? (int64_t)k : N);
    unsigned slices = (unsigned)((N + 256 * 64 - 1) / (256 * 64));
    if (slices > 2048) slices = 2048;
    for (int q0 = 0; q0 < nq; q0 += ADC_EXACT_QX) {
        const int nx = nq - q0 < ADC_EXACT_QX ? nq - q0 : ADC_EXACT_QX;
        const float* lq = lut + (size_t)q0 * M * RC_K;
        switch (M) {
#define ADC_EXACT_CASE(MM, QQ) case MM: rc = adc_exact_scores<MM, QQ>(h, codes, N, lq, nx, sc, s); break;
            ADC_EXACT_CASE(8, 4) ADC_EXACT_CASE(12, 4) ADC_EXACT_CASE(16, 4) ADC_EXACT_CASE(24, 4) ADC_EXACT_CASE(32, 4)
            ADC_EXACT_CASE(48, 2) ADC_EXACT_CASE(64, 2) ADC_EXACT_CASE(96, 1)
#undef ADC_EXACT_CASE
            default: return RC_ESHAPE;
        }
        if (rc != RC_OK) return rc;
        hipLaunchKernelGGL(adc_exact_init_kernel, dim3((unsigned)nx), dim3(256), 0, s, hist, prefix, rank, cnt, want);
        RC_LAUNCH_CHECK(h);
        for (int pass = 0; pass < 8; ++pass) {
            hipLaunchKernelGGL(adc_exact_hist_kernel, dim3(slices, (unsigned)nx), dim3(256), 0, s, (const float*)sc, N,
                               (const unsigned long long*)prefix, pass, hist);
            RC_LAUNCH_CHECK(h);
            hipLaunchKernelGGL(adc_exact_pick_kernel, dim3((unsigned)nx), dim3(256), 0, s, hist, prefix, rank, pass);
            RC_LAUNCH_CHECK(h);
        }
        hipLaunchKernelGGL(adc_exact_collect_kernel, dim3(slices, (unsigned)nx), dim3(256), 0, s, (const float*)sc, N,
                           (const unsigned long long*)prefix, cnt, cand);
        RC_LAUNCH_CHECK(h);
        rc = rc_adc_launch_select(h, cand, cnt, nx, N, k, id_offset, scores + (size_t)q0 * k, ids + (size_t)q0 * k, status, s);
        if (rc != RC_OK) return rc;
    }
    return RC_OK;
}

// =============================================================================================== IVF, list-centric
// Search of the cell-major IVF index (csrc/ivf_search.hip; a build-side extension, the reference has one list) with the
// machinery of the flat search.  Round 1 scanned every probed cell once per query (one block per query slice, fp32
// tables, dense score write-out + radix select over it).  Here the work is organised by CELL: all queries probing a cell
// are split into groups of up to 8, one block per (cell, group) TASK runs the conflict-free 8-bit screen over the cell's
// rows — 8 queries share every gather, the cell's codes are read once per group — and the survivors are re-scored
// exactly and selected like in the flat search:
//   1. adc_lut                     fp32 tables of every query (caller)
//   2. ivf_sample_scan_kernel      exact scores of every SS-th row of the query's probed cells -> sample[q][..]
//   3. ivf_rank_select_kernel      tau_q = rank_q-th largest sample score (rank 0: -inf, every probed row is a candidate)
//   4. adc_qstats_kernel + adc_qbyte_write_kernel   per query: 8-bit tables in slot layout + integer threshold
//   5. adc_screen_cf_kernel<IVF>   per task: transpose the 8 queries' byte tables into LDS, screen the cell's rows
//   6. adc_rescore_kernel          exact fp32 score of the survivors, keys carry the corpus position of the row
//   7. adc_select_kernel           top-k, (score desc, corpus id asc) — the tie rule of the flat search
// The host builds the task list (cells sorted, 8 queries per task) and the sample ranks; status bit0 = a query kept
// fewer than min(k, rows probed) candidates (retry with more slack), bit1 = a list overflowed (less slack).

// ------------------------------------------------------------------------------------ 5'. pipelined IVF screen (round 3)
// A wall-clock trace of the screen above on the BASELINE configs[3] shape (M = 96, 5000 cells of ~1770 rows, nprobe 128:
// 19 k tasks of 8 queries; tools/_exp/ivf_trace.py) showed where a task's 15.9 us go: 1.6 us of dependent scalar loads
// (task -> queries -> thresholds), 2.9 + 4.3 us for the two synchronous table fills (128 KiB each: loads from the
// memory-side cache, byte transposes, a block-wide barrier either side), 2.2 + 1.6 us of gathers and 3.1 us for the
// returning atomics of the survivor slots — with one 128 KiB block per CU nothing overlaps any of it.  Two blocks per CU
// (three 64 KiB phases) measured the same: more fills and barriers eat what the overlap gives.
// This kernel keeps ONE persistent block per CU and overlaps by construction:
//   * table phases of 32 sub-quantisers (+ one of 16 for M = 16 / 48): 64 KiB, TWO buffers.  The next stage's tables are
//     requested (global loads into 16 registers) before the current stage's gathers start and are transposed into the other
//     buffer after them: one barrier per stage, no load latency on the critical path;
//   * the block walks its tasks (XCD x owns a contiguous eighth of the cell-ordered task list, its blocks take the tasks
//     round-robin so that the tasks of one cell run side by side in one L2); task descriptors are read two tasks ahead,
//     thresholds one task ahead;
//   * the codes of the next stage are requested right after the current stage's last gather (same registers);
//   * survivors: the wave writes them to its LDS list, issues ONE atomic per (wave, query) for the slots and moves on; the
//     list is copied out one task later, when the atomic has long returned.  (A wave that keeps more than its list holds —
//     queries that keep every row — takes the synchronous path.)
// The per-query byte tables are stored biased (b ^ 0x80) by ivf_qbyte_write_kernel; image: [row][phase][g][step].
#define IVFS_WAVES 16
#define IVFS_THREADS (64 * IVFS_WAVES)
#define IVFS_R 8
#define IVFS_BUF 65536
#define IVFS_MAX_BLOCKS 256      // persistent blocks (one per CU); sizes the survivor streams of the workspace
#ifndef IVFS_PRIO
#define IVFS_PRIO 1
#endif
__host__ __device__ constexpr int ivfs_phases(int M) { return (M + 31) / 32; }
__host__ __device__ constexpr int ivfs_pm(int M, int p) { return (M - 32 * p) >= 32 ? 32 : 16; }

// Image of the list-centric IVF search, blocked by chunks of 16 rows (the unit a wave gathers for): chunk n / 16 holds
// [phase p][lane quarter g][row n mod 16][step s] = codes[n][32 p + m(s; n mod 16, g)], i.e. a wave's load of one chunk and
// phase is 64 lanes x PMp / 4 bytes of CONTIGUOUS memory (with row-major rows it was sixteen 32-byte pieces 96 bytes apart:
// 12-16 cache lines per instruction, and the sixteen waves of a block issue theirs at the same moment).
__host__ __device__ inline int64_t ivfs_image_at(int M, int64_t n, int p, int g, int st) {
    const int PM = ivfs_pm(M, p);
    return (n >> 4) * (int64_t)(16 * M) + (int64_t)(16 * 32 * p) + (int64_t)((g * 16 + (int)(n & 15)) * (PM / 4) + st);
}
__global__ __launch_bounds__(256) void ivfs_image_kernel(const uint8_t* __restrict__ codes, int64_t n0, int64_t cnt, int M,
                                                         uint8_t* __restrict__ image) {
    const int64_t total = cnt * M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = n0 + i / M;
        const int pos = (int)(i % M);
        const int p = pos / 32, rem = pos % 32, PM = ivfs_pm(M, p);
        const int g = rem / (PM / 4), st = rem % (PM / 4);
        int slot, m;
        adc_cf_step(PM, st, (int)(n & 15), g, slot, m);
        image[ivfs_image_at(M, n, p, g, st)] = codes[n * M + 32 * p + m];
    }
}

// per-query byte tables, [phase][code][PMp] one biased byte per sub-quantiser (phase p starts at byte 256 * 32 p)
__global__ __launch_bounds__(RC_K) void ivfs_qbyte_write_kernel(const float* __restrict__ lut, const float* __restrict__ qstat,
                                                                int M, uint8_t* __restrict__ qbyte) {
    const int qi = blockIdx.x, c = threadIdx.x;
    const float* lq = lut + (size_t)qi * M * RC_K;
    const float* st = qstat + (size_t)qi * ADC_QSTAT_STRIDE;
    const float delta = st[ADC_QSTAT_STRIDE - 1];
    for (int b16 = 0; b16 < M / 16; ++b16) {
        const int p = b16 / 2, PM = ivfs_pm(M, p), j0 = 16 * (b16 & 1);
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = 32 * p + j0 + j;
            w[j >> 2] |= (adc_quant8(lq[m * RC_K + c], st[m], delta) ^ 0x80u) << (8 * (j & 3));
        }
        *reinterpret_cast<uint4*>(qbyte + (size_t)qi * M * RC_K + (size_t)RC_K * 32 * p + (size_t)c * PM + j0) =
            make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Round 4: adc_qstats_kernel + ivfs_qbyte_write_kernel in one pass over the query's LUT.  Block (256 codes, M / 16): thread
// (c, b) keeps lut[16 b + j][c], j < 16, in registers; lo / hi per sub-quantiser by wave reductions + LDS, delta = the
// largest range / 255 (the arithmetic of adc_qstats_kernel), then the bytes are quantised from the registers.  The integer
// threshold needs tau and is computed where tau is (ivf_rank_select_kernel).  One read of the LUT instead of two, one launch
// instead of two, 6 x the threads (26 + 42 -> ~25 us per 1200 queries at M = 96).
__global__ __launch_bounds__(1024) void ivfs_qprep_kernel(const float* __restrict__ lut, int M, float* __restrict__ qstat,
                                                          uint8_t* __restrict__ qbyte) {
    __shared__ float s_lo[16][ADC_QSTAT_STRIDE], s_hi[16][ADC_QSTAT_STRIDE];
    __shared__ float s_mlo[ADC_QSTAT_STRIDE];
    __shared__ float s_delta;
    // block (256 codes, ceil(M / 32)): thread (c, y) holds the 16-blocks b = 2 y and 2 y + 1 (= table phase y of the screen)
    const int qi = blockIdx.x, c = threadIdx.x, y = threadIdx.y, lane = c & 63, wc = c >> 6;
    const float* lq = lut + (size_t)qi * M * RC_K;
    const int nb = (M / 16 - 2 * y) < 2 ? (M / 16 - 2 * y) : 2;       // 16-blocks of this thread row: 1 or 2
    float v[2][16];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[h][j] = (h < nb) ? lq[(32 * y + 16 * h + j) * RC_K + c] : 0.f;
    // min / max over the 256 codes: DPP rotations inside each row of 16 lanes (plain VALU; a butterfly of __shfl_xor is 12
    // LDS-crossbar operations per value), then 16 partials per sub-quantiser through LDS
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h < nb) {                                                  // uniform over the thread row
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float lo = v[h][j], hi = v[h][j];
                lo = fminf(lo, __int_as_float(rc_dpp_row_ror<8>(__float_as_int(lo))));
                hi = fmaxf(hi, __int_as_float(rc_dpp_row_ror<8>(__float_as_int(hi))));
                lo = fminf(lo, __int_as_float(rc_dpp_row_ror<4>(__float_as_int(lo))));
                hi = fmaxf(hi, __int_as_float(rc_dpp_row_ror<4>(__float_as_int(hi))));
                lo = fminf(lo, __int_as_float(rc_dpp_row_ror<2>(__float_as_int(lo))));
                hi = fmaxf(hi, __int_as_float(rc_dpp_row_ror<2>(__float_as_int(hi))));
                lo = fminf(lo, __int_as_float(rc_dpp_row_ror<1>(__float_as_int(lo))));
                hi = fmaxf(hi, __int_as_float(rc_dpp_row_ror<1>(__float_as_int(hi))));
                if ((lane & 15) == 0) {
                    s_lo[4 * wc + (lane >> 4)][32 * y + 16 * h + j] = lo;
                    s_hi[4 * wc + (lane >> 4)][32 * y + 16 * h + j] = hi;
                }
            }
        }
    }
    __syncthreads();
    const int t = y * RC_K + c;
    if (t < M) {
        float lo = s_lo[0][t], hi = s_hi[0][t];
#pragma unroll
        for (int r = 1; r < 16; ++r) { lo = fminf(lo, s_lo[r][t]); hi = fmaxf(hi, s_hi[r][t]); }
        s_mlo[t] = lo;
        qstat[(size_t)qi * ADC_QSTAT_STRIDE + t] = lo;
        s_lo[0][t] = hi - lo;
    }
    __syncthreads();
    if (t == 0) {
        float maxrange = 0.f;
        double A = 0.0;
        for (int m = 0; m < M; ++m) {
            maxrange = fmaxf(maxrange, s_lo[0][m]);
            A += (double)s_mlo[m];
        }
        float delta = maxrange / 255.0f;
        if (!(delta > 0.f)) delta = 1.0f;
        qstat[(size_t)qi * ADC_QSTAT_STRIDE + ADC_QSTAT_STRIDE - 1] = delta;
        *reinterpret_cast<double*>(qstat + (size_t)qi * ADC_QSTAT_STRIDE + ADC_QSTAT_STRIDE - 4) = A;   // sum of lo, m ascending
        s_delta = delta;
    }
    __syncthreads();
    const float delta = s_delta;
    const int PM = ivfs_pm(M, y);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h < nb) {
            unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 16; ++j)
                w[j >> 2] |= (adc_quant8(v[h][j], s_mlo[32 * y + 16 * h + j], delta) ^ 0x80u) << (8 * (j & 3));
            *reinterpret_cast<uint4*>(qbyte + (size_t)qi * M * RC_K + (size_t)RC_K * 32 * y + (size_t)c * PM + 16 * h) =
                make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// the integer threshold of a query from tau and the statistics of its tables (the arithmetic of adc_qstats_kernel)
__device__ __forceinline__ int adc_tint_from(float t, const float* __restrict__ st, int M) {
    if (t == -INFINITY) return INT_MIN;
    const double A = *reinterpret_cast<const double*>(st + ADC_QSTAT_STRIDE - 4);     // written by ivfs_qprep_kernel
    const double delta = (double)st[ADC_QSTAT_STRIDE - 1];
    const double v = ceil(((double)t - A) / delta - 0.5 * (double)M) - 2.0;   // entries rounded to NEAREST: |error| <= 1/2 each
    return v < -2.0e9 ? INT_MIN : (v > 2.0e9 ? INT_MAX : (int)v);
}

struct ivfs_task {
    int valid;
    int qid[8];
    unsigned t0;              // first (16-aligned) row of the range
    unsigned row_lo, nrows;   // rows [row_lo, nrows) counted from t0 are the cell's (nrows = 0: nothing to scan)
};

// LW = 0: every wave gathers and takes its share of the table fills.  LW = 4 (wave specialisation, default): the block's last
// four waves do nothing but fetch, transpose and store the NEXT stage's tables while the other twelve gather — the fill runs
// beside the gathers instead of after them (the sixteen waves of the LW = 0 form do the same thing at the same time).
// Development aid (tools/ivf_timeline.py builds a variant library with -DRC_IVF_TRACE): wall-clock stamps of every wave at the
// stage boundaries of the first tasks of every block, read back with rc_debug_ivfs_trace.  Off in the shipped library.
#ifdef RC_IVF_TRACE
__device__ unsigned long long ivfs_trace[256 * 8 * 3 * 16 * 4];
extern "C" int rc_debug_ivfs_trace(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ivfs_trace), sizeof(ivfs_trace));
}
#define IVFS_TSTAMP(i)                                                                                                 \
    do {                                                                                                               \
        if (l == 0 && k < 8u && rd == 0 && blockIdx.x < 256u)                                                          \
            ivfs_trace[(((blockIdx.x * 8u + k) * 3u + (unsigned)P) * 16u + (unsigned)wv) * 4u + (i)] = wall_clock64(); \
    } while (0)
#else
#define IVFS_TSTAMP(i) do { } while (0)
#endif
template <int M, int LW>
__global__ __launch_bounds__(IVFS_THREADS, 4) void ivfs_screen_kernel(const uint8_t* __restrict__ image,
                                                                      const int* __restrict__ tint,
                                                                      unsigned* __restrict__ stream_cnt,
                                                                      unsigned* __restrict__ stream, unsigned stream_cap,
                                                                      int* __restrict__ status, adc_ivf_tasks T,
                                                                      int ntasks_arg) {
    constexpr int GW = IVFS_WAVES - LW;                       // gathering waves
    constexpr int R = (LW == 4) ? 10 : IVFS_R;              // twelve gathering waves: ten chunks each cover a 1920-row round
    constexpr int NPH = ivfs_phases(M), ROUND = GW * R * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    const int l = (int)(tid & 63u), wv = __builtin_amdgcn_readfirstlane((int)(tid >> 6)), r = l & 15, g = l >> 4;
    // ---- this block's tasks
    const unsigned total = (unsigned)__builtin_amdgcn_readfirstlane(T.ntasks ? *T.ntasks : ntasks_arg);
    const unsigned xcd = blockIdx.x % 8u, jb = blockIdx.x / 8u, pxb = (gridDim.x - xcd + 7u) / 8u;
    const unsigned tq8 = total / 8u, tr8 = total % 8u;
    const unsigned lo = xcd < tr8 ? xcd * (tq8 + 1u) : tr8 * (tq8 + 1u) + (xcd - tr8) * tq8, cnt = tq8 + (xcd < tr8 ? 1u : 0u);
    auto load_task = [&](unsigned k) {
        ivfs_task d;
        const unsigned at = jb + k * pxb;
        d.valid = at < cnt ? 1 : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) d.qid[j] = -1;
        d.t0 = 0; d.row_lo = 0; d.nrows = 0;
        if (d.valid) {
            // (block-uniform values; the loads are vector loads - the kernel also stores - so pin them to scalars)
            auto sc = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
            const unsigned task = lo + at;
            const int qs = sc(T.task_qstart[task]), qc = sc(T.task_qcnt[task]), cell = sc(T.task_list[task]);
#pragma unroll
            for (int j = 0; j < 8; ++j) d.qid[j] = (j < qc) ? sc(T.sorted_q[qs + j]) : -1;
            const unsigned a = (unsigned)sc((int)T.list_off[cell]), b = (unsigned)sc((int)T.list_off[cell + 1]);   // N < 2^32
            if (qc > 0 && b > a) {
                const unsigned t0 = a & ~15u;
                d.t0 = t0; d.row_lo = a - t0; d.nrows = b - t0;
            }
        }
        return d;
    };
    auto rounds_of = [&](const ivfs_task& d) { return d.nrows ? (int)((d.nrows + ROUND - 1) / ROUND) : 1; };
    // threshold and query id of this lane's column (r < 8) for a task
    auto lane_q = [&](const ivfs_task& d) {
        int q = -1;
#pragma unroll
        for (int j = 0; j < 8; ++j) q = (r == j) ? d.qid[j] : q;
        return q;
    };
    auto lane_thr = [&](int q) {
        if (q < 0) return INT_MAX;
        const int t = tint[q];
        return (t == INT_MIN) ? INT_MIN : t - 128 * M;
    };
    // ---- tables: global -> registers -> (byte transpose) -> LDS
    // dword i of a query's phase table ([code][PM] bytes) = sub-quantisers 4 u .. 4 u + 3 of code i / (PM / 4); its LDS
    // entries are slots 4 u .. 4 u + 3 of that code's row (256 bytes = 32 slots x 8 queries; a 16-block is stored twice)
    constexpr int DD = 2;                                     // 2048 dwords per query and 32-phase / 1024 threads
    // Buffer loads: ONE vector offset (tid * 4) for all eight queries, the query's table comes in through the scalar offset
    // (with flat pointers the compiler forms eight 64-bit vector addresses, hoists them and spills)
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)T.qbyte, 0, -1, 0x00020000);
    auto load_tables = [&](auto PMc, int p, const ivfs_task& d, unsigned (&dd)[DD][8]) {
        constexpr int PM = decltype(PMc)::value;
        constexpr int FI = RC_K * PM / 4 / IVFS_THREADS;      // 2 (PM = 32) or 1
        // (an empty slot reads query 0's table: its column is masked by the threshold INT_MAX)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned so = (unsigned)(d.qid[j] < 0 ? 0 : d.qid[j]) * (unsigned)(M * RC_K) + (unsigned)(RC_K * 32 * p);
            if constexpr (FI == 2) {                           // dwords 2 tid, 2 tid + 1 of the query's phase table in one load
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(qrsrc, tid * 8u, so, 0);
                dd[0][j] = v.x; dd[1][j] = v.y;
            } else {
                dd[0][j] = __builtin_amdgcn_raw_buffer_load_b32(qrsrc, tid * 4u, so, 0);
            }
        }
    };
    // byte transpose of dword i of the eight queries' phase tables -> the 32 bytes of LDS entries 4 u .. 4 u + 3 of its code
    auto emit_entry = [&](auto PMc, const unsigned (&d)[8], unsigned i, unsigned bufoff) {
        constexpr int PM = decltype(PMc)::value;
        unsigned o[8];                                       // o[2 t] = queries 0-3 of entry t, o[2 t + 1] = queries 4-7
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            const unsigned a0 = d[4 * hq], a1 = d[4 * hq + 1], a2 = d[4 * hq + 2], a3 = d[4 * hq + 3];
            const unsigned t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
            const unsigned u0 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), u1 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
            o[0 + hq] = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
            o[2 + hq] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
            o[4 + hq] = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
            o[6 + hq] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
        }
        const uint4 lo4 = make_uint4(o[0], o[1], o[2], o[3]), hi4 = make_uint4(o[4], o[5], o[6], o[7]);
        if constexpr (PM == 32) {
            uint4* e = reinterpret_cast<uint4*>(smem + (bufoff + i * 32u));
            e[0] = lo4;
            e[1] = hi4;
        } else {
            uint4* e = reinterpret_cast<uint4*>(smem + (bufoff + (i >> 2) * 256u + (i & 3u) * 32u));
            e[0] = lo4;
            e[1] = hi4;
            e[8] = lo4;                                      // second copy, 16 slots further
            e[9] = hi4;
        }
    };
    auto write_tables = [&](auto PMc, const unsigned (&dd)[DD][8], unsigned bufoff) {
        constexpr int PM = decltype(PMc)::value;
        constexpr int FI = RC_K * PM / 4 / IVFS_THREADS;
#pragma unroll
        for (int f = 0; f < FI; ++f) emit_entry(PMc, dd[f], FI == 2 ? 2u * tid + (unsigned)f : tid, bufoff);
    };
    // loader waves (LW > 0): the whole phase by LW * 64 threads, 64 table registers per batch.
    // 32-phase: consecutive lanes take consecutive dwords (4-byte loads), so lane l's entry is 32 bytes at 32 i, i = l (mod 64).
    // Written as lo half then hi half by every lane, the 16 lanes the LDS serves together ({0-3, 12-15, 20-27}, ...) hit 8
    // bank quads twice (and with the 8-byte loads of the first version, 64 bytes per lane, four times: PMC showed 39 % of
    // the kernel's LDS cycles as bank conflicts).  Lanes with bit 3 set write their HI half first: the two lanes of a group that
    // share i mod 8 then differ in the half, 16 distinct quads per group.
    auto loader_fill = [&](auto PMc, int p, const ivfs_task& d, unsigned bufoff) {
        constexpr int PM = decltype(PMc)::value;
        constexpr int LT = (LW > 0 ? LW : 1) * 64;
        const unsigned lt = tid - (unsigned)(GW * 64);
        unsigned so[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) so[j] = (unsigned)(d.qid[j] < 0 ? 0 : d.qid[j]) * (unsigned)(M * RC_K) + (unsigned)(RC_K * 32 * p);
        if constexpr (PM == 32) {
            constexpr int NDW = RC_K * PM / 4, ITER = NDW / LT, BATCH = ITER < 8 ? ITER : 8;
            static_assert(NDW % LT == 0 && ITER % BATCH == 0, "whole batches");
            const bool hi_first = ((lt >> 3) & 1u) != 0;
#pragma unroll
            for (int b0 = 0; b0 < ITER; b0 += BATCH) {
                unsigned dq[BATCH][8];
#pragma unroll
                for (int it = 0; it < BATCH; ++it)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        dq[it][j] = __builtin_amdgcn_raw_buffer_load_b32(qrsrc, ((unsigned)((b0 + it) * LT) + lt) * 4u, so[j], 0);
#pragma unroll
                for (int it = 0; it < BATCH; ++it) {
                    const unsigned i = (unsigned)((b0 + it) * LT) + lt;
                    const unsigned (&dv)[8] = dq[it];
                    unsigned o[8];
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq) {
                        const unsigned a0 = dv[4 * hq], a1 = dv[4 * hq + 1], a2 = dv[4 * hq + 2], a3 = dv[4 * hq + 3];
                        const unsigned t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
                        const unsigned u0 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), u1 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
                        o[0 + hq] = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
                        o[2 + hq] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
                        o[4 + hq] = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
                        o[6 + hq] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
                    }
                    const uint4 first = hi_first ? make_uint4(o[4], o[5], o[6], o[7]) : make_uint4(o[0], o[1], o[2], o[3]);
                    const uint4 second = hi_first ? make_uint4(o[0], o[1], o[2], o[3]) : make_uint4(o[4], o[5], o[6], o[7]);
                    unsigned char* e = smem + (bufoff + i * 32u);
                    *reinterpret_cast<uint4*>(e + (hi_first ? 16 : 0)) = first;
                    *reinterpret_cast<uint4*>(e + (hi_first ? 0 : 16)) = second;
                }
            }
        } else {
            constexpr int NPAIR = RC_K * PM / 8, ITER = NPAIR / LT, BATCH = ITER < 4 ? ITER : 4;
            static_assert(NPAIR % LT == 0 && ITER % BATCH == 0, "whole batches");
#pragma unroll
            for (int b0 = 0; b0 < ITER; b0 += BATCH) {
                unsigned dq[BATCH][2][8];
#pragma unroll
                for (int it = 0; it < BATCH; ++it)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(qrsrc, ((unsigned)((b0 + it) * LT) + lt) * 8u, so[j], 0);
                        dq[it][0][j] = v.x; dq[it][1][j] = v.y;
                    }
#pragma unroll
                for (int it = 0; it < BATCH; ++it)
#pragma unroll
                    for (int f = 0; f < 2; ++f) emit_entry(PMc, dq[it][f], 2u * ((unsigned)((b0 + it) * LT) + lt) + (unsigned)f, bufoff);
            }
        }
    };
    // ---- codes of one stage: chunk c of wave wv is chunk 16 c + wv of the round (the waves share a short cell evenly:
    // a cell of 1770 rows = 111 chunks costs every wave 7 chunks, not the first 14 waves 8); PM / 16 dwords per lane and chunk
    auto chunks_of = [&](unsigned nrows, int rd) {            // chunks this wave owns in round rd (wave-uniform, 0 .. R)
        const unsigned done = (unsigned)rd * ROUND;
        if (nrows <= done) return 0;
        unsigned nc = (nrows - done + 15u) / 16u;             // chunks of the round that hold rows of the cell
        if (nc > (unsigned)(ROUND / 16)) nc = ROUND / 16;
        if (wv >= GW) return 0;                                // a loader wave
        const int mine = ((int)nc - wv + GW - 1) / GW;
        return mine < 0 ? 0 : mine;
    };
    auto load_codes = [&](auto PMc, int p, unsigned t0, unsigned nrows, int rd, unsigned (&w)[R][2]) {
        constexpr int PM = decltype(PMc)::value;
        constexpr int NW = PM / 16;
        const int reff = chunks_of(nrows, rd);
        if (reff == 0) return;
        // t0 is a multiple of 16: the cell's first chunk; a chunk and phase = 64 lanes x PM / 4 contiguous bytes.  Rows of the
        // last chunk past the cell's end are another cell's (or, past the index, the padding of the last chunk): masked later
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(image + (size_t)t0 * M), 0, -1, 0x00020000);
        const unsigned lane_at = (unsigned)((g * 16 + r) * (PM / 4));
        const unsigned first = ((unsigned)rd * (unsigned)(ROUND / 16) + (unsigned)wv) * (unsigned)(16 * M) + (unsigned)(16 * 32 * p);
#pragma unroll
        for (int c = 0; c < R; ++c) {
            if (c < reff) {                                    // wave-uniform
                const unsigned so = first + (unsigned)(c * GW * 16 * M);
                if constexpr (NW == 2) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_at, so, 0);
                    w[c][0] = v.x; w[c][1] = v.y;
                } else {
                    w[c][0] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane_at, so, 0);
                }
            }
        }
    };
    adc_i32x4v bsel = {0, 0, 0, 0};                          // B[k][j = r] = [k % 8 == r]
    if (r < 8) {
        const int one = 1 << (8 * (r & 3));
        bsel[r >> 2] = one;
        bsel[2 + (r >> 2)] = one;
    }
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    if (lds0 & 0xFFFFu) __builtin_trap();                    // the one-instruction gather address needs 64 KiB-aligned table buffers
    adc_i32x4v acc[R];
    // ---- gathers + folds of one stage
    auto gathers = [&](auto PMc, bool first, const unsigned (&w)[R][2], unsigned bufoff, int reff) {
        constexpr int PM = decltype(PMc)::value;
        constexpr int STEPS = PM / 4;
        unsigned off[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            int slot, m;
            adc_cf_step(PM, s, r, g, slot, m);
            off[s] = lds0 + bufoff + (unsigned)slot * 8u;
        }
        // units of 4 gathers (half a chunk of a 32-phase, a chunk of a 16-phase) = 2 MFMAs; the gathers of the next unit are
        // issued before the MFMAs of the current one (8 gathers per wave in flight; 16 did not fit the 128 registers of 4 waves/SIMD)
        constexpr int UPC = STEPS / 4;
        uint2 ea[4], eb[4];
        auto gather = [&](int c, int hh, uint2 (&e)[4]) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                // buffer base (0 / 64 KiB: bytes 2-3) | code << 8 | slot offset (< 256): one v_perm_b32 (see the 16-query screen)
                const unsigned addr = __builtin_amdgcn_perm(w[c][hh], off[4 * hh + s4], 0x03020000u | ((4u + (unsigned)s4) << 8));
                typedef unsigned adc_u32x2 __attribute__((ext_vector_type(2)));
                const adc_u32x2 v = *reinterpret_cast<const adc_u32x2 __attribute__((address_space(3)))*>(addr);
                e[s4] = make_uint2(v.x, v.y);
            }
        };
        auto fold = [&](int c, int hh, const uint2 (&e)[4]) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const adc_i32x4v a = {(int)e[2 * s2].x, (int)e[2 * s2].y, (int)e[2 * s2 + 1].x, (int)e[2 * s2 + 1].y};
                if (hh == 0 && s2 == 0 && first) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, adc_i32x4v{0, 0, 0, 0}, 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, acc[c], 0, 0, 0);
            }
        };
        if (reff <= 0) return;                                // wave-uniform
        gather(0, 0, ea);
#pragma unroll
        for (int c = 0; c < R; ++c) {
            if (c < reff) {                                   // wave-uniform
#if IVFS_PRIO
                // progress-proportional priority (see the 16-query screen): a wave that is behind in its stage outranks one ahead
                if (c == 0) __builtin_amdgcn_s_setprio(3);
                else if (c == R / 4) __builtin_amdgcn_s_setprio(2);
                else if (c == R / 2) __builtin_amdgcn_s_setprio(1);
                else if (c == 3 * R / 4) __builtin_amdgcn_s_setprio(0);
#endif
                if constexpr (UPC == 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    gather(c, 1, eb);
                    __builtin_amdgcn_sched_barrier(0);
                    fold(c, 0, ea);
                    __builtin_amdgcn_sched_barrier(0);
                    if (c + 1 < R && c + 1 < reff) gather(c + 1, 0, ea);
                    __builtin_amdgcn_sched_barrier(0);
                    fold(c, 1, eb);
                } else {
                    __builtin_amdgcn_sched_barrier(0);
                    if (c + 1 < R && c + 1 < reff) gather(c + 1, 0, (c & 1) ? ea : eb);
                    __builtin_amdgcn_sched_barrier(0);
                    fold(c, 0, (c & 1) ? eb : ea);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#if IVFS_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    // ---- survivors
    // No atomics here: a returning atomic costs the wave its round trip at the next vmcnt wait on anything older (the
    // counter is in-order), ~1-3 us per task with sixteen waves meeting at the next barrier.  Every wave appends (query, row)
    // pairs to its OWN stream in global memory (stream_cap pairs, running offset in a scalar); ivfs_bucket_kernel deals the
    // streams to the per-query id lists afterwards.
    // One branch-free pass over the wave's 32 sums per lane builds a bit mask of the lane's survivors (a divergent branch per
    // sum cost 3 us per task); the lanes' counts give the positions (query column major: a stream holds runs of equal
    // query ids), then the lanes write out one survivor per trip of a wave-uniform loop (max count over the lanes: 1-3 trips).
    const __amdgpu_buffer_rsrc_t strsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(stream + (size_t)(blockIdx.x * IVFS_WAVES + (unsigned)wv) * stream_cap * 2u), 0, -1, 0x00020000);
    unsigned woff = 0;                                        // wave-uniform: pairs in the wave's stream
    typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
    static_assert(R * 4 <= 64, "one mask bit per sum");
    typedef typename std::conditional<(R * 4 <= 32), unsigned, unsigned long long>::type mask_t;
    auto epilogue = [&](unsigned t0, unsigned row_lo, unsigned nrows, int rd, int tq, int myq, int reff) {
        if (reff <= 0) return;                                // wave-uniform: no rows of the cell in this wave's share
        const unsigned rb = (unsigned)rd * ROUND + (unsigned)(wv * 16);      // first row of the wave's chunk 0
        mask_t m = 0;                                         // bit 4 c + e: D[row 4 g + e of chunk c][column r] survives
#pragma unroll
        for (int c = 0; c < R; ++c) {
            if (c < reff) {
#pragma unroll
                for (int e = 0; e < 4; ++e) m |= (acc[c][e] >= tq) ? ((mask_t)1 << (4 * c + e)) : (mask_t)0;
            }
        }
        // rows outside the cell (before its first row in the first chunk, after its last in the last): never survivors
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const unsigned cb = rb + (unsigned)(16 * GW * c);
            if (c < reff && (cb < row_lo || cb + 16u > nrows)) {           // wave-uniform, rare
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned n = cb + 4u * g + e;
                    if (n < row_lo || n >= nrows) m &= ~((mask_t)1 << (4 * c + e));
                }
            }
        }
        const unsigned cnt = (unsigned)__popcll((unsigned long long)m);
        if (!__ballot(cnt != 0)) return;
        const unsigned c0 = __shfl(cnt, r), c1 = __shfl(cnt, r + 16), c2 = __shfl(cnt, r + 32), c3 = __shfl(cnt, r + 48);
        const unsigned tot = c0 + c1 + c2 + c3;
        const unsigned lane_first = (g > 0 ? c0 : 0u) + (g > 1 ? c1 : 0u) + (g > 2 ? c2 : 0u);
        unsigned inc = tot;                                   // inclusive prefix over the query columns r of the lane's row
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {                      // (columns 8 .. 15 hold nothing)
            const unsigned t = __shfl_up(inc, o, 16);
            if (r >= o) inc += t;
        }
        const unsigned wtotal = (unsigned)__builtin_amdgcn_readlane((int)inc, 7);
        if (woff + wtotal > stream_cap) {                     // wave-uniform; status bit 2: a stream filled up (no query to blame)
            if (l == 0) atomicOr(status, 4);
            return;
        }
        unsigned at = (woff + (inc - tot) + lane_first) * 8u;  // byte offset of the lane's first pair
        const unsigned row0 = t0 + rb + 4u * (unsigned)g;
        while (__ballot(m != 0)) {                             // wave-uniform
            if (m) {
                const unsigned idx = (unsigned)__builtin_ctzll((unsigned long long)m);
                m &= m - (mask_t)1;
                const u32x2s v = {(unsigned)myq, row0 + (idx >> 2) * (unsigned)(16 * GW) + (idx & 3u)};
                __builtin_amdgcn_raw_buffer_store_b64(v, strsrc, at, 0, 0);
                at += 8u;
            }
        }
        woff += wtotal;
    };
    auto block_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- prologue: every thread helps with the first tables
    const ivfs_task first = load_task(0);
    if (!first.valid) return;                                 // block-uniform
    using P0 = std::integral_constant<int, ivfs_pm(M, 0)>;
    {
        unsigned dd[DD][8];
        load_tables(P0{}, 0, first, dd);
        write_tables(P0{}, dd, 0u);
    }
    // ---- the walk over (task, round, phase) stages, once per role: a loader wave runs its own copy of the loop — it meets the
    // gathering waves at every barrier but never holds their sums / codes (as one loop with a branch per stage, the compiler
    // keeps those 60 registers live through the loader's branch and spills 300 bytes per lane)
    auto walk = [&](auto ROLEc) {
        constexpr bool LOADER = decltype(ROLEc)::value == 1;
        ivfs_task cur = first;
        int myq = -1, tq = INT_MAX;
        unsigned dd[DD][8];
        unsigned w[R][2];
        if constexpr (!LOADER) {
            myq = lane_q(cur); tq = lane_thr(myq);
            load_codes(P0{}, 0, cur.t0, cur.nrows, 0, w);
        }
        unsigned bufoff = 0;
        unsigned k = 0;
        for (;;) {                                            // tasks of this block
            const ivfs_task nxt = load_task(k + 1);           // used in this task's LAST stage (and for its thresholds after)
            const int nrounds = rounds_of(cur);
            for (int rd = 0; rd < nrounds; ++rd) {
                const bool more = rd + 1 < nrounds;           // block-uniform
                auto stage = [&](auto Pc) {
                    constexpr int P = decltype(Pc)::value;
                    constexpr bool LASTP = (P == NPH - 1);
                    constexpr int PN = LASTP ? 0 : P + 1;     // phase of the next stage
                    using PMc = std::integral_constant<int, ivfs_pm(M, P)>;
                    using PMn = std::integral_constant<int, ivfs_pm(M, PN)>;
                    IVFS_TSTAMP(3);                               // arrival at the barrier that ends the previous stage
                    block_sync();
                    IVFS_TSTAMP(0);
                    // the next stage: same task (next phase / next round) or the next task's first
                    const bool to_next = LASTP && !more;      // block-uniform
                    const bool has_next = !to_next || nxt.valid;
                    ivfs_task nd;
#pragma unroll
                    for (int j = 0; j < 8; ++j) nd.qid[j] = to_next ? nxt.qid[j] : cur.qid[j];
                    nd.t0 = to_next ? nxt.t0 : cur.t0;
                    nd.nrows = to_next ? nxt.nrows : cur.nrows;
                    const int nrd = to_next ? 0 : (LASTP ? rd + 1 : rd);
                    if constexpr (LOADER) {
                        if (has_next) loader_fill(PMn{}, PN, nd, bufoff ^ (unsigned)IVFS_BUF);
                    } else {
                        // (LW = 0) the next tables are requested now and transposed after this stage's gathers
                        if (LW == 0 && has_next) load_tables(PMn{}, PN, nd, dd);
                        const int reff = chunks_of(cur.nrows, rd);
                        gathers(PMc{}, P == 0, w, bufoff, reff);
                        IVFS_TSTAMP(1);
                        // the codes of the next stage go into the registers the gathers just released (last phase: after the
                        // survivor pass, whose few waits would otherwise also wait for them)
                        if constexpr (!LASTP) { if (has_next) load_codes(PMn{}, PN, nd.t0, nd.nrows, nrd, w); }
                        if constexpr (LASTP) {
                            epilogue(cur.t0, cur.row_lo, cur.nrows, rd, tq, myq, reff);
                            if (has_next) load_codes(PMn{}, PN, nd.t0, nd.nrows, nrd, w);
                        }
                        if (LW == 0 && has_next) write_tables(PMn{}, dd, bufoff ^ (unsigned)IVFS_BUF);
                    }
                    IVFS_TSTAMP(2);
                    bufoff ^= (unsigned)IVFS_BUF;
                };
                stage(std::integral_constant<int, 0>{});
                if constexpr (NPH > 1) stage(std::integral_constant<int, 1>{});
                if constexpr (NPH > 2) stage(std::integral_constant<int, 2>{});
            }
            if (!nxt.valid) break;
            cur = nxt;
            if constexpr (!LOADER) { myq = lane_q(cur); tq = lane_thr(myq); }
            ++k;
        }
    };
    if (LW > 0 && wv >= GW) {                                 // wave-uniform
        walk(std::integral_constant<int, 1>{});
        return;                                               // (its stream stays empty: stream_cnt was cleared by the host)
    }
    walk(std::integral_constant<int, 0>{});
    if (l == 0) stream_cnt[blockIdx.x * IVFS_WAVES + (unsigned)wv] = woff;
}

// Deal the waves' (query, row) streams to the per-query id lists.  An atomic on one address takes ~0.2 us and the atomics
// of one address do not overlap: 2.4 M runs (one per wave, task and query) on 1200 counters cost 0.44 ms however many waves
// issue them.  The sixteen streams of ONE screen block hold the same (task, query) pairs, so one bucket block takes them
// all: a histogram over the queries in LDS (pass 1), ONE global atomic per query present (~600 of 1200 per block: 128 per
// counter over the whole grid), then every pair finds its slot with an LDS atomic (pass 2).
#define IVFS_BUCKET_THREADS 1024
__global__ __launch_bounds__(IVFS_BUCKET_THREADS) void ivfs_bucket_kernel(const unsigned* __restrict__ stream_cnt,
                                                                          const unsigned* __restrict__ stream, unsigned stream_cap,
                                                                          int nq, unsigned* __restrict__ id_count,
                                                                          unsigned* __restrict__ ids) {
    extern __shared__ unsigned bk_hist[];                     // [nq] pairs of the query in this block's streams, then its first slot
    const unsigned tid = threadIdx.x, wv = tid >> 6, l = tid & 63u;
    for (int q = (int)tid; q < nq; q += IVFS_BUCKET_THREADS) bk_hist[q] = 0u;
    __syncthreads();
    const unsigned sidx = blockIdx.x * IVFS_WAVES + wv;       // wave w of the bucket block reads stream w of the screen block
    const unsigned n = stream_cnt[sidx];
    const uint2* st = reinterpret_cast<const uint2*>(stream) + (size_t)sidx * stream_cap;
    for (unsigned i = l; i < n; i += 64u) atomicAdd(&bk_hist[st[i].x], 1u);
    __syncthreads();
    for (int q = (int)tid; q < nq; q += IVFS_BUCKET_THREADS) {
        const unsigned c = bk_hist[q];
        if (c) bk_hist[q] = atomicAdd(id_count + q, c);
    }
    __syncthreads();
    for (unsigned i = l; i < n; i += 64u) {
        const uint2 e = st[i];
        const unsigned slot = atomicAdd(&bk_hist[e.x], 1u);
        if (slot < ADC_ID_CAP) ids[(size_t)e.x * ADC_ID_CAP + slot] = e.y;
    }
}

// grid (nq, slices): the query's sample entries 0 .. scount[qi] are dealt to the threads of its blocks; an entry finds its
// cell by binary search over the query's sbase row (no per-cell loop: a probed cell contributes only a few dozen sampled
// rows, and walking the cells one after the other would serialise two dependent loads per cell).
// 1024 threads: the 4 M 256-byte table takes the CU's LDS, so the block is also the CU's whole occupancy.
// Round 4: the query's plan (sbase, first row of every probed cell) is staged in LDS beside the table — the binary search
// was log2(nprobe) DEPENDENT global loads per entry, most of a block's 12 us —, table and codes move in 16-byte pieces, two
// entries per thread are in flight, and a query gets one slice (one staging of its 4 M 256 bytes) unless the grid would
// not fill the chip.
#define IVF_SAMPLE_THREADS 1024
#define IVF_SAMPLE_PLAN_MAX 2048     // probes whose plan fits in LDS beside a 96 KiB table
template <int M>
__global__ __launch_bounds__(IVF_SAMPLE_THREADS) void ivf_sample_scan_kernel(const uint8_t* __restrict__ codes,
                                                              const int64_t* __restrict__ list_off,
                                                              const float* __restrict__ lut, const int* __restrict__ probes,
                                                              const int* __restrict__ sbase, const int* __restrict__ scount,
                                                              int nprobe, int64_t sstride, int ss,
                                                              float* __restrict__ sample) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tab = reinterpret_cast<float*>(smem);   // [M][256]
    int64_t* s_lo = reinterpret_cast<int64_t*>(smem + (size_t)M * RC_K * sizeof(float));   // [nprobe] first row of the cell
    int* s_sb = reinterpret_cast<int*>(s_lo + nprobe);                                      // [nprobe]
    const int qi = blockIdx.x, tid = threadIdx.x;
    const int n = scount[qi];
    const int* sb = sbase + (size_t)qi * nprobe;
    const int* pr = probes + (size_t)qi * nprobe;
    const bool plan_lds = nprobe <= IVF_SAMPLE_PLAN_MAX;     // block-uniform
    if (plan_lds)
        for (int p = tid; p < nprobe; p += IVF_SAMPLE_THREADS) { s_sb[p] = sb[p]; s_lo[p] = list_off[pr[p]]; }
    {
        const float4* l4 = reinterpret_cast<const float4*>(lut + (size_t)qi * M * RC_K);
        float4* t4 = reinterpret_cast<float4*>(tab);
        for (int i = tid; i < M * RC_K / 4; i += IVF_SAMPLE_THREADS) t4[i] = l4[i];
    }
    __syncthreads();
    const int step = gridDim.y * IVF_SAMPLE_THREADS;
    for (int i = blockIdx.y * IVF_SAMPLE_THREADS + tid; i < n; i += 2 * step) {
        int64_t row[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ih = (i + h * step < n) ? i + h * step : i;
            int lo = 0, hi = nprobe - 1;                              // last probe p with sbase[p] <= ih
            if (plan_lds) {
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (s_sb[mid] <= ih) lo = mid; else hi = mid - 1;
                }
                const int off = ih - s_sb[lo];
                // the sample of a cell: runs of 16 consecutive rows (coalesced reads), one run every 16 * ss rows
                row[h] = s_lo[lo] + (int64_t)(off >> 4) * 16 * ss + (off & 15);
            } else {
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (sb[mid] <= ih) lo = mid; else hi = mid - 1;
                }
                const int off = ih - sb[lo];
                row[h] = list_off[pr[lo]] + (int64_t)(off >> 4) * 16 * ss + (off & 15);
            }
        }
        const float s0 = adc_rescore_row<M>(codes + row[0] * M, tab);
        const float s1 = adc_rescore_row<M>(codes + row[1] * M, tab);
        sample[(size_t)qi * sstride + i] = s0;
        if (i + step < n) sample[(size_t)qi * sstride + i + step] = s1;
    }
}

// thr[qi] = rank[qi]-th largest of sample[qi][0 .. scount[qi]); rank <= 0 or > scount: -inf.  One block per query,
// 8 bits per pass over global memory (the sample is 1/SS of the probed rows).
__global__ __launch_bounds__(1024) void ivf_rank_select_kernel(const float* __restrict__ sample, const int* __restrict__ scount,
                                                               const int* __restrict__ rank, int64_t sstride,
                                                               float* __restrict__ thr, const float* __restrict__ qstat = nullptr,
                                                               int M = 0, int* __restrict__ tint = nullptr) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_aux[8];
    __shared__ unsigned s_scan[4];
    const int qi = blockIdx.x, tid = threadIdx.x;
    const int n = scount[qi], k = rank[qi];
    if (k <= 0 || k > n) {
        if (tid == 0) {
            thr[qi] = -INFINITY;
            if (tint) tint[qi] = INT_MIN;
        }
        return;
    }
    // a few thousand scores in global memory: four plain passes (the value-space cut of adc_kth_largest_v costs more barriers
    // and one more pass than it saves at this length: 18 -> 26 us per 1200 queries at nprobe 8)
    const float* row = sample + (size_t)qi * sstride;
    unsigned* s_sel = s_aux;
    if (tid == 0) { s_sel[0] = 0u; s_sel[1] = (unsigned)k; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = s_sel[0], need = s_sel[1];
        const unsigned himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < n; i += 1024) {
            const unsigned key = adc_order_key(row[i]);
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        adc_pick_bin(hist, need, prefix, shift, s_scan, &s_sel[0], &s_sel[1]);
    }
    const unsigned kth = s_sel[0];
    if (tid == 0) {
        const float t = adc_unorder_key(kth);
        thr[qi] = t;
        if (tint) tint[qi] = adc_tint_from(t, qstat + (size_t)qi * ADC_QSTAT_STRIDE, M);
    }
}

__global__ void ivf_check_kernel(const unsigned* __restrict__ cand_count, const int* __restrict__ rows, int nq, int k,
                                 int* __restrict__ status, int* __restrict__ qstatus) {
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    const int want = rows[qi] < k ? rows[qi] : k;
    if ((int)cand_count[qi] < want) {
        atomicOr(status, 1);
        if (qstatus) atomicOr(qstatus + qi, 1);
    }
}

namespace {
struct ivfl_ws {
    size_t sample, thr, tint, qstat, qbyte, idcnt, ids, cnt, cand, stream_cnt, counters_end, stream, stream_cap, total;
};
ivfl_ws ivfl_layout(int M, int nq, int64_t sstride) {
    ivfl_ws L;
    size_t o = 0;
    L.sample = o; o += rc_align_up((size_t)nq * (size_t)sstride * sizeof(float), 256);
    L.thr = o;    o += rc_align_up((size_t)nq * sizeof(float), 256);
    L.tint = o;   o += rc_align_up((size_t)nq * sizeof(int), 256);
    L.qstat = o;  o += rc_align_up((size_t)nq * ADC_QSTAT_STRIDE * sizeof(float), 256);
    L.qbyte = o;  o += rc_align_up((size_t)nq * M * RC_K, 256);                 // compact per-query byte tables
    // the three counter arrays sit side by side: ONE memset clears them (idcnt | cnt | stream_cnt)
    L.idcnt = o;  o += rc_align_up((size_t)nq * sizeof(unsigned), 256);
    L.cnt = o;    o += rc_align_up((size_t)nq * sizeof(unsigned), 256);
    L.stream_cnt = o; o += rc_align_up((size_t)IVFS_MAX_BLOCKS * IVFS_WAVES * sizeof(unsigned), 256);
    L.counters_end = o;
    L.ids = o;    o += rc_align_up((size_t)nq * ADC_ID_CAP * sizeof(unsigned), 256);
    L.cand = o;   o += rc_align_up((size_t)nq * ADC_CAND_CAP * sizeof(unsigned long long), 256);
    // (query, row) streams of the pipelined screen: one per wave of its <= IVFS_MAX_BLOCKS persistent blocks
    size_t cap = (size_t)nq * (ADC_ID_CAP / 2) / (IVFS_MAX_BLOCKS * IVFS_WAVES);
    if (cap < 4096) cap = 4096;
    if (const char* e = getenv("RC_IVF_STREAM_CAP")) {      // tests: provoke the overflow path (status bit 2 -> less slack -> scan)
        const long v = atol(e);
        if (v > 0) cap = (size_t)v;
    }
    L.stream_cap = cap;
    L.stream = o;     o += rc_align_up((size_t)IVFS_MAX_BLOCKS * IVFS_WAVES * cap * 8, 256);
    L.total = o;
    return L;
}

template <int M>
int ivfl_launch(rc_handle_t h, const uint8_t* codes, const uint8_t* image, const int64_t* list_off, const int64_t* rowmap,
                int64_t N, const float* lut, int nq, const int* probes, const int* sbase, const int* scount,
                const int* rows, const int* rank, int nprobe, int64_t sstride, int ss, const adc_ivf_tasks& T, int ntasks,
                int k, float* scores, int64_t* out_ids, int* status, char* w, const ivfl_ws& L, hipStream_t s,
                int* qstatus = nullptr) {
    constexpr int NP = (M == 96) ? 2 : 1, PM = M / NP;
    constexpr int R = (NP > 1) ? 8 : (M == 64 ? 2 : 4);
    constexpr int TH = ADC_THREADS;
    float* sample = (float*)(w + L.sample);
    float* thr = (float*)(w + L.thr);
    int* tint = (int*)(w + L.tint);
    float* qstat = (float*)(w + L.qstat);
    uint8_t* qbyte = (uint8_t*)(w + L.qbyte);
    unsigned* idcnt = (unsigned*)(w + L.idcnt);
    unsigned* ids = (unsigned*)(w + L.ids);
    unsigned* cnt = (unsigned*)(w + L.cnt);
    unsigned long long* cand = (unsigned long long*)(w + L.cand);
    {
        auto kern = ivf_sample_scan_kernel<M>;
        const size_t lds = (size_t)M * RC_K * sizeof(float) + (nprobe <= IVF_SAMPLE_PLAN_MAX ? (size_t)nprobe * 12 : 0);
        RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)((size_t)M * RC_K * sizeof(float) + IVF_SAMPLE_PLAN_MAX * 12)));
        // every block stages the query's 4 M 256-byte fp32 table: one slice per query unless the grid would not fill the chip
        int64_t slices = (sstride + 2047) / 2048;
        const int64_t fill = (2 * (int64_t)(h->num_cus > 0 ? h->num_cus : 256) + nq - 1) / nq;
        if (slices > fill) slices = fill;
        if (slices > 16) slices = 16;
        hipLaunchKernelGGL(kern, dim3((unsigned)nq, (unsigned)(slices < 1 ? 1 : slices)), dim3(IVF_SAMPLE_THREADS), lds, s, codes, list_off, lut,
                           probes, sbase, scount, nprobe, sstride, ss, sample);
        RC_LAUNCH_CHECK(h);
    }
    if (ivf_pipe()) {
        // tables first (they need no threshold), then tau and the integer threshold in one kernel
        hipLaunchKernelGGL(ivfs_qprep_kernel, dim3((unsigned)nq), dim3(RC_K, (M + 31) / 32), 0, s, lut, M, qstat, qbyte);
        RC_LAUNCH_CHECK(h);
        hipLaunchKernelGGL(ivf_rank_select_kernel, dim3((unsigned)nq), dim3(1024), 0, s, (const float*)sample, scount, rank, sstride, thr,
                           (const float*)qstat, M, tint);
        RC_LAUNCH_CHECK(h);
    } else {
        hipLaunchKernelGGL(ivf_rank_select_kernel, dim3((unsigned)nq), dim3(1024), 0, s, (const float*)sample, scount, rank, sstride, thr);
        RC_LAUNCH_CHECK(h);
        hipLaunchKernelGGL(adc_qstats_kernel, dim3((unsigned)nq), dim3(RC_K), 0, s, lut, (const float*)thr, M, qstat, tint);
        RC_LAUNCH_CHECK(h);
        hipLaunchKernelGGL(adc_qbyte_write_kernel<PM>, dim3((unsigned)nq), dim3(RC_K), 0, s, lut, (const float*)qstat, M, qbyte);
        RC_LAUNCH_CHECK(h);
    }
    RC_HIP_CHECK(h, hipMemsetAsync(w + L.idcnt, 0, L.counters_end - L.idcnt, s));      // idcnt, cnt, stream_cnt
    if (ivf_pipe()) {
        static const int lw = [] { const char* e = getenv("RC_IVF_LW"); return (e && e[0] == '0') ? 0 : 4; }();   // 0: no loader waves
        auto kern = lw ? ivfs_screen_kernel<M, 4> : ivfs_screen_kernel<M, 0>;
        constexpr int sl = 2 * IVFS_BUF;
        RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, sl));
        adc_ivf_tasks TT = T;
        TT.qbyte = qbyte;
        int blocks = h->num_cus > 0 ? h->num_cus : 256;       // persistent: one block per CU
        if (blocks > IVFS_MAX_BLOCKS) blocks = IVFS_MAX_BLOCKS;
        if (!T.ntasks && ntasks < blocks) blocks = ntasks;
        unsigned* stream_cnt = (unsigned*)(w + L.stream_cnt);
        unsigned* stream = (unsigned*)(w + L.stream);
        const unsigned nstreams = (unsigned)blocks * IVFS_WAVES;
        rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(IVFS_THREADS), sl, s, image, (const int*)tint, stream_cnt, stream,
                           (unsigned)L.stream_cap, status, TT, ntasks);
        rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
        RC_LAUNCH_CHECK(h);
        {
            const size_t bl = (size_t)nq * sizeof(unsigned);
            RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)ivfs_bucket_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bl));
            hipLaunchKernelGGL(ivfs_bucket_kernel, dim3((unsigned)blocks), dim3(IVFS_BUCKET_THREADS), bl, s, (const unsigned*)stream_cnt,
                               (const unsigned*)stream, (unsigned)L.stream_cap, nq, idcnt, ids);
        }
        RC_LAUNCH_CHECK(h);
    } else {
        auto kern = adc_screen_cf_kernel<M, NP, R, true, TH>;
        constexpr int sl = adc_cf<PM>::TABLE_BYTES;
        RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, sl));
        adc_ivf_tasks TT = T;
        TT.qbyte = qbyte;
        rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
        hipLaunchKernelGGL(kern, dim3((unsigned)ntasks), dim3(TH), sl, s, image, N, (const uint8_t*)nullptr,
                           (const int*)tint, nq, idcnt, ids, TT, adc_part_args{});
        rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
        RC_LAUNCH_CHECK(h);
    }
    {
        auto krescore = adc_rescore_kernel<M>;
        const size_t rl = (size_t)M * RC_K * sizeof(float);
        RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)krescore, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl));
        hipLaunchKernelGGL(krescore, dim3((unsigned)nq), dim3(adc_rescore_threads(M)), rl, s, codes, lut, (const float*)thr, (const unsigned*)idcnt,
                           (const unsigned*)ids, cnt, cand, status, rowmap, qstatus);
        RC_LAUNCH_CHECK(h);
    }
    hipLaunchKernelGGL(ivf_check_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, (const unsigned*)cnt, rows, nq, k, status,
                       qstatus);
    RC_LAUNCH_CHECK(h);
    if (rc_env_set("RC_IVF_DEBUG")) {                          // development: list lengths of this search (synchronises)
        std::vector<unsigned> a(nq), b(nq);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(a.data(), idcnt, nq * sizeof(unsigned), hipMemcpyDeviceToHost);
        (void)hipMemcpy(b.data(), cnt, nq * sizeof(unsigned), hipMemcpyDeviceToHost);
        double sa = 0, sb = 0; unsigned ma = 0, mb = 0;
        for (int i = 0; i < nq; ++i) { sa += a[i]; sb += b[i]; ma = a[i] > ma ? a[i] : ma; mb = b[i] > mb ? b[i] : mb; }
        fprintf(stderr, "[ivf debug] nq %d nprobe %d ss %d sstride %lld: screened ids mean %.0f max %u, candidates mean %.0f max %u\n",
                nq, nprobe, ss, (long long)sstride, sa / nq, ma, sb / nq, mb);
    }
    // N = 0: fewer than k rows is legitimate (small cells); too FEW CANDIDATES is what ivf_check_kernel reports
    return rc_adc_launch_select(h, cand, cnt, nq, 0, k, 0, scores, out_ids, status, s, qstatus);
}
}  // namespace

extern "C" size_t rc_ivf_search_lists_ws_bytes(int M, int nq, int64_t sstride) {
    if (!adc_cf_supported(M) || nq <= 0 || sstride <= 0) return 0;
    return ivfl_layout(M, nq, sstride).total;
}

// codes / image: [N,M] cell-major canonical codes and their permuted image; list_off [nlist+1]; rowmap [N] corpus position
// of every row; lut [nq,M,256] (rc_adc_lut); probes / sbase [nq,nprobe]: probed cells and the position of each probe's
// first SAMPLED row in the query's sample array (a cell of n rows is sampled in runs of 16 rows every 16 ss rows:
// 16 floor(n / (16 ss)) + min(16, n mod (16 ss)) entries);
// scount [nq] sampled rows, rows [nq] probed rows, rank [nq] rank of the sample score used as threshold (0: keep all);
// tasks: task_list / task_qstart / task_qcnt [ntasks] and sorted_q (query ids ordered by probed cell).
extern "C" int rc_ivf_search_lists(rc_handle_t h, const uint8_t* codes, const uint8_t* image, const int64_t* list_off,
                                   const int64_t* rowmap, int64_t N, int M, int K, const float* lut, int nq,
                                   const int* probes, const int* sbase, const int* scount, const int* rows, const int* rank,
                                   int nprobe, int64_t sstride, int ss, const int* task_list, const int* task_qstart,
                                   const int* task_qcnt, const int* sorted_q, int ntasks, int k, float* scores,
                                   int64_t* out_ids, int* status, void* ws, size_t ws_bytes, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !image || !list_off || !rowmap || !lut || !probes || !sbase || !scount || !rows || !rank ||
        !task_list || !task_qstart || !task_qcnt || !sorted_q || !scores || !out_ids || !status || N <= 0 || nq < 0 ||
        nprobe <= 0 || sstride <= 0 || ss <= 0 || ntasks < 0 || k <= 0)
        return RC_EINVAL;
    if (K != RC_K || !adc_cf_supported(M) || k > ADC_CAND_CAP / 2 || N > 0xFFFFFFFFll || RC_ADC_IMG16 || nq > 32768) return RC_ESHAPE;
    if (nq == 0) return RC_OK;
    const ivfl_ws L = ivfl_layout(M, nq, sstride);
    if (!ws || ws_bytes < L.total) return RC_EWORKSPACE;
    adc_ivf_tasks T = {task_list, task_qstart, task_qcnt, sorted_q, list_off, nullptr, nullptr};
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)ws;
    if (ntasks == 0) {                                        // nothing probed: empty results through the select kernel
        RC_HIP_CHECK(h, hipMemsetAsync(w + L.cnt, 0, (size_t)nq * sizeof(unsigned), s));
        return rc_adc_launch_select(h, (unsigned long long*)(w + L.cand), (const unsigned*)(w + L.cnt), nq, 0, k, 0, scores,
                                    out_ids, status, s);
    }
    switch (M) {
#define IVFL_CASE(MM)                                                                                                   \
        case MM: return ivfl_launch<MM>(h, codes, image, list_off, rowmap, N, lut, nq, probes, sbase, scount, rows, rank, \
                                        nprobe, sstride, ss, T, ntasks, k, scores, out_ids, status, w, L, s);
        IVFL_CASE(16) IVFL_CASE(32) IVFL_CASE(48) IVFL_CASE(64) IVFL_CASE(96)
#undef IVFL_CASE
        default: return RC_ESHAPE;
    }
}

// ------------------------------------------------------------------------------------ device-side plan of the search
// rc_ivf_search_probes: everything rc_ivf_search_lists expects from its caller (sample layout, ranks, the task list) is
// derived on the device from the probes alone - four small kernels instead of ~40 framework launches and two host
// synchronisations (task count, sample stride) per search.
namespace {
struct ivfp_ws {
    size_t sbase, scount, rows, rank, per_cell, cell_start, first_task, cursor, ntasks, sorted_q, task_list, task_qstart,
        task_qcnt, total;
    int64_t ub;
};
ivfp_ws ivfp_layout(size_t base, int nq, int nprobe, int nlist) {
    ivfp_ws P;
    const size_t pairs = (size_t)nq * nprobe;
    size_t ub = (size_t)nlist + pairs / 8 + 1;                // tasks: at most one partly filled group per probed cell
    if (ub > pairs) ub = pairs;
    P.ub = (int64_t)ub;
    size_t o = base;
    auto take = [&](size_t n) { const size_t at = o; o += rc_align_up(n * sizeof(int), 256); return at; };
    P.sbase = take(pairs); P.scount = take(nq); P.rows = take(nq); P.rank = take(nq);
    P.per_cell = take(nlist); P.cursor = take(nlist);         // adjacent: one memset clears both
    P.cell_start = take(nlist); P.first_task = take(nlist); P.ntasks = take(1);
    P.sorted_q = take(pairs); P.task_list = take(ub); P.task_qstart = take(ub); P.task_qcnt = take(ub);
    P.total = o;
    return P;
}
}  // namespace

// exclusive scan of one int per thread over a 256-thread block; returns the block total through `total`
__device__ __forceinline__ int ivfp_block_scan256(int v, int* s_wave, int& total) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) before += (j < wv) ? s_wave[j] : 0;
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return before + inc - v;
}

// One block per query: sample layout of its probes, totals, threshold rank; counts the queries probing every cell.
__global__ __launch_bounds__(256) void ivf_plan_query_kernel(const int64_t* __restrict__ list_off, const int* __restrict__ probes,
                                                             int nprobe, int ss, int k, double slack, int keep_all_rows,
                                                             int* __restrict__ sbase, int* __restrict__ scount,
                                                             int* __restrict__ rows, int* __restrict__ rank,
                                                             int* __restrict__ per_cell) {
    __shared__ int s_wave[4];
    __shared__ long long s_rows[4];
    const int qi = blockIdx.x, tid = threadIdx.x;
    int carry = 0;
    long long rsum = 0;
    for (int b0 = 0; b0 < nprobe; b0 += 256) {                // block-uniform
        const int p = b0 + tid;
        int ssz = 0;
        long long size = 0;
        if (p < nprobe) {
            const int c = probes[(size_t)qi * nprobe + p];
            size = list_off[c + 1] - list_off[c];
            const long long run = 16ll * ss, rem = size % run;
            ssz = (int)(16ll * (size / run) + (rem < 16 ? rem : 16));
            atomicAdd(per_cell + c, 1);
        }
        int total;
        const int ex = ivfp_block_scan256(ssz, s_wave, total);
        if (p < nprobe) sbase[(size_t)qi * nprobe + p] = carry + ex;
        carry += total;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) size += __shfl_xor(size, o);
        if ((tid & 63) == 0) s_rows[tid >> 6] = size;             // per-wave totals of the row counts
        __syncthreads();
        rsum += s_rows[0] + s_rows[1] + s_rows[2] + s_rows[3];
        __syncthreads();
    }
    if (tid == 0) {
        const long long r = rsum > 0x7FFFFFFFll ? 0x7FFFFFFFll : rsum;
        scount[qi] = carry;
        rows[qi] = (int)r;
        // rank of the sample score used as threshold: mu = expected number of the k best among the sampled rows; queries
        // whose probed rows fit the candidate list comfortably keep every row (rank 0 -> threshold -inf)
        const double den = (double)(r > 0 ? r : 1);
        const double mu = (double)k * (double)carry / den;
        double rk = floor(mu + slack * sqrt(mu + 1.0) + 4.0) + 1.0;
        const double cap = floor(0.8 * (double)ADC_CAND_CAP * (double)carry / den);
        if (rk > cap && cap >= mu + 2.5 * sqrt(mu + 1.0) + 2.0) rk = cap;
        if (rk > (double)carry) rk = (double)carry;
        if (rk < 0.0) rk = 0.0;
        rank[qi] = (r <= keep_all_rows) ? 0 : (int)rk;
    }
}

// One block: exclusive prefix sums over the cells of (queries probing the cell) and of (tasks of the cell).
__global__ __launch_bounds__(256) void ivf_plan_cells_kernel(const int* __restrict__ per_cell, int nlist,
                                                             int* __restrict__ cell_start, int* __restrict__ first_task,
                                                             int* __restrict__ ntasks) {
    __shared__ int s_wave[4];
    int cq = 0, ct = 0;
    for (int b0 = 0; b0 < nlist; b0 += 256) {
        const int c = b0 + (int)threadIdx.x;
        const int n = c < nlist ? per_cell[c] : 0, t = (n + 7) / 8;
        int tq, tt;
        const int eq = ivfp_block_scan256(n, s_wave, tq);
        const int et = ivfp_block_scan256(t, s_wave, tt);
        if (c < nlist) { cell_start[c] = cq + eq; first_task[c] = ct + et; }
        cq += tq;
        ct += tt;
    }
    if (threadIdx.x == 0) *ntasks = ct;
}

// (query, probe) pairs bucketed by cell; the order inside a cell is whatever the atomics give — it only decides which
// queries share a task, never a result.
__global__ __launch_bounds__(256) void ivf_plan_scatter_kernel(const int* __restrict__ probes, int64_t pairs, int nprobe,
                                                               const int* __restrict__ cell_start, int* __restrict__ cursor,
                                                               int* __restrict__ sorted_q) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pairs) return;
    const int c = probes[i];
    sorted_q[cell_start[c] + atomicAdd(cursor + c, 1)] = (int)(i / nprobe);
}

// task t -> (cell, first entry in sorted_q, number of queries); tasks past the device-side count get 0 queries
__global__ __launch_bounds__(256) void ivf_plan_tasks_kernel(const int* __restrict__ per_cell, const int* __restrict__ cell_start,
                                                             const int* __restrict__ first_task, const int* __restrict__ ntasks,
                                                             int nlist, int64_t ub, int* __restrict__ task_list,
                                                             int* __restrict__ task_qstart, int* __restrict__ task_qcnt) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ub) return;
    int cell = 0, qs = 0, qc = 0;
    if (t < *ntasks) {
        int lo = 0, hi = nlist;                               // last cell with first_task <= t (the non-empty one of a plateau)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (first_task[mid] <= (int)t) lo = mid; else hi = mid;
        }
        cell = lo;
        const int within = (int)t - first_task[cell];
        qs = cell_start[cell] + 8 * within;
        qc = per_cell[cell] - 8 * within;
        qc = qc > 8 ? 8 : qc;
    }
    task_list[t] = cell;
    task_qstart[t] = qs;
    task_qcnt[t] = qc;
}

// Probe selection: the nprobe cells with the largest coarse score of every query (ties at the boundary: lower cell id),
// written in ascending cell order — the search needs the SET of probed cells, not their ranking.  One block per query: the
// nlist scores as order-preserving keys in LDS, 4-pass radix select of the nprobe-th largest key, ordered compaction.
// (The framework's topk + sort + gather + argsort chain cost 0.15 ms per 1200 queries, a tenth of a search at nprobe 32.)
__global__ __launch_bounds__(1024) void ivf_probe_select_kernel(const float* __restrict__ scores, int nlist, int nprobe,
                                                                int* __restrict__ probes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* keys = reinterpret_cast<unsigned*>(smem);  // [nlist]
    __shared__ unsigned hist[256];
    __shared__ unsigned sel_prefix, sel_rank;
    __shared__ unsigned s_scan[4];
    __shared__ int s_gt[16], s_eq[16];
    const int qi = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < nlist; i += 1024) keys[i] = adc_order_key(scores[(size_t)qi * nlist + i]);
    if (tid == 0) { sel_prefix = 0u; sel_rank = (unsigned)nprobe; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = sel_prefix;
        const unsigned himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < nlist; i += 1024) {
            const unsigned k = keys[i];
            if ((k & himask) == prefix) atomicAdd(&hist[(k >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        adc_pick_bin(hist, sel_rank, prefix, shift, s_scan, &sel_prefix, &sel_rank);
    }
    const unsigned T = sel_prefix;
    const int need = (int)sel_rank;                        // how many of the cells tied at T belong to the selection
    const int chunk = (nlist + 1023) / 1024;
    const int c0 = tid * chunk, c1 = (c0 + chunk < nlist) ? c0 + chunk : nlist;
    int gt = 0, eq = 0;
    for (int c = c0; c < c1; ++c) {
        const unsigned k = keys[c];
        gt += (k > T) ? 1 : 0;
        eq += (k == T) ? 1 : 0;
    }
    // exclusive prefix sums of (gt, eq) over the 1024 threads
    const int lane = tid & 63, wv = tid >> 6;
    int igt = gt, ieq = eq;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int a = __shfl_up(igt, o), b = __shfl_up(ieq, o);
        if (lane >= o) { igt += a; ieq += b; }
    }
    if (lane == 63) { s_gt[wv] = igt; s_eq[wv] = ieq; }
    __syncthreads();
    int bgt = 0, beq = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        bgt += (j < wv) ? s_gt[j] : 0;
        beq += (j < wv) ? s_eq[j] : 0;
    }
    const int gt_before = bgt + igt - gt;
    int eq_seen = beq + ieq - eq;
    int pos = gt_before + (eq_seen < need ? eq_seen : need);
    int* out = probes + (size_t)qi * nprobe;
    for (int c = c0; c < c1; ++c) {
        const unsigned k = keys[c];
        if (k > T) {
            out[pos++] = c;
        } else if (k == T) {
            if (eq_seen < need) out[pos++] = c;
            ++eq_seen;
        }
    }
}

// scores: [nq, nlist] fp32 coarse scores (larger = closer); probes: [nq, nprobe] int32, ascending cell ids.
extern "C" int rc_ivf_select_probes(rc_handle_t h, const float* scores, int nq, int nlist, int nprobe, int* probes,
                                    rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !scores || !probes || nq < 0 || nlist <= 0 || nprobe <= 0 || nprobe > nlist) return RC_EINVAL;
    if (nlist > 16384) return RC_ESHAPE;                    // the keys of a query live in 64 KiB of LDS
    if (nq == 0) return RC_OK;
    const size_t lds = (size_t)nlist * sizeof(unsigned);
    RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)ivf_probe_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ivf_probe_select_kernel, dim3((unsigned)nq), dim3(1024), lds, (hipStream_t)stream, scores, nlist, nprobe,
                       probes);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

extern "C" size_t rc_ivf_search_probes_ws_bytes(int M, int nq, int nprobe, int nlist, int64_t sstride) {
    if (!adc_cf_supported(M) || nq <= 0 || nprobe <= 0 || nlist <= 0 || sstride <= 0) return 0;
    return ivfp_layout(ivfl_layout(M, nq, sstride).total, nq, nprobe, nlist).total;
}

// rc_ivf_search_lists with the plan made on the device.  probes [nq, nprobe]: distinct cells per query; sstride: capacity of
// a query's sample array, >= the largest possible number of sampled rows of nprobe cells (a cell of n rows contributes
// 16 floor(n / 16 ss) + min(16, n mod 16 ss)); sel_slack: standard deviations of head-room in the threshold rank;
// keep_all_rows: queries probing no more rows than this re-score every row.  Same status bits, same results.
extern "C" int rc_ivf_search_probes_q(rc_handle_t h, const uint8_t* codes, const uint8_t* image, const int64_t* list_off,
                                      const int64_t* rowmap, int64_t N, int nlist, int M, int K, const float* lut, int nq,
                                      const int* probes, int nprobe, int64_t sstride, int ss, int k, double sel_slack,
                                      int keep_all_rows, float* scores, int64_t* out_ids, int* status, int* qstatus, void* ws,
                                      size_t ws_bytes, rc_stream_t stream);
extern "C" int rc_ivf_search_probes(rc_handle_t h, const uint8_t* codes, const uint8_t* image, const int64_t* list_off,
                                    const int64_t* rowmap, int64_t N, int nlist, int M, int K, const float* lut, int nq,
                                    const int* probes, int nprobe, int64_t sstride, int ss, int k, double sel_slack,
                                    int keep_all_rows, float* scores, int64_t* out_ids, int* status, void* ws,
                                    size_t ws_bytes, rc_stream_t stream) {
    return rc_ivf_search_probes_q(h, codes, image, list_off, rowmap, N, nlist, M, K, lut, nq, probes, nprobe, sstride, ss, k,
                                  sel_slack, keep_all_rows, scores, out_ids, status, nullptr, ws, ws_bytes, stream);
}
// ... with per-query status words (qstatus [nq] int32, zeroed by the caller; may be NULL): bit 0 = the query kept fewer than
// min(k, rows probed) candidates, bit 1 = its id list overflowed.  The other queries' results stand: a caller answers only
// the flagged ones again (IVFPQIndex.search: by the per-query exact scan).  A survivor STREAM that filled up (status bit 2)
// is not attributable to a query and may have dropped anybody's rows: repeat the call with less slack.
extern "C" int rc_ivf_search_probes_q(rc_handle_t h, const uint8_t* codes, const uint8_t* image, const int64_t* list_off,
                                      const int64_t* rowmap, int64_t N, int nlist, int M, int K, const float* lut, int nq,
                                      const int* probes, int nprobe, int64_t sstride, int ss, int k, double sel_slack,
                                      int keep_all_rows, float* scores, int64_t* out_ids, int* status, int* qstatus, void* ws,
                                      size_t ws_bytes, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !image || !list_off || !rowmap || !lut || !probes || !scores || !out_ids || !status || N <= 0 ||
        nq < 0 || nprobe <= 0 || nlist <= 0 || nprobe > nlist || sstride <= 0 || ss <= 0 || k <= 0)
        return RC_EINVAL;
    if (K != RC_K || !adc_cf_supported(M) || k > ADC_CAND_CAP / 2 || N > 0xFFFFFFFFll || RC_ADC_IMG16 || nq > 32768) return RC_ESHAPE;
    if (nq == 0) return RC_OK;
    const ivfl_ws L = ivfl_layout(M, nq, sstride);
    const ivfp_ws P = ivfp_layout(L.total, nq, nprobe, nlist);
    if (!ws || ws_bytes < P.total) return RC_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)ws;
    auto I = [&](size_t off) { return (int*)(w + off); };
    const int64_t pairs = (int64_t)nq * nprobe;
    RC_HIP_CHECK(h, hipMemsetAsync(w + P.per_cell, 0, P.cell_start - P.per_cell, s));      // per_cell and cursor
    hipLaunchKernelGGL(ivf_plan_query_kernel, dim3((unsigned)nq), dim3(256), 0, s, list_off, probes, nprobe, ss, k, sel_slack,
                       keep_all_rows, I(P.sbase), I(P.scount), I(P.rows), I(P.rank), I(P.per_cell));
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(ivf_plan_cells_kernel, dim3(1), dim3(256), 0, s, (const int*)I(P.per_cell), nlist, I(P.cell_start),
                       I(P.first_task), I(P.ntasks));
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(ivf_plan_scatter_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, probes, pairs, nprobe,
                       (const int*)I(P.cell_start), I(P.cursor), I(P.sorted_q));
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(ivf_plan_tasks_kernel, dim3((unsigned)((P.ub + 255) / 256)), dim3(256), 0, s, (const int*)I(P.per_cell),
                       (const int*)I(P.cell_start), (const int*)I(P.first_task), (const int*)I(P.ntasks), nlist, P.ub,
                       I(P.task_list), I(P.task_qstart), I(P.task_qcnt));
    RC_LAUNCH_CHECK(h);
    adc_ivf_tasks T = {I(P.task_list), I(P.task_qstart), I(P.task_qcnt), I(P.sorted_q), list_off, nullptr, I(P.ntasks)};
    switch (M) {
#define IVFP_CASE(MM)                                                                                                  \
        case MM: return ivfl_launch<MM>(h, codes, image, list_off, rowmap, N, lut, nq, probes, I(P.sbase), I(P.scount),  \
                                        I(P.rows), I(P.rank), nprobe, sstride, ss, T, (int)P.ub, k, scores, out_ids,  \
                                        status, w, L, s, qstatus);
        IVFP_CASE(16) IVFP_CASE(32) IVFP_CASE(48) IVFP_CASE(64) IVFP_CASE(96)
#undef IVFP_CASE
        default: return RC_ESHAPE;
    }
}

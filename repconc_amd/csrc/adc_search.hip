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
//   4. full scan, rows with score >= tau_q are appended to a per-query candidate list of 64-bit keys.  Two
//      implementations with IDENTICAL output:
//        a. adc_scan_kernel<FILTER>: exact fp32 scores for every row (small indexes);
//        b. integer screening (N >= 2^18): each query's tables are quantised to 8 bits with a common step Delta_q
//           (l = round((LUT - min_m)/Delta_q)); the screen sums the bytes and keeps every row with S_int >= T_q, where
//           T_q = ceil((tau_q - sum_m min_m)/Delta_q - M/2) - 2 is a RIGOROUS lower bound (each of the M entries is off by
//           at most half a step, plus float rounding), so no row with exact score >= tau_q is ever lost;
//           adc_rescore_kernel (adc_common.h) then computes the exact fp32 score of the survivors (~1.4x the final
//           candidates) and applies the exact test.  The candidate set, hence the result, is the same as (a).
//           Screens: adc_screen_q16_kernel for the M with a permuted image (16, 32, 48, 64, 96: 16 queries per conflict-free
//           ds_read_b128 gather, phases of 16 sub-quantisers, i8 MFMA accumulation); for the other M (8, 24 / 12) and as the
//           tests' A/B partner (RC_ADC_OLD_SCREEN=1) the round-1 screens on the canonical codes: adc_screen_mfma_kernel
//           (v_mfma_i32_32x32x32_i8 against a selection matrix) and adc_screen_kernel (VALU, M % 8 != 0).  The screen
//           generations in between (round-2 8-query conflict-free screen, two-phase / two-pass M = 96 forms) were removed
//           in round 4; the list-centric IVF search lives in ivf_lists.hip.
//   5. adc_select_kernel     per query: radix-select cut to the k best scores (+ties), bitonic sort in LDS, emit top-k
//
// The scan is the hot kernel.  A block keeps the LUTs of QT queries in LDS, interleaved
// [m][k][QT] so ONE ds_read_b64 / b128 gather serves QT queries, and streams a tile of codes
// (consecutive lanes = consecutive rows, 16-byte loads).  Blocks that share a code tile are
// adjacent in the grid, so a tile is fetched from HBM about once per XCD and re-read from L2.
#include "adc_common.h"
#include <stdio.h>
#include <string.h>

#include <type_traits>


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
#ifndef ADC_Q16_LATE_TEST
#define ADC_Q16_LATE_TEST 1        // 1: a round's survivor test runs one step later, pair by pair, in front of the MFMAs that restart the sums
#endif
#define ADC_Q16_SCAP 256           // survivor entries per wave held in LDS between flushes
// adc_q16_pos / adc_q16_slot: adc_common.h (shared with the 16-query IVF screen)

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
    // wv in an SGPR: everything derived from it (table pieces, code addresses, survivor list) is scalar + lane offset
    const int tid = threadIdx.x, l = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = l & 15, g = l >> 4;
    const unsigned xcd = blockIdx.x & 7u, jx = blockIdx.x >> 3;
    unsigned group, btile;
    if (!(flags & 2)) {                                       // XCD x owns the groups == x (mod 8), every tile: the XCD's ~10
        const unsigned gpx = (unsigned)(groups + 7) / 8u;     // groups' tables stay in its L2, a tile is read once per XCD
        group = (jx % gpx) * 8u + xcd;
        btile = jx / gpx;
        if (group >= (unsigned)groups) return;                // block-uniform
    } else {                                                  // few queries (JPQ steps, validation): every XCD takes ALL the
        group = jx % (unsigned)groups;                        // groups (their tables fit its L2) and the tiles == x (mod 8): the
        btile = (jx / (unsigned)groups) * 8u + xcd;           // image is read ONCE from HBM instead of once per XCD
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
    int held[2] = {phase_of(0), -1};                         // block-uniform: the phase each table buffer holds (or is being sent)
    // per-wave survivor list in LDS: entries (row in tile << 4 | query column); flushed to the per-query id lists (one
    // global atomic per entry, all of a flush in flight together) when 64 more might not fit, and at the end
    constexpr int SCAP = ADC_Q16_SCAP;
    unsigned* sbuf = reinterpret_cast<unsigned*>(smem + 2 * BUF) + wv * SCAP;
    int scount = 0;                                          // wave-uniform
    bool flushed = false;                                    // wave-uniform: this wave issued stores / atomics since the last phase change
    auto flush_survivors = [&]() {
        flushed = flushed || scount > 0;
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
    // survivor test of one chunk's sums (D[row = 4 g + e][column = r]); survivors go to the wave's LDS list.  Round 6: this path
    // is not rare — ~9 k survivors per query of 8.84 M rows at k = 1000 = 0.25 per chunk of 16 rows x 16 queries, one chunk in
    // five has one — and with it compiled out the kernel is 5-10 % (M = 48) to 15 % (M = 32) faster.  A branch-free form (four
    // compares into scalar masks, one capacity check, range test only for the index's last chunks) measured 4 % SLOWER than
    // this one (128 VGPRs, the masks of all four sums computed for every tested chunk): profiles/r06c_adc_keep_chunk.txt.
    auto test_chunk = [&](const adc_i32x4v& v, unsigned rbase) {
#ifdef ADC_EXP_NOSURV          // A/B timing only (tools/adc_ab.sh): the sums are consumed by one cheap instruction, no test
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
        return;
#endif
        const int top = max(max(v[0], v[1]), max(v[2], v[3]));
        if (__ballot(top >= tq)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned n = rbase + 4u * g + e;
                const bool hit = v[e] >= tq && n < nrows;
                const unsigned long long mask = __ballot(hit);
                if (mask) {                                       // wave-uniform
                    if (scount + 64 > SCAP) flush_survivors();
                    // position among the hits below this lane: v_mbcnt (no lane-mask registers kept live through the chunk loop)
                    if (hit) sbuf[scount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))] =
                        (n << 4) | (unsigned)r;
                    scount += (int)__popcll(mask);
                }
            }
        }
    };
    // ONE global_load_dwordx4 per call, and the phase change's `s_waitcnt vmcnt(NV)` counts exactly the NV calls of a step as
    // the youngest loads in flight: the immediate of that wait IS NV (same constant), the type is 16 bytes (one instruction),
    // and the loads must stay behind the asm waits (they are volatile asm with a memory clobber; the loads are plain C++).
    static_assert(sizeof(adc_u32x4v) == 16 && NV == R / 4, "vmcnt(NV) at the phase change counts one dwordx4 load per load_quad");
    auto load_quad = [&](int it, int v) {
        return *reinterpret_cast<const adc_u32x4v*>(tile + ((size_t)phase_of(it) * (TILE * 16) + (size_t)(it / NPH) * (ROUND * 16) +
                                                            lane_at + 16 * v));
    };
    auto run_step = [&](int it, adc_u32x4v (&w)[NV]) {
        if (it > 0 && phase_of(it) != phase_of(it - 1)) {
            if constexpr (NPH <= 2) {
                // both phases' tables stay in the two buffers (M = 32): a phase change is an address offset, nothing else
                buf ^= 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) offb[j] = off[j] + (unsigned)buf * (unsigned)BUF;
            } else {
            // phase change: this phase's tables were requested into the other buffer one segment ago
            // my pieces have landed: everything but the NV code loads of the step just finished, which are younger than any table
            // piece (a piece is requested at a phase change, the codes after it) — round 5: vmcnt(0) also waited for those
            // (vmcnt counts loads in order; a flush's stores and atomics may complete out of order with respect to them, and the
            // tile's last step follows a step that requested no codes: both cases wait for everything)
            if (it + 1 < nsteps && !flushed) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            flushed = false;
            // everybody's have, and everybody is done with the old buffer (every LDS read of the step has returned: its last
            // gather_wait is lgkmcnt(0)).  The bare instruction: __syncthreads() is fence + barrier, and the fence is
            // s_waitcnt vmcnt(0) — it would wait for the code loads issued a few chunks ago after all (M = 64 / 96: -1 %).
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            buf ^= 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) offb[j] = off[j] + (unsigned)buf * (unsigned)BUF;
            // the buffer just vacated gets the table of the segment after this one - unless it still holds it: with the phases
            // visited 0 .. P-1 | P-1 .. 0 | ... the segment before and the segment after a turning point are the same phase
            // (M = 48: every other segment is phase 1 and phase 1 never leaves buffer 1: one 64 KiB refill per round instead of
            // two).  The refill's LDS writes compete with the gathers: 8.2 -> 7.8 ms per 1200 queries at M = 48.
            const int nx = next_change(it);
            if (nx < nsteps && phase_of(nx) != held[buf ^ 1]) {
                stage(phase_of(nx), buf ^ 1);
                held[buf ^ 1] = phase_of(nx);
            }
            }
        }
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
                // issued by hand so that the wait can be counted by hand (gather_wait): the compiler's own bookkeeping waits
                // for lgkmcnt(0) before a chunk's first MFMA, i.e. also for the next chunk's gathers it has just issued
                asm volatile("ds_read_b128 %0, %1" : "=v"(e[j]) : "v"(addr));
            }
        };
        const bool first = (it % NPH == 0);                   // block-uniform: the round's first step starts from zero
        const bool last = (it % NPH == NPH - 1);
        const unsigned r0 = (unsigned)(it / NPH) * ROUND + (unsigned)(wv * R * 16);
        // LDS reads return in order: with `newer` gathers issued after this chunk's four, lgkmcnt(newer) means these four have
        // landed (any other outstanding LDS / scalar operation only makes the wait longer, never shorter than needed)
        auto gather_wait = [&](adc_u32x4v (&e)[4], bool more_in_flight) {
            if (more_in_flight) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
        };
        auto fold = [&](int c, const adc_u32x4v (&e)[4]) {
            if constexpr (!PACK) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const adc_i32x4v a = {(int)e[j][0], (int)e[j][1], (int)e[j][2], (int)e[j][3]};
                    acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, acc[c], 0, 0, 0);
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
        // A round's sums start from explicitly cleared accumulators (32 v_mov per 96 gathers): round 3 gave the round's first
        // MFMA a literal-zero C instead, which put a scalar branch into every chunk and the MFMAs into basic blocks of their
        // own — the loop body is straight-line now (8.62 -> 8.38 ms per 1200 queries at M = 48; with the hand-counted waits 8.26).
        constexpr bool LATE = ADC_Q16_LATE_TEST != 0 && !PACK;
        if constexpr (!PACK && !LATE) {
            if (first) {
#pragma unroll
                for (int c = 0; c < R; ++c) acc[c] = adc_i32x4v{0, 0, 0, 0};
            }
        }
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
            // Code loads roll TWO steps ahead (round 5; one step ahead into a second register set before): the sixteen bytes
            // of chunks c - 2 .. c + 1 are used up, so the same chunks' codes of the step after next go there.  No extra
            // registers, and the phase change no longer waits for a load issued at the start of the step it ends:
            // 7.95 -> 7.62 ms per 1200 queries at M = 48 (M = 96: -3 %; M = 32, which has no phase change: unchanged).
            if ((c & 3) == 2 && it + 2 < nsteps) w[c >> 2] = load_quad(it + 2, c >> 2);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LATE) {
                // Round 6.  The survivor test of a round used to follow its last MFMAs directly: every wave then sat through the
                // matrix pipe's latency (four dependent MFMAs per chunk, queued behind the SIMD's other waves) with no gather in
                // flight, once per round.  Now the sums of chunks c, c + 1 of the PREVIOUS round are tested here, in the
                // round's first step, behind the gathers just issued and right before the MFMAs that restart them from zero:
                // their MFMAs are a whole step old, nothing is waited for, and the test's VALU work overlaps LDS latency.
                // Worth 1-2 % (M = 48: 7.41 -> 7.28, 7.50 -> 7.38 ms; M = 96: 14.7 -> 14.4; profiles/r06b_adc_late_test.txt):
                // the cost of the survivor path is its instructions (see test_chunk), not this wait.
                if (first) {                                      // block-uniform
                    if (it > 0) {
                        const int top = max(max(max(acc[c][0], acc[c][1]), max(acc[c][2], acc[c][3])),
                                            max(max(acc[c + 1][0], acc[c + 1][1]), max(acc[c + 1][2], acc[c + 1][3])));
                        if (__ballot(top >= tq)) {
                            test_chunk(acc[c], r0 - (unsigned)ROUND + 16u * c);
                            test_chunk(acc[c + 1], r0 - (unsigned)ROUND + 16u * (c + 1));
                        }
                    }
                    acc[c] = adc_i32x4v{0, 0, 0, 0};
                    acc[c + 1] = adc_i32x4v{0, 0, 0, 0};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            gather_wait(ea, true);
            fold(c, ea);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < R) gather(c + 2, ea);
            __builtin_amdgcn_sched_barrier(0);
            gather_wait(eb, c + 2 < R);
            fold(c + 1, eb);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!PACK && !LATE) {
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
    if (nsteps > 1) load_step(1, wb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (NPH > 2) {
        const int nx = next_change(0);
        if (nx < nsteps) {
            stage(phase_of(nx), 1);
            held[1] = phase_of(nx);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) offb[j] = off[j];
    for (int it = 0; it < nsteps; it += 2) {                 // block-uniform
        run_step(it, wa);                                    // even steps live in wa, odd ones in wb
        if (it + 1 < nsteps) run_step(it + 1, wb);
    }
    if constexpr (ADC_Q16_LATE_TEST != 0 && !PACK) {        // the tile's last round has no next step to be tested in
        if (nsteps > 0) {
            const unsigned rl = (unsigned)(nrounds - 1) * ROUND + (unsigned)(wv * R * 16);
#pragma unroll
            for (int c = 0; c < R; ++c) test_chunk(acc[c], rl + 16u * c);
        }
    }
    flush_survivors();
}

// ------------------------------------------------------------------------------------------ host
extern "C" size_t rc_adc_scan_image_bytes(int64_t N, int M);
struct adc_ws_layout {
    size_t lut, sample, thr, cnt, cand, qlut, tint, qstat, idcnt, ids, image, total;
    int64_t S;
};
static int adc_qs_for(int M) { (void)M; return 16; }   // table groups are sized for 16 queries (covers the 8- and 4-query kernels)
// M with a permuted image = the M of the 16-query screen (and of the IVF screen): adc_cf_supported (adc_common.h)
static bool adc_use_image(int64_t N, int M) {
    return N >= ADC_SCREEN_MIN_N && adc_cf_supported(M) && !rc_env_set("RC_ADC_VALU_SCREEN") && !rc_env_set("RC_ADC_OLD_SCREEN");
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
        const size_t qb = (size_t)((nq + QS - 1) / QS) * M * RC_K * QS;
        L.qlut = o;  o += rc_align_up(qb, 256);
        L.tint = o;  o += rc_align_up((size_t)nq * sizeof(int), 256);
        L.qstat = o; o += rc_align_up((size_t)nq * ADC_QSTAT_STRIDE * sizeof(float), 256);
        L.idcnt = o; o += rc_align_up((size_t)nq * sizeof(unsigned), 256);
        L.ids = o;   o += rc_align_up((size_t)nq * ADC_ID_CAP * sizeof(unsigned), 256);
    }
    L.image = o;
    if (own_image && N >= ADC_SCREEN_MIN_N && adc_cf_supported(M)) o += rc_align_up(rc_adc_scan_image_bytes(N, M), 256);
    L.total = o;
    return L;
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
// bytes of the permuted code image of an N-row index (0: this M has no image — M = 8, 12, 24 run the round-1 screens on the
// canonical codes): whole tiles of ADC_Q16_TILE rows, [tile][phase][round][wave][lane][chunk][step] (adc_q16_image_at)
// Where a finished search left its per-query counts inside the caller's workspace (measurement only: SURVEY 8d asks for the
// screen's survivors next to queries/s): *survivors_off = byte offset of unsigned[nq] rows that passed the 8-bit screen
// (0 when the index is too small for a screen), *candidates_off = unsigned[nq] rows kept by the exact rescoring.
// own_image: the workspace of rc_adc_search_ws_bytes (1) or rc_adc_search_img_ws_bytes (0) — the offsets are the same.
extern "C" int rc_adc_search_ws_counts(int64_t N, int M, int K, int nq, size_t* survivors_off, size_t* candidates_off) {
    if (N <= 0 || M <= 0 || K != RC_K || nq <= 0 || !survivors_off || !candidates_off) return RC_EINVAL;
    const adc_ws_layout L = adc_layout(N, M, nq, false);
    *survivors_off = (N >= ADC_SCREEN_MIN_N) ? L.idcnt : 0;
    *candidates_off = L.cnt;
    return RC_OK;
}

extern "C" size_t rc_adc_scan_image_bytes(int64_t N, int M) {
    if (N < 0 || !adc_cf_supported(M)) return 0;
    const int64_t T = ADC_Q16_TILE;
    return (size_t)((N + T - 1) / T * T) * M;
}
// Host-side description of the conflict-free slot rule (adc_common.h; no GPU involved; what tests/test_abi.py checks): for
// lane `lane` (0..63) of a wave and gather step `step` (0 .. steps_per_lane-1) of a table phase of PM sub-quantisers (PM = M,
// 48 for M = 96): the 8-byte LDS slot it reads and the phase-relative sub-quantiser that slot belongs to.  Returns the number
// of steps per lane, or RC_ESHAPE.  (The IVF screen applies the rule to phases of 32 / 16: ivfs_pm, ivf_lists.hip.)
extern "C" int rc_adc_cf_describe(int M, int lane, int step, int* slot, int* m, int* slots_per_code, int* phases) {
    if (!adc_cf_supported(M)) return RC_ESHAPE;
    const int PM = M == 96 ? 48 : M;
    if (lane < 0 || lane > 63 || step < 0 || step >= PM / 4 || !slot || !m) return RC_EINVAL;
    adc_cf_step(PM, step, lane & 15, lane >> 4, *slot, *m);
    if (slots_per_code) *slots_per_code = 32 * (PM / 32 + (PM % 32) / 16);
    if (phases) *phases = M / PM;
    return PM / 4;
}

// (Re)build rows [n0, n0 + n) of the flat-search image (what rc_adc_search_img / rc_adc_search_q take) from the canonical
// codes [N, M] (both pointers = row 0 of the index): rc_adc_scan_image_bytes(N, M) bytes
extern "C" int rc_adc_scan_image(rc_handle_t h, const uint8_t* codes, int64_t n0, int64_t n, int M, uint8_t* image,
                                 rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !image || n0 < 0 || n < 0) return RC_EINVAL;
    if (!adc_cf_supported(M)) return RC_ESHAPE;
    if (n == 0) return RC_OK;
    int64_t blocks = (n * M + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(adc_q16_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, codes, n0, n, M, image);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}
// host-side description of the 16-query screen's layout (tests): slot read by `lane` in step j, or RC_ESHAPE
extern "C" int rc_adc_q16_describe(int M, int lane, int step, int* slot) {
    if (!adc_cf_supported(M)) return RC_ESHAPE;
    if (lane < 0 || lane > 63 || step < 0 || step > 3 || !slot) return RC_EINVAL;
    *slot = adc_q16_slot(lane, step);
    return 1;                                                // every M with an image uses the layout
}

static int adc_qt_for(int M) {
    if (M <= 32) return 4;   // <= 128 KiB of tables
    if (M <= 64) return 2;   // M=48: 96 KiB, M=64: 128 KiB
    return 1;                // M=96: 96 KiB
}

struct adc_bufs {
    float* lut; float* sample; float* thr; unsigned* cnt; unsigned long long* cand;
    uint8_t* qlut; int* tint; unsigned* idcnt; unsigned* ids; float* qstat;
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
    if (CF && image != nullptr) {
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
            // block -> (group, tile).  Many groups: XCD x owns the groups == x (mod 8) and walks every tile — each XCD streams the
            // whole image (8 x N M bytes from HBM per launch: nothing beside 1200 queries' gathers).  Few groups (round 6): with
            // 8 groups that traffic IS the launch (3.4 GB per 128-query search: 1.04 ms where 8 / 75 of a full launch is 0.78), so
            // while all the groups' tables fit an XCD's L2 (groups x M / 16 x 64 KiB <= 4 MiB) the TILES are dealt to the XCDs and
            // every XCD takes all groups: the image is read once.  [MI355X] ms per search, k = 200, M = 48: 128 queries 1.27 -> 1.15,
            // 256: 1.91 -> 1.80, 384: 2.71 -> 2.63, 512: 3.30 -> 3.34 (not used there); M = 32 / 96 at 128 queries: 1.02 -> 0.94,
            // 2.20 -> 2.04 (profiles/r06s_adc_small_batches.txt).
            const int by_tiles_max = rc_env_int("RC_ADC_Q16_TILESPLIT_GROUPS", 64 / (M / 16));
            const bool by_tiles = groups < 8 || groups <= by_tiles_max;
            const unsigned nblocks = !by_tiles ? 8u * gpx * tiles16 : 8u * (unsigned)groups * ((tiles16 + 7u) / 8u);
            rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
            hipLaunchKernelGGL(kern, dim3(nblocks), dim3(TH), sl, s, image, N, b.qlut, b.tint, nq, groups, b.idcnt, b.ids,
                               (rc_env_int("RC_ADC_Q16_PRIO", 1) ? 1 : 0) | (by_tiles ? 2 : 0));
            rc_prof_mark(h, RC_PROF_ADC_SCAN, s);
            RC_LAUNCH_CHECK(h);
        }
    } else if constexpr (M % 8 == 0) {
        // M without an image (8, 24; or RC_ADC_OLD_SCREEN=1, the tests' A/B partner): the round-1 screens on the canonical codes
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
                         int64_t id_offset, float* scores, int64_t* ids, int* status, hipStream_t s, int* qstatus) {
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
    if (adc_use_image(N, M)) {
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
                           (int*)(w + L.tint), (unsigned*)(w + L.idcnt), (unsigned*)(w + L.ids), (float*)(w + L.qstat)};
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

// Any other M (the reference's IndexPQ takes every divisor of the hidden size: modeling_repconc.py:41, evaluate_repconc.py:81):
// exact scores with a run-time width — thread = row, the row's codes walked once for up to ADC_EXACT_QX queries, tables read
// through the caches (M x 1 KiB per query: no LDS size fits every M), m-ascending fp32 sums like every other scoring path.
__global__ __launch_bounds__(256) void adc_scan_rt_kernel(const uint8_t* __restrict__ codes, int64_t N, int M,
                                                          const float* __restrict__ lut, int nq, float* __restrict__ sc) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const uint8_t* cp = codes + n * M;
    float s[ADC_EXACT_QX];
#pragma unroll
    for (int t = 0; t < ADC_EXACT_QX; ++t) s[t] = 0.f;
    for (int m = 0; m < M; ++m) {
        const unsigned c = cp[m];
#pragma unroll
        for (int t = 0; t < ADC_EXACT_QX; ++t)
            if (t < nq) s[t] = s[t] + lut[((size_t)t * M + m) * RC_K + c];
    }
#pragma unroll
    for (int t = 0; t < ADC_EXACT_QX; ++t)
        if (t < nq) sc[(size_t)t * N + n] = s[t];
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
            default:
                hipLaunchKernelGGL(adc_scan_rt_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, codes, N, M, lq, nx, sc);
                RC_LAUNCH_CHECK(h);
                rc = RC_OK;
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

// Shared pieces of the flat ADC search (adc_search.hip) and the list-centric IVF search (ivf_lists.hip): limits, the
// order-preserving score key, the k-th-largest selection, the conflict-free slot rule, the 8-bit quantiser and the exact
// rescoring kernel.  Device functions and templates only (no -fgpu-rdc: every translation unit compiles its own copy).
#pragma once
#include "rc_common.h"
#include <limits.h>

#define ADC_THREADS 1024
#define ADC_SAMPLE_MAX 32768
#define ADC_KTH_LIST 4096            // members of the selected value bin kept in LDS by adc_kth_largest_v
#define ADC_CAND_CAP 16384
#define ADC_TILE_DOCS 32768
#define ADC_SCREEN_MIN_N (1 << 18)
#define ADC_ID_CAP 32768
#define ADC_QSTAT_STRIDE 128          // floats per query: lo[0..M), sum of lo as a double at [124], delta at [127]

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

// ---- conflict-free slot rule (round 2; today the table phases of the IVF screen, ivf_lists.hip) ------------------
// The sum over sub-quantisers is commutative, so the lanes of a wave need not visit them in the same order: byte tables are
// laid out [code][slot][8 queries] with one 8-byte SLOT per sub-quantiser (a slot's LDS bank pair is slot mod 32 whatever the
// code) and in every step the 32 lanes the LDS services together read 32 DIFFERENT slots mod 32: lane (r, g) of a 16-row
// chunk walks block-relative sub-quantiser (r + (S/4) tau(g) + j) mod S in step j, S = 32 or 16 the size of the block of
// sub-quantisers, tau(g) = 2 (g & 1) + (g >> 1); a 16-block is stored twice (slots 16 apart), lanes 16-31 of the group use
// the second copy.  Conflict-free BY CONSTRUCTION, for any codes; a lane receives its codes in its own visiting order from a
// permuted image of the code matrix (the permutation of row n depends on n mod 16 only).
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

__device__ __forceinline__ unsigned adc_quant8(float v, float lo, float delta) {
    int l = (int)floorf((v - lo) / delta + 0.5f);           // nearest: the screen's one-sided slack is M / 2 + 2 steps, not M + 2
    l = l < 0 ? 0 : (l > 255 ? 255 : l);
    return (unsigned)l;
}

typedef int adc_i32x4v __attribute__((ext_vector_type(4)));
typedef unsigned adc_u32x2v __attribute__((ext_vector_type(2)));

// Tasks of the list-centric IVF search: a task = (coarse cell, up to 8 of the queries that probe it)
// 16-query gathers (ds_read_b128; adc_search.hip 4b'', ivfs_screen16.h): position of a lane inside its service group of 16
// lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32), and the sub-quantiser (within its phase of 16) = LDS slot
// that lane `lane` of a wave reads in step j — distinct inside every service group, the four lanes of a row cover all 16
__host__ __device__ constexpr int adc_q16_pos(int h32) {
    return (h32 < 4) ? h32 : (h32 < 12) ? h32 - 4 : (h32 < 16) ? h32 - 8 : (h32 < 20) ? h32 - 8 : (h32 < 28) ? h32 - 12 : h32 - 16;
}
__host__ __device__ constexpr int adc_q16_slot(int lane, int j) { return (adc_q16_pos(lane & 31) + j + 4 * (lane >> 5)) & 15; }

struct adc_ivf_tasks {
    const int* task_list;        // [tasks] cell of the task
    const int* task_qstart;      // [tasks] first entry of the task's queries in sorted_q
    const int* task_qcnt;        // [tasks] 1 .. 8 queries (1 .. 16 for the 16-query screen)
    const int* sorted_q;         // query ids ordered by probed cell
    const int64_t* list_off;     // [nlist + 1] row ranges of the cells
    const uint8_t* qbyte;        // [nq][NP][256][PM] per-query byte tables, one byte per sub-quantiser
    const int* ntasks;           // device-side task count when the list is padded (rc_ivf_search_probes), else NULL
};

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

// widths the screened flat search is compiled for (rc_adc_search*: ADC_CASE); every other divisor of D is answered by the exact
// scan with a run-time width (rc_adc_search_exact, adc_scan_rt_kernel)
static inline bool adc_search_supported(int M) {
    return M == 8 || M == 12 || M == 16 || M == 24 || M == 32 || M == 48 || M == 64 || M == 96;
}
static inline bool adc_cf_supported(int M) { return M == 16 || M == 32 || M == 48 || M == 64 || M == 96; }

// adc_search.hip: sort + emit of the per-query key lists
int rc_adc_launch_select(rc_handle_t h, unsigned long long* cand, const unsigned* cnt, int nq, int64_t N, int k,
                         int64_t id_offset, float* scores, int64_t* ids, int* status, hipStream_t s, int* qstatus = nullptr);

// Lloyd / k-means kernels of the warm-up and the index build (train/run_warmup.py:85-132, inside Faiss in the reference):
// sufficient statistics (exact fixed-point, order-independent), centroid update, Faiss's empty-cluster rule.
#include "rc_common.h"
#include <string.h>

#include <math.h>

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
// the lanes at work, 1.3 ms for 65 536 rows.  Integer addition does not care about the order: every value is split as
//   x 2^s = hi + r,  hi = rne(x 2^s),  lo = rne(r 2^38)
// (s sized so that n values below the bound 2^eb cannot overflow 61 bits; both parts exact for every x down to 2^(eb-51)),
// (hi, lo) are summed as 64-bit INTEGERS — LDS accumulators per (strip, sub-quantiser), then per-strip partials, then one
// integer sum over the strips — and the exact 128-bit total hi 2^38 + lo is rounded to fp64 ONCE.  While every part is
// exact the result is the correctly rounded real sum: it depends on nothing but the data — not on the order, the strips,
// the launch geometry or the scale s.
//
// Round 4 (the round-3 kernel ran at 0.15 of the HBM roof): (1) the split is integer arithmetic on the fp32 bits (shift
// of the 24-bit significand) instead of fp64 multiply / rint / 64-bit conversions, ~3x fewer VALU instructions per value;
// (2) accumulators laid out [e][k][q] so that a row's four lanes hit four different bank pairs (5.1 -> 2.6 expected
// passes per LDS atomic with random codes); (3) ONE round of equal blocks — (strip, m) pairs, as many as the chip has
// slots, XCD x owns a contiguous range of pairs so that the 64-byte halves of a line and the rows' codes meet in one L2 —
// which store their partials plainly (round 3: 24 576 blocks x 4096 memory-side 64-bit atomics = 100 M atomics per 2^20
// rows); (4) no separate max|x| pass over the data: the scale comes from a per-handle HINT (device word, the bound of
// the previous call + one bit), every block tracks max|x| and inexact splits while it works, and the (normally idle,
// device-gated) second pass repeats the work with the tight bound of THIS data when the hint was exceeded or a part had
// to be rounded.  The answer is a pure function of the data either way.
// (Rounds 3-4 sent non-finite input through the strip kernels; round 5 keeps it on this path, see below.  The strip kernels
// remain for rows that are not float4-addressable, widths that are not multiples of 4, and 2^24 rows or more per call.)
#define KM_FX_LO_BITS 38
#define KM_FX_U 4                                       // rows in flight per thread
#define KM_FX_EB0 4                                     // first hint of a handle: |x| < 16
//
// Round 5 (0.34 of the roof at M = 64 at every size; seven launches per call, four of them idle):
//  * a block's share of a row is a PIECE of 32 consecutive floats at a 128-byte boundary whatever the sub-vector width — whole
//    cache lines for every M (dsub = 12 / 24 / 48 used to give 96- or 64-byte pieces: each line fetched by two blocks); lane q of
//    a piece (one float4) belongs to sub-quantiser (32 p + 4 q) / dsub and adds into the LDS accumulators of ITS code;
//  * three launches, the middle one never idle: A adds the rows and stores per-strip partials, B reads every piece's verdict
//    ("did the hint hold, was nothing rounded" — decided PER PIECE, on the piece's own data) and either takes its share of the
//    piece's finish (integer sum over the strips, one rounding, into sums / counts) or repeats the piece with the tight bound of
//    its data, C finishes the repeated pieces (normally none: every block leaves at once) and closes the call;
//  * the low parts and the non-finite marks of a block are written (and read back) only when it used them;
//  * inf / NaN no longer leave the fixed-point path: an accumulator carries three marks (+inf, -inf, NaN seen) and the total is
//    what IEEE addition gives in ANY order (NaN if NaN or both infinities, else the infinity, else the exact finite sum).
// Built and measured on the way (profiles/r05*_kmeans*.txt): the finish done by each piece's last-arriving block (one launch, as
// the Sinkhorn sweep's reducer) costs 55 us at 65 536 rows — the 24 reducing CUs read 650 KB each past their L2 at ~12-25 GB/s,
// one CU's share of the memory system — against 11 us for launch B, whose finish runs on every CU.
// The control block lives on the handle (device memory): nothing depends on a host-side call count, so a captured sequence of
// calls (the warm-up's round graph) replays correctly from any state.
#define KM_PX_TPR 8                                     // float4 lanes per row piece
#define KM_PX_MAXM 8                                    // sub-quantisers a piece can touch (dsub >= 4)
#define KM_PX_MAXP 64                                   // pieces of a row (D <= 2048)

// Control block (device memory on the handle).  Nothing in it is ever reset: the per-piece words carry the call's sequence
// number in their high bits and are combined with atomicMax, so what an earlier call left behind is simply smaller; `seq` and
// `hint` are advanced by the block of the call's last launch that arrives last (every block of that launch has read them by then).
struct km_piece_ctl {
    unsigned long long am;                               // max over the blocks of (seq << 8 | largest finite exponent field)
    unsigned long long fl;                               // max of (seq << 1 | a low part was rounded)
    unsigned long long redo;                             // (seq << 16 | tight bound + 32768) when launch B repeated the piece
    unsigned long long pad[5];
};
struct km_ctl {
    int hint;                                            // |x| < 2^hint expected by the next call
    unsigned carrive;                                    // blocks of launch C that have read what they need (the last one closes the call)
    unsigned long long seq;                              // calls completed on this handle
    unsigned long long pad[6];
    km_piece_ctl piece[KM_PX_MAXP];
};

__device__ __forceinline__ int km_tight_eb(unsigned am) { return am == 0u ? -126 : (int)(am >> 23) - 126; }   // |x| < 2^eb

// x 2^sexp = hi + lo 2^-38 on the bits of x (finite).  Returns true when lo had to be rounded.
__device__ __forceinline__ bool km_fx_split(unsigned b, int sexp, long long& hi, long long& lo) {
    int e = (int)((b >> 23) & 0xFFu);
    unsigned m = b & 0x7FFFFFu;
    m |= e ? 0x800000u : 0u;
    e = e ? e : 1;
    const int sh = e - 150 + sexp;                              // |x| 2^sexp = m 2^sh
    long long h, l = 0;
    bool inexact = false;
    if (sh >= 0) {
        h = (long long)((unsigned long long)m << (sh > 40 ? 40 : sh));     // > 40: above the bound, the pass is discarded
    } else {
        const int t = -sh;
        if (t <= 24) {
            unsigned h0 = m >> t;
            const unsigned rem = m & ((1u << t) - 1u), half = 1u << (t - 1);
            h0 += (rem > half || (rem == half && (h0 & 1u))) ? 1u : 0u;
            const int r = (int)m - (int)(h0 << t);              // |r| <= 2^(t-1)
            h = (long long)h0;
            l = (long long)r << (KM_FX_LO_BITS - t);
        } else if (t <= KM_FX_LO_BITS) {
            h = 0;
            l = (long long)m << (KM_FX_LO_BITS - t);
        } else {
            h = 0;
            const int u = t - KM_FX_LO_BITS;                    // lo = rne(m 2^-u)
            if (u >= 26) {
                l = 0;
                inexact = m != 0u;
            } else {
                unsigned l0 = m >> u;
                const unsigned rem = m & ((1u << u) - 1u), half = 1u << (u - 1);
                l0 += (rem > half || (rem == half && (l0 & 1u))) ? 1u : 0u;
                l = (long long)l0;
                inexact = rem != 0u;
            }
        }
    }
    const long long sm = -(long long)(b >> 31);                 // 0 or -1
    hi = (h ^ sm) - sm;
    lo = (l ^ sm) - sm;
    return inexact;
}

// One row's float4 (lane q of the piece) into the LDS accumulators of centroid k: accumulator (k, q, e) at e ESTRIDE + 8 k + q.
// Fast path (every value's significand is shifted LEFT: nothing below the integer grid): per value 3 + 3 + 1 + 1 integer
// instructions and one ds_add_u64 at an immediate offset; otherwise the general split.  emax collects the exponent fields of
// the FINITE values; bit 0 of the result: a low part was rounded, bit 1: a low part was used, bit 2: a non-finite value.
__device__ __forceinline__ unsigned km_px_row(const float4 v, int k, int q, int sexp, unsigned long long* __restrict__ hi,
                                              unsigned long long* __restrict__ lo, unsigned* __restrict__ nfl, unsigned& emax) {
    constexpr int ESTRIDE = RC_K * KM_PX_TPR;
    const unsigned b[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    unsigned long long* a = hi + (k * KM_PX_TPR + q);
    unsigned e[4];
    int sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        e[i] = (b[i] >> 23) & 0xFFu;
        sh[i] = (int)(e[i] ? e[i] : 1u) - 150 + sexp;       // |x| 2^sexp = significand 2^sh
    }
    const int shmin = min(min(sh[0], sh[1]), min(sh[2], sh[3]));
    const int shmax = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
    unsigned what = 0u;
    const unsigned e4 = max(max(e[0], e[1]), max(e[2], e[3]));
    // (e4 != 0xFF: with a very large hint an inf / NaN would pass the shift test and be added as a finite significand)
    if (shmin >= 0 && shmax <= 40 && e4 != 0xFFu) {
        emax = max(emax, e4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sig = (int)((b[i] & 0x7FFFFFu) | ((e[i] ? 1u : 0u) << 23));
            const int sg = (int)b[i] >> 31;                                  // 0 / -1
            const long long sv = (long long)((sig ^ sg) - sg);              // signed significand
            atomicAdd(a + i * ESTRIDE, (unsigned long long)(sv << sh[i]));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (e[i] == 0xFFu) {                                             // inf / NaN: a mark on the accumulator, no value
                const unsigned mark = (b[i] & 0x7FFFFFu) ? 4u : ((b[i] >> 31) ? 2u : 1u);
                const int acc = i * ESTRIDE + k * KM_PX_TPR + q;
                atomicOr(nfl + (acc >> 2), mark << (8 * (acc & 3)));
                what |= 4u;
                continue;
            }
            emax = max(emax, e[i]);
            long long hl, ll;
            if (km_fx_split(b[i], sexp, hl, ll)) what |= 1u;
            atomicAdd(a + i * ESTRIDE, (unsigned long long)hl);
            if (ll) {
                atomicAdd(lo + (k * KM_PX_TPR + q) + i * ESTRIDE, (unsigned long long)ll);
                what |= 2u;
            }
        }
    }
    return what;
}

// (hi 2^38 + lo) 2^(-sexp-38), rounded to nearest-even once
__device__ __forceinline__ double km_fx_to_double(long long H, long long L, int sexp) {
    __int128 t = ((__int128)H << KM_FX_LO_BITS) + (__int128)L;
    const bool neg = t < 0;
    unsigned __int128 u = neg ? (unsigned __int128)(-t) : (unsigned __int128)t;
    const unsigned long long uh = (unsigned long long)(u >> 64), ul = (unsigned long long)u;
    if ((uh | ul) == 0ull) return 0.0;
    const int msb = uh ? 127 - __clzll((long long)uh) : 63 - __clzll((long long)ul);
    double r;
    if (msb <= 52) {
        r = (double)ul;
    } else {
        const int sh = msb - 52;
        unsigned long long qq = (unsigned long long)(u >> sh);                    // 53 bits
        const unsigned __int128 rem = u & ((((unsigned __int128)1) << sh) - 1), half = ((unsigned __int128)1) << (sh - 1);
        qq += (rem > half || (rem == half && (qq & 1ull))) ? 1ull : 0ull;         // <= 2^53: exact in fp64
        r = ldexp((double)qq, sh);
    }
    r = ldexp(r, -sexp - KM_FX_LO_BITS);
    return neg ? -r : r;
}

// The piece's share of the finish that block `strip` takes: entries [strip per, (strip + 1) per) of the piece's 256 x 32, summed
// over the strips as integers (any order), rounded once, added into the caller's sums / counts.  Plain loads: the partials
// were written by the previous launch.
__device__ __forceinline__ void km_px_finish_share(int p, int strip, int strips, int D, int M, int dsub, int P, int sexp,
                                                   size_t sstride, size_t cstride, const long long* __restrict__ phi,
                                                   const long long* __restrict__ plo, const unsigned* __restrict__ pcnt,
                                                   const unsigned char* __restrict__ pnf, const unsigned* __restrict__ sfl,
                                                   double* __restrict__ sums, unsigned long long* __restrict__ counts) {
    constexpr int SB = 16;
    const int tid = threadIdx.x, nthr = blockDim.x;
    unsigned sf_any = ((tid & 63) < strips) ? sfl[(size_t)(tid & 63) * P + p] : 0u;           // strips <= 64 (host)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sf_any |= (unsigned)__shfl_xor((int)sf_any, o);
    const int per = (RC_K * 32 + strips - 1) / strips;
    const int e1 = min(RC_K * 32, (strip + 1) * per);
    for (int i = strip * per + tid; i < e1; i += nthr) {
        const int f = i & 31, k = i >> 5, fl = 32 * p + f;
        if (fl >= D) continue;
        const int mm = fl / dsub, j = fl % dsub;
        const size_t o = ((size_t)mm * RC_K + k) * dsub + j;
        long long H = 0, L = 0;
        unsigned marks = 0u;
        for (int t0 = 0; t0 < strips; t0 += SB) {
            long long hv[SB];                                  // a strip index past the end is clamped and its value masked: the
#pragma unroll                                                 // loads of a thread go out side by side
            for (int w = 0; w < SB; ++w) hv[w] = phi[(size_t)(t0 + w < strips ? t0 + w : strips - 1) * sstride + o];
#pragma unroll
            for (int w = 0; w < SB; ++w) H += (t0 + w < strips) ? hv[w] : 0ll;
        }
        if (sf_any & 6u) {                                     // rare: a block of the piece used low parts or met inf / NaN
            for (int t = 0; t < strips; ++t) {
                const unsigned sf = sfl[(size_t)t * P + p];
                if (sf & 2u) L += plo[(size_t)t * sstride + o];
                if (sf & 4u) marks |= pnf[(size_t)t * sstride + o];
            }
        }
        double r;
        if ((marks & 4u) || (marks & 3u) == 3u) r = __builtin_nan("");
        else if (marks & 1u) r = INFINITY;
        else if (marks & 2u) r = -INFINITY;
        else r = km_fx_to_double(H, L, sexp);
        sums[o] += r;
        if (j == 0) {
            unsigned long long c = 0;
            for (int t0 = 0; t0 < strips; t0 += SB) {
                unsigned cv[SB];
#pragma unroll
                for (int w = 0; w < SB; ++w) cv[w] = pcnt[(size_t)(t0 + w < strips ? t0 + w : strips - 1) * cstride + (size_t)mm * RC_K + k];
#pragma unroll
                for (int w = 0; w < SB; ++w) c += (t0 + w < strips) ? cv[w] : 0u;
            }
            counts[(size_t)mm * RC_K + k] += c;
        }
    }
}

// 1-D grid of strips x P blocks of 1024 threads (128 rows x 8 lanes per trip), three launches per call:
//   PASS 0 (A)  every block adds its rows (scale from the hint) and stores its partials; the exponent maximum and the "a low part
//               was rounded" flag of every piece go to the control block;
//   PASS 1 (B)  every block reads its piece's verdict.  Hint held, nothing rounded (the normal case): it takes its share of the
//               piece's finish.  Otherwise the piece is repeated: the block adds its rows again with the tight bound of the
//               piece's own data and stores its partials again;
//   PASS 2 (C)  blocks of repeated pieces take their share of the finish, all others leave at once; the block that arrives last
//               moves the hint to this call's bound + one bit and closes the call (seq).
// A piece is judged on its own data: another piece exceeding the hint does not touch these sums.
// dynamic LDS: hi[4][256][8] | lo[4][256][8] (int64) | cnt[KM_PX_MAXM][256] (u32) | nfl[4 x 256 x 8 bytes] | a few words.  The row
// loop is software-pipelined: the next KM_FX_U rows per thread are requested before the current ones are split and added.
// scratch: phi / plo [strips][sstride] int64, pcnt [strips][cstride] u32, pnf [strips][sstride] bytes, sfl [strips][P] u32.
template <int PASS>
__global__ __launch_bounds__(1024) void kmeans_stats_px_kernel(const float* __restrict__ x, int64_t ldx,
                                                               const uint8_t* __restrict__ codes, int64_t n, int D, int M,
                                                               int dsub, int P, int strips, int64_t rows_per_strip,
                                                               km_ctl* __restrict__ ctl, int log2n, size_t sstride, size_t cstride,
                                                               long long* __restrict__ phi,
                                                               long long* __restrict__ plo, unsigned* __restrict__ pcnt,
                                                               unsigned char* __restrict__ pnf, unsigned* __restrict__ sfl,
                                                               double* __restrict__ sums,
                                                               unsigned long long* __restrict__ counts) {
    constexpr int TPR = KM_PX_TPR, ESTRIDE = RC_K * TPR, NACC = 4 * ESTRIDE;
    extern __shared__ __attribute__((aligned(16))) unsigned char km_smem[];
    unsigned long long* hi = reinterpret_cast<unsigned long long*>(km_smem);
    unsigned long long* lo = hi + NACC;
    unsigned* cnt = reinterpret_cast<unsigned*>(lo + NACC);
    unsigned* nfl = cnt + KM_PX_MAXM * RC_K;                  // NACC bytes
    unsigned* s_w = nfl + NACC / 4;                           // [0] what-bits of the block
    const int tid = threadIdx.x, nthr = blockDim.x;
    // XCD x (blocks L = x mod 8) walks a contiguous range of (strip, piece) pairs, piece fastest: the blocks that share a
    // strip's codes (and the two 64-byte halves of every line) meet in one L2
    const unsigned Lb = blockIdx.x, T = gridDim.x, per = T / 8u;
    const unsigned v = (Lb < per * 8u) ? (Lb % 8u) * per + Lb / 8u : Lb;
    const int p = (int)(v % (unsigned)P), strip = (int)(v / (unsigned)P);
    km_piece_ctl* pc = &ctl->piece[p];
    const unsigned long long seq = ctl->seq;               // stable during A, B and all of C but its last instruction
    const int hint = ctl->hint;
    int sexp = 61 - log2n - hint;
    if (PASS >= 1) {
        // the piece's verdict, the same in every block of the piece: am / fl are complete (launch A is over)
        const unsigned long long am = pc->am, fl = pc->fl;
        const int tight = ((am >> 8) == seq) ? km_tight_eb((unsigned)(am & 0xFFu) << 23) : -126;
        const bool inexact = ((fl >> 1) == seq) && (fl & 1ull);
        const bool accept = tight <= hint && !inexact;
        if (PASS == 1) {
            if (accept) {
                km_px_finish_share(p, strip, strips, D, M, dsub, P, sexp, sstride, cstride, phi, plo, pcnt, pnf, sfl, sums, counts);
                return;
            }
            if (strip == 0 && tid == 0) pc->redo = (seq << 16) | (unsigned long long)(tight + 32768);
            sexp = 61 - log2n - tight;                     // repeat the piece with the bound of its own data
        } else {
            const unsigned long long rd = pc->redo;
            if ((rd >> 16) == seq)                          // launch B repeated this piece: its partials are the new ones
                km_px_finish_share(p, strip, strips, D, M, dsub, P, 61 - log2n - ((int)(rd & 0xFFFFull) - 32768), sstride, cstride, phi,
                                   plo, pcnt, pnf, sfl, sums, counts);
            // every wave of this block has read ctl->seq / hint / its piece's redo word (and done its finish share) before
            // the block counts as arrived: the last arriver advances seq, and a wave that loaded it afterwards would take the
            // piece for one of the NEXT call and skip its share
            __syncthreads();
            if (tid == 0 && __hip_atomic_fetch_add(&ctl->carrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
                // the last block of the call's last launch (every other one has read hint and seq): the bound of this call's
                // data (all pieces) + one bit becomes the next call's hint; then the call is closed
                __hip_atomic_store(&ctl->carrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int eb = -126;
                for (int q = 0; q < P; ++q) {
                    const unsigned long long a2 = ctl->piece[q].am;
                    if ((a2 >> 8) == seq) eb = max(eb, km_tight_eb((unsigned)(a2 & 0xFFu) << 23));
                }
                ctl->hint = eb + 1 > 127 ? 127 : eb + 1;
                ctl->seq = seq + 1ull;
            }
            return;
        }
    }
    for (int i = tid; i < 2 * NACC; i += nthr) hi[i] = 0ull;
    for (int i = tid; i < KM_PX_MAXM * RC_K + NACC / 4; i += nthr) cnt[i] = 0u;
    if (tid < 4) s_w[tid] = 0u;
    __syncthreads();
    const int64_t r0 = (int64_t)strip * rows_per_strip;
    const int64_t r1 = (r0 + rows_per_strip < n) ? r0 + rows_per_strip : n;
    const int rpi = nthr / TPR;                            // rows per block iteration (one float4 per thread)
    const int q = tid % TPR;
    const int fl0 = 32 * p + 4 * q;                        // this lane's first float of the row
    const int mfirst = (32 * p) / dsub;
    const bool lane_on = fl0 < D;
    const int m = lane_on ? fl0 / dsub : 0;
    const bool count = lane_on && (fl0 % dsub) == 0;
    unsigned emax = 0u, what = 0u;
    unsigned* mycnt = cnt + (m - mfirst) * RC_K;
    if (lane_on && tid < rpi * TPR) {
        int64_t r = r0 + tid / TPR;
        const float* xp = x + r * ldx + fl0;
        const uint8_t* cp = codes + r * M + m;
        const int64_t xstep = (int64_t)rpi * ldx, cstep = (int64_t)rpi * M;
        int kc[KM_FX_U];
        float4 vc[KM_FX_U];
#pragma unroll
        for (int u = 0; u < KM_FX_U; ++u) {
            const bool ok = r + (int64_t)u * rpi < r1;
            kc[u] = ok ? (int)cp[u * cstep] : -1;
            vc[u] = ok ? *reinterpret_cast<const float4*>(xp + u * xstep) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        while (r < r1) {
            // the wave priority rotates with the 100 MHz clock (the arbiter serves the oldest wave of a SIMD first, so the
            // sixteen waves of a block would finish their rows at different times; sinkhorn.hip, sk_setprio): -3 % at M = 48
            switch (((tid >> 8) + (int)(wall_clock64() >> 10)) & 3) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
            int kn[KM_FX_U];
            float4 vn[KM_FX_U];
            r += (int64_t)KM_FX_U * rpi;
            xp += KM_FX_U * xstep;
            cp += KM_FX_U * cstep;
#pragma unroll
            for (int u = 0; u < KM_FX_U; ++u) {
                const bool ok = r + (int64_t)u * rpi < r1;
                kn[u] = ok ? (int)cp[u * cstep] : -1;
                vn[u] = ok ? *reinterpret_cast<const float4*>(xp + u * xstep) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < KM_FX_U; ++u) {
                if (kc[u] >= 0) {
                    what |= km_px_row(vc[u], kc[u], q, sexp, hi, lo, nfl, emax);
                    if (count) atomicAdd(&mycnt[kc[u]], 1u);
                }
            }
#pragma unroll
            for (int u = 0; u < KM_FX_U; ++u) { kc[u] = kn[u]; vc[u] = vn[u]; }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned t = (unsigned)__shfl_xor((int)emax, o);
            emax = t > emax ? t : emax;
            what |= (unsigned)__shfl_xor((int)what, o);
        }
        if ((tid & 63) == 0) {
            if (emax) atomicMax(&s_w[1], emax);
            if (what) atomicOr(&s_w[0], what);
        }
    }
    __syncthreads();
    if (PASS == 0 && tid == 0) {
        // one fire-and-forget atomic per block (sixteen returning ones per block queued at 24 words: ~2 us of launch A's tail)
        (void)atomicMax(&pc->am, (seq << 8) | (unsigned long long)s_w[1]);                  // <= 254: finite values only
        if (s_w[0] & 1u) (void)atomicMax(&pc->fl, (seq << 1) | 1ull);
    }
    const unsigned bw = s_w[0];                            // bit 1: low parts used, bit 2: non-finite marks set (by this block)
    // ---- partials of this (strip, piece), [strip][m][k][j].  Strips are `sstride` entries apart, not a power-of-two-ish 1.5 MiB.
    for (int i = tid; i < RC_K * 32; i += nthr) {
        const int f = i & 31, k = i >> 5, fl = 32 * p + f;
        if (fl >= D) continue;
        const int a = (f & 3) * ESTRIDE + k * TPR + (f >> 2);
        const size_t o = (size_t)strip * sstride + ((size_t)(fl / dsub) * RC_K + k) * dsub + (fl % dsub);
        phi[o] = (long long)hi[a];
        if (bw & 2u) plo[o] = (long long)lo[a];
        if (bw & 4u) pnf[o] = reinterpret_cast<const unsigned char*>(nfl)[a];
    }
    {
        const int nm = (min(32 * p + 31, D - 1)) / dsub - mfirst + 1;          // sub-quantisers this piece touches
        for (int i = tid; i < nm * RC_K; i += nthr) {
            const int mm = mfirst + i / RC_K;
            if ((mm * dsub) >= 32 * p && (mm * dsub) < 32 * p + 32)            // ... and counts: the one whose first float is here
                pcnt[(size_t)strip * cstride + (size_t)mm * RC_K + (i % RC_K)] = cnt[i];
        }
    }
    if (tid == 0) sfl[(size_t)strip * P + p] = bw;
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
    // exact fixed-point path (see above): float4 rows, n < 2^24 per call
    const int P = (D + 31) / 32;
    const bool fx = vec && n < (1ll << 24) && P <= KM_PX_MAXP && !rc_env_set("RC_KMEANS_STRIPS");
    if (fx) {
        int log2n = 0;
        while ((1ll << log2n) < n) ++log2n;
        if (!h->km_ctl) {                                           // control block: zero = idle, first hint
            void* c = nullptr;
            RC_HIP_CHECK(h, hipMalloc(&c, sizeof(km_ctl)));
            km_ctl init;
            memset(&init, 0, sizeof init);
            init.hint = KM_FX_EB0;
            init.seq = 1ull;                                        // (0 is what the zeroed per-piece words carry)
            RC_HIP_CHECK(h, hipMemcpy(c, &init, sizeof init, hipMemcpyHostToDevice));
            h->km_ctl = c;
        }
        km_ctl* ctl = (km_ctl*)h->km_ctl;
        constexpr int NACC = 4 * RC_K * KM_PX_TPR;
        const size_t lds = (size_t)2 * NACC * sizeof(unsigned long long) + (size_t)KM_PX_MAXM * RC_K * sizeof(unsigned) + NACC + 64;
        const int slots = h->num_cus;                               // one 1024-thread block (134 KiB of LDS) per CU
        int fstrips = slots / P;                                    // one round of equal blocks
        if (fstrips < 1) fstrips = 1;
        if (fstrips > 64) fstrips = 64;                             // (the reducer fetches a piece's strip flags with one wave)
        if (fstrips > (int)((n + 255) / 256)) fstrips = (int)((n + 255) / 256);
        const int64_t frps = (n + fstrips - 1) / fstrips;
        fstrips = (int)((n + frps - 1) / frps);
        // strip strides: the dense size plus an odd number of 256-byte lines, so that an entry's partials of consecutive strips
        // fall into different HBM channels (see the kernel)
        const size_t sstride = (size_t)per_strip + 37 * 32;          // int64 entries (also the byte stride of the marks)
        const size_t cstride = (size_t)M * RC_K + 37 * 64;           // u32 entries
        const size_t lbytes = rc_align_up((size_t)fstrips * sstride * sizeof(long long), 256);
        const size_t cb = rc_align_up((size_t)fstrips * cstride * sizeof(unsigned), 256);
        const size_t nb = rc_align_up((size_t)fstrips * sstride, 256);
        const size_t fb = rc_align_up((size_t)fstrips * P * sizeof(unsigned), 256);
        char* ws = (char*)rc_scratch(h, 2 * lbytes + cb + nb + fb);
        if (!ws) return RC_EHIP;
        long long* phi = (long long*)ws;
        long long* plo = (long long*)(ws + lbytes);
        unsigned* pcn = (unsigned*)(ws + 2 * lbytes);
        unsigned char* pnf = (unsigned char*)(ws + 2 * lbytes + cb);
        unsigned* sfl = (unsigned*)(ws + 2 * lbytes + cb + nb);
#define KM_PX_GO(PASS)                                                                                                          \
    RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kmeans_stats_px_kernel<PASS>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                        (int)lds));                                                                             \
    hipLaunchKernelGGL(kmeans_stats_px_kernel<PASS>, dim3((unsigned)(fstrips * P)), dim3(1024), (PASS == 2 ? 0 : lds), s, x, ldx, codes, \
                       n, D, M, dsub, P, fstrips, frps, ctl, log2n, sstride, cstride, phi, plo, pcn, pnf, sfl, sums,            \
                       reinterpret_cast<unsigned long long*>(counts));                                                          \
    RC_LAUNCH_CHECK(h)
        KM_PX_GO(0);
        KM_PX_GO(1);
        KM_PX_GO(2);
#undef KM_PX_GO
        return RC_OK;
    }
    // fixed-order strip kernels: rows that are not float4-addressable, sub-vector widths that are not multiples of 4, 2^24 rows
    // or more in one call, RC_KMEANS_STRIPS=1
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
    const unsigned* gate = nullptr;
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

// ------------------------------------------------------------------------------------------ IVF coarse quantiser
// Centroid update of the coarse k-means (repconc_amd/ivf.py::coarse_kmeans; BASELINE configs[3], nlist = 5000): the
// statistics of `nlist` cells x D columns do not fit a block's LDS the way a sub-quantiser's 256 x dsub do, so the rows are
// brought into cell order first and every cell is summed by ONE block, in ascending row order, in fp64 — deterministic
// whatever the launch geometry (round 3: torch index_add_ — fp32 atomics — and bincount, two host synchronisations and a
// host RNG per Lloyd iteration).  Stable counting sort, three kernels, no atomics:
//   ivfc_tile_hist_kernel   per tile of IVFC_TILE consecutive rows: histogram of the cells            -> hist[tile][cell]
//   ivfc_cell_scan_kernel   per cell: running sum over the tiles (hist becomes the tile's base), count; then the exclusive
//                           scan of the counts over the cells (last block to finish)                    -> start[cell]
//   ivfc_scatter_kernel     per tile: row r goes to start[cell] + base[tile][cell] + (number of earlier rows of the tile in
//                           the same cell)                                                              -> perm[n]
// then ivfc_cell_mean_kernel: cent[cell] = sum / count (fp32 of the fp64 mean); an EMPTY cell takes the row
// splitmix64(seed, iteration, cell) mod n — counter-based, the same on every rank, nothing read back.
#define IVFC_TILE 2048
__device__ __forceinline__ unsigned long long ivfc_mix(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(1024) void ivfc_tile_hist_kernel(const int* __restrict__ assign, int64_t n, int nlist,
                                                              unsigned* __restrict__ hist) {
    extern __shared__ unsigned ivfc_h[];
    const int tile = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < nlist; i += 1024) ivfc_h[i] = 0u;
    __syncthreads();
    const int64_t r0 = (int64_t)tile * IVFC_TILE;
    for (int i = tid; i < IVFC_TILE; i += 1024) {
        const int64_t r = r0 + i;
        if (r < n) {
            const int c = assign[r];
            if (c >= 0 && c < nlist) atomicAdd(&ivfc_h[c], 1u);           // integer counts: order-free
        }
    }
    __syncthreads();
    for (int i = tid; i < nlist; i += 1024) hist[(size_t)tile * nlist + i] = ivfc_h[i];
}

__global__ __launch_bounds__(256) void ivfc_cell_scan_kernel(unsigned* __restrict__ hist, int tiles, int nlist,
                                                             unsigned* __restrict__ count, unsigned* __restrict__ start,
                                                             unsigned* __restrict__ done) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < nlist) {
        unsigned run = 0;
        for (int t = 0; t < tiles; ++t) {
            const unsigned v = hist[(size_t)t * nlist + c];
            hist[(size_t)t * nlist + c] = run;
            run += v;
        }
        count[c] = run;
    }
    // the last block to arrive scans the counts (one pass of 256 threads over 256-cell strips)
    __shared__ unsigned s_last, s_w[4], s_carry;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(done, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) { s_carry = 0u; *done = 0u; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c0 = 0; c0 < nlist; c0 += 256) {
        const int i = c0 + threadIdx.x;
        const unsigned v = i < nlist ? __hip_atomic_load(&count[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        unsigned incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = (unsigned)__shfl_up((int)incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        unsigned before = s_carry;
        for (int w = 0; w < wv; ++w) before += s_w[w];
        if (i < nlist) start[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = before + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void ivfc_scatter_kernel(const int* __restrict__ assign, int64_t n, int nlist,
                                                            const unsigned* __restrict__ base, const unsigned* __restrict__ start,
                                                            unsigned* __restrict__ perm) {
    __shared__ int s_c[IVFC_TILE];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int64_t r0 = (int64_t)tile * IVFC_TILE;
    for (int i = tid; i < IVFC_TILE; i += 1024) s_c[i] = (r0 + i < n) ? assign[r0 + i] : -1;
    __syncthreads();
    for (int i = tid; i < IVFC_TILE; i += 1024) {
        const int c = s_c[i];
        if (c < 0 || c >= nlist) continue;
        unsigned rank = 0;
        for (int j = 0; j < i; ++j) rank += (s_c[j] == c) ? 1u : 0u;        // rows of the tile ahead of this one in its cell
        perm[start[c] + base[(size_t)tile * nlist + c] + rank] = (unsigned)(r0 + i);
    }
}

// grid nlist, block (D / 4, IVFC_RL): thread (j4, rl) sums the float4 column group j4 of the cell's rows i = rl, rl + RL, ...
// (ascending) in fp64, IVFC_U of them in flight; the RL partial sums are added in the order ((p0 + p1) + p2) + p3.  The order
// is a function of the cell's rows alone, so the result does not depend on the launch.  (One row at a time per cell made a
// 2000-row cell a chain of 500 dependent HBM round trips: 1.3 ms for a skewed 5000-cell assignment of 2^18 rows.)
#define IVFC_RL 4
#define IVFC_U 8
__global__ __launch_bounds__(1024) void ivfc_cell_mean_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int D,
                                                              const unsigned* __restrict__ perm, const unsigned* __restrict__ start,
                                                              const unsigned* __restrict__ count, float* __restrict__ cent,
                                                              unsigned long long seed, int iter) {
    extern __shared__ double ivfc_part[];                    // [RL - 1][D]
    const int c = blockIdx.x, j4 = threadIdx.x, rl = threadIdx.y;
    const unsigned cnt = count[c], s0 = start[c];
    float4* out = reinterpret_cast<float4*>(cent + (size_t)c * D) + j4;
    if (cnt == 0u) {                                         // empty cell: a counter-based random row
        if (rl == 0) {
            const int64_t r = (int64_t)(ivfc_mix(seed ^ ivfc_mix(((unsigned long long)(unsigned)iter << 32) | (unsigned)c)) % (unsigned long long)n);
            *out = *(reinterpret_cast<const float4*>(x + r * ldx) + j4);
        }
        return;
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    unsigned i = (unsigned)rl;
    for (; i + (IVFC_U - 1) * IVFC_RL < cnt; i += IVFC_U * IVFC_RL) {
        float4 v[IVFC_U];
#pragma unroll
        for (int u = 0; u < IVFC_U; ++u) v[u] = *(reinterpret_cast<const float4*>(x + (int64_t)perm[s0 + i + u * IVFC_RL] * ldx) + j4);
#pragma unroll
        for (int u = 0; u < IVFC_U; ++u) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
    }
    for (; i < cnt; i += IVFC_RL) {
        const float4 v = *(reinterpret_cast<const float4*>(x + (int64_t)perm[s0 + i] * ldx) + j4);
        a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
    }
    if (rl > 0) {
        double* p = ivfc_part + ((size_t)(rl - 1) * D + 4 * j4);
        p[0] = a0; p[1] = a1; p[2] = a2; p[3] = a3;
    }
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int r = 1; r < IVFC_RL; ++r) {
            const double* p = ivfc_part + ((size_t)(r - 1) * D + 4 * j4);
            a0 += p[0]; a1 += p[1]; a2 += p[2]; a3 += p[3];
        }
        const double inv = (double)cnt;
        *out = make_float4((float)(a0 / inv), (float)(a1 / inv), (float)(a2 / inv), (float)(a3 / inv));
    }
}

extern "C" size_t rc_ivf_coarse_update_ws_bytes(int64_t n, int nlist) {
    if (n <= 0 || nlist <= 0) return 0;
    const size_t tiles = (size_t)((n + IVFC_TILE - 1) / IVFC_TILE);
    return rc_align_up(tiles * nlist * sizeof(unsigned), 256) + 2 * rc_align_up((size_t)nlist * sizeof(unsigned), 256) +
           rc_align_up((size_t)n * sizeof(unsigned), 256) + 256;
}

// cent [nlist, D] <- mean of the rows assigned to each cell (assign [n] int32, e.g. from rc_ivf_coarse_assign); rows with an
// assignment outside [0, nlist) are ignored; counts_out (optional) receives the cell sizes.
extern "C" int rc_ivf_coarse_update(rc_handle_t h, const float* x, int64_t ldx, const int* assign, int64_t n, int D, int nlist,
                                    float* cent, unsigned* counts_out, uint64_t seed, int iter, void* ws, size_t ws_bytes,
                                    rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !x || !assign || !cent || n <= 0 || D <= 0 || nlist <= 0 || ldx < D) return RC_EINVAL;
    if (D % 4 != 0 || D > 1024 || nlist > 16384 || n > 0xFFFFFFFFll || (ldx % 4) != 0 || ((uintptr_t)x & 15) || ((uintptr_t)cent & 15))
        return RC_ESHAPE;
    if (!ws || ws_bytes < rc_ivf_coarse_update_ws_bytes(n, nlist)) return RC_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = (int)((n + IVFC_TILE - 1) / IVFC_TILE);
    char* w = (char*)ws;
    unsigned* hist = (unsigned*)w;            w += rc_align_up((size_t)tiles * nlist * sizeof(unsigned), 256);
    unsigned* count = (unsigned*)w;           w += rc_align_up((size_t)nlist * sizeof(unsigned), 256);
    unsigned* start = (unsigned*)w;           w += rc_align_up((size_t)nlist * sizeof(unsigned), 256);
    unsigned* perm = (unsigned*)w;            w += rc_align_up((size_t)n * sizeof(unsigned), 256);
    unsigned* done = (unsigned*)w;
    RC_HIP_CHECK(h, hipMemsetAsync(done, 0, sizeof(unsigned), s));
    const size_t lds = (size_t)nlist * sizeof(unsigned);
    RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)ivfc_tile_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ivfc_tile_hist_kernel, dim3((unsigned)tiles), dim3(1024), lds, s, assign, n, nlist, hist);
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(ivfc_cell_scan_kernel, dim3((unsigned)((nlist + 255) / 256)), dim3(256), 0, s, hist, tiles, nlist, count, start, done);
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(ivfc_scatter_kernel, dim3((unsigned)tiles), dim3(1024), 0, s, assign, n, nlist, (const unsigned*)hist,
                       (const unsigned*)start, perm);
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(ivfc_cell_mean_kernel, dim3((unsigned)nlist), dim3((unsigned)(D / 4), IVFC_RL), (size_t)(IVFC_RL - 1) * D * sizeof(double),
                       s, x, ldx, n, D, (const unsigned*)perm, (const unsigned*)start, (const unsigned*)count, cent, (unsigned long long)seed,
                       iter);
    RC_LAUNCH_CHECK(h);
    if (counts_out) RC_HIP_CHECK(h, hipMemcpyAsync(counts_out, count, (size_t)nlist * sizeof(unsigned), hipMemcpyDeviceToDevice, s));
    return RC_OK;
}

// List-centric IVF search (BASELINE configs[3]: nlist = 5000, M = 96): the pipelined 8-query screen over the cells' rows,
// its image and tables, sample / rank-select thresholds, the device-side plan, and the entry points rc_ivf_search_lists /
// rc_ivf_search_probes.  Split from adc_search.hip in round 4; shares adc_common.h with the flat search.
#include "adc_common.h"
#include <stdio.h>
#include <string.h>
#include <vector>

#include <type_traits>

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
//   4. ivfs_qprep_kernel           per query: table statistics + 8-bit tables in slot layout (the integer threshold: step 3)
//   5. ivfs_screen_kernel          persistent blocks walk the tasks: tables transposed into LDS by loader waves, 8-bit screen
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
#ifndef IVFS_PREFETCH_CODES
#define IVFS_PREFETCH_CODES 0   // 1: the 8-query screen requests the next stage's codes before its gathers, as the 16-query screen does.
                                // Twenty registers per set instead of ten: 128 VGPRs + spills, 4.74 -> 5.34 ms per 6 980-query search at
                                // nprobe 128 (profiles/r06i_ivf8_prefetch_ab.txt): off.
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

// bytes of the IVF image of N rows (whole chunks of 16 rows)
extern "C" size_t rc_adc_scan_image_rows_bytes(int64_t N, int M) {
    if (!adc_cf_supported(M) || N < 0) return 0;
    return (size_t)((N + 15) / 16 * 16) * M;
}
// host-side description of that image (no GPU involved): byte offset of codes[n][m], or -1
extern "C" int64_t rc_adc_scan_image_rows_at(int M, int64_t n, int m) {
    if (!adc_cf_supported(M) || n < 0 || m < 0 || m >= M) return -1;
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

// per-query byte tables, [phase][code][PMp] one biased byte per sub-quantiser (phase p starts at byte 256 * 32 p):
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
                        if constexpr (LW > 0 && IVFS_PREFETCH_CODES) {
                            // round 6 (from the 16-query screen): the codes of the NEXT stage are requested BEFORE this stage's
                            // gathers into a second register set and copied over afterwards — a stage used to begin by waiting
                            // 0.7-1 us for codes requested after the previous stage's last gather
                            unsigned wn[R][2];
                            if (has_next) load_codes(PMn{}, PN, nd.t0, nd.nrows, nrd, wn);
                            gathers(PMc{}, P == 0, w, bufoff, reff);
                            IVFS_TSTAMP(1);
                            if constexpr (LASTP) epilogue(cur.t0, cur.row_lo, cur.nrows, rd, tq, myq, reff);
                            if (has_next) {
#pragma unroll
                                for (int c = 0; c < R; ++c) { w[c][0] = wn[c][0]; w[c][1] = wn[c][1]; }
                            }
                        } else {
                        gathers(PMc{}, P == 0, w, bufoff, reff);
                        IVFS_TSTAMP(1);
                        // the codes of the next stage go into the registers the gathers just released (last phase: after the
                        // survivor pass, whose few waits would otherwise also wait for them)
                        if constexpr (!LASTP) { if (has_next) load_codes(PMn{}, PN, nd.t0, nd.nrows, nrd, w); }
                        if constexpr (LASTP) {
                            epilogue(cur.t0, cur.row_lo, cur.nrows, rd, tq, myq, reff);
                            if (has_next) load_codes(PMn{}, PN, nd.t0, nd.nrows, nrd, w);
                        }
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

#include "ivfs_screen16.h"

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
                int* qstatus = nullptr, int width = 8) {
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
    // tables first (they need no threshold), then tau and the integer threshold in one kernel
    hipLaunchKernelGGL(ivfs_qprep_kernel, dim3((unsigned)nq), dim3(RC_K, (M + 31) / 32), 0, s, lut, M, qstat, qbyte);
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(ivf_rank_select_kernel, dim3((unsigned)nq), dim3(1024), 0, s, (const float*)sample, scount, rank, sstride, thr,
                       (const float*)qstat, M, tint);
    RC_LAUNCH_CHECK(h);
    RC_HIP_CHECK(h, hipMemsetAsync(w + L.idcnt, 0, L.counters_end - L.idcnt, s));      // idcnt, cnt, stream_cnt
    {
        // four loader waves (DESIGN_HISTORY 7: without them 1.04 vs 0.95 ms at nprobe 128); width 16: the 16-query screen
        // (ivfs_screen16.h; `image` is then the rows16 image and the tasks hold up to 16 queries)
        auto kern = width == 16 ? ivfs_screen16_kernel<M, 4> : ivfs_screen_kernel<M, 4>;
        constexpr int sl = 2 * IVFS_BUF;
        RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, sl));
        adc_ivf_tasks TT = T;
        TT.qbyte = qbyte;
        int blocks = h->num_cus > 0 ? h->num_cus : 256;       // persistent: one block per CU
        if (blocks > IVFS_MAX_BLOCKS) blocks = IVFS_MAX_BLOCKS;
        if (!T.ntasks && ntasks < blocks) blocks = ntasks;
        unsigned* stream_cnt = (unsigned*)(w + L.stream_cnt);
        unsigned* stream = (unsigned*)(w + L.stream);
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
    if (K != RC_K || !adc_cf_supported(M) || k > ADC_CAND_CAP / 2 || N > 0xFFFFFFFFll || nq > 32768) return RC_ESHAPE;
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
                                                             int* __restrict__ ntasks, int tw) {
    __shared__ int s_wave[4];
    int cq = 0, ct = 0;
    for (int b0 = 0; b0 < nlist; b0 += 256) {
        const int c = b0 + (int)threadIdx.x;
        const int n = c < nlist ? per_cell[c] : 0, t = (n + tw - 1) / tw;     // tw = queries per task (8 | 16)
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
                                                             int* __restrict__ task_qstart, int* __restrict__ task_qcnt, int tw) {
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
        qs = cell_start[cell] + tw * within;
        qc = per_cell[cell] - tw * within;
        qc = qc > tw ? tw : qc;
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
static int ivf_search_probes_w(rc_handle_t h, const uint8_t* codes, const uint8_t* image, const int64_t* list_off,
                               const int64_t* rowmap, int64_t N, int nlist, int M, int K, const float* lut, int nq,
                               const int* probes, int nprobe, int64_t sstride, int ss, int k, double sel_slack,
                               int keep_all_rows, float* scores, int64_t* out_ids, int* status, int* qstatus, void* ws,
                               size_t ws_bytes, rc_stream_t stream, int width) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !image || !list_off || !rowmap || !lut || !probes || !scores || !out_ids || !status || N <= 0 ||
        nq < 0 || nprobe <= 0 || nlist <= 0 || nprobe > nlist || sstride <= 0 || ss <= 0 || k <= 0)
        return RC_EINVAL;
    if (K != RC_K || !adc_cf_supported(M) || k > ADC_CAND_CAP / 2 || N > 0xFFFFFFFFll || nq > 32768) return RC_ESHAPE;
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
                       I(P.first_task), I(P.ntasks), width);
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(ivf_plan_scatter_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, probes, pairs, nprobe,
                       (const int*)I(P.cell_start), I(P.cursor), I(P.sorted_q));
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(ivf_plan_tasks_kernel, dim3((unsigned)((P.ub + 255) / 256)), dim3(256), 0, s, (const int*)I(P.per_cell),
                       (const int*)I(P.cell_start), (const int*)I(P.first_task), (const int*)I(P.ntasks), nlist, P.ub,
                       I(P.task_list), I(P.task_qstart), I(P.task_qcnt), width);
    RC_LAUNCH_CHECK(h);
    adc_ivf_tasks T = {I(P.task_list), I(P.task_qstart), I(P.task_qcnt), I(P.sorted_q), list_off, nullptr, I(P.ntasks)};
    switch (M) {
#define IVFP_CASE(MM)                                                                                                  \
        case MM: return ivfl_launch<MM>(h, codes, image, list_off, rowmap, N, lut, nq, probes, I(P.sbase), I(P.scount),  \
                                        I(P.rows), I(P.rank), nprobe, sstride, ss, T, (int)P.ub, k, scores, out_ids,  \
                                        status, w, L, s, qstatus, width);
        IVFP_CASE(16) IVFP_CASE(32) IVFP_CASE(48) IVFP_CASE(64) IVFP_CASE(96)
#undef IVFP_CASE
        default: return RC_ESHAPE;
    }
}
extern "C" int rc_ivf_search_probes_q(rc_handle_t h, const uint8_t* codes, const uint8_t* image, const int64_t* list_off,
                                      const int64_t* rowmap, int64_t N, int nlist, int M, int K, const float* lut, int nq,
                                      const int* probes, int nprobe, int64_t sstride, int ss, int k, double sel_slack,
                                      int keep_all_rows, float* scores, int64_t* out_ids, int* status, int* qstatus, void* ws,
                                      size_t ws_bytes, rc_stream_t stream) {
    return ivf_search_probes_w(h, codes, image, list_off, rowmap, N, nlist, M, K, lut, nq, probes, nprobe, sstride, ss, k, sel_slack,
                               keep_all_rows, scores, out_ids, status, qstatus, ws, ws_bytes, stream, 8);
}
// The same search on the 16-QUERY screen (round 6, ivfs_screen16.h): tasks of up to 16 queries per probed cell, one
// ds_read_b128 gather per (16 rows, 4 sub-quantisers, 16 queries).  `image16`: rc_adc_scan_image_rows16 of the cell-major codes
// (rc_adc_scan_image_rows16_bytes(N, M) bytes).  Same workspace, status bits and RESULTS as rc_ivf_search_probes_q; pays when a
// probed cell is shared by more than ~8 queries of the call (nq x nprobe / nlist; IVFPQIndex.search decides).
extern "C" int rc_ivf_search_probes_q16(rc_handle_t h, const uint8_t* codes, const uint8_t* image16, const int64_t* list_off,
                                        const int64_t* rowmap, int64_t N, int nlist, int M, int K, const float* lut, int nq,
                                        const int* probes, int nprobe, int64_t sstride, int ss, int k, double sel_slack,
                                        int keep_all_rows, float* scores, int64_t* out_ids, int* status, int* qstatus, void* ws,
                                        size_t ws_bytes, rc_stream_t stream) {
    return ivf_search_probes_w(h, codes, image16, list_off, rowmap, N, nlist, M, K, lut, nq, probes, nprobe, sstride, ss, k,
                               sel_slack, keep_all_rows, scores, out_ids, status, qstatus, ws, ws_bytes, stream, 16);
}
extern "C" size_t rc_adc_scan_image_rows16_bytes(int64_t N, int M) {
    if (!adc_cf_supported(M) || N < 0) return 0;
    return (size_t)((N + 15) / 16 * 16) * M;
}
// host-side description of that image (no GPU involved): byte offset of codes[n][m], or -1
extern "C" int64_t rc_adc_scan_image_rows16_at(int M, int64_t n, int m) {
    if (!adc_cf_supported(M) || n < 0 || m < 0 || m >= M) return -1;
    const int p16 = m / 16;
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < 4; ++j)
            if (adc_q16_slot((int)(n & 15) + 16 * g, j) == m % 16) return ivfs16_image_at(M, n, p16, g, j);
    return -1;
}
extern "C" int rc_adc_scan_image_rows16(rc_handle_t h, const uint8_t* codes, int64_t n0, int64_t n, int M, uint8_t* image,
                                        rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !image || n0 < 0 || n < 0) return RC_EINVAL;
    if (!adc_cf_supported(M)) return RC_ESHAPE;
    if (n == 0) return RC_OK;
    int64_t blocks = (n * M + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(ivfs16_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, codes, n0, n, M, image);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

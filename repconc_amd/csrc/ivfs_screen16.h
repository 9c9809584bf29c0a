// 16-query form of the pipelined IVF screen (round 6; included by ivf_lists.hip after ivfs_* / adc_ivf_tasks are defined).
//
// The 8-query screen (ivfs_screen_kernel) gathers with ds_read_b64: 8 queries per gather at ~5 cycles per gather in the real
// kernel, and its loader waves need as long for a 64 KiB table phase (2.4-3.1 us) as the twelve gathering waves for its
// gathers (2.1 us + 0.7 us of code loads; profiles/r03g_ivf_timeline.txt).  With whole-query-set calls a probed cell is
// shared by tens of queries (6 980 queries x nprobe 32 / 5000 cells = 45), so the flat search's 16-query gather applies:
// tables [code][16 slots][16 queries] (a code's row = 256 B = all 64 banks once, conflict-free whatever the codes), phases of
// 16 sub-quantisers (64 KiB), ONE ds_read_b128 + ONE i8 MFMA per (16 rows, 4 sub-quantisers, 16 queries).  Per query a task
// moves the same table bytes as before and issues half the gathers.
//   * tasks hold up to 16 queries (ivf_plan_*_kernel with width 16); the queries of a task are read from sorted_q when needed
//     (the loader's 16 table offsets, the gathering lanes' own column) instead of living in 2 x 16 scalar registers;
//   * image: chunk n / 16 holds [phase p16][lane = (n mod 16) + 16 g][step j] = codes[n][16 p16 + slot(lane, j)]
//     (adc_q16_slot): a wave's code load for one chunk and phase is 64 lanes x 4 B = 256 contiguous bytes;
//   * tables: a loader thread takes item (code c, quad u): one dword (4 sub-quantisers) from each of the 16 queries' byte tables
//     (consecutive lanes = consecutive dwords), byte-transposes them with v_perm_b32 into the four 16-byte entries
//     (c, 4 u .. 4 u + 3) and stores them with ds_write_b128.  A wave-instruction of b128 stores is served 8 lanes at a time over a
//     128-byte window, and a code's row is 256 B: lanes that stored "entry i of my item" together would hit two windows' worth of
//     banks four times over.  Lane l therefore stores entry (j + (l >> 1)) mod 4 in step j — eight distinct 16-byte slots per
//     group — and gets that rotation for free: the selector of the first transposition step is a per-lane register.
#define IVFS16_R 10              // chunks of 16 rows per gathering wave and round
typedef unsigned adc_u32x4s __attribute__((ext_vector_type(4)));

__host__ __device__ inline int64_t ivfs16_image_at(int M, int64_t n, int p16, int g, int j) {
    return (n >> 4) * (int64_t)(16 * M) + (int64_t)(256 * p16) + (int64_t)((((int)(n & 15)) + 16 * g) * 4 + j);
}
__global__ __launch_bounds__(256) void ivfs16_image_kernel(const uint8_t* __restrict__ codes, int64_t n0, int64_t cnt, int M,
                                                           uint8_t* __restrict__ image) {
    const int64_t total = cnt * M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = n0 + i / M;
        const int pos = (int)(i % M);
        const int p16 = pos >> 4, g = (pos >> 2) & 3, j = pos & 3;
        const int m = 16 * p16 + adc_q16_slot((int)(n & 15) + 16 * g, j);
        image[ivfs16_image_at(M, n, p16, g, j)] = codes[n * M + m];
    }
}

// Development aid (tools/ivf16_timeline.py builds a variant library with -DRC_IVF_TRACE): wall-clock stamps of every wave at
// three points of each of a block's first 64 stages.  Off in the shipped library.
#ifdef RC_IVF_TRACE
__device__ unsigned long long ivfs16_trace[256 * 64 * 16 * 4];
extern "C" int rc_debug_ivfs16_trace(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ivfs16_trace), sizeof(ivfs16_trace));
}
#define IVFS16_TSTAMP(stage, i)                                                                                          \
    do {                                                                                                                 \
        if (l == 0 && (stage) < 64u && blockIdx.x < 256u)                                                                 \
            ivfs16_trace[((blockIdx.x * 64u + (stage)) * 16u + (unsigned)wv) * 4u + (i)] = wall_clock64();               \
    } while (0)
#else
#define IVFS16_TSTAMP(stage, i) do { } while (0)
#endif

struct ivfs_task16 {
    int valid;
    int qs, qc;               // the task's queries: sorted_q[qs .. qs + qc), 1 <= qc <= 16
    unsigned t0;              // first (16-aligned) row of the range
    unsigned row_lo, nrows;   // rows [row_lo, nrows) counted from t0 are the cell's (nrows = 0: nothing to scan)
};

template <int M, int LW>
__global__ __launch_bounds__(IVFS_THREADS, 4) void ivfs_screen16_kernel(const uint8_t* __restrict__ image,
                                                                        const int* __restrict__ tint,
                                                                        unsigned* __restrict__ stream_cnt,
                                                                        unsigned* __restrict__ stream, unsigned stream_cap,
                                                                        int* __restrict__ status, adc_ivf_tasks T,
                                                                        int ntasks_arg) {
    static_assert(LW > 0 && M % 16 == 0, "loader waves; whole phases of 16 sub-quantisers");
    constexpr int GW = IVFS_WAVES - LW;                       // gathering waves
    constexpr int R = IVFS16_R;
    constexpr int NPH = M / 16, ROUND = GW * R * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    const int l = (int)(tid & 63u), wv = __builtin_amdgcn_readfirstlane((int)(tid >> 6)), r = l & 15, g = l >> 4;
    // ---- this block's tasks (as in the 8-query screen: XCD x owns a contiguous eighth of the cell-ordered list)
    const unsigned total = (unsigned)__builtin_amdgcn_readfirstlane(T.ntasks ? *T.ntasks : ntasks_arg);
    const unsigned xcd = blockIdx.x % 8u, jb = blockIdx.x / 8u, pxb = (gridDim.x - xcd + 7u) / 8u;
    const unsigned tq8 = total / 8u, tr8 = total % 8u;
    const unsigned lo = xcd < tr8 ? xcd * (tq8 + 1u) : tr8 * (tq8 + 1u) + (xcd - tr8) * tq8, cnt = tq8 + (xcd < tr8 ? 1u : 0u);
    auto load_task = [&](unsigned k) {
        ivfs_task16 d;
        const unsigned at = jb + k * pxb;
        d.valid = at < cnt ? 1 : 0;
        d.qs = 0; d.qc = 0; d.t0 = 0; d.row_lo = 0; d.nrows = 0;
        if (d.valid) {
            auto sc = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
            const unsigned task = lo + at;
            d.qs = sc(T.task_qstart[task]);
            d.qc = sc(T.task_qcnt[task]);
            const int cell = sc(T.task_list[task]);
            const unsigned a = (unsigned)sc((int)T.list_off[cell]), b = (unsigned)sc((int)T.list_off[cell + 1]);   // N < 2^32
            if (d.qc > 0 && b > a) {
                const unsigned t0 = a & ~15u;
                d.t0 = t0; d.row_lo = a - t0; d.nrows = b - t0;
            }
        }
        return d;
    };
    auto rounds_of = [&](const ivfs_task16& d) { return d.nrows ? (int)((d.nrows + ROUND - 1) / ROUND) : 1; };
    auto lane_q = [&](const ivfs_task16& d) { return (r < d.qc) ? T.sorted_q[d.qs + r] : -1; };
    auto lane_thr = [&](int q) {
        if (q < 0) return INT_MAX;
        const int t = tint[q];
        return (t == INT_MIN) ? INT_MIN : t - 128 * M;
    };
    // ---- tables: phase p16 of a task, by threads t = 0 .. NTHR - 1 (t = this thread's number), in two halves so that a loader
    // can request a phase one stage before it transposes it (the loads' ~1 us from the L2 / memory-side cache is then nobody's
    // critical path: with request, wait, transposition and stores inside one stage the four loader waves needed ~3 us per 64 KiB
    // phase and set the stage period — the 16-query gathers, half as many per query, gained 2 %)
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)T.qbyte, 0, -1, 0x00020000);
    const unsigned rot = ((unsigned)l >> 1) & 3u;
    // the task's 16 table offsets: lane j < 16 reads query j's id, the wave reads them back lane by lane
    // (an empty slot reads query 0's table: its column is masked by the threshold INT_MAX)
    auto task_offsets = [&](const ivfs_task16& d, unsigned (&so)[16]) {
        int qv = 0;
        if (l < 16 && l < d.qc) qv = T.sorted_q[d.qs + l];
#pragma unroll
        for (int j = 0; j < 16; ++j) so[j] = (unsigned)__builtin_amdgcn_readlane(qv, j) * (unsigned)(M * RC_K);
    };
    auto fill_load = [&](auto NTHRc, int p16, const unsigned (&so)[16], unsigned t, auto& a) {
        constexpr unsigned NTHR = decltype(NTHRc)::value;
        constexpr int ITEMS = 1024 / NTHR;                    // items (code, quad of sub-quantisers) per thread
        static_assert(1024 % NTHR == 0 && NTHR % 64 == 0, "whole items per thread, whole waves");
        const int y = p16 >> 1, hh = p16 & 1;
        const unsigned PM = (unsigned)ivfs_pm(M, y);          // the byte tables are stored in phases of 32 (last one: 16)
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const unsigned item = (unsigned)it * NTHR + t, c = item >> 2, u = item & 3u;
            const unsigned voff = (unsigned)(RC_K * 32 * y) + c * PM + 16u * (unsigned)hh + 4u * u;
#pragma unroll
            for (int j = 0; j < 16; ++j) a[it][j] = __builtin_amdgcn_raw_buffer_load_b32(qrsrc, voff, so[j], 0);
        }
    };
    auto fill_store = [&](auto NTHRc, unsigned bufoff, unsigned t, const auto& a) {
        constexpr unsigned NTHR = decltype(NTHRc)::value;
        constexpr int ITEMS = 1024 / NTHR;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const unsigned item = (unsigned)it * NTHR + t;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned i = ((unsigned)j + rot) & 3u;                      // the entry this lane stores in step j
                const unsigned sel = ((4u + i) << 8) | i;                        // (a_even.b_i, a_odd.b_i, -, -)
                adc_u32x4s e;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const unsigned p01 = __builtin_amdgcn_perm(a[it][4 * gq + 1], a[it][4 * gq], sel);
                    const unsigned p23 = __builtin_amdgcn_perm(a[it][4 * gq + 3], a[it][4 * gq + 2], sel);
                    e[gq] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);      // queries 4 gq .. 4 gq + 3 of entry (c, 4 u + i)
                }
                *reinterpret_cast<adc_u32x4s*>(smem + (bufoff + item * 64u + i * 16u)) = e;
            }
        }
    };
    // The loader waves' form (256 threads): thread = code c, all four quads in ONE 16-byte load per query — 16 load instructions per
    // thread and phase instead of 64.  The wall-clock trace (tools/ivf16_timeline.py, profiles/r06j_ivf16_timeline.txt) showed the
    // loader waves as the stage's critical path: 1.3 us transposing and storing + 1.5 us ISSUING the 64 dword requests of the next
    // phase, against 2.2-2.5 us of gathers + hand-over.  The price: a thread's sixteen entries are one whole 256-byte row, so the
    // eight lanes of a store group can only be spread over FOUR distinct 16-byte slots (rotation l & 3 of the entry inside its quad):
    // two-way bank conflicts on the 16 stores, 256 instead of 128 LDS cycles per wave and phase.
    // (Requesting the phase in two 8-byte halves — the half just stored frees its registers for the same half of the phase after
    // next, so the first request leaves a quarter into the loader's stage — was built and is SLOWER: 32 load instructions per
    // thread cost the loader 1.2-1.4 us of issue against 0.6 for 16, 6.6 ms against 4.1 per 6 980-query search at nprobe 128
    // (profiles/r06j_ivf16_timeline_half_sets.txt).  The loader waves stay the stage's critical path at ~2.9 us in the traced build:
    // ~0.8 us waiting for the phase requested at the end of the previous stage, 1.1 us transposing and storing, 0.6 us requesting.)
    typedef unsigned adc_u32x4l __attribute__((ext_vector_type(4)));
    auto fill_load4 = [&](int p16, const unsigned (&so)[16], unsigned c, adc_u32x4l (&a)[16]) {
        const int y = p16 >> 1, hh = p16 & 1;
        const unsigned PM = (unsigned)ivfs_pm(M, y);
        const unsigned voff = (unsigned)(RC_K * 32 * y) + c * PM + 16u * (unsigned)hh;
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, voff, so[j], 0);
    };
    auto fill_store4 = [&](unsigned bufoff, unsigned c, const adc_u32x4l (&a)[16]) {
        const unsigned rot4 = (unsigned)l & 3u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned i = ((unsigned)j + rot4) & 3u;
                const unsigned sel = ((4u + i) << 8) | i;
                adc_u32x4s e;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const unsigned p01 = __builtin_amdgcn_perm(a[4 * gq + 1][u], a[4 * gq][u], sel);
                    const unsigned p23 = __builtin_amdgcn_perm(a[4 * gq + 3][u], a[4 * gq + 2][u], sel);
                    e[gq] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                }
                *reinterpret_cast<adc_u32x4s*>(smem + (bufoff + c * 256u + (unsigned)(4 * u) * 16u + i * 16u)) = e;
            }
        }
    };
    // ---- codes of one stage: chunk c of wave wv is chunk GW c + wv of the round; one dword per lane and chunk
    auto chunks_of = [&](unsigned nrows, int rd) {            // chunks this wave owns in round rd (wave-uniform, 0 .. R)
        const unsigned done = (unsigned)rd * ROUND;
        if (nrows <= done) return 0;
        unsigned nc = (nrows - done + 15u) / 16u;
        if (nc > (unsigned)(ROUND / 16)) nc = ROUND / 16;
        if (wv >= GW) return 0;                                // a loader wave
        const int mine = ((int)nc - wv + GW - 1) / GW;
        return mine < 0 ? 0 : mine;
    };
    auto load_codes = [&](int p16, unsigned t0, unsigned nrows, int rd, unsigned (&w)[R]) {
        const int reff = chunks_of(nrows, rd);
        if (reff == 0) return;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(image + (size_t)t0 * M), 0, -1, 0x00020000);
        const unsigned first = ((unsigned)rd * (unsigned)(ROUND / 16) + (unsigned)wv) * (unsigned)(16 * M) + (unsigned)(256 * p16);
#pragma unroll
        for (int c = 0; c < R; ++c)
            if (c < reff) w[c] = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)l * 4u, first + (unsigned)(c * GW * 16 * M), 0);
    };
    adc_i32x4v bsel = {0, 0, 0, 0};                          // B[k][n = r] = [k mod 16 == r]
    bsel[r >> 2] = 1 << (8 * (r & 3));
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    if (lds0 & 0xFFFFu) __builtin_trap();                    // the one-instruction gather address needs 64 KiB-aligned table buffers
    unsigned off[4];                                          // this lane's slot addresses in the CURRENT table buffer (bit 16 toggles)
#pragma unroll
    for (int j = 0; j < 4; ++j) off[j] = lds0 + (unsigned)adc_q16_slot(l, j) * 16u;
    adc_i32x4v acc[R];
    // ---- gathers + folds of one stage: the four gathers of chunk c + 1 are issued before the four MFMAs of chunk c
    auto gathers = [&](bool first, const unsigned (&w)[R], int reff) {
        if (reff <= 0) return;                                // wave-uniform
        adc_u32x4s ea[4], eb[4];
        auto gather = [&](int c, adc_u32x4s (&e)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // buffer base (0 / 64 KiB: bytes 2-3) | code << 8 | slot offset (< 256): one v_perm_b32
                const unsigned addr = __builtin_amdgcn_perm(w[c], off[j], 0x03020000u | ((4u + (unsigned)j) << 8));
                e[j] = *reinterpret_cast<const adc_u32x4s __attribute__((address_space(3)))*>(addr);
            }
        };
        auto fold = [&](int c, const adc_u32x4s (&e)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const adc_i32x4v a = {(int)e[j][0], (int)e[j][1], (int)e[j][2], (int)e[j][3]};
                if (j == 0 && first) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, adc_i32x4v{0, 0, 0, 0}, 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, acc[c], 0, 0, 0);
            }
        };
        gather(0, ea);
#pragma unroll
        for (int c = 0; c < R; ++c) {
            if (c < reff) {                                   // wave-uniform
#if IVFS_PRIO
                if (c == 0) __builtin_amdgcn_s_setprio(3);
                else if (c == R / 4) __builtin_amdgcn_s_setprio(2);
                else if (c == R / 2) __builtin_amdgcn_s_setprio(1);
                else if (c == 3 * R / 4) __builtin_amdgcn_s_setprio(0);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < R && c + 1 < reff) gather(c + 1, (c & 1) ? ea : eb);
                __builtin_amdgcn_sched_barrier(0);
                fold(c, (c & 1) ? eb : ea);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#if IVFS_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    // ---- survivors: (query, row) pairs appended to the wave's own stream (see the 8-query screen); 16 query columns here
    const __amdgpu_buffer_rsrc_t strsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(stream + (size_t)(blockIdx.x * IVFS_WAVES + (unsigned)wv) * stream_cap * 2u), 0, -1, 0x00020000);
    unsigned woff = 0;                                        // wave-uniform: pairs in the wave's stream
    typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
    static_assert(R * 4 <= 64, "one mask bit per sum");
    typedef typename std::conditional<(R * 4 <= 32), unsigned, unsigned long long>::type mask_t;
    auto epilogue = [&](unsigned t0, unsigned row_lo, unsigned nrows, int rd, int tq, int myq, int reff) {
        if (reff <= 0) return;
        const unsigned rb = (unsigned)rd * ROUND + (unsigned)(wv * 16);      // first row of the wave's chunk 0
        mask_t m = 0;                                         // bit 4 c + e: D[row 4 g + e of chunk c][column r] survives
#pragma unroll
        for (int c = 0; c < R; ++c) {
            if (c < reff) {
#pragma unroll
                for (int e = 0; e < 4; ++e) m |= (acc[c][e] >= tq) ? ((mask_t)1 << (4 * c + e)) : (mask_t)0;
            }
        }
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const unsigned cb = rb + (unsigned)(16 * GW * c);
            if (c < reff && (cb < row_lo || cb + 16u > nrows)) {           // wave-uniform, rare: rows outside the cell
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned n = cb + 4u * g + e;
                    if (n < row_lo || n >= nrows) m &= ~((mask_t)1 << (4 * c + e));
                }
            }
        }
        const unsigned cnt1 = (unsigned)__popcll((unsigned long long)m);
        if (!__ballot(cnt1 != 0)) return;
        const unsigned c0 = __shfl(cnt1, r), c1 = __shfl(cnt1, r + 16), c2 = __shfl(cnt1, r + 32), c3 = __shfl(cnt1, r + 48);
        const unsigned tot = c0 + c1 + c2 + c3;
        const unsigned lane_first = (g > 0 ? c0 : 0u) + (g > 1 ? c1 : 0u) + (g > 2 ? c2 : 0u);
        unsigned inc = tot;                                   // inclusive prefix over the 16 query columns of the lane's row group
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const unsigned t = __shfl_up(inc, o, 16);
            if (r >= o) inc += t;
        }
        const unsigned wtotal = (unsigned)__builtin_amdgcn_readlane((int)inc, 15);
        if (woff + wtotal > stream_cap) {                     // wave-uniform; status bit 2: a stream filled up
            if (l == 0) atomicOr(status, 4);
            return;
        }
        unsigned at = (woff + (inc - tot) + lane_first) * 8u;
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

    // ---- prologue: every thread takes one item of the first tables
    const ivfs_task16 first = load_task(0);
    if (!first.valid) return;                                 // block-uniform
    {
        unsigned so[16], a0[1][16];
        task_offsets(first, so);
        fill_load(std::integral_constant<unsigned, IVFS_THREADS>{}, 0, so, tid, a0);
        fill_store(std::integral_constant<unsigned, IVFS_THREADS>{}, 0u, tid, a0);
    }
    // ---- loader waves: their own walk over the stages, two stages ahead with the requests, one with the stores.  Exactly one
    // barrier per stage, as the gathering waves.
    if (wv >= GW) {                                           // wave-uniform
        static_assert(LW * 64 == RC_K, "one loader thread per code");
        const unsigned lt = tid - (unsigned)(GW * 64);
        struct pos { ivfs_task16 d; unsigned k; int rd, P; };
        auto next = [&](const pos& p) {
            pos n = p;
            if (!p.d.valid) return n;
            if (++n.P == NPH) {
                n.P = 0;
                if (++n.rd == rounds_of(p.d)) { n.rd = 0; n.k = p.k + 1; n.d = load_task(n.k); }
            }
            return n;
        };
        unsigned so[16];
        adc_u32x4l a[16];
        unsigned so_k = 0xFFFFFFFFu;                          // the task `so` belongs to
        auto request = [&](const pos& p) {
            if (p.k != so_k) { task_offsets(p.d, so); so_k = p.k; }
            fill_load4(p.P, so, lt, a);
        };
        pos p1 = next(pos{first, 0u, 0, 0});                   // the stage AFTER the one that is about to run
        if (p1.d.valid) request(p1);
        unsigned bufoff = (unsigned)IVFS_BUF;                   // where p1's tables go
        unsigned sidx = 0;
        for (;;) {
            block_sync();                                     // a stage begins: everybody is done with the other buffer
            IVFS16_TSTAMP(sidx, 0);
            if (!p1.d.valid) break;
            fill_store4(bufoff, lt, a);
            IVFS16_TSTAMP(sidx, 1);
            p1 = next(p1);
            if (p1.d.valid) request(p1);
            IVFS16_TSTAMP(sidx, 2);
            bufoff ^= (unsigned)IVFS_BUF;
            ++sidx;
        }
        return;                                               // (its stream stays empty: stream_cnt was cleared by the host)
    }
    // ---- the gathering waves' walk over (task, round, phase) stages
    auto walk = [&](auto ROLEc) {
        constexpr bool LOADER = decltype(ROLEc)::value == 1;
        ivfs_task16 cur = first;
        int myq = -1, tq = INT_MAX;
        unsigned w[R];
        if constexpr (!LOADER) {
            myq = lane_q(cur); tq = lane_thr(myq);
            load_codes(0, cur.t0, cur.nrows, 0, w);
        }
        unsigned k = 0;
        unsigned sidx = 0;                                    // stage counter (trace builds only)
        (void)sidx;
        for (;;) {                                            // tasks of this block
            const ivfs_task16 nxt = load_task(k + 1);
            const int nrounds = rounds_of(cur);
            for (int rd = 0; rd < nrounds; ++rd) {
                const bool more = rd + 1 < nrounds;           // block-uniform
                auto stage = [&](auto Pc) {
                    constexpr int P = decltype(Pc)::value;
                    constexpr bool LASTP = (P == NPH - 1);
                    constexpr int PN = LASTP ? 0 : P + 1;     // phase of the next stage
                    block_sync();
                    IVFS16_TSTAMP(sidx, 0);
                    const bool to_next = LASTP && !more;      // block-uniform
                    const bool has_next = !to_next || nxt.valid;
                    const ivfs_task16 nd = to_next ? nxt : cur;
                    const int nrd = to_next ? 0 : (LASTP ? rd + 1 : rd);
                    {
                        // the codes of the NEXT stage are requested before this stage's gathers, into a second set of ten
                        // registers (the 8-query screen needs twenty per set and requests them after its last gather: every
                        // stage then begins by waiting 0.7-1 us for them, profiles/r03g_ivf_timeline.txt)
                        const int reff = chunks_of(cur.nrows, rd);
                        unsigned wn[R];
                        if (has_next) load_codes(PN, nd.t0, nd.nrows, nrd, wn);
                        gathers(P == 0, w, reff);
                        IVFS16_TSTAMP(sidx, 1);
                        if constexpr (LASTP) epilogue(cur.t0, cur.row_lo, cur.nrows, rd, tq, myq, reff);
                        if (has_next) {
#pragma unroll
                            for (int c = 0; c < R; ++c) w[c] = wn[c];
                        }
                        IVFS16_TSTAMP(sidx, 2);
                        ++sidx;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) off[j] ^= (unsigned)IVFS_BUF;     // the other table buffer (64 KiB-aligned bases)
                };
                stage(std::integral_constant<int, 0>{});
                if constexpr (NPH > 1) stage(std::integral_constant<int, 1>{});
                if constexpr (NPH > 2) stage(std::integral_constant<int, 2>{});
                if constexpr (NPH > 3) stage(std::integral_constant<int, 3>{});
                if constexpr (NPH > 4) stage(std::integral_constant<int, 4>{});
                if constexpr (NPH > 5) stage(std::integral_constant<int, 5>{});
                static_assert(NPH <= 6, "M <= 96");
            }
            if (!nxt.valid) break;
            cur = nxt;
            if constexpr (!LOADER) { myq = lane_q(cur); tq = lane_thr(myq); }
            ++k;
        }
    };
    walk(std::integral_constant<int, 0>{});
    if (l == 0) stream_cnt[blockIdx.x * IVFS_WAVES + (unsigned)wv] = woff;
}

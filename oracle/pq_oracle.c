/*
 * pq_oracle.c — plain-C (OpenMP) restatement of the RepCONC PQ hot path.  TEST INFRASTRUCTURE:
 * used by tests/ as a second checker at sizes numpy is too slow for, and by bench.py's
 * `cpu_baseline` leg (kind "port").  Never linked into or called by the product (repconc_amd).
 *
 * Parity: rows a-1…a-6 PINNED — tests/test_oracle_golden.py checks this library against the
 * golden vectors generated from the reference (oracle/gen_golden.py).  ADC / k-means rows:
 * PARITY UNPINNED (Faiss not available; see oracle/pq_oracle.py header).
 *
 * Build: oracle/Makefile  (gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math: every fp32
 * sub/mul/add of the distance table is rounded separately, as torch-CPU does).
 * All citations: /root/reference/src/repconc/models/repconc/modeling_repconc.py unless noted.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define K 256

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* torch-CPU `sum(-1)` order over a contiguous fp32 row of length dsub (aten/native/cpu/SumKernel.cpp; SURVEY.md §8 a-1;
   the numpy twin in pq_oracle.py spells the rule out).  dsub >= 8: 8-wide vectors, four ILP accumulators fed round by round
   with torch's cascade (after every 16 rounds the running sums move one level up; levels merged lowest first), left-over
   vectors to accumulator 0, 0 += 1, 2, 3, then (tail scalars from 0) + lane 0 .. lane 7.  dsub < 8: the scalar twin. */
static float rowsum_torch_order(const float* sq, int dsub) {
    if (dsub < 8) {
        float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const int full = dsub / 4;
        if (full)
            for (int j = 0; j < 4; ++j) p[j] = p[j] + sq[j];
        for (int j = 4 * full; j < dsub; ++j) p[0] = p[0] + sq[j];
        float r = p[0] + p[1];
        r = r + p[2];
        return r + p[3];
    }
    const int nv = dsub / 8, tail = dsub % 8, full = nv / 4;
    float acc[4][4][8];                                  /* [level][ilp][lane] */
    memset(acc, 0, sizeof(acc));
    int i = 0;
    while (i + 16 <= full) {
        for (int t = 0; t < 16; ++t, ++i)
            for (int j = 0; j < 4; ++j)
                for (int l = 0; l < 8; ++l) acc[0][j][l] = acc[0][j][l] + sq[8 * (4 * i + j) + l];
        for (int lv = 1; lv < 4; ++lv) {
            for (int j = 0; j < 4; ++j)
                for (int l = 0; l < 8; ++l) { acc[lv][j][l] = acc[lv][j][l] + acc[lv - 1][j][l]; acc[lv - 1][j][l] = 0.0f; }
            if (i & (15 << (4 * lv))) break;
        }
    }
    for (; i < full; ++i)
        for (int j = 0; j < 4; ++j)
            for (int l = 0; l < 8; ++l) acc[0][j][l] = acc[0][j][l] + sq[8 * (4 * i + j) + l];
    for (int lv = 1; lv < 4; ++lv)
        for (int j = 0; j < 4; ++j)
            for (int l = 0; l < 8; ++l) acc[0][j][l] = acc[0][j][l] + acc[lv][j][l];
    for (int v = 4 * full; v < nv; ++v)
        for (int l = 0; l < 8; ++l) acc[0][0][l] = acc[0][0][l] + sq[8 * v + l];
    float a[8];
    for (int l = 0; l < 8; ++l) {
        float t = acc[0][0][l] + acc[0][1][l];
        t = t + acc[0][2][l];
        a[l] = t + acc[0][3][l];
    }
    float r = 0.0f;
    for (int j = 0; j < tail; ++j) r = r + sq[nv * 8 + j];
    for (int l = 0; l < 8; ++l) r = r + a[l];
    return r;
}

/* d[m][b][k] = sum_j (x[b][m*dsub+j] - C[m][k][j])^2.  :49-50 */
void orc_dist_table(const float* x, int64_t ldx, const float* C, int64_t B, int M, int dsub, float* d) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
        for (int64_t b = 0; b < B; ++b) {
            const float* xr = x + b * ldx + (int64_t)m * dsub;
            float sq[1024];
            for (int k = 0; k < K; ++k) {
                const float* c = C + ((int64_t)m * K + k) * dsub;
                for (int j = 0; j < dsub; ++j) {
                    const float t = xr[j] - c[j];
                    sq[j] = t * t;
                }
                d[((int64_t)m * B + b) * K + k] = rowsum_torch_order(sq, dsub);
            }
        }
}

/* per-m max (minmax[0..M)) and min (minmax[M..2M)).  :76-77 */
void orc_minmax(const float* d, int64_t B, int M, float* minmax) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        float mx = -INFINITY, mn = INFINITY;
        const float* p = d + (int64_t)m * B * K;
        for (int64_t i = 0; i < B * K; ++i) {
            if (p[i] > mx) mx = p[i];
            if (p[i] < mn) mn = p[i];
        }
        minmax[m] = mx;
        minmax[M + m] = mn;
    }
}

/* in place (d-mid)/amp.  :81-84 */
void orc_centre(float* d, const float* minmax, int64_t B, int M) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float mx = minmax[m], mn = minmax[M + m];
        const float mid = (mx + mn) / 2.0f;
        const float amp = (mx - mid) + 1e-5f;
        float* p = d + (int64_t)m * B * K;
        for (int64_t i = 0; i < B * K; ++i) p[i] = (p[i] - mid) / amp;
    }
}

/* codes[b][m] = argmin_k d[m][b][k], first minimum.  :52,:66 */
void orc_argmin(const float* d, int64_t B, int M, uint8_t* codes) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
        for (int64_t b = 0; b < B; ++b) {
            const float* p = d + ((int64_t)m * B + b) * K;
            int bi = 0;
            float best = p[0];
            for (int k = 1; k < K; ++k)
                if (p[k] < best) { best = p[k]; bi = k; }
            codes[b * M + m] = (uint8_t)bi;
        }
}

/* sinkhorn_algorithm in the reference's in-place Q form, single rank, :137-165, followed by the
 * argmax of :63.  dc: centred distances [M][B][K] fp32.  Q: [M][K][B] fp64 (the reference's
 * layout), every loop nest parallel over (m, row) or (m, column block) so all host cores work.
 * Returns flags: bit0 NaN seen, bit1 Inf seen (:64-65). */
#define CB 256 /* column block */
int orc_sinkhorn_codes(const float* dc, int64_t B, int M, double eps, int iters, uint8_t* codes) {
    int flags = 0;
    double* Q = (double*)malloc(sizeof(double) * (size_t)M * K * B);
    double* rowsum = (double*)malloc(sizeof(double) * (size_t)M * K);
    const int64_t nblk = (B + CB - 1) / CB;
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            double* q = Q + ((int64_t)m * K + k) * B;
            const float* dm = dc + (int64_t)m * B * K + k;
            double s = 0.0;
            for (int64_t b = 0; b < B; ++b) {
                q[b] = exp((-(double)dm[b * K]) / eps); /* :141 on out=-centred (:57) */
                s += q[b];
            }
            rowsum[m * K + k] = s;
        }
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) { /* :148,:152 */
        double tot = 0.0;
        for (int k = 0; k < K; ++k) tot += rowsum[m * K + k];
        double* q = Q + (int64_t)m * K * B;
        for (int64_t i = 0; i < (int64_t)K * B; ++i) q[i] /= tot;
    }
    for (int it = 0; it < iters; ++it) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int m = 0; m < M; ++m) /* :155-159 */
            for (int k = 0; k < K; ++k) {
                double* q = Q + ((int64_t)m * K + k) * B;
                double s = 0.0;
                for (int64_t b = 0; b < B; ++b) s += q[b];
                for (int64_t b = 0; b < B; ++b) { q[b] /= s; q[b] /= K; }
            }
#pragma omp parallel for collapse(2) schedule(static)
        for (int m = 0; m < M; ++m) /* :162-163 */
            for (int64_t blk = 0; blk < nblk; ++blk) {
                const int64_t b0 = blk * CB, b1 = (b0 + CB < B) ? b0 + CB : B;
                double col[CB];
                for (int64_t b = b0; b < b1; ++b) col[b - b0] = 0.0;
                for (int k = 0; k < K; ++k) {
                    const double* q = Q + ((int64_t)m * K + k) * B;
                    for (int64_t b = b0; b < b1; ++b) col[b - b0] += q[b];
                }
                for (int k = 0; k < K; ++k) {
                    double* q = Q + ((int64_t)m * K + k) * B;
                    for (int64_t b = b0; b < b1; ++b) { q[b] /= col[b - b0]; q[b] /= (double)B; }
                }
            }
    }
#pragma omp parallel for collapse(2) schedule(static) reduction(| : flags)
    for (int m = 0; m < M; ++m)
        for (int64_t b = 0; b < B; ++b) { /* :164 then :63 (argmax over k, first maximum) */
            const double* q = Q + (int64_t)m * K * B + b;
            int bi = 0;
            double best = q[0] * (double)B;
            if (isnan(best)) flags |= 1;
            if (isinf(best)) flags |= 2;
            for (int k = 1; k < K; ++k) {
                const double v = q[(int64_t)k * B] * (double)B;
                if (isnan(v)) flags |= 1;
                if (isinf(v)) flags |= 2;
                if (v > best) { best = v; bi = k; }
            }
            codes[b * M + m] = (uint8_t)bi;
        }
    free(Q);
    free(rowsum);
    return flags;
}

/* RepCONC.quantize on one rank (:47-67).  ws must hold M*B*K floats. */
int orc_quantize(const float* x, int64_t ldx, const float* C, int64_t B, int M, int dsub, int use_constraint,
                 double eps, int iters, uint8_t* codes, float* ws) {
    orc_dist_table(x, ldx, C, B, M, dsub, ws);
    if (!use_constraint) {
        orc_argmin(ws, B, M, codes);
        return 0;
    }
    float* mm = (float*)malloc(sizeof(float) * 2 * M);
    orc_minmax(ws, B, M, mm);
    orc_centre(ws, mm, B, M);
    free(mm);
    return orc_sinkhorn_codes(ws, B, M, eps, iters, codes);
}

/* decode :168-175 */
void orc_decode(const uint8_t* codes, const float* C, int64_t n, int M, int dsub, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        for (int m = 0; m < M; ++m)
            memcpy(out + (i * M + m) * dsub, C + ((int64_t)m * K + codes[i * M + m]) * dsub, sizeof(float) * dsub);
}

/* ---- Faiss IndexPQ(IP) search restated: per-query LUT, linear scan, size-k heap keeping the
 * largest, output (score desc, id asc).  evaluate_repconc.py:182.  PARITY UNPINNED. */
typedef struct { float s; int64_t id; } hit_t;
static int hit_worse(hit_t a, hit_t b) { /* a ranks after b */
    return a.s < b.s || (a.s == b.s && a.id > b.id);
}
static void heap_sift_down(hit_t* h, int n, int i) { /* min-heap on "better": root = worst kept */
    for (;;) {
        int l = 2 * i + 1, r = l + 1, w = i;
        if (l < n && hit_worse(h[l], h[w])) w = l;
        if (r < n && hit_worse(h[r], h[w])) w = r;
        if (w == i) return;
        hit_t t = h[i]; h[i] = h[w]; h[w] = t;
        i = w;
    }
}
static int hit_cmp_desc(const void* pa, const void* pb) {
    const hit_t a = *(const hit_t*)pa, b = *(const hit_t*)pb;
    if (hit_worse(a, b)) return 1;
    if (hit_worse(b, a)) return -1;
    return 0;
}

void orc_adc_search(const uint8_t* codes, int64_t N, int M, int dsub, const float* C, const float* q, int nq, int k,
                    float* scores, int64_t* ids) {
#pragma omp parallel
    {
        float* lut = (float*)malloc(sizeof(float) * M * K);
        hit_t* heap = (hit_t*)malloc(sizeof(hit_t) * k);
#pragma omp for schedule(dynamic)
        for (int qi = 0; qi < nq; ++qi) {
            for (int m = 0; m < M; ++m)
                for (int kk = 0; kk < K; ++kk) {
                    const float* qs = q + (int64_t)qi * M * dsub + m * dsub;
                    const float* c = C + ((int64_t)m * K + kk) * dsub;
                    float s = 0.f;
                    for (int j = 0; j < dsub; ++j) s = s + qs[j] * c[j];
                    lut[m * K + kk] = s;
                }
            int hn = 0;
            /* four rows at a time (independent accumulators, like Faiss's PQ scanner): each row's sum keeps its own
             * m-ascending order, so the scores are bit-identical to the one-row loop */
            float s4[4];
            for (int64_t n = 0; n < N; ++n) {
                if ((n & 3) == 0) {
                    const int64_t left = N - n < 4 ? N - n : 4;
                    const uint8_t* c0 = codes + n * M;
                    const uint8_t* c1 = c0 + (left > 1 ? M : 0);
                    const uint8_t* c2 = c0 + (left > 2 ? 2 * M : 0);
                    const uint8_t* c3 = c0 + (left > 3 ? 3 * M : 0);
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                    for (int m = 0; m < M; ++m) {
                        const float* t = lut + m * K;
                        a0 = a0 + t[c0[m]];
                        a1 = a1 + t[c1[m]];
                        a2 = a2 + t[c2[m]];
                        a3 = a3 + t[c3[m]];
                    }
                    s4[0] = a0; s4[1] = a1; s4[2] = a2; s4[3] = a3;
                }
                const float s = s4[n & 3];
                hit_t h = {s, n};
                if (hn < k) {
                    heap[hn++] = h;
                    if (hn == k)
                        for (int i = k / 2 - 1; i >= 0; --i) heap_sift_down(heap, k, i);
                } else if (hit_worse(heap[0], h)) {
                    heap[0] = h;
                    heap_sift_down(heap, k, 0);
                }
            }
            qsort(heap, hn, sizeof(hit_t), hit_cmp_desc);
            for (int j = 0; j < k; ++j) {
                scores[(int64_t)qi * k + j] = j < hn ? heap[j].s : -INFINITY;
                ids[(int64_t)qi * k + j] = j < hn ? heap[j].id : -1;
            }
        }
        free(lut);
        free(heap);
    }
}

/* The same search, cache-blocked for many queries (bench.py's CPU baseline on a many-core host): rows are visited in tiles
 * of `tile` rows (outer loop) and the queries are spread over the threads inside a tile, so a tile's codes are read from DRAM
 * once per nq queries instead of once per query — Faiss's own IndexPQ search parallelises over queries only and streams the
 * whole code array per query, which on a two-socket host is bound by the memory one thread's first touch put on ONE node
 * (42 GB/s for 128 threads).  Every query still sees the rows in ascending order with its own heap, so the
 * results are those of orc_adc_search bit for bit (same sums, same tie rule).  tile <= 0: 16384 rows. */
void orc_adc_search_tiled(const uint8_t* codes, int64_t N, int M, int dsub, const float* C, const float* q, int nq, int k,
                          int64_t tile, float* scores, int64_t* ids) {
    if (tile <= 0) tile = 16384;
    tile = (tile + 3) & ~(int64_t)3;          /* whole groups of four rows per tile */
    float* luts = (float*)malloc(sizeof(float) * (size_t)nq * M * K);
    hit_t* heaps = (hit_t*)malloc(sizeof(hit_t) * (size_t)nq * k);
    int* hns = (int*)calloc((size_t)nq, sizeof(int));
#pragma omp parallel for schedule(static)
    for (int qi = 0; qi < nq; ++qi) {
        float* lut = luts + (size_t)qi * M * K;
        for (int m = 0; m < M; ++m)
            for (int kk = 0; kk < K; ++kk) {
                const float* qs = q + (int64_t)qi * M * dsub + m * dsub;
                const float* c = C + ((int64_t)m * K + kk) * dsub;
                float s = 0.f;
                for (int j = 0; j < dsub; ++j) s = s + qs[j] * c[j];
                lut[m * K + kk] = s;
            }
    }
    for (int64_t n0 = 0; n0 < N; n0 += tile) {
        const int64_t n1 = n0 + tile < N ? n0 + tile : N;
#pragma omp parallel for schedule(static)
        for (int qi = 0; qi < nq; ++qi) {
            const float* lut = luts + (size_t)qi * M * K;
            hit_t* heap = heaps + (size_t)qi * k;
            int hn = hns[qi];
            float s4[4];
            for (int64_t n = n0; n < n1; ++n) {
                if (((n - n0) & 3) == 0) {      /* tiles start at multiples of 4: the same groups of four rows as above */
                    const int64_t left = N - n < 4 ? N - n : 4;
                    const uint8_t* c0 = codes + n * M;
                    const uint8_t* c1 = c0 + (left > 1 ? M : 0);
                    const uint8_t* c2 = c0 + (left > 2 ? 2 * M : 0);
                    const uint8_t* c3 = c0 + (left > 3 ? 3 * M : 0);
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                    for (int m = 0; m < M; ++m) {
                        const float* t = lut + m * K;
                        a0 = a0 + t[c0[m]];
                        a1 = a1 + t[c1[m]];
                        a2 = a2 + t[c2[m]];
                        a3 = a3 + t[c3[m]];
                    }
                    s4[0] = a0; s4[1] = a1; s4[2] = a2; s4[3] = a3;
                }
                const float sc = s4[(n - n0) & 3];
                hit_t h = {sc, n};
                if (hn < k) {
                    heap[hn++] = h;
                    if (hn == k)
                        for (int i = k / 2 - 1; i >= 0; --i) heap_sift_down(heap, k, i);
                } else if (hit_worse(heap[0], h)) {
                    heap[0] = h;
                    heap_sift_down(heap, k, 0);
                }
            }
            hns[qi] = hn;
        }
    }
#pragma omp parallel for schedule(static)
    for (int qi = 0; qi < nq; ++qi) {
        hit_t* heap = heaps + (size_t)qi * k;
        const int hn = hns[qi];
        qsort(heap, hn, sizeof(hit_t), hit_cmp_desc);
        for (int j = 0; j < k; ++j) {
            scores[(int64_t)qi * k + j] = j < hn ? heap[j].s : -INFINITY;
            ids[(int64_t)qi * k + j] = j < hn ? heap[j].id : -1;
        }
    }
    free(luts);
    free(heaps);
    free(hns);
}

/* Copy with a parallel first touch: dst (freshly allocated, untouched) gets its pages from the memory nodes of the threads that
 * copy them (static schedule over 2 MiB pieces), instead of all from the node of the one thread that happened to write the
 * array first — the many-thread scans of bench.py's CPU baseline then read from every memory controller of the host. */
void orc_first_touch_copy(void* dst, const void* src, int64_t bytes) {
    const int64_t piece = 2 << 20, n = (bytes + piece - 1) / piece;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t o = i * piece, len = (o + piece <= bytes) ? piece : bytes - o;
        memcpy((char*)dst + o, (const char*)src + o, (size_t)len);
    }
}

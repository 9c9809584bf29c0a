// IVF extension of the PQ index: the code rows are stored list-major (sorted by coarse cell) and a query scans
// only the rows of its `nprobe` best cells.
//
// The reference never builds more than ONE list (models/repconc/evaluate_repconc.py:101-118: nlist = 1, zero coarse
// centroid) — BASELINE.json's "IVF nlist=5000" config has no counterpart in its code.  This is the
// build-side extension SURVEY §8d asks to measure; codes are NOT residual-encoded (by_residual = False), so RepCONC's
// codes stay valid, and probing every list must (and does, tests) return exactly the flat search result.
//
//   ivf_scan_kernel<M>     block = (query, slice of its probes): the query's fp32 LUT in LDS, exact m-ascending scores
//                          of every row of the probed lists -> dense[qi][pos], rowid[qi][pos]
//   ivf_kth_kernel         per query: exact k-th largest of its dense scores (4-pass radix select over global memory)
//   ivf_filter_kernel      per query: rows with score >= that value -> 64-bit keys (ordered score, ~id)
//   adc_select_kernel      (adc_search.hip) sorts the keys and emits the top-k — same tie rule as the flat search
#include "rc_common.h"

#define IVF_CAND_CAP 16384

__device__ __forceinline__ unsigned ivf_order_key(float s) {
    const unsigned u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ivf_unorder_key(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// grid (nq, slices).  probes [nq][nprobe] list ids, base [nq][nprobe] position of each probe's first row in the
// query's dense arrays.
template <int M>
__global__ __launch_bounds__(256) void ivf_scan_kernel(const uint8_t* __restrict__ codes, const int64_t* __restrict__ list_off,
                                                       const float* __restrict__ lut, const int* __restrict__ probes,
                                                       const int* __restrict__ base, int nprobe, int64_t stride,
                                                       float* __restrict__ dense, int* __restrict__ rowid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tab = reinterpret_cast<float*>(smem);   // [M][256]
    const int qi = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < M * RC_K; i += 256) tab[i] = lut[(size_t)qi * M * RC_K + i];
    __syncthreads();
    for (int p = blockIdx.y; p < nprobe; p += gridDim.y) {
        const int l = probes[(size_t)qi * nprobe + p];
        const int64_t r0 = list_off[l], r1 = list_off[l + 1];
        const int64_t o = (int64_t)qi * stride + base[(size_t)qi * nprobe + p];
        for (int64_t r = r0 + tid; r < r1; r += 256) {
            const unsigned* cp = reinterpret_cast<const unsigned*>(codes + r * M);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < M / 4; ++j) {
                const unsigned w = cp[j];
#pragma unroll
                for (int b = 0; b < 4; ++b) s = s + tab[(4 * j + b) * RC_K + ((w >> (8 * b)) & 0xFFu)];
            }
            dense[o + (r - r0)] = s;
            rowid[o + (r - r0)] = (int)r;
        }
    }
}

// thr[qi] = k-th largest of dense[qi][0..count[qi]) (or -inf when count < k).  One block per query.
__global__ __launch_bounds__(1024) void ivf_kth_kernel(const float* __restrict__ dense, const int* __restrict__ count,
                                                       int64_t stride, int k, float* __restrict__ thr) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sel_prefix, sel_rank;
    const int qi = blockIdx.x, tid = threadIdx.x;
    const int n = count[qi];
    if (n < k || k <= 0) {
        if (tid == 0) thr[qi] = -INFINITY;
        return;
    }
    const float* row = dense + (size_t)qi * stride;
    if (tid == 0) { sel_prefix = 0u; sel_rank = (unsigned)k; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = sel_prefix;
        const unsigned himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < n; i += 1024) {
            const unsigned key = ivf_order_key(row[i]);
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned need = sel_rank, b = 255;
            for (;; --b) {
                if (hist[b] >= need) break;
                need -= hist[b];
                if (b == 0) break;
            }
            sel_prefix = prefix | (b << shift);
            sel_rank = need;
        }
        __syncthreads();
    }
    if (tid == 0) thr[qi] = ivf_unorder_key(sel_prefix);
}

// grid (nq, chunks).  Keys of the rows with score >= thr[qi]; id = ids[rowid] (the corpus position of the row).
__global__ __launch_bounds__(256) void ivf_filter_kernel(const float* __restrict__ dense, const int* __restrict__ rowid,
                                                         const int* __restrict__ count, int64_t stride,
                                                         const float* __restrict__ thr, const int64_t* __restrict__ ids,
                                                         unsigned* __restrict__ cand_count,
                                                         unsigned long long* __restrict__ cand) {
    const int qi = blockIdx.x;
    const int n = count[qi];
    const float tau = thr[qi];
    const float* row = dense + (size_t)qi * stride;
    const int* rid = rowid + (size_t)qi * stride;
    const int per = (n + gridDim.y - 1) / gridDim.y;
    const int i0 = blockIdx.y * per, i1 = (i0 + per < n) ? i0 + per : n;
    for (int ib = i0; ib < i1; ib += 256) {
        const int i = ib + threadIdx.x;
        const bool live = i < i1;
        const float s = live ? row[i] : -INFINITY;
        const bool pass = live && (s >= tau);
        const unsigned long long mask = __ballot(pass);
        if (mask) {
            const int lane = threadIdx.x & 63;
            const int rank = __popcll(mask & ((1ull << lane) - 1ull));
            unsigned b = 0;
            if (lane == (int)__builtin_ctzll(mask)) b = atomicAdd(cand_count + qi, (unsigned)__popcll(mask));
            b = __shfl(b, (int)__builtin_ctzll(mask));
            const unsigned slot = b + rank;
            if (pass && slot < IVF_CAND_CAP) {
                const unsigned id = (unsigned)ids[rid[i]];
                cand[(size_t)qi * IVF_CAND_CAP + slot] =
                    ((unsigned long long)ivf_order_key(s) << 32) | (unsigned long long)(0xFFFFFFFFu - id);
            }
        }
    }
}

// adc_search.hip: sort the candidate keys and emit the top-k
int rc_adc_launch_select(rc_handle_t h, unsigned long long* cand, const unsigned* cnt, int nq, int64_t N, int k,
                         int64_t id_offset, float* scores, int64_t* ids, int* status, hipStream_t s, int* qstatus = nullptr);

extern "C" size_t rc_ivf_search_ws_bytes(int nq, int64_t stride) {
    if (nq <= 0 || stride <= 0) return 0;
    size_t o = 0;
    o += rc_align_up((size_t)nq * stride * sizeof(float), 256);   // dense
    o += rc_align_up((size_t)nq * stride * sizeof(int), 256);     // rowid
    o += rc_align_up((size_t)nq * sizeof(float), 256);            // thr
    o += rc_align_up((size_t)nq * sizeof(unsigned), 256);         // cand_count
    o += rc_align_up((size_t)nq * IVF_CAND_CAP * sizeof(unsigned long long), 256);
    return o;
}

template <int M>
static int ivf_launch_scan(rc_handle_t h, const uint8_t* codes, const int64_t* list_off, const float* lut,
                           const int* probes, const int* base, int nq, int nprobe, int64_t stride, float* dense,
                           int* rowid, hipStream_t s) {
    auto kern = ivf_scan_kernel<M>;
    const size_t lds = (size_t)M * RC_K * sizeof(float);
    RC_HIP_CHECK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int slices = (nprobe + 7) / 8;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)nq, (unsigned)slices), dim3(256), lds, s, codes, list_off, lut, probes, base,
                       nprobe, stride, dense, rowid);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// codes: [N,M] uint8 list-major; list_off: [nlist+1] int64 row offsets; ids: [N] int64 corpus id of every row;
// lut: [nq,M,256] fp32 (rc_adc_lut); probes/base: [nq,nprobe] int32; count: [nq] int32 rows scanned per query
// (= base of the last probe + its size); stride >= max count.  Outputs as rc_adc_search.
extern "C" int rc_ivf_search(rc_handle_t h, const uint8_t* codes, const int64_t* list_off, const int64_t* ids, int64_t N,
                             int M, int K, const float* lut, const int* probes, const int* base, const int* count, int nq,
                             int nprobe, int64_t stride, int k, float* scores, int64_t* out_ids, int* status, void* ws,
                             size_t ws_bytes, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !codes || !list_off || !ids || !lut || !probes || !base || !count || !scores || !out_ids || !status ||
        N <= 0 || nq < 0 || nprobe <= 0 || stride <= 0 || k <= 0)
        return RC_EINVAL;
    if (K != RC_K || k > IVF_CAND_CAP / 2 || N > 0xFFFFFFFFll) return RC_ESHAPE;
    if (nq == 0) return RC_OK;
    if (!ws || ws_bytes < rc_ivf_search_ws_bytes(nq, stride)) return RC_EWORKSPACE;
    char* w = (char*)ws;
    size_t o = 0;
    float* dense = (float*)(w + o); o += rc_align_up((size_t)nq * stride * sizeof(float), 256);
    int* rowid = (int*)(w + o);     o += rc_align_up((size_t)nq * stride * sizeof(int), 256);
    float* thr = (float*)(w + o);   o += rc_align_up((size_t)nq * sizeof(float), 256);
    unsigned* cnt = (unsigned*)(w + o); o += rc_align_up((size_t)nq * sizeof(unsigned), 256);
    unsigned long long* cand = (unsigned long long*)(w + o);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (M) {
        case 8:  rc = ivf_launch_scan<8>(h, codes, list_off, lut, probes, base, nq, nprobe, stride, dense, rowid, s); break;
        case 12: rc = ivf_launch_scan<12>(h, codes, list_off, lut, probes, base, nq, nprobe, stride, dense, rowid, s); break;
        case 16: rc = ivf_launch_scan<16>(h, codes, list_off, lut, probes, base, nq, nprobe, stride, dense, rowid, s); break;
        case 24: rc = ivf_launch_scan<24>(h, codes, list_off, lut, probes, base, nq, nprobe, stride, dense, rowid, s); break;
        case 32: rc = ivf_launch_scan<32>(h, codes, list_off, lut, probes, base, nq, nprobe, stride, dense, rowid, s); break;
        case 48: rc = ivf_launch_scan<48>(h, codes, list_off, lut, probes, base, nq, nprobe, stride, dense, rowid, s); break;
        case 64: rc = ivf_launch_scan<64>(h, codes, list_off, lut, probes, base, nq, nprobe, stride, dense, rowid, s); break;
        case 96: rc = ivf_launch_scan<96>(h, codes, list_off, lut, probes, base, nq, nprobe, stride, dense, rowid, s); break;
        default: return RC_ESHAPE;
    }
    if (rc != RC_OK) return rc;
    hipLaunchKernelGGL(ivf_kth_kernel, dim3((unsigned)nq), dim3(1024), 0, s, dense, count, stride, k, thr);
    RC_LAUNCH_CHECK(h);
    RC_HIP_CHECK(h, hipMemsetAsync(cnt, 0, (size_t)nq * sizeof(unsigned), s));
    int chunks = (int)((stride + 16383) / 16384);
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(ivf_filter_kernel, dim3((unsigned)nq, (unsigned)chunks), dim3(256), 0, s, dense, rowid, count, stride,
                       thr, ids, cnt, cand);
    RC_LAUNCH_CHECK(h);
    // N = 0: finding fewer than k rows is legitimate here (the probed lists may hold fewer), only overflow is an error
    return rc_adc_launch_select(h, cand, cnt, nq, 0, k, 0, scores, out_ids, status, s);
}

// ------------------------------------------------------------------------------------------------ coarse quantiser
// Cell of every document: argmin_l ||x - c_l||^2 = argmin_l (||c_l||^2 - 2 <x, c_l>), first minimum, for nlist coarse
// centroids of the FULL dimension D (the reference has no coarse quantiser, evaluate_repconc.py:101-118 builds one list;
// this is the build side of BASELINE.json's nlist = 5000 configuration).  GEMM-shaped ([B, D] x [D, nlist], 2 B nlist D
// flop = 68 Tflop for the 8.84 M-passage corpus at nlist = 5000), so it runs on the matrix cores:
// v_mfma_f32_32x32x2_f32 — fp32 inputs, exact fp32 products, fp32 accumulation in k order (bit-identical to an fmaf
// chain), 157 TFLOP/s peak — with the argmin fused into the epilogue so the [B, nlist] score matrix never exists.
//
// Block = 4 waves = 128 documents x (all list tiles of 128 centroids, one after the other); wave (wr, wc) owns the
// 64 x 64 corner (centroids wr*64.., documents wc*64..) as 2 x 2 MFMA tiles (rows of D = centroids, columns = documents:
// a lane holds 16 centroids of ONE document per tile, so the running argmin needs no cross-lane traffic).  K is walked in
// chunks of 16 staged in LDS (rows padded to 17 floats: conflict-free column reads).
typedef float ivf_f32x16 __attribute__((ext_vector_type(16)));
#define IVFC_TILE 128
#define IVFC_KC 16
#define IVFC_LD (IVFC_KC + 1)

__global__ __launch_bounds__(256) void ivf_coarse_assign_kernel(const float* __restrict__ x, int64_t ldx,
                                                                const float* __restrict__ cent,
                                                                const float* __restrict__ cnorm, int64_t B, int D,
                                                                int nlist, int* __restrict__ cell) {
    __shared__ float sa[2][IVFC_TILE * IVFC_LD];     // centroid chunk  [128][16 (+1)]
    __shared__ float sb[2][IVFC_TILE * IVFC_LD];     // document chunk  [128][16 (+1)]
    __shared__ float s_best[2][IVFC_TILE];
    __shared__ int s_idx[2][IVFC_TILE];
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const int col = l & 31, half = l >> 5;
    const int64_t d0 = (int64_t)blockIdx.x * IVFC_TILE;
    // loader mapping: thread -> (row = tid / 2, 8 consecutive k = (tid & 1) * 8)
    const int lrow = tid >> 1, lk = (tid & 1) * 8;
    const int64_t drow = (d0 + lrow < B) ? d0 + lrow : B - 1;
    float best[2] = {INFINITY, INFINITY};
    int bidx[2] = {0, 0};
    const int nkc = D / IVFC_KC;
    for (int lt = 0; lt < nlist; lt += IVFC_TILE) {
        const int crow = (lt + lrow < nlist) ? lt + lrow : nlist - 1;
        ivf_f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        float4 ra[2], rb[2];
        auto gload = [&](int kc) {
            const float4* pa = reinterpret_cast<const float4*>(cent + (size_t)crow * D + kc * IVFC_KC + lk);
            const float4* pb = reinterpret_cast<const float4*>(x + drow * ldx + kc * IVFC_KC + lk);
            ra[0] = pa[0]; ra[1] = pa[1];
            rb[0] = pb[0]; rb[1] = pb[1];
        };
        auto sstore = [&](int buf) {
            float* da = &sa[buf][lrow * IVFC_LD + lk];
            float* db = &sb[buf][lrow * IVFC_LD + lk];
            da[0] = ra[0].x; da[1] = ra[0].y; da[2] = ra[0].z; da[3] = ra[0].w;
            da[4] = ra[1].x; da[5] = ra[1].y; da[6] = ra[1].z; da[7] = ra[1].w;
            db[0] = rb[0].x; db[1] = rb[0].y; db[2] = rb[0].z; db[3] = rb[0].w;
            db[4] = rb[1].x; db[5] = rb[1].y; db[6] = rb[1].z; db[7] = rb[1].w;
        };
        gload(0);
        sstore(0);
        __syncthreads();
        for (int kc = 0; kc < nkc; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < nkc) gload(kc + 1);
#pragma unroll
            for (int ks = 0; ks < IVFC_KC / 2; ++ks) {
                float fa[2], fb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[t] = sa[buf][(wr * 64 + t * 32 + col) * IVFC_LD + 2 * ks + half];   // A[i = col][k = half]
                    fb[t] = sb[buf][(wc * 64 + t * 32 + col) * IVFC_LD + 2 * ks + half];   // B[k = half][j = col]
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
            if (kc + 1 < nkc) sstore(buf ^ 1);
            __syncthreads();
        }
        // epilogue: this lane's document of column tile b is wc*64 + b*32 + col; its 16 centroids of row tile a are
        // lt + wr*64 + a*32 + (r & 3) + 8 (r >> 2) + 4 half, ascending in r — strict < keeps the first minimum
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = lt + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (c < nlist) {
                        const float s = cnorm[c] - 2.0f * acc[a][b][r];
                        if (s < best[b] || (s == best[b] && c < bidx[b])) { best[b] = s; bidx[b] = c; }
                    }
                }
        __syncthreads();                                           // the next list tile overwrites buffer 0
    }
    // merge: the two half-waves of a lane pair, then the two waves (wr = 0, 1) that share the documents
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float ob = __shfl_xor(best[b], 32);
        const int oi = __shfl_xor(bidx[b], 32);
        if (ob < best[b] || (ob == best[b] && oi < bidx[b])) { best[b] = ob; bidx[b] = oi; }
    }
    if (half == 0) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            s_best[wr][wc * 64 + b * 32 + col] = best[b];
            s_idx[wr][wc * 64 + b * 32 + col] = bidx[b];
        }
    }
    __syncthreads();
    if (tid < IVFC_TILE && d0 + tid < B) {
        float v = s_best[0][tid];
        int i = s_idx[0][tid];
        if (s_best[1][tid] < v || (s_best[1][tid] == v && s_idx[1][tid] < i)) i = s_idx[1][tid];
        cell[d0 + tid] = i;
    }
}

__global__ __launch_bounds__(256) void ivf_cnorm_kernel(const float* __restrict__ cent, int D, int nlist, float* __restrict__ cnorm) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (c >= nlist) return;
    float s = 0.f;
    for (int j = l; j < D; j += 64) s = __builtin_fmaf(cent[(size_t)c * D + j], cent[(size_t)c * D + j], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (l == 0) cnorm[c] = s;
}

// x: [B, ldx >= D] fp32 (16-byte aligned rows, ldx % 4 == 0), cent: [nlist, D] fp32, cell: [B] int32.
// ws: nlist floats (centroid norms).  D % 16 == 0.
extern "C" size_t rc_ivf_coarse_assign_ws_bytes(int nlist) { return nlist > 0 ? rc_align_up((size_t)nlist * sizeof(float), 256) : 0; }

extern "C" int rc_ivf_coarse_assign(rc_handle_t h, const float* x, int64_t ldx, const float* cent, int64_t B, int D,
                                    int nlist, int* cell, void* ws, size_t ws_bytes, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !x || !cent || !cell || B < 0 || D <= 0 || nlist <= 0 || ldx < D) return RC_EINVAL;
    if (D % IVFC_KC != 0 || ldx % 4 != 0) return RC_ESHAPE;
    if (!ws || ws_bytes < rc_ivf_coarse_assign_ws_bytes(nlist)) return RC_EWORKSPACE;
    if (B == 0) return RC_OK;
    hipStream_t s = (hipStream_t)stream;
    float* cnorm = (float*)ws;
    hipLaunchKernelGGL(ivf_cnorm_kernel, dim3((unsigned)((nlist + 3) / 4)), dim3(256), 0, s, cent, D, nlist, cnorm);
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(ivf_coarse_assign_kernel, dim3((unsigned)((B + IVFC_TILE - 1) / IVFC_TILE)), dim3(256), 0, s, x, ldx,
                       cent, (const float*)cnorm, B, D, nlist, cell);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

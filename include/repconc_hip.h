/*
 * repconc_hip.h — C ABI of librepconc_hip.so, the MI355X (gfx950) implementation of the
 * RepCONC product-quantisation hot path.
 *
 * The reference (jingtaozhan/RepCONC) is pure Python: the "plugin interface" of this path is
 * the Python surface of `repconc.models.repconc`.  This header is the C-level
 * boundary a maintainer binds with ctypes from those Python functions (INTEGRATION.md shows the
 * stubs).  Each entry point names the reference lines it replaces; paths are relative to
 * /root/reference/src/repconc/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the handle's HIP device unless suffixed `_host`;
 *   - the caller allocates every buffer (sizes given below / by the *_bytes helpers);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); functions only
 *     ENQUEUE work on it and return, they never synchronise;
 *   - return value: RC_OK (0) or a negative RC_E* code, never throws; rc_error_string() decodes;
 *   - no global mutable state outside the handle; handles are independent and re-entrant;
 *   - K (centroids per sub-quantiser) must be 256 (evaluate_repconc.py:80, run_warmup.py:90);
 *     M may be any divisor of D (modeling_repconc.py:41): dsub = D/M in {8,12,16,24,32,48,64,96} (the recipes' widths)
 *     runs on specialised kernels, any other width on run-time-width kernels with the same arithmetic — the fp32
 *     summation order of the torch-CPU oracle, SURVEY.md §8 a-1, pinned for all 18 divisors of 768;
 *     search: rc_index_search, rc_adc_search_exact and the Python boundary serve ANY M (round 5); the raw screening
 *     entries rc_adc_search / _img / _q have kernels for M in {8,12,16,24,32,48,64,96} and return RC_ESHAPE for any other M
 *     BEFORE anything is enqueued — the caller's route is then rc_adc_search_exact (same results, exact scan with a
 *     run-time width), which is what rc_index_search does itself; the list-centric IVF entries need M in {16,32,48,64,96};
 *   - fp32 arithmetic follows the reference bit for bit (no FMA contraction, IEEE division);
 *     the fp64 Sinkhorn stage is evaluated with potentials and is specified
 *     on its OUTPUT, the codes.
 */
#ifndef REPCONC_HIP_H
#define REPCONC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_OK 0
#define RC_EINVAL (-1)     /* bad argument (null pointer, negative size, …)            */
#define RC_ESHAPE (-2)     /* unsupported K / dsub / M                                 */
#define RC_EHIP (-3)       /* a HIP runtime call failed (see rc_last_hip_error)        */
#define RC_EWORKSPACE (-4) /* workspace smaller than the matching *_ws_bytes()         */
#define RC_ECOMM (-5)      /* RCCL unavailable or a collective call failed             */
#define RC_ESELECT (-6)    /* ADC candidate selection did not converge (thousands of identical codes) */

#define RC_CODE_U8 0
#define RC_CODE_I64 1

/* bits of the `flags` word written by the Sinkhorn entry points */
#define RC_FLAG_NONFINITE 1 /* a row/column sum became 0, inf or NaN — the reference's   */
                            /* "Sinkhorn Algorithm returns nan/inf values" warning,      */
                            /* models/repconc/modeling_repconc.py:64-65                  */
#define RC_FLAG_COMM 4      /* IPC transport: a peer's signal did not arrive within RC_IPC_TIMEOUT_MS (default 600000): */
                            /* the wait gave up instead of hanging the GPU; the codes of this call are garbage       */
#define RC_FLAG_RANGE 2     /* |(L + f) N/ln2| left the range the sweep's integer split  */
                            /* covers (eps < ~3e-4 on centred distances: the reference's */
                            /* own exp(1/eps) overflows fp64 long before, at eps < 1.4e-3) */

typedef struct rc_handle_s* rc_handle_t;
typedef void* rc_stream_t;
typedef struct rc_index_s* rc_index_t;   /* stateful PQ index, see rc_index_* below */

int rc_version(void);
const char* rc_error_string(int code);
int rc_create(rc_handle_t* out, int device);
int rc_destroy(rc_handle_t h);
int rc_last_hip_error(rc_handle_t h); /* hipError_t of the last RC_EHIP on this handle */
int rc_num_cus(rc_handle_t h);

/* ------------------------------------------------------------------ measurement hook
 * With profiling enabled every launch of a hot kernel is bracketed by a pair of hipEvents recorded
 * on the launch stream.  rc_profile_collect() waits for the recorded events of one kernel class,
 * returns the number of launches and the summed device time in milliseconds, and clears them.
 * Classes: 0 = Sinkhorn sweep (sk_sweep_kernel, t >= 1), 1 = ADC filter scan, 2 = nearest assignment,
 * 3 = distance table.  bench.py's `roofline.achieved` comes from this.
 * on = 1: one event pair per launch (the Sinkhorn solve then runs its eager launch loop).  on = 2 ("bracket"): ONE pair
 * around the whole run of sweeps t >= 2 of a solve — eager or replayed from the captured hipGraph — counted as that
 * many launches: two event records per solve, the inter-launch gaps (~1.5 us each) are inside the measured time. */
int rc_profile_enable(rc_handle_t h, int on);
int rc_profile_collect(rc_handle_t h, int kernel_class, int* launches, double* total_ms);

/* ------------------------------------------------------------------ a-1 / a-5
 * Nearest-centroid codes (index build).  Replaces RepCONC.quantize with use_constraint=False,
 * models/repconc/modeling_repconc.py:49-52,66: code[b,m] = argmin_k sum_j (x[b,m*dsub+j]-C[m,k,j])^2,
 * first minimum wins.  x: [B, ldx>=D] fp32 row-major; C: [M,K,dsub] fp32.
 * Either output may be NULL: codes_u8 [B,M] (what evaluate_repconc.py:69 casts to) and
 * codes_i64 [B,M] (what quantize returns). */
int rc_pq_assign_nearest(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B,
                         int D, int M, int K, uint8_t* codes_u8, int64_t* codes_i64,
                         rc_stream_t stream);

/* Same codes, bit for bit, ~5.7x faster: the bf16 matrix cores (v_mfma_f32_32x32x16_bf16 on operands split into two
 * bf16 pieces, three products) screen with the GEMM form ||c||^2 - 2<c,x>; every (row, m) whose best/second-best gap is
 * inside the rounding bound of that form is recomputed with the reference's exact arithmetic and first-minimum rule
 * (csrc/pq_assign_mfma.hip).  x must be 16-byte aligned with ldx % 4 == 0 and B*M < 2^32.
 * ws: rc_pq_assign_nearest_fast_ws_bytes(B, M) bytes (one doubt-list slot per pair: the list cannot overflow; plus the
 * staged bf16 image of the centroids, written once per call by assign_prep_kernel and fetched by every block).
 * Asynchronous and complete — no follow-up call is needed.  rc_pq_assign_nearest_fast_overflow (synchronises) only
 * reports the number of doubtful pairs in *doubtful and returns 0. */
size_t rc_pq_assign_nearest_fast_ws_bytes(int64_t B, int M);
int rc_pq_assign_nearest_fast(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B,
                              int D, int M, int K, uint8_t* codes_u8, int64_t* codes_i64, void* ws,
                              size_t ws_bytes, rc_stream_t stream);
int rc_pq_assign_nearest_fast_overflow(rc_handle_t h, const void* ws, int64_t B, int M, int* doubtful);

/* ------------------------------------------------------------------ a-1
 * Distance table d[M,B,K] fp32 (modeling_repconc.py:50) and, if minmax != NULL, the per-m
 * maximum (minmax[0..M)) and minimum (minmax[M..2M)) over (b,k) (:76-77).
 * ws: rc_pq_dist_table_ws_bytes(B, M) bytes of scratch (block partials).  x must be 16-byte aligned with ldx % 4 == 0
 * (the same holds for the assign_sinkhorn entry points, which start with this table). */
size_t rc_pq_dist_table_ws_bytes(int64_t B, int M);
int rc_pq_dist_table(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B, int D,
                     int M, int K, float* d, float* minmax, void* ws, size_t ws_bytes,
                     rc_stream_t stream);

/* ------------------------------------------------------------------ a-2
 * In place (d - mid)/amp with mid=(mx+mn)/2, amp=(mx-mid)+1e-5f per m
 * (RepCONC.center_distance_for_constraint, modeling_repconc.py:81-84).  `minmax` holds the
 * values AFTER the cross-rank all_reduce(MAX/MIN) of :78-80, which the host performs. */
int rc_pq_centre(rc_handle_t h, float* d, const float* minmax, int64_t B, int M, int K,
                 rc_stream_t stream);

/* ------------------------------------------------------------------ a-3 / a-4, staged form
 * Sinkhorn-Knopp of sinkhorn_algorithm (modeling_repconc.py:137-165) on L = -d/eps, evaluated with
 * potentials f[M,K], g[M,B] (fp64) instead of the in-place matrix Q.  One launch per sweep over d:
 *
 *   rc_sk_sweep(t = 0):   rows[m,k] = sum_b exp(L[m,b,k])                                  (:141,:155)
 *   rc_sk_sweep(t >= 1):  f -= log(sum_r rows_prev[r])   (f = 0 before t = 1)              (:157-158)
 *                         g -= log(colsum of sweep t-1)  (g = 0 in sweep 1)                (:162)
 *                         w = exp(L + f_k + g_b); colsum_b = sum_k w; rows[m,k] = sum_b w/colsum_b
 *   rc_sk_argmax(t = T):  f -= log(sum_r rows_prev[r]);  code[b,m] = argmax_k (L + f), first max (:63,:66)
 *
 * T = sinkhorn_iterations is: sweeps t = 0 … T-1, then rc_sk_argmax(t = T) — T+1 reads of d.  Between two
 * sweeps the host all-gathers `rows_out` of every rank into `rows_prev` [G,M,K] (rank-major; the rank sum of
 * :157 is taken inside the next launch in rank order; on one rank pass rows_out itself, G = 1).  The
 * constants /K, /B, the global normalisation of :152 and the last column normalisation cancel in the argmax.
 *
 * f2:       [2,M,K] fp64, potentials double-buffered by sweep parity (owned by the solve, no init needed)
 * g,colsum: [M,B] fp64-sized scratch each (no init needed).  The column potentials never reach the result (w/colsum and
 *           the argmax are independent of g), so the default sweep keeps them as int32 column exponents inside `g` and
 *           leaves `colsum` untouched; the round-1 sweep (RC_SK_V1=1) stores fp64 g and colsum.  d must be a CENTRED
 *           table (|d| <= 1, rc_pq_centre) — a wider table is rescaled together with eps by a power of two first.
 * rows_out: [M,K] fp64, this rank's row sums of the sweep
 * ws:       rc_sk_ws_bytes(B, M, K) bytes (block partials + arrival counters; sweep 0 resets the counters)
 * flags:    one int, OR-ed with RC_FLAG_* (caller zeroes it before sweep 0)
 */
size_t rc_sk_ws_bytes(int64_t B, int M, int K);
int rc_sk_sweep(rc_handle_t h, const float* d, const double* rows_prev, int G, double* f2, double* g,
                double* colsum, double* rows_out, int64_t B, int M, int K, double eps, int t, int* flags,
                void* ws, size_t ws_bytes, rc_stream_t stream);
int rc_sk_argmax(rc_handle_t h, const float* d, const double* rows_prev, int G, const double* f2, int64_t B,
                 int M, int K, double eps, int t, uint8_t* codes_u8, int64_t* codes_i64, int* flags,
                 rc_stream_t stream);

/* ------------------------------------------------------------------ a-3 for ANY fp64 cost tensor (module boundary)
 * `sinkhorn_algorithm(out, epsilon, sinkhorn_iterations, use_distrib_train)` (modeling_repconc.py:137-165) takes any fp64
 * tensor out[M,K,B]; RepCONC.quantize passes the negated centred fp32 table, which the streaming sweep above serves.  A
 * tensor that is NOT exactly representable in fp32 is served by these two entries on the caller's fp64 data (log-domain
 * potentials, every log-sum-exp max-subtracted; deterministic; not a hot path):
 *   rc_sk64_rows: lse[m,k] = log sum_b exp(out[m,k,b]/eps + g[m,b])      (g = NULL: zeros)            (:155)
 *   rc_sk64_cols: f[m,k] = -log sum_r exp(lse_gathered[r,m,k])  (ranks in order: the all-reduce of :157),
 *                 g_out[m,b] = -log sum_k exp(out[m,k,b]/eps + f[m,k])  (NULL: potentials only)        (:158-163)
 * T iterations = rows, (cols, rows) x (T-1), cols(g_out = NULL); the plan is Q[:,:,b] = softmax_k(out/eps + f).
 * out: [M,K,B] fp64 contiguous (this rank's columns), lse / f: [M,K] fp64, g: [M,B] fp64, lse_gathered: [G,M,K] fp64. */
int rc_sk64_rows(rc_handle_t h, const double* out, const double* g, int64_t B, int M, int K, double eps, double* lse,
                 rc_stream_t stream);
int rc_sk64_cols(rc_handle_t h, const double* out, const double* lse_gathered, int G, int64_t B, int M, int K, double eps,
                 double* f_out, double* g_out, rc_stream_t stream);

/* ------------------------------------------------------------------ a-1 … a-4, one call
 * RepCONC.quantize with use_constraint=True on ONE rank (modeling_repconc.py:47-67,
 * dist.is_initialized()==False): distance table -> centring -> `iters` Sinkhorn iterations ->
 * argmax.  ws: rc_pq_assign_sinkhorn_ws_bytes(B, M, K) bytes. */
size_t rc_pq_assign_sinkhorn_ws_bytes(int64_t B, int M, int K);
int rc_pq_assign_sinkhorn(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B,
                          int D, int M, int K, double eps, int iters, uint8_t* codes_u8,
                          int64_t* codes_i64, int* flags, void* ws, size_t ws_bytes,
                          rc_stream_t stream);

/* ------------------------------------------------------------------ a-1 … a-4, one call, N ranks
 * RepCONC.quantize with use_constraint=True in the dist.is_initialized() branch (modeling_repconc.py:47-67 with
 * :78-80 and :149-157): every rank passes its equal row block of the batch and receives the codes of its rows; the
 * uniform-assignment constraint is over the global batch.  The exchanges (MAX/MIN of the distance range once, one
 * all-gather of the [M,256] fp64 row sums per iteration) run inside the call on the transport the handle was given:
 * IPC peer stores (below; the default of the Python boundary) — ONE chain of all M sub-quantisers, the exchange fused
 * into the sweep kernel — or RCCL (rc_comm_init), where the sub-quantisers are solved as TWO independent chains on two
 * streams so that one chain's all-gather overlaps the other's sweep (comm.hip; RC_DIST_SPLIT=0/1 overrides the rule,
 * rc_solve_num_chains_on says what a call will use).
 *
 * rc_comm_unique_ids: fill ids_host[2*128] on ONE rank (ncclGetUniqueId x2); the caller broadcasts the 256 bytes.
 * rc_comm_init: every rank, same ids; creates two communicators on the handle's device.  RC_ECOMM if RCCL
 * (librccl.so.1, resolved with dlopen at run time) is missing.
 * ws: rc_pq_assign_sinkhorn_dist_ws_bytes(B_local, M, K, world) bytes. */
int rc_comm_unique_ids(void* ids_host);
int rc_comm_init(rc_handle_t h, const void* ids_host, int rank, int world);
int rc_comm_destroy(rc_handle_t h);
int rc_comm_world(rc_handle_t h);

/* IPC transport (round 3; default of the Python boundary, RC_COMM=rccl selects RCCL): the same exchange steps — the
 * dist.all_reduce calls of modeling_repconc.py:78-80,149-157 — as hand-written peer stores.  Every rank owns one receive
 * buffer in device memory, exports it with hipIpcGetMemHandle and maps the other ranks' buffers (ranks may be processes
 * on ONE GPU — what a one-GPU box can test — or one process per GPU of a node, where the stores travel over xGMI; the
 * code is the same).  An all-gather is ONE kernel that stores this rank's slice into slot `rank` of every peer's buffer
 * and bumps the peer's arrival counter (system-scope release), followed by a one-thread kernel that waits for the local
 * counter and re-arms it: no library call, no communicator, capturable in a hipGraph like any kernel; buffers and
 * counters alternate between two parities so consecutive exchanges need no further handshake.  The wait gives up after
 * RC_IPC_TIMEOUT_MS (default 600000; flags |= RC_FLAG_COMM) instead of hanging the device when a peer has died.
 * Round 5: inside rc_pq_assign_sinkhorn_dist the per-iteration exchange (modeling_repconc.py:155-157) is part of the
 * sweep kernel itself — the block that finishes a sub-quantiser's row sums stores them and a sequence-numbered flag at
 * every peer, the next sweep's blocks of that sub-quantiser wait for the peers' flags in their prologue (after their
 * long-latency loads are on their way; a flag-wait kernel instead when ranks share a device): ONE launch per Sinkhorn
 * iteration on N ranks, one chain of all M sub-quantisers (RC_IPC_XSWEEP=0: the push + wait kernels of rounds 3-4).
 * One chain moves [M, 256] fp64 through a 256 KiB slot: M <= 128 on this transport (RC_ESHAPE before anything is
 * enqueued; RC_COMM=rccl has no limit).
 *
 * rc_comm_ipc_export: allocate this rank's buffer, write its RC_IPC_BLOB_BYTES-byte descriptor (IPC handle, device
 *   identity) to blob_host.  The caller gathers the descriptors of all ranks, rank order, through any channel
 *   (torch.distributed, a file, a pipe) and hands them to
 * rc_comm_ipc_connect: map the peers.  After it rc_pq_assign_sinkhorn_dist / rc_comm_allgather use the IPC transport.
 * rc_comm_allgather: generic all-gather of `bytes` bytes per rank into dst [world][bytes] on `stream` (either
 *   transport; per-shard k-means statistics run_warmup.py:102-113, row-sharded search results).  Every rank must call
 *   it with the same `bytes`, in the same order relative to its other collectives.
 * rc_comm_kind: 0 none, 1 RCCL, 2 IPC.  rc_comm_destroy releases either; the caller makes sure (barrier) that no peer
 *   still stores into this rank's buffer. */
#define RC_IPC_BLOB_BYTES 128
#define RC_IPC_MAX_WORLD 16
int rc_comm_ipc_export(rc_handle_t h, int rank, int world, void* blob_host);
int rc_comm_ipc_connect(rc_handle_t h, const void* blobs_host);
int rc_comm_allgather(rc_handle_t h, const void* src, void* dst, size_t bytes, int* flags, rc_stream_t stream);
int rc_comm_kind(rc_handle_t h);
int rc_comm_status(rc_handle_t h); /* IPC transport: RC_FLAG_COMM once a wait has timed out (synchronises the device) */
int rc_solve_num_chains(int world, int M); /* 1 or 2: launches per sweep (measurement bookkeeping; RCCL transport) */
int rc_solve_num_chains_on(rc_handle_t h, int world, int M); /* ... on h's transport (IPC, fused exchange: one chain) */
size_t rc_pq_assign_sinkhorn_dist_ws_bytes(int64_t B_local, int M, int K, int world);
int rc_pq_assign_sinkhorn_dist(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B_local,
                               int D, int M, int K, double eps, int iters, uint8_t* codes_u8,
                               int64_t* codes_i64, int* flags, void* ws, size_t ws_bytes, rc_stream_t stream);

/* ------------------------------------------------------------------ a-6
 * decode (modeling_repconc.py:168-175): out[n, m*dsub:(m+1)*dsub] = C[m, codes[n,m], :], and
 * its gradient w.r.t. C (autograd of the gather at :175): grad_C[m,codes[n,m],:] += grad_out[n,m,:].
 * code_dtype: RC_CODE_U8 or RC_CODE_I64; codes are [n, M] row-major. */
int rc_pq_decode(rc_handle_t h, const void* codes, int code_dtype, const float* C, int64_t n,
                 int M, int K, int dsub, float* out, rc_stream_t stream);
int rc_pq_decode_bwd(rc_handle_t h, const void* codes, int code_dtype, const float* grad_out,
                     int64_t n, int M, int K, int dsub, float* grad_C, rc_stream_t stream);

/* ------------------------------------------------------------------ a-8
 * RepCONC.normalize_centrodis (modeling_repconc.py:112-116): C <- C / max(||C||_2, 1e-12). */
int rc_normalize_centroids(rc_handle_t h, float* C, int M, int K, int dsub, rc_stream_t stream);

/* ------------------------------------------------------------------ a-13
 * hist[m,k] = #{n : codes[n,m]==k} — eval_balance's 256 `.sum().item()` round trips
 * (models/repconc/finetune_repconc.py:588-592) for all sub-quantisers in one launch.
 * hist: [M,K] int32, overwritten. */
int rc_code_hist(rc_handle_t h, const void* codes, int code_dtype, int64_t n, int M, int K,
                 int32_t* hist, rc_stream_t stream);

/* ------------------------------------------------------------------ a-12 (centroid update)
 * Lloyd sufficient statistics of the k-means inside Faiss `index.train`
 * (train/run_warmup.py:113): sums[m,k,:] += x[n, m*dsub:(m+1)*dsub], counts[m,k] += 1 for
 * k = codes[n,m].  sums: [M,K,dsub] fp64, counts: [M,K] int64; ACCUMULATED into (caller
 * zeroes, so shards / ranks can be summed).  rc_kmeans_update: C = sums/counts where
 * counts>0, unchanged otherwise. */
int rc_kmeans_stats(rc_handle_t h, const float* x, int64_t ldx, const uint8_t* codes, int64_t n,
                    int D, int M, int K, double* sums, int64_t* counts, rc_stream_t stream);
int rc_kmeans_update(rc_handle_t h, const double* sums, const int64_t* counts, float* C, int M,
                     int K, int dsub, rc_stream_t stream);
/* Empty clusters after an update, by Faiss's published rule (Clustering.cpp `split_clusters`, run by `index.train`,
 * train/run_warmup.py:113): per sub-quantiser a std::mt19937(1234) walk picks a donor with probability proportional to
 * its size, the empty centroid becomes the donor's scaled by (1 +- 1/1024), the donor the opposite.  On the device (no host
 * synchronisation inside a Lloyd iteration).  counts: [M,K] int64 of the update just made; nsplit: optional device int,
 * incremented by the number of splits. */
int rc_kmeans_split_empty(rc_handle_t h, float* C, const int64_t* counts, int M, int K, int dsub, int* nsplit,
                          rc_stream_t stream);

/* ------------------------------------------------------------------ a-9 … a-11
 * PQ asymmetric-distance (inner product) top-k over raw codes — what `index.search` of the
 * faiss.IndexPQ / 1-list IndexIVFPQ does at models/repconc/evaluate_repconc.py:182 and
 * models/jpq/finetune_jpq.py:176.  Per query: LUT[m][k] = <q_m, C[m,k]>, score(n) = sum_m
 * LUT[m][codes[n,m]] (fp32, m ascending), the k largest scores sorted (score desc, id asc).
 *
 * codes: [N,M] uint8 (the index; stays resident, evaluate_repconc.py:89-98)
 * q: [nq,D] fp32;  scores: [nq,k] fp32;  ids: [nq,k] int64 = row + id_offset, -1 and -inf
 * score when fewer than k rows exist.  status_host semantics are reported through `status`
 * (device int, caller zeroes): bit0 = some query collected fewer than k candidates,
 * bit1 = candidate buffer overflowed; the host retries with another `sel_slack` (see
 * repconc_amd.index).  sel_slack: head-room of the sampled candidate threshold in standard deviations of the sample rank —
 * the threshold is the r-th best of 32768 exactly scored sample rows, r = floor(mu + sel_slack sqrt(mu + 1) + 4) + 1,
 * mu = k 32768 / N; the Python wrapper passes 3 (repconc_amd.ops.ADC_SEL_SLACK: 8e-6 repeated queries per query at k = 1000
 * over 8.84 M rows); larger = fewer repeats, more rows screened in.  ws: rc_adc_search_ws_bytes(N, M, K, nq, k) bytes. */
size_t rc_adc_search_ws_bytes(int64_t N, int M, int K, int nq, int k);
int rc_adc_search(rc_handle_t h, const uint8_t* codes, int64_t N, int M, int K, const float* C,
                  int D, const float* q, int nq, int k, int64_t id_offset, double sel_slack,
                  float* scores, int64_t* ids, int* status, void* ws, size_t ws_bytes,
                  rc_stream_t stream);
/* The same search with the index's PERMUTED CODE IMAGE supplied by the caller.  For M in {16,32,48,64,96} and
 * N >= 2^18 the integer screen reads a second copy of the code matrix in which the bytes of row n are stored in the
 * order its lanes visit the sub-quantisers (a fixed permutation that depends on n mod 16; csrc/adc_search.hip,
 * "conflict-free screen": every LDS gather of the byte tables is then bank-conflict-free whatever the codes).
 * rc_adc_scan_image_bytes(N, M): size of that image (0 = this M does not use one); rc_adc_scan_image converts rows
 * [n0, n0+n) (call it after appending rows — evaluate_repconc.py:89-98); rc_adc_search_img takes it (NULL = rebuild it
 * in the workspace on every call, which is what rc_adc_search does).  Results are identical with or without an image. */
size_t rc_adc_scan_image_bytes(int64_t N, int M);
/* layout introspection (host only, for tests): slot / phase-relative sub-quantiser read by `lane` in gather `step` */
int rc_adc_cf_describe(int M, int lane, int step, int* slot, int* m, int* slots_per_code, int* phases);
int rc_adc_scan_image(rc_handle_t h, const uint8_t* codes, int64_t n0, int64_t n, int M, uint8_t* image,
                      rc_stream_t stream);
/* Layouts.  The flat-search image (rc_adc_scan_image; rc_adc_scan_image_bytes(N, M) bytes, what rc_adc_search_img takes) is
 * row-major [N][M] for the one-phase M (16, 32, 48, 64) and, for M = 96, stored in tiles of 32768 rows, phase-major inside a
 * tile — [n / T][phase][n % T][48] — so each of the two screen passes streams dense 48-byte rows; the buffer holds whole
 * tiles, the position of a row does not depend on the buffer's capacity (rows can be appended).  The list-centric IVF search
 * (rc_ivf_search_lists / _probes) takes its own image for every M (rc_adc_scan_image_rows; layout below). */
int rc_adc_scan_image_rows(rc_handle_t h, const uint8_t* codes, int64_t n0, int64_t n, int M, uint8_t* image,
                           rc_stream_t stream);
/* The IVF image is blocked by chunks of 16 rows (round 3: a wave's load of one chunk and table phase is contiguous), so
 * its buffer holds whole chunks: rc_adc_scan_image_rows_bytes(N, M) bytes.  rc_adc_scan_image_rows_at (host only, no GPU)
 * = byte offset of codes[n][m] in it, -1 for arguments out of range: image[at(n, m)] == codes[n][m] is the whole contract. */
size_t rc_adc_scan_image_rows_bytes(int64_t N, int M);
int64_t rc_adc_scan_image_rows_at(int M, int64_t n, int m);
/* Round 6: the image of the 16-QUERY list-centric screen (rc_ivf_search_probes_q16).  Same blocking by chunks of 16 rows, but a
 * chunk holds [phase p16 = m / 16][lane = (n mod 16) + 16 g][step j] = codes[n][16 p16 + slot(lane, j)] — the order in which the
 * lanes of a ds_read_b128 gather visit the sub-quantisers (rc_adc_q16_describe), 4 bytes per lane and phase.
 * rc_adc_scan_image_rows16_bytes(N, M) bytes; rc_adc_scan_image_rows16_at = byte offset of codes[n][m] (host only). */
int rc_adc_scan_image_rows16(rc_handle_t h, const uint8_t* codes, int64_t n0, int64_t n, int M, uint8_t* image,
                             rc_stream_t stream);
size_t rc_adc_scan_image_rows16_bytes(int64_t N, int M);
int64_t rc_adc_scan_image_rows16_at(int M, int64_t n, int m);
/* Round 3: for the M whose flat search runs the 16-query screen (adc_screen_q16_kernel; M = 48 and 96 unless RC_ADC_Q16
 * says otherwise — read once per process) the flat-search image is [n / 32768][phase = m / 16][n % 32768][16 bytes], the 16
 * bytes of a (row, phase) ordered [lane quarter g][step j] = code of sub-quantiser 16 phase + slot(lane = (n & 15) + 16 g, j).
 * rc_adc_q16_describe (host only, tests): *slot = slot(lane, step); returns 1 if this M's flat search uses the layout, 0 if
 * not, RC_ESHAPE for an M without an image. */
int rc_adc_q16_describe(int M, int lane, int step, int* slot);
size_t rc_adc_search_img_ws_bytes(int64_t N, int M, int K, int nq, int k);
/* measurement: byte offsets, inside the workspace a search was given, of its per-query counts — unsigned[nq] rows that passed
 * the 8-bit screen (0: no screen at this size) and unsigned[nq] rows the exact rescoring kept */
int rc_adc_search_ws_counts(int64_t N, int M, int K, int nq, size_t* survivors_off, size_t* candidates_off);
int rc_adc_search_img(rc_handle_t h, const uint8_t* codes, const uint8_t* scan_image, int64_t N, int M, int K,
                      const float* C, int D, const float* q, int nq, int k, int64_t id_offset, double sel_slack,
                      float* scores, int64_t* ids, int* status, void* ws, size_t ws_bytes, rc_stream_t stream);
/* Search that cannot fail (evaluate_repconc.py:180-185: Faiss's IndexPQ.search returns for ANY index content).
 * rc_adc_search_q = rc_adc_search_img + qstatus: NULL, or nq ints zeroed by the caller that receive the status bits PER
 *   QUERY, so a caller repeats (other sel_slack) or re-routes only the queries concerned — one degenerate query does not
 *   tax the others of its batch.
 * rc_adc_search_exact: exact fp32 score of every row, the min(k, N) best by an 8-pass radix select over the 64-bit keys
 *   (ordered(score) << 32 | ~row), sorted (score desc, id asc).  No sample, no threshold, no status: terminates with the
 *   same answer as the fast path for any codes (all rows identical, k = N, ...).  Slower (N x 4 bytes of scores per query,
 *   ~10 passes): the route for the queries the fast path gives up on.  ws: rc_adc_search_exact_ws_bytes(N, M, K, nq, k). */
int rc_adc_search_q(rc_handle_t h, const uint8_t* codes, const uint8_t* scan_image, int64_t N, int M, int K,
                    const float* C, int D, const float* q, int nq, int k, int64_t id_offset, double sel_slack,
                    float* scores, int64_t* ids, int* status, int* qstatus, void* ws, size_t ws_bytes, rc_stream_t stream);
size_t rc_adc_search_exact_ws_bytes(int64_t N, int M, int K, int nq, int k);
int rc_adc_search_exact(rc_handle_t h, const uint8_t* codes, int64_t N, int M, int K, const float* C, int D,
                        const float* q, int nq, int k, int64_t id_offset, float* scores, int64_t* ids, void* ws,
                        size_t ws_bytes, rc_stream_t stream);
/* the look-up tables alone (test hook): lut [nq,M,K] fp32 */
int rc_adc_lut(rc_handle_t h, const float* C, const float* q, int nq, int D, int M, int K,
               float* lut, rc_stream_t stream);

/* ------------------------------------------------------------------ a-9 … a-11, stateful form
 * The index object the reference keeps inside Faiss (initialize_index / add_docs / index.search,
 * models/repconc/evaluate_repconc.py:78-98,182; JPQ's per-step synchronize_model_index, models/jpq/finetune_jpq.py:209-214),
 * with the device memory owned by the library: codes [ntotal, M] uint8 in one allocation whose capacity doubles,
 * the centroid table [M,256,D/M] (set_centroids rewrites it in place: 786 KB per JPQ step instead of re-cloning the
 * index), and the search workspace.  All pointers are device pointers.  rc_index_search is synchronous: it loops over
 * rc_adc_search until the status word is clean (RC_ESELECT after 4 attempts).  Empty index: -inf scores, -1 ids.
 * One index per handle/device; not thread-safe. */
int rc_index_create(rc_handle_t h, int D, int M, int K, rc_index_t* out);
int rc_index_destroy(rc_index_t idx);
int rc_index_set_centroids(rc_index_t idx, const float* C, rc_stream_t stream);
int rc_index_reserve(rc_index_t idx, int64_t rows, rc_stream_t stream);
int rc_index_add_codes(rc_index_t idx, const uint8_t* codes, int64_t n, rc_stream_t stream);
int rc_index_reset(rc_index_t idx);
int64_t rc_index_ntotal(rc_index_t idx);
const uint8_t* rc_index_codes(rc_index_t idx);
const float* rc_index_centroids(rc_index_t idx);
int rc_index_search(rc_index_t idx, const float* q, int nq, int k, float* scores, int64_t* ids,
                    rc_stream_t stream);

/* ------------------------------------------------------------------ IVF extension (SURVEY §8d input D)
 * The reference only ever builds a 1-list IVFPQ (evaluate_repconc.py:101-118); BASELINE.json's nlist=5000 config is a
 * build-side extension.  Code rows are stored list-major (`codes` [N,M] sorted by coarse cell, `list_off` [nlist+1]
 * row offsets, `ids` [N] corpus position of every row), NOT residual-encoded.  For query qi the rows of lists
 * probes[qi, 0..nprobe) are scored exactly (same fp32 m-ascending sum as rc_adc_search) and the top-k returned with
 * the same order (score desc, id asc); probing every list gives exactly the flat result.
 * lut: [nq,M,256] from rc_adc_lut; base[qi,p] = rows scanned before probe p; count[qi] = rows scanned in total;
 * stride >= max count.  ids -1 / score -inf pad queries whose probed lists hold fewer than k rows.
 * ws: rc_ivf_search_ws_bytes(nq, stride).  status bit1: more than 16384 rows tie at the k-th score. */
/* Build side: cell[b] = argmin_l ||x_b - cent_l||^2 (first minimum) for nlist coarse centroids of the full dimension D,
 * evaluated as ||c||^2 - 2<x,c> on the fp32 matrix cores with the argmin fused in (the [B, nlist] score matrix is never
 * written).  x: [B, ldx >= D] fp32, 16-byte aligned rows, ldx % 4 == 0, D % 16 == 0; cent: [nlist, D]; cell: [B] int32;
 * ws: rc_ivf_coarse_assign_ws_bytes(nlist). */
size_t rc_ivf_coarse_assign_ws_bytes(int nlist);
int rc_ivf_coarse_assign(rc_handle_t h, const float* x, int64_t ldx, const float* cent, int64_t B, int D, int nlist,
                         int* cell, void* ws, size_t ws_bytes, rc_stream_t stream);

/* Centroid update of the coarse k-means (the Lloyd step after rc_ivf_coarse_assign; repconc_amd/ivf.py::coarse_kmeans —
 * a build-side extension, BASELINE configs[3] "IVF nlist=5000"): cent[l] <- mean of the rows with assign[r] == l, summed in
 * a fixed order in fp64 by one block per cell after a stable counting sort (four interleaved row lanes, each ascending, then
 * ((p0 + p1) + p2) + p3: deterministic, no atomics on values);
 * an empty cell takes row splitmix64(seed, iter, l) mod n.  assign: int32 [n] (entries outside [0, nlist) are ignored);
 * counts_out: optional uint32 [nlist].  D % 4 == 0, D <= 1024, nlist <= 16384, 16-byte aligned x / cent rows.
 * ws: rc_ivf_coarse_update_ws_bytes(n, nlist). */
size_t rc_ivf_coarse_update_ws_bytes(int64_t n, int nlist);
int rc_ivf_coarse_update(rc_handle_t h, const float* x, int64_t ldx, const int* assign, int64_t n, int D, int nlist, float* cent,
                         unsigned* counts_out, uint64_t seed, int iter, void* ws, size_t ws_bytes, rc_stream_t stream);
/* List-centric search of the same index (M in {16,32,48,64,96}): all queries probing a cell are split into groups of up
 * to 8, one block per (cell, group) task runs the conflict-free 8-bit screen of the flat search over the cell's rows,
 * survivors are re-scored exactly — results identical to rc_ivf_search.  image: rc_adc_scan_image of the cell-major codes.
 * Per query: probes / sbase [nq,nprobe] (probed cells; position of each probe's first SAMPLED row — a cell of n rows is
 * sampled in runs of 16 rows every 16 ss rows, 16 floor(n / 16 ss) + min(16, n mod 16 ss) entries — in the query's sample
 * array of stride sstride), scount (sampled rows), rows (probed rows), rank (rank of the
 * sample score used as threshold, 0 = keep every probed row).  Tasks: task_list / task_qstart / task_qcnt [ntasks] over
 * sorted_q (query ids ordered by probed cell).  status bit0: a query kept fewer than min(k, rows) candidates (retry with
 * larger ranks), bit1: a candidate list overflowed. */
size_t rc_ivf_search_lists_ws_bytes(int M, int nq, int64_t sstride);
int rc_ivf_search_lists(rc_handle_t h, const uint8_t* codes, const uint8_t* image, const int64_t* list_off,
                        const int64_t* rowmap, int64_t N, int M, int K, const float* lut, int nq, const int* probes,
                        const int* sbase, const int* scount, const int* rows, const int* rank, int nprobe,
                        int64_t sstride, int ss, const int* task_list, const int* task_qstart, const int* task_qcnt,
                        const int* sorted_q, int ntasks, int k, float* scores, int64_t* out_ids, int* status, void* ws,
                        size_t ws_bytes, rc_stream_t stream);
/* The same search with the plan made on the device: sample layout, threshold ranks and the task list are derived from
 * `probes` by four small kernels (no host round trip).  probes [nq,nprobe]: distinct cells per query, nprobe <= nlist;
 * sstride: capacity of a query's sample array, >= the largest possible number of sampled rows of nprobe cells;
 * sel_slack: standard deviations of head-room in the threshold rank (the Python wrapper passes 4: IVFPQIndex.SEL_SLACK);
 * keep_all_rows: queries probing no more rows than this re-score every probed row.  Status bits and results as above.
 * ws: rc_ivf_search_probes_ws_bytes(M, nq, nprobe, nlist, sstride). */
/* Probe selection for the calls below: per query the nprobe cells with the largest coarse score (scores [nq,nlist] fp32,
 * e.g. q @ coarse^T from a library GEMM; ties at the boundary go to the lower cell id), written in ASCENDING CELL ORDER to
 * probes [nq,nprobe] int32 — the searches need the set of cells, not their ranking.  nlist <= 16384. */
int rc_ivf_select_probes(rc_handle_t h, const float* scores, int nq, int nlist, int nprobe, int* probes,
                         rc_stream_t stream);
size_t rc_ivf_search_probes_ws_bytes(int M, int nq, int nprobe, int nlist, int64_t sstride);
int rc_ivf_search_probes(rc_handle_t h, const uint8_t* codes, const uint8_t* image, const int64_t* list_off,
                         const int64_t* rowmap, int64_t N, int nlist, int M, int K, const float* lut, int nq,
                         const int* probes, int nprobe, int64_t sstride, int ss, int k, double sel_slack,
                         int keep_all_rows, float* scores, int64_t* out_ids, int* status, void* ws, size_t ws_bytes,
                         rc_stream_t stream);
/* rc_ivf_search_probes with per-query status words: qstatus [nq] int32 (zeroed by the caller, may be NULL) gets bit 0 for a
 * query that kept fewer than min(k, rows probed) candidates and bit 1 for one whose id list overflowed; the other queries'
 * results stand, so a caller answers only the flagged ones again (repconc_amd.ivf: by the per-query exact scan).  status bit 2
 * (value 4, both entries): a survivor stream of the screen filled up — nobody's results are reliable, repeat with a smaller
 * sel_slack. */
int rc_ivf_search_probes_q(rc_handle_t h, const uint8_t* codes, const uint8_t* image, const int64_t* list_off,
                           const int64_t* rowmap, int64_t N, int nlist, int M, int K, const float* lut, int nq,
                           const int* probes, int nprobe, int64_t sstride, int ss, int k, double sel_slack,
                           int keep_all_rows, float* scores, int64_t* out_ids, int* status, int* qstatus, void* ws,
                           size_t ws_bytes, rc_stream_t stream);
/* The same search on the 16-query screen (round 6; evaluate_repconc.py:180-206 with a list-centric index and whole-query-set
 * calls): tasks of up to 16 queries per probed cell, one ds_read_b128 gather + one i8 MFMA per (16 rows, 4 sub-quantisers,
 * 16 queries) — per query the same table bytes as the 8-query screen and half the gathers.  image16: rc_adc_scan_image_rows16.
 * Same workspace (rc_ivf_search_probes_ws_bytes), status bits and RESULTS as rc_ivf_search_probes_q; it pays when a probed
 * cell is shared by more than ~8 queries of the call (nq x nprobe / nlist). */
int rc_ivf_search_probes_q16(rc_handle_t h, const uint8_t* codes, const uint8_t* image16, const int64_t* list_off,
                             const int64_t* rowmap, int64_t N, int nlist, int M, int K, const float* lut, int nq,
                             const int* probes, int nprobe, int64_t sstride, int ss, int k, double sel_slack,
                             int keep_all_rows, float* scores, int64_t* out_ids, int* status, int* qstatus, void* ws,
                             size_t ws_bytes, rc_stream_t stream);
size_t rc_ivf_search_ws_bytes(int nq, int64_t stride);
int rc_ivf_search(rc_handle_t h, const uint8_t* codes, const int64_t* list_off, const int64_t* ids, int64_t N,
                  int M, int K, const float* lut, const int* probes, const int* base, const int* count, int nq,
                  int nprobe, int64_t stride, int k, float* scores, int64_t* out_ids, int* status, void* ws,
                  size_t ws_bytes, rc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* REPCONC_HIP_H */

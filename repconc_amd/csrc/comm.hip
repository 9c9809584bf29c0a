// Multi-GPU constrained assignment behind the C ABI: the whole batch-sharded Sinkhorn solve of one rank in ONE
// call, with the cross-rank exchange on RCCL (xGMI) issued from C between the sweep launches.
//
// Reference: the dist.is_initialized() branch of RepCONC.quantize — all_reduce(MAX/MIN) of the per-m distance range
// (models/repconc/modeling_repconc.py:78-80), all_reduce(SUM) of the row sums every iteration (:155-157; the global
// total of :149-152 cancels in the argmax).  Mapping on MI355X:
//   * per iteration each rank's [M,256] fp64 row sums (98 KB at M = 48) are ALL-GATHERED (ncclAllGather) and the next
//     sweep's prologue adds them in rank order, so all ranks compute bit-identical potentials;
//   * the M sub-quantisers are independent problems: they are solved as TWO chains (m < M/2 and m >= M/2), each on its
//     own HIP stream with its own communicator, so while one chain's 98 KB all-gather (pure latency on xGMI) is in
//     flight the other chain's sweep has the CUs — the collective leaves the critical path;
//   * no Python between iterations: 100 x (1 kernel launch + 1 ncclAllGather) per chain are enqueued by this function.
//
// RCCL is resolved at run time with dlopen/dlsym ("librccl.so.1"): inside a PyTorch process that is the copy torch has
// already loaded (RTLD_NOLOAD first), so there is exactly one RCCL in the process.
#include "rc_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <stdlib.h>

namespace {
struct nccl_api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};

nccl_api* nccl() {
    static nccl_api api;
    static bool tried = false;
    if (tried) return api.ok ? &api : nullptr;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (api.lib) break;
    }
    if (!api.lib)
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
    if (!api.lib) return nullptr;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.AllReduce;
    return api.ok ? &api : nullptr;
}
}  // namespace

#define RC_NCCL_CHECK(h, expr)                                 \
    do {                                                       \
        ncclResult_t _r = (expr);                              \
        if (_r != ncclSuccess) {                               \
            if (h) (h)->last_hip_error = 100000 + (int)_r;     \
            return RC_ECOMM;                                   \
        }                                                      \
    } while (0)

// ids_host: 2 x NCCL_UNIQUE_ID_BYTES, filled on ONE rank and broadcast by the caller (any channel).
extern "C" int rc_comm_unique_ids(void* ids_host) {
    nccl_api* n = nccl();
    if (!n || !ids_host) return n ? RC_EINVAL : RC_ECOMM;
    for (int i = 0; i < 2; ++i)
        if (n->GetUniqueId((ncclUniqueId*)((char*)ids_host + i * NCCL_UNIQUE_ID_BYTES)) != ncclSuccess) return RC_ECOMM;
    return RC_OK;
}

extern "C" int rc_comm_init(rc_handle_t h, const void* ids_host, int rank, int world) {
    rc_device_guard device_guard_(h);
    nccl_api* n = nccl();
    if (!n) return RC_ECOMM;
    if (!h || !ids_host || world < 1 || rank < 0 || rank >= world) return RC_EINVAL;
    if (h->comm[0]) return RC_EINVAL;   // already initialised
    RC_HIP_CHECK(h, hipSetDevice(h->device));
    for (int i = 0; i < 2; ++i) {
        ncclUniqueId id;
        memcpy(&id, (const char*)ids_host + i * NCCL_UNIQUE_ID_BYTES, NCCL_UNIQUE_ID_BYTES);
        ncclComm_t c = nullptr;
        RC_NCCL_CHECK(h, n->CommInitRank(&c, world, id, rank));
        h->comm[i] = (void*)c;
    }
    h->comm_rank = rank;
    h->comm_world = world;
    return RC_OK;
}

extern "C" int rc_comm_destroy(rc_handle_t h) {
    if (!h) return RC_EINVAL;
    rc_device_guard device_guard_(h);
    nccl_api* n = nccl();
    for (int i = 0; i < 2; ++i)
        if (h->comm[i] && n) { (void)n->CommDestroy((ncclComm_t)h->comm[i]); h->comm[i] = nullptr; }
    // cached iteration graphs hold the communicators and the side stream
    for (auto& g : h->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
        g.exec = nullptr; g.graph = nullptr; g.ws = nullptr; g.stamp = 0;
    }
    if (h->side_stream) { (void)hipStreamDestroy(h->side_stream); h->side_stream = nullptr; }
    if (h->ev_fork) { (void)hipEventDestroy(h->ev_fork); h->ev_fork = nullptr; }
    if (h->ev_join) { (void)hipEventDestroy(h->ev_join); h->ev_join = nullptr; }
    h->comm_world = 0;
    return RC_OK;
}

extern "C" int rc_comm_world(rc_handle_t h) { return h ? h->comm_world : 0; }

static int ensure_side_stream(rc_handle_t h) {
    if (h->side_stream) return RC_OK;
    RC_HIP_CHECK(h, hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    RC_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    RC_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    return RC_OK;
}

// ---- workspace of the distributed solve -------------------------------------------------------------------------
namespace {
struct chain_ws {
    size_t f2, g, colsum, rows, gathered, sweep, end;
};
struct dist_ws {
    size_t d, minmax, dist_ws;
    chain_ws ch[2];
    int m0[2], mc[2], nch;
    size_t total;
};
dist_ws dist_layout(int64_t B, int M, int world, bool split) {
    dist_ws L;
    size_t o = 0;
    L.d = o;       o += rc_align_up((size_t)M * B * RC_K * sizeof(float), 256);
    L.minmax = o;  o += rc_align_up((size_t)2 * M * sizeof(float), 256);
    L.dist_ws = o; o += rc_pq_dist_table_ws_bytes(B, M);
    L.nch = (split && M >= 2) ? 2 : 1;
    L.m0[0] = 0;
    L.mc[0] = (L.nch == 2) ? M / 2 : M;
    L.m0[1] = L.mc[0];
    L.mc[1] = M - L.mc[0];
    for (int c = 0; c < L.nch; ++c) {
        chain_ws& w = L.ch[c];
        const int mc = L.mc[c];
        w.f2 = o;       o += rc_align_up((size_t)2 * mc * RC_K * sizeof(double), 256);
        w.g = o;        o += rc_align_up((size_t)mc * B * sizeof(double), 256);
        w.colsum = o;   o += rc_align_up((size_t)mc * B * sizeof(double), 256);
        w.rows = o;     o += rc_align_up((size_t)mc * RC_K * sizeof(double), 256);
        w.gathered = o; o += rc_align_up((size_t)2 * world * mc * RC_K * sizeof(double), 256);   // ping-pong
        w.sweep = o;    o += rc_sk_ws_bytes(B, mc, RC_K);
        w.end = o;
    }
    L.total = o;
    return L;
}
bool want_split(int world) {
    // Default: two chains as soon as there is a collective to hide.  On one GPU the split only buys ~3 % (the tail
    // of one sweep overlaps the head of the other: 53.1 -> 51.6 ms per 49152-row step) and makes per-launch timings
    // overlap, so it stays off unless RC_DIST_SPLIT=1.
    const char* e = getenv("RC_DIST_SPLIT");
    if (e) return atoi(e) != 0;
    return world > 1;
}
}  // namespace

size_t rc_solve_ws_bytes(int64_t B, int M, int world) {
    return dist_layout(B > 0 ? B : 1, M, world, true).total + 256;   // + the solve's own flag word
}

namespace {
__global__ void solve_fill_range_kernel(float* __restrict__ minmax, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) { minmax[i] = -INFINITY; minmax[M + i] = INFINITY; }   // neutral elements of the MAX / MIN all-reduce
}
__global__ void solve_merge_flags_kernel(const int* __restrict__ own, int* __restrict__ out) {
    if (*own) atomicOr(out, *own);
}

// One iteration (sweep t of every chain + its all-gather).  Shared by the eager loop and the capture.
struct solve_ctx {
    rc_handle_t h; nccl_api* n; const dist_ws* L; char* w; float* d; float* minmax; int64_t B; int M, G; double eps;
    int* flags; hipStream_t st[2]; bool fuse_centre, coll;
};
int solve_iteration(const solve_ctx& c, int t) {
    const dist_ws& L = *c.L;
    for (int ch = 0; ch < L.nch; ++ch) {
        const chain_ws& cw = L.ch[ch];
        const int mc = L.mc[ch];
        float* dc = c.d + (size_t)L.m0[ch] * c.B * RC_K;
        double* gath = (double*)(c.w + cw.gathered);
        const size_t gsz = (size_t)c.G * mc * RC_K;
        const double* prev = gath + (size_t)((t + 1) & 1) * gsz;   // gathered row sums of sweep t-1
        double* out = gath + (size_t)(t & 1) * gsz;
        // one rank: the sweep writes its row sums straight into the "gathered" slot
        double* rows = c.coll ? (double*)(c.w + cw.rows) : out;
        int rc = RC_OK;
        if (c.B == 0) {
            // a rank without rows contributes zero row sums (already zeroed) and only takes part in the exchange
        } else if (t == 0 && c.fuse_centre) {
            rc = rc_sk_sweep0_centre(c.h, dc, c.minmax + L.m0[ch], c.minmax + c.M + L.m0[ch], (double*)(c.w + cw.g),
                                     (double*)(c.w + cw.colsum), rows, c.B, mc, c.eps, c.flags, c.w + cw.sweep,
                                     rc_sk_ws_bytes(c.B, mc, RC_K), c.st[ch]);
        } else {
            rc = rc_sk_sweep(c.h, dc, prev, c.G, (double*)(c.w + cw.f2), (double*)(c.w + cw.g), (double*)(c.w + cw.colsum),
                             rows, c.B, mc, RC_K, c.eps, t, c.flags, c.w + cw.sweep, rc_sk_ws_bytes(c.B, mc, RC_K),
                             (rc_stream_t)c.st[ch]);
        }
        if (rc != RC_OK) return rc;
        if (c.coll)
            RC_NCCL_CHECK(c.h, c.n->AllGather(rows, out, (size_t)mc * RC_K, ncclDouble, (ncclComm_t)c.h->comm[ch], c.st[ch]));
    }
    return RC_OK;
}
}  // namespace

// The whole constrained assignment of this rank's rows on `world` ranks (world == 1: no RCCL involved).
//
// Ordering invariant (multi-rank): every rank enqueues exactly the same sequence of collectives — per iteration chain 0's
// all-gather on communicator 0 / stream 0, then chain 1's on communicator 1 / stream 1 — from ONE host thread (or from
// one captured graph, whose node order is that same sequence).  Two communicators are only ever driven concurrently in
// that fixed order; a rank with no rows (B == 0) still issues every collective.  RC_DIST_SPLIT=0 falls back to one chain.
//
// hipGraph: sweeps t = 2 .. T-1 with their all-gathers (198 kernel launches + 198 collectives per step at T = 100, two
// chains) are captured once per (workspace, shape, eps, T, world) and replayed with one hipGraphLaunch; t = 0 and 1 stay
// eager (RCCL finishes its lazy set-up outside the capture).  RC_GRAPH=0, an active rc_profile_enable (per-launch event
// marks) or a failed capture select the eager loop.
int rc_solve_chains(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B, int D, int M, double eps,
                    int iters, int world, uint8_t* codes_u8, int64_t* codes_i64, int* flags, void* ws, size_t ws_bytes,
                    hipStream_t s0) {
    const int G = world;
    // RC_DIST_FORCE_COLL=1 (tests on a one-GPU box): a one-rank communicator still issues every RCCL call of the loop
    const bool coll = G > 1 || (h->comm[0] && rc_env_int("RC_DIST_FORCE_COLL", 0) != 0);
    nccl_api* n = coll ? nccl() : nullptr;
    if (coll && (!n || !h->comm[0])) return RC_ECOMM;
    if (G == 1 && B == 1) {      // a global batch of one row: exact K-way tie, the reference returns code 0
        if (codes_u8) RC_HIP_CHECK(h, hipMemsetAsync(codes_u8, 0, (size_t)M, s0));
        if (codes_i64) RC_HIP_CHECK(h, hipMemsetAsync(codes_i64, 0, (size_t)M * sizeof(int64_t), s0));
        return RC_OK;
    }
    const int64_t Bw = B > 0 ? B : 1;                        // workspace geometry of an empty rank
    const dist_ws L = dist_layout(Bw, M, G, want_split(G));
    if (!ws || ws_bytes < L.total + 256) return RC_EWORKSPACE;
    char* w = (char*)ws;
    float* d = (float*)(w + L.d);
    float* minmax = (float*)(w + L.minmax);
    int* own_flags = (int*)(w + L.total);                     // captured kernels flag here, merged into `flags` at the end
    RC_HIP_CHECK(h, hipMemsetAsync(own_flags, 0, sizeof(int), s0));
    int rc;
    if (B > 0) {
        if ((rc = rc_pq_dist_table(h, x, ldx, C, B, D, M, RC_K, d, minmax, w + L.dist_ws, rc_pq_dist_table_ws_bytes(B, M),
                                   (rc_stream_t)s0)) != RC_OK) return rc;
    } else {
        hipLaunchKernelGGL(solve_fill_range_kernel, dim3((M + 63) / 64), dim3(64), 0, s0, minmax, M);
        RC_LAUNCH_CHECK(h);
        for (int c = 0; c < L.nch; ++c)
            RC_HIP_CHECK(h, hipMemsetAsync(w + L.ch[c].rows, 0, (size_t)L.mc[c] * RC_K * sizeof(double), s0));
    }
    if (coll) {   // modeling_repconc.py:79-80
        RC_NCCL_CHECK(h, n->AllReduce(minmax, minmax, (size_t)M, ncclFloat, ncclMax, (ncclComm_t)h->comm[0], s0));
        RC_NCCL_CHECK(h, n->AllReduce(minmax + M, minmax + M, (size_t)M, ncclFloat, ncclMin, (ncclComm_t)h->comm[0], s0));
    }
    // centring: fused into the first sweep (one pass over the table less); RC_FUSE_CENTRE=0 keeps the separate kernel
    const bool fuse_centre = rc_env_int("RC_FUSE_CENTRE", 1) != 0;
    if (B > 0 && !fuse_centre && (rc = rc_pq_centre(h, d, minmax, B, M, RC_K, (rc_stream_t)s0)) != RC_OK) return rc;

    solve_ctx cx = {h, n, &L, w, d, minmax, B, M, G, eps, own_flags, {s0, s0}, fuse_centre, coll};
    if (L.nch == 2) {
        if ((rc = ensure_side_stream(h)) != RC_OK) return rc;
        cx.st[1] = h->side_stream;
        RC_HIP_CHECK(h, hipEventRecord(h->ev_fork, s0));
        RC_HIP_CHECK(h, hipStreamWaitEvent(cx.st[1], h->ev_fork, 0));
    }
    auto join = [&]() -> int {
        if (L.nch == 2) {
            RC_HIP_CHECK(h, hipEventRecord(h->ev_join, cx.st[1]));
            RC_HIP_CHECK(h, hipStreamWaitEvent(s0, h->ev_join, 0));
        }
        return RC_OK;
    };
    auto fork = [&]() -> int {
        if (L.nch == 2) {
            RC_HIP_CHECK(h, hipEventRecord(h->ev_fork, s0));
            RC_HIP_CHECK(h, hipStreamWaitEvent(cx.st[1], h->ev_fork, 0));
        }
        return RC_OK;
    };
    // sweeps t = 0 .. iters-1, the two chains enqueued alternately so both streams stay fed
    const int variant = rc_env_int("RC_SK_V1", 0) | (rc_env_int("RC_SK_FKLDS", 1) << 1) | (rc_env_int("RC_SK_NB", 0) << 2) |
                        (rc_env_int("RC_SK_CPB", 0) << 14) | ((int)coll << 24);
    // per-launch event marks (profile mode 1) need the eager loop; the bracket mode (2) times the whole run of sweeps
    const bool want_graph = rc_env_int("RC_GRAPH", 1) != 0 && h->profile_on != 1 && !h->graph_broken && iters > 4;
    const bool bracket = h->profile_on == 2 && L.nch == 1 && B > 0;     // one chain: launches are back to back on s0
    int t = 0;
    const int t_eager = (want_graph || bracket) ? 2 : iters;
    for (; t < t_eager && t < iters; ++t)
        if ((rc = solve_iteration(cx, t)) != RC_OK) return rc;
    if (t < iters) {
        // ---- sweeps t = 2 .. iters-1 from a cached graph
        rc_handle_s::solve_graph* hit = nullptr;
        rc_handle_s::solve_graph* victim = &h->graphs[0];
        if (want_graph)
        for (auto& g : h->graphs) {
            if (g.exec && g.ws == ws && g.B == B && g.M == M && g.iters == iters && g.world == G && g.nch == L.nch &&
                g.variant == variant && g.eps == eps) hit = &g;
            if (g.stamp < victim->stamp) victim = &g;
        }
        if ((rc = join()) != RC_OK) return rc;               // the graph is launched on s0 and forks inside
        if (!hit && want_graph) {
            hipGraph_t graph = nullptr;
            bool ok = hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                h->capturing = 1;
                int crc = fork();
                for (int tt = t; crc == RC_OK && tt < iters; ++tt) crc = solve_iteration(cx, tt);
                if (crc == RC_OK) crc = join();
                h->capturing = 0;
                const hipError_t e = hipStreamEndCapture(s0, &graph);
                ok = (crc == RC_OK) && e == hipSuccess && graph != nullptr;
            }
            hipGraphExec_t exec = nullptr;
            if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                if (graph) (void)hipGraphDestroy(graph);
                h->graph_broken = 1;                         // stay eager on this handle
            } else {
                if (victim->exec) (void)hipGraphExecDestroy(victim->exec);
                if (victim->graph) (void)hipGraphDestroy(victim->graph);
                *victim = {ws, B, M, iters, G, L.nch, variant, eps, graph, exec, 0};
                hit = victim;
            }
        }
        const int nsweeps = iters - t;
        if (bracket) rc_prof_bracket(h, RC_PROF_SK_PASS, s0, true, nsweeps);
        if (hit) {
            hit->stamp = ++h->graph_stamp;
            RC_HIP_CHECK(h, hipGraphLaunch(hit->exec, s0));
            if (bracket) rc_prof_bracket(h, RC_PROF_SK_PASS, s0, false, nsweeps);
            if ((rc = fork()) != RC_OK) return rc;           // the argmax launches below use both streams again
        } else {
            if ((rc = fork()) != RC_OK) return rc;
            for (; t < iters; ++t)
                if ((rc = solve_iteration(cx, t)) != RC_OK) return rc;
            if (bracket) rc_prof_bracket(h, RC_PROF_SK_PASS, s0, false, nsweeps);
        }
    }
    if (B > 0)
        for (int c = 0; c < L.nch; ++c) {
            const chain_ws& cw = L.ch[c];
            const int mc = L.mc[c];
            const double* gath = (const double*)(w + cw.gathered) + (size_t)((iters - 1) & 1) * G * mc * RC_K;
            if ((rc = rc_sk_argmax_strided(h, d + (size_t)L.m0[c] * B * RC_K, gath, G, (const double*)(w + cw.f2), B, mc, eps,
                                           iters, M, L.m0[c], codes_u8, codes_i64, own_flags, cx.st[c])) != RC_OK) return rc;
        }
    if ((rc = join()) != RC_OK) return rc;
    hipLaunchKernelGGL(solve_merge_flags_kernel, dim3(1), dim3(1), 0, s0, (const int*)own_flags, flags);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// number of independent chains (launches per sweep) the solve uses for `world` ranks and M sub-quantisers
extern "C" int rc_solve_num_chains(int world, int M) { return (want_split(world) && M >= 2) ? 2 : 1; }

extern "C" size_t rc_pq_assign_sinkhorn_dist_ws_bytes(int64_t B_local, int M, int K, int world) {
    if (B_local < 0 || M <= 0 || K != RC_K || world < 1) return 0;
    return rc_solve_ws_bytes(B_local, M, world);
}

extern "C" int rc_pq_assign_sinkhorn_dist(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B, int D,
                                          int M, int K, double eps, int iters, uint8_t* codes_u8, int64_t* codes_i64,
                                          int* flags, void* ws, size_t ws_bytes, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !h->comm[0] || !C || !flags || B < 0 || (B > 0 && !x) || M <= 0 || iters < 1 || !(eps > 0.0) ||
        (B > 0 && !codes_u8 && !codes_i64))
        return RC_EINVAL;
    if (K != RC_K || D % M != 0 || !rc_dsub_supported(D / M)) return RC_ESHAPE;
    return rc_solve_chains(h, x, ldx, C, B, D, M, eps, iters, h->comm_world, codes_u8, codes_i64, flags, ws, ws_bytes,
                           (hipStream_t)stream);
}

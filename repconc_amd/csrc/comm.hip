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
    nccl_api* n = nccl();
    for (int i = 0; i < 2; ++i)
        if (h->comm[i] && n) { (void)n->CommDestroy((ncclComm_t)h->comm[i]); h->comm[i] = nullptr; }
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

size_t rc_solve_ws_bytes(int64_t B, int M, int world) { return dist_layout(B, M, world, true).total; }

// The whole constrained assignment of this rank's rows on `world` ranks (world == 1: no RCCL involved).
int rc_solve_chains(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B, int D, int M, double eps,
                    int iters, int world, uint8_t* codes_u8, int64_t* codes_i64, int* flags, void* ws, size_t ws_bytes,
                    hipStream_t s0) {
    const int G = world;
    nccl_api* n = (G > 1) ? nccl() : nullptr;
    if (G > 1 && (!n || !h->comm[0])) return RC_ECOMM;
    if ((int64_t)G * B == 1) {   // a global batch of one row: exact K-way tie, the reference returns code 0
        if (codes_u8) RC_HIP_CHECK(h, hipMemsetAsync(codes_u8, 0, (size_t)M, s0));
        if (codes_i64) RC_HIP_CHECK(h, hipMemsetAsync(codes_i64, 0, (size_t)M * sizeof(int64_t), s0));
        return RC_OK;
    }
    const dist_ws L = dist_layout(B, M, G, want_split(G));
    if (!ws || ws_bytes < L.total) return RC_EWORKSPACE;
    char* w = (char*)ws;
    float* d = (float*)(w + L.d);
    float* minmax = (float*)(w + L.minmax);
    int rc;
    if ((rc = rc_pq_dist_table(h, x, ldx, C, B, D, M, RC_K, d, minmax, w + L.dist_ws, rc_pq_dist_table_ws_bytes(B, M),
                               (rc_stream_t)s0)) != RC_OK) return rc;
    if (G > 1) {   // modeling_repconc.py:79-80
        RC_NCCL_CHECK(h, n->AllReduce(minmax, minmax, (size_t)M, ncclFloat, ncclMax, (ncclComm_t)h->comm[0], s0));
        RC_NCCL_CHECK(h, n->AllReduce(minmax + M, minmax + M, (size_t)M, ncclFloat, ncclMin, (ncclComm_t)h->comm[0], s0));
    }
    // centring: fused into the first sweep (one pass over the table less); RC_FUSE_CENTRE=0 keeps the separate kernel
    const bool fuse_centre = rc_env_int("RC_FUSE_CENTRE", 1) != 0;
    if (!fuse_centre && (rc = rc_pq_centre(h, d, minmax, B, M, RC_K, (rc_stream_t)s0)) != RC_OK) return rc;

    hipStream_t st[2] = {s0, s0};
    if (L.nch == 2) {
        if ((rc = ensure_side_stream(h)) != RC_OK) return rc;
        st[1] = h->side_stream;
        RC_HIP_CHECK(h, hipEventRecord(h->ev_fork, s0));
        RC_HIP_CHECK(h, hipStreamWaitEvent(st[1], h->ev_fork, 0));
    }
    // sweeps t = 0 .. iters-1, the two chains enqueued alternately so both streams stay fed
    for (int t = 0; t < iters; ++t) {
        for (int c = 0; c < L.nch; ++c) {
            const chain_ws& cw = L.ch[c];
            const int mc = L.mc[c];
            const float* dc = d + (size_t)L.m0[c] * B * RC_K;
            double* gath = (double*)(w + cw.gathered);
            const size_t gsz = (size_t)G * mc * RC_K;
            const double* prev = gath + (size_t)((t + 1) & 1) * gsz;   // gathered row sums of sweep t-1
            double* out = gath + (size_t)(t & 1) * gsz;
            // one rank: the sweep writes its row sums straight into the "gathered" slot
            double* rows = (G > 1) ? (double*)(w + cw.rows) : out;
            if (t == 0 && fuse_centre)
                rc = rc_sk_sweep0_centre(h, d + (size_t)L.m0[c] * B * RC_K, minmax + L.m0[c], minmax + M + L.m0[c],
                                         (double*)(w + cw.g), (double*)(w + cw.colsum), rows, B, mc, eps, flags,
                                         w + cw.sweep, rc_sk_ws_bytes(B, mc, RC_K), st[c]);
            else
                rc = rc_sk_sweep(h, dc, prev, G, (double*)(w + cw.f2), (double*)(w + cw.g), (double*)(w + cw.colsum), rows,
                                 B, mc, RC_K, eps, t, flags, w + cw.sweep, rc_sk_ws_bytes(B, mc, RC_K), (rc_stream_t)st[c]);
            if (rc != RC_OK) return rc;
            if (G > 1)
                RC_NCCL_CHECK(h, n->AllGather(rows, out, (size_t)mc * RC_K, ncclDouble, (ncclComm_t)h->comm[c], st[c]));
        }
    }
    for (int c = 0; c < L.nch; ++c) {
        const chain_ws& cw = L.ch[c];
        const int mc = L.mc[c];
        const double* gath = (const double*)(w + cw.gathered) + (size_t)((iters - 1) & 1) * G * mc * RC_K;
        if ((rc = rc_sk_argmax_strided(h, d + (size_t)L.m0[c] * B * RC_K, gath, G, (const double*)(w + cw.f2), B, mc, eps,
                                       iters, M, L.m0[c], codes_u8, codes_i64, flags, st[c])) != RC_OK) return rc;
    }
    if (L.nch == 2) {
        RC_HIP_CHECK(h, hipEventRecord(h->ev_join, st[1]));
        RC_HIP_CHECK(h, hipStreamWaitEvent(s0, h->ev_join, 0));
    }
    return RC_OK;
}

// number of independent chains (launches per sweep) the solve uses for `world` ranks and M sub-quantisers
extern "C" int rc_solve_num_chains(int world, int M) { return (want_split(world) && M >= 2) ? 2 : 1; }

extern "C" size_t rc_pq_assign_sinkhorn_dist_ws_bytes(int64_t B_local, int M, int K, int world) {
    if (B_local <= 0 || M <= 0 || K != RC_K || world < 1) return 0;
    return rc_solve_ws_bytes(B_local, M, world);
}

extern "C" int rc_pq_assign_sinkhorn_dist(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B, int D,
                                          int M, int K, double eps, int iters, uint8_t* codes_u8, int64_t* codes_i64,
                                          int* flags, void* ws, size_t ws_bytes, rc_stream_t stream) {
    if (!h || !h->comm[0] || !x || !C || !flags || B <= 0 || M <= 0 || iters < 1 || !(eps > 0.0) ||
        (!codes_u8 && !codes_i64))
        return RC_EINVAL;
    if (K != RC_K || D % M != 0 || !rc_dsub_supported(D / M)) return RC_ESHAPE;
    return rc_solve_chains(h, x, ldx, C, B, D, M, eps, iters, h->comm_world, codes_u8, codes_i64, flags, ws, ws_bytes,
                           (hipStream_t)stream);
}

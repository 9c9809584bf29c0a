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
#include <unistd.h>

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
    if (h->comm[0] || h->ipc.on || h->ipc.exported) return RC_EINVAL;   // already initialised
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

namespace { void ipc_release(rc_handle_t h); }

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
    if (h->ipc.on || h->ipc.exported) { (void)hipDeviceSynchronize(); ipc_release(h); }
    h->comm_world = 0;
    return RC_OK;
}

extern "C" int rc_comm_world(rc_handle_t h) { return h ? h->comm_world : 0; }


// ---- IPC transport ------------------------------------------------------------------------------------------------
// The exchange steps of the solve as hand-written peer stores (include/repconc_hip.h, rc_comm_ipc_*).  Receive buffer of
// a rank:  data [channel 0..2][parity 0..1][world x IPC_SLOT bytes]  |  counters [channel][parity] (one u64 per 64 B)  |
// status word.  Channels 0 / 1 belong to the two Sinkhorn chains (one stream each), channel 2 to everything else on the
// caller's stream (distance-range all-reduce, rc_comm_allgather).  Exchange number n of a channel uses parity n & 1:
//   push   one kernel, grid (IPC_PUSH_BLOCKS, world): block (b, p) stores its share of this rank's slice into peer p's
//          region at offset rank * bytes (so a region holds the dense [world][bytes] all-gather result) and, after a
//          system-scope fence, adds 1 to peer p's counter of (channel, parity);
//   wait   one thread: spins (s_sleep) until the local counter shows IPC_PUSH_BLOCKS * world arrivals, re-arms it to 0.
// Why two parities are enough: a rank's push n + 2 (same parity as n) is stream-ordered after its wait n + 1, which
// needs every peer's push n + 1, which that peer issued after ITS wait n (counter re-armed) and after the kernels that
// read region n (stream order) — so neither the region nor the counter of exchange n can still be in use.
// No CU is held while waiting beyond one wave, so ranks that share a GPU (the one-GPU test boxes) cannot starve each
// other; a peer that died is noticed after RC_IPC_TIMEOUT_MS (default 10 min; flags |= RC_FLAG_COMM, the transport stays
// broken: ipc_wait_one) instead of hanging the queue.
namespace {
constexpr size_t IPC_SLOT = 256 * 1024;     // most bytes one rank contributes per exchange ([96, 256] fp64 row sums = 192 KiB)
constexpr int IPC_CHANNELS = 3;
constexpr int IPC_PUSH_BLOCKS = 4;
constexpr unsigned IPC_MAGIC = 0x52434950u;  // "RCIP"

struct ipc_blob {                            // RC_IPC_BLOB_BYTES on the wire
    hipIpcMemHandle_t handle;                // 64 bytes
    unsigned magic;
    int rank, world, pid;
    int pci_domain, pci_bus, pci_device;
    char pad[RC_IPC_BLOB_BYTES - 64 - 7 * 4];
};
static_assert(sizeof(ipc_blob) == RC_IPC_BLOB_BYTES, "blob layout");

// after the data: 4 KiB of control words — counters [channel][parity] (64 B apart) | status word at 2048 | first exchange
// number of the running solve per chain at 3072 / 3136 (sk_xchg::seq_base) | this process's peer pointers at 3328 (the
// kernels index them with a run-time rank) — then the arrival flags of the fused exchange (sinkhorn.hip, sk_xchg):
// u64 [chain 0..1][parity][m < IPC_XMAX_M][RC_IPC_MAX_WORLD], one 128-byte line per (chain, parity, m)
constexpr int IPC_XMAX_M = 128;              // = IPC_SLOT / (256 * 8): sub-quantisers one chain can exchange
constexpr size_t IPC_XFLAG_BYTES = (size_t)IPC_XMAX_M * RC_IPC_MAX_WORLD * sizeof(unsigned long long);
size_t ipc_data_bytes(int world) { return (size_t)IPC_CHANNELS * 2 * world * IPC_SLOT; }
size_t ipc_total_bytes(int world) { return ipc_data_bytes(world) + 4096 + 4 * IPC_XFLAG_BYTES; }
size_t ipc_region_off(int world, int ch, int par) { return ((size_t)ch * 2 + par) * world * IPC_SLOT; }
size_t ipc_counter_off(int world, int ch, int par) { return ipc_data_bytes(world) + ((size_t)ch * 2 + par) * 64; }
size_t ipc_status_off(int world) { return ipc_data_bytes(world) + 2048; }
size_t ipc_seq_off(int world, int ch) { return ipc_data_bytes(world) + 3072 + (size_t)ch * 64; }
size_t ipc_peers_off(int world) { return ipc_data_bytes(world) + 3328; }
size_t ipc_xflag_off(int world, int ch, int par) { return ipc_data_bytes(world) + 4096 + ((size_t)ch * 2 + par) * IPC_XFLAG_BYTES; }
static_assert(3328 + RC_IPC_MAX_WORLD * sizeof(char*) <= 4096, "control block");

struct ipc_peers { char* p[RC_IPC_MAX_WORLD]; };

// A transport that has timed out once (status & RC_FLAG_COMM) is broken for good: its rank neither waits NOR pushes any more
// (a push without the matching wait has no back-pressure — a late but healthy peer would count it towards an exchange whose
// region is still being rewritten); the peers run into their own time-out and flag their results.
__device__ __forceinline__ bool ipc_broken(const int* __restrict__ status) {
    return (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & RC_FLAG_COMM) != 0;
}

__global__ __launch_bounds__(256) void ipc_push_kernel(const char* __restrict__ src, size_t bytes, ipc_peers P,
                                                       size_t region_off, size_t slot_off, size_t counter_off,
                                                       const int* __restrict__ status) {
    __shared__ int s_broken;
    if (threadIdx.x == 0) s_broken = ipc_broken(status);
    __syncthreads();
    if (s_broken) return;
    char* dst = P.p[blockIdx.y] + region_off + slot_off;
    const size_t n16 = bytes / 16;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d4[i] = s4[i];
        if (blockIdx.x == 0)
            for (size_t i = n16 * 16 + threadIdx.x; i < bytes; i += blockDim.x) dst[i] = src[i];
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < bytes; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    }
    __threadfence_system();          // this thread's stores are visible system-wide before the barrier ...
    __syncthreads();                 // ... so after it every store of the block is
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(P.p[blockIdx.y] + counter_off), 1ull, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
}

// One thread waits for `want` arrivals on the local counter of (channel, parity) and re-arms it.  A wait that times out
// (a peer died, or is more than RC_IPC_TIMEOUT_MS behind) BREAKS the transport: the status word keeps RC_FLAG_COMM, the
// counter is left as it is (re-arming it would count the late arrivals towards the next exchange of this parity, which
// would then complete early on stale data), and every later wait of this handle leaves at once with the flag set — the
// caller sees RC_FLAG_COMM on every result after the break, never a silently wrong one (round 3 re-armed and carried on).
__device__ __forceinline__ void ipc_wait_one(unsigned long long* __restrict__ counter, unsigned long long want,
                                             int* __restrict__ flags, int* __restrict__ status, long long timeout_ticks) {
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & RC_FLAG_COMM) {
        if (flags) atomicOr(flags, RC_FLAG_COMM);
        return;
    }
    const long long t0 = wall_clock64();                     // constant 100 MHz
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > timeout_ticks) {
            if (flags) atomicOr(flags, RC_FLAG_COMM);
            atomicOr(status, RC_FLAG_COMM);
            return;
        }
    }
    __hip_atomic_store(counter, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// push and wait in ONE launch (the solve's exchange: one kernel per iteration and chain instead of two): every block pushes
// its share and signals as ipc_push_kernel does; block (0, 0) then waits for the LOCAL counter like ipc_wait_kernel.  The
// waiting block holds one wave; all pushes of this rank are issued before it starts to wait, so two ranks cannot wait for
// each other's pushes.
__global__ __launch_bounds__(256) void ipc_pushwait_kernel(const char* __restrict__ src, size_t bytes, ipc_peers P, size_t region_off,
                                                           size_t slot_off, size_t counter_off, unsigned long long* __restrict__ counter,
                                                           unsigned long long want, int* __restrict__ flags, int* __restrict__ status,
                                                           long long timeout_ticks) {
    __shared__ int s_broken;
    if (threadIdx.x == 0) s_broken = ipc_broken(status);
    __syncthreads();
    if (s_broken) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && flags) atomicOr(flags, RC_FLAG_COMM);
        return;
    }
    char* dst = P.p[blockIdx.y] + region_off + slot_off;
    const size_t n16 = bytes / 16;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);       // the solve's row sums: 16-byte aligned, a multiple of 16 bytes
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d4[i] = s4[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(P.p[blockIdx.y] + counter_off), 1ull, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ipc_wait_one(counter, want, flags, status, timeout_ticks);
}

__global__ void ipc_wait_kernel(unsigned long long* __restrict__ counter, unsigned long long want, int* __restrict__ flags,
                                int* __restrict__ status, long long timeout_ticks) {
    ipc_wait_one(counter, want, flags, status, timeout_ticks);
}

// ---- fused exchange of the Sinkhorn row sums (sk_xchg, sinkhorn.hip): the pieces that are not inside the sweep ---------------
// first exchange number of the solve on each chain, read by the (possibly replayed) sweeps through sk_xchg::seq_base
__global__ void xchg_set_seq_kernel(unsigned long long* __restrict__ s0, unsigned long long v0, unsigned long long* __restrict__ s1,
                                    unsigned long long v1) {
    *s0 = v0;
    *s1 = v1;
}
// One thread per (m, rank) flag of exchange number `t` of the solve (flag value seq_base + t + 1): used where no sweep follows
// that could wait in its prologue (before the argmax pass, on a rank without rows) and instead of the in-prologue wait when ranks
// share a device (a sweep whose blocks spin for a peer would keep that peer's sweep off the CUs).
__global__ __launch_bounds__(256) void xchg_flagwait_kernel(const unsigned long long* __restrict__ xflags, int M, int world,
                                                            const unsigned long long* __restrict__ seq_base, int t,
                                                            int* __restrict__ flags, int* __restrict__ status,
                                                            long long timeout_ticks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * world) return;
    const unsigned long long* fp = xflags + (size_t)(i / world) * RC_IPC_MAX_WORLD + (i % world);
    const unsigned long long want = *seq_base + (unsigned long long)t + 1ull;
    // the flag is read with ACQUIRE at system scope (pairs with the pusher's RELEASE store): the row sums behind it are visible
    // to whatever this rank reads afterwards, by the HIP memory model and not only by the sc0 sc1 bits of gfx942 / gfx950
    if (__hip_atomic_load(fp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= want) return;
    if (ipc_broken(status)) { atomicOr(flags, RC_FLAG_COMM); return; }
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > timeout_ticks) {
            atomicOr(flags, RC_FLAG_COMM);
            atomicOr(status, RC_FLAG_COMM);
            return;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);               // system scope
}
// a rank without rows: block m stores `rows` (zeros) [M][K] and the flags exactly as a sweep's reducer would
__global__ __launch_bounds__(RC_K) void xchg_rowpush_kernel(const double* __restrict__ rows, int M, int t, const sk_xchg x,
                                                            int* __restrict__ flags) {
    __shared__ int s_broken;
    if (threadIdx.x == 0) s_broken = ipc_broken(x.status);
    __syncthreads();
    if (s_broken) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(flags, RC_FLAG_COMM);
        return;
    }
    const int m = blockIdx.x, tid = threadIdx.x;
    const double v = rows[(size_t)m * RC_K + tid];
    const size_t off = x.push_data_off + (((size_t)x.rank * M + m) * RC_K + tid) * sizeof(double);
    for (int p = 0; p < x.world; ++p)
        __hip_atomic_store(reinterpret_cast<double*>(x.peers[p] + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < x.world)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(x.peers[tid] + x.push_flag_off) + (size_t)m * RC_IPC_MAX_WORLD + x.rank,
                           *x.seq_base + (unsigned long long)t + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// distance range over ranks: gathered [world][2M] (max then min per rank) -> minmax [2M]
__global__ void ipc_minmax_kernel(const float* __restrict__ gathered, int world, int M, float* __restrict__ minmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * M) return;
    float v = gathered[i];
    for (int r = 1; r < world; ++r) {
        const float o = gathered[(size_t)r * 2 * M + i];
        // NaN ranges propagate like torch's all_reduce(MAX / MIN) would not guarantee; keep the first NaN seen
        v = (i < M) ? ((o > v || o != o) ? o : v) : ((o < v || o != o) ? o : v);
    }
    minmax[i] = v;
}

// default 10 minutes, the order of torch.distributed's process-group time-out: a rank that checkpoints or evaluates while
// its peers already wait must not break the channel (round 3: 30 s)
long long ipc_timeout_ticks() { return (long long)rc_env_int("RC_IPC_TIMEOUT_MS", 600000) * 100000ll; }

// exchange number `n` of channel `ch`: every rank contributes `bytes` from `src`; returns this rank's dense
// [world][bytes] region in *region.  Kernels only (capturable).
int ipc_exchange(rc_handle_t h, int ch, unsigned long long n, const void* src, size_t bytes, const char** region, int* flags,
                 hipStream_t s) {
    const int world = h->comm_world, rank = h->comm_rank, par = (int)(n & 1);
    if (bytes > IPC_SLOT) return RC_EINVAL;
    ipc_peers P;
    for (int r = 0; r < RC_IPC_MAX_WORLD; ++r) P.p[r] = r < world ? h->ipc.peer[r] : nullptr;
    const size_t roff = ipc_region_off(world, ch, par), coff = ipc_counter_off(world, ch, par);
    if (bytes && bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && rc_env_int("RC_IPC_FUSED", 1) != 0) {
        hipLaunchKernelGGL(ipc_pushwait_kernel, dim3(IPC_PUSH_BLOCKS, world), dim3(256), 0, s, (const char*)src, bytes, P, roff,
                           (size_t)rank * bytes, coff, (unsigned long long*)(h->ipc.mine + coff),
                           (unsigned long long)IPC_PUSH_BLOCKS * world, flags, (int*)(h->ipc.mine + ipc_status_off(world)),
                           ipc_timeout_ticks());
        RC_LAUNCH_CHECK(h);
        *region = h->ipc.mine + roff;
        return RC_OK;
    }
    if (bytes)
        hipLaunchKernelGGL(ipc_push_kernel, dim3(IPC_PUSH_BLOCKS, world), dim3(256), 0, s, (const char*)src, bytes, P, roff,
                           (size_t)rank * bytes, coff, (const int*)(h->ipc.mine + ipc_status_off(world)));
    else   // nothing to send still signals (keeps the counters of all ranks in step)
        hipLaunchKernelGGL(ipc_push_kernel, dim3(IPC_PUSH_BLOCKS, world), dim3(256), 0, s, (const char*)h->ipc.mine, (size_t)0, P,
                           roff, (size_t)0, coff, (const int*)(h->ipc.mine + ipc_status_off(world)));
    RC_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(ipc_wait_kernel, dim3(1), dim3(1), 0, s, (unsigned long long*)(h->ipc.mine + coff),
                       (unsigned long long)IPC_PUSH_BLOCKS * world, flags, (int*)(h->ipc.mine + ipc_status_off(world)),
                       ipc_timeout_ticks());
    RC_LAUNCH_CHECK(h);
    *region = h->ipc.mine + roff;
    return RC_OK;
}

// Channel 2 serves every caller stream (rc_comm_allgather on the caller's stream, the solve's range exchange on its own):
// the two-parity argument needs the exchanges of a channel to be ordered, so each one starts after the end of the previous
// one whatever stream that ran on (an event on the handle; same stream: a no-op).  The exchange numbering itself is host
// state of the handle: ONE host thread drives a handle's exchanges (as with a communicator).
int ipc_ch2_begin(rc_handle_t h, hipStream_t s) {
    if (h->capturing) return RC_OK;                       // inside a capture the stream order of the capture is the order
    if (h->ipc.ev_ch2_set) RC_HIP_CHECK(h, hipStreamWaitEvent(s, h->ipc.ev_ch2, 0));
    return RC_OK;
}
int ipc_ch2_end(rc_handle_t h, hipStream_t s) {
    if (h->capturing) return RC_OK;
    if (!h->ipc.ev_ch2) RC_HIP_CHECK(h, hipEventCreateWithFlags(&h->ipc.ev_ch2, hipEventDisableTiming));
    RC_HIP_CHECK(h, hipEventRecord(h->ipc.ev_ch2, s));
    h->ipc.ev_ch2_set = 1;
    return RC_OK;
}

void ipc_release(rc_handle_t h) {
    if (h->ipc.ev_ch2) (void)hipEventDestroy(h->ipc.ev_ch2);
    if (h->ipc.on)
        for (int r = 0; r < h->comm_world && r < RC_IPC_MAX_WORLD; ++r)
            if (r != h->comm_rank && h->ipc.peer[r]) (void)hipIpcCloseMemHandle(h->ipc.peer[r]);
    if (h->ipc.mine) (void)hipFree(h->ipc.mine);
    memset(&h->ipc, 0, sizeof(h->ipc));
}
}  // namespace

extern "C" int rc_comm_ipc_export(rc_handle_t h, int rank, int world, void* blob_host) {
    if (!h || !blob_host || world < 1 || world > RC_IPC_MAX_WORLD || rank < 0 || rank >= world) return RC_EINVAL;
    if (h->comm[0] || h->ipc.on || h->ipc.exported) return RC_EINVAL;   // one transport per handle; rc_comm_destroy first
    rc_device_guard device_guard_(h);
    RC_HIP_CHECK(h, hipSetDevice(h->device));
    // fine-grained device memory: peer stores and the system-scope counters stay coherent while kernels of several
    // agents touch it (RC_IPC_ALLOC=plain|uncached for experiments)
    const char* kind = getenv("RC_IPC_ALLOC");
    void* buf = nullptr;
    const size_t total = ipc_total_bytes(world);
    if (kind && !strcmp(kind, "plain")) RC_HIP_CHECK(h, hipMalloc(&buf, total));
    else RC_HIP_CHECK(h, hipExtMallocWithFlags(&buf, total, (kind && !strcmp(kind, "uncached")) ? hipDeviceMallocUncached
                                                                                               : hipDeviceMallocFinegrained));
    RC_HIP_CHECK(h, hipMemset(buf, 0, total));
    RC_HIP_CHECK(h, hipDeviceSynchronize());
    ipc_blob b;
    memset(&b, 0, sizeof b);
    hipError_t e = hipIpcGetMemHandle(&b.handle, buf);
    if (e != hipSuccess) { (void)hipFree(buf); h->last_hip_error = (int)e; return RC_EHIP; }
    hipDeviceProp_t prop;
    RC_HIP_CHECK(h, hipGetDeviceProperties(&prop, h->device));
    b.magic = IPC_MAGIC; b.rank = rank; b.world = world; b.pid = (int)getpid();
    b.pci_domain = prop.pciDomainID; b.pci_bus = prop.pciBusID; b.pci_device = prop.pciDeviceID;
    memcpy(blob_host, &b, sizeof b);
    h->ipc.mine = (char*)buf;
    h->ipc.exported = 1;
    h->comm_rank = rank;
    h->comm_world = world;
    return RC_OK;
}

extern "C" int rc_comm_ipc_connect(rc_handle_t h, const void* blobs_host) {
    if (!h || !blobs_host || !h->ipc.exported || h->ipc.on) return RC_EINVAL;
    rc_device_guard device_guard_(h);
    const int world = h->comm_world, rank = h->comm_rank;
    const ipc_blob* B = (const ipc_blob*)blobs_host;
    for (int r = 0; r < world; ++r)
        if (B[r].magic != IPC_MAGIC || B[r].rank != r || B[r].world != world) return RC_EINVAL;
    h->ipc.shared_device = 0;
    for (int r = 0; r < world; ++r) {
        if (r == rank) { h->ipc.peer[r] = h->ipc.mine; continue; }
        if (B[r].pci_domain == B[rank].pci_domain && B[r].pci_bus == B[rank].pci_bus && B[r].pci_device == B[rank].pci_device)
            h->ipc.shared_device = 1;
        void* q = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&q, B[r].handle, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int k = 0; k < r; ++k)
                if (k != rank && h->ipc.peer[k]) { (void)hipIpcCloseMemHandle(h->ipc.peer[k]); h->ipc.peer[k] = nullptr; }
            h->last_hip_error = (int)e;
            return RC_ECOMM;
        }
        h->ipc.peer[r] = (char*)q;
    }
    // the peer pointers as this process maps them, for kernels that index them with a run-time rank (sk_xchg::peers)
    RC_HIP_CHECK(h, hipMemcpy(h->ipc.mine + ipc_peers_off(world), h->ipc.peer, (size_t)world * sizeof(char*), hipMemcpyHostToDevice));
    h->ipc.on = 1;
    return RC_OK;
}

extern "C" int rc_comm_kind(rc_handle_t h) { return !h ? 0 : h->ipc.on ? 2 : h->comm[0] ? 1 : 0; }

// flags word of the IPC transport (RC_FLAG_COMM after a timed-out wait); synchronises the device
extern "C" int rc_comm_status(rc_handle_t h) {
    if (!h || !h->ipc.on) return 0;
    rc_device_guard device_guard_(h);
    int v = 0;
    RC_HIP_CHECK(h, hipMemcpy(&v, h->ipc.mine + ipc_status_off(h->comm_world), sizeof v, hipMemcpyDeviceToHost));
    return v;
}

extern "C" int rc_comm_allgather(rc_handle_t h, const void* src, void* dst, size_t bytes, int* flags, rc_stream_t stream) {
    if (!h) return RC_EINVAL;
    if (bytes == 0) return RC_OK;                 // every rank passes the same size: nothing to exchange anywhere
    if (!dst || !src) return RC_EINVAL;
    rc_device_guard device_guard_(h);
    hipStream_t s = (hipStream_t)stream;
    if (h->ipc.on) {
        const int world = h->comm_world;
        int rc = ipc_ch2_begin(h, s);
        if (rc != RC_OK) return rc;
        for (size_t off = 0; off < bytes; off += IPC_SLOT) {
            const size_t chunk = bytes - off < IPC_SLOT ? bytes - off : IPC_SLOT;
            const char* region = nullptr;
            rc = ipc_exchange(h, 2, h->ipc.seq[2]++, (const char*)src + off, chunk, &region, flags, s);
            if (rc != RC_OK) return rc;
            // region [world][chunk] -> dst [world][bytes] at column offset off
            RC_HIP_CHECK(h, hipMemcpy2DAsync((char*)dst + off, bytes, region, chunk, chunk, (size_t)world,
                                             hipMemcpyDeviceToDevice, s));
        }
        return ipc_ch2_end(h, s);
    }
    if (h->comm[0]) {
        nccl_api* n = nccl();
        if (!n) return RC_ECOMM;
        RC_NCCL_CHECK(h, n->AllGather(src, dst, bytes, ncclChar, (ncclComm_t)h->comm[0], s));
        return RC_OK;
    }
    if (bytes) RC_HIP_CHECK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    return RC_OK;
}

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
// The exchange of the row sums inside the sweep kernels (sk_xchg): IPC transport + version-2 sweep.  RC_IPC_XSWEEP=0: the
// round-3/4 form (sweep, then a push + wait kernel per chain and iteration).
bool want_xsweep(rc_handle_t h) { return h->ipc.on && rc_sk_xchg_capable() && rc_env_int("RC_IPC_XSWEEP", 1) != 0; }
// ... and the wait for the previous exchange inside the next sweep's prologue: only when no peer shares this device — blocks
// that spin for a peer's row sums would hold the CU slots that peer's sweep needs (the one-GPU test boxes).  RC_IPC_INWAIT
// forces it either way (tests: small grids that fit the device together, short time-out).
bool want_inwait(rc_handle_t h) {
    const char* e = getenv("RC_IPC_INWAIT");
    if (e && *e) return atoi(e) != 0;
    return !h->ipc.shared_device;
}
// One chain or two (RC_DIST_SPLIT overrides).  A rule every rank evaluates identically (world, transport): two chains exist
// to hide a collective behind the other chain's sweep and cost 18-20 us per iteration on their own (twice the per-block
// prologue and hand-over, DESIGN_HISTORY 9.15); with the exchange fused into the sweep there is no collective launch left to hide, so
// the IPC transport runs ONE chain at every batch size; RCCL (a ~20-30 us collective per iteration) keeps two.  On one GPU the
// split buys nothing on top of the rotating wave priority of the sweep (round 4) and makes per-launch timings overlap.
bool want_split(rc_handle_t h, int world) {
    const char* e = getenv("RC_DIST_SPLIT");
    if (e && *e) return atoi(e) != 0;
    if (world <= 1) return false;
    return !(h && want_xsweep(h));
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
    int* flags; hipStream_t st[2]; bool fuse_centre, coll, ipc;
    unsigned long long ipc_base[2];      // exchange number of sweep 0 on channel 0 / 1 (IPC transport)
    bool xsweep, inwait; int iters;      // exchange fused into the sweeps (sk_xchg); wait in the next sweep's prologue
};
// sk_xchg of sweep t of chain ch (exchange number ipc_base[ch] + t; the previous one has the other parity)
sk_xchg xchg_of(const solve_ctx& c, int ch, int t) {
    const int world = c.G, par = (int)((c.ipc_base[ch] + (unsigned long long)t) & 1);
    char* mine = c.h->ipc.mine;
    sk_xchg x;
    x.push = 1;
    x.wait = (t > 0 && c.inwait) ? 1 : 0;
    x.rank = c.h->comm_rank;
    x.world = world;
    x.peers = reinterpret_cast<char* const*>(mine + ipc_peers_off(world));
    x.push_data_off = ipc_region_off(world, ch, par);
    x.push_flag_off = ipc_xflag_off(world, ch, par);
    x.wait_flags = reinterpret_cast<const unsigned long long*>(mine + ipc_xflag_off(world, ch, par ^ 1));
    x.seq_base = reinterpret_cast<const unsigned long long*>(mine + ipc_seq_off(world, ch));
    x.status = reinterpret_cast<int*>(mine + ipc_status_off(world));
    x.timeout_ticks = ipc_timeout_ticks();
    return x;
}
// where the gathered [G, mc, K] row sums of sweep t of chain ch live: the workspace ping-pong, or (IPC transport) this
// rank's receive region of that exchange
const double* gathered_rows(const solve_ctx& c, int ch, int t) {
    if (c.ipc) return (const double*)(c.h->ipc.mine + ipc_region_off(c.G, ch, (int)((c.ipc_base[ch] + (unsigned long long)t) & 1)));
    return (const double*)(c.w + c.L->ch[ch].gathered) + (size_t)(t & 1) * c.G * c.L->mc[ch] * RC_K;
}
int solve_iteration(const solve_ctx& c, int t) {
    const dist_ws& L = *c.L;
    for (int ch = 0; ch < L.nch; ++ch) {
        const chain_ws& cw = L.ch[ch];
        const int mc = L.mc[ch];
        float* dc = c.d + (size_t)L.m0[ch] * c.B * RC_K;
        const double* prev = t > 0 ? gathered_rows(c, ch, t - 1) : nullptr;   // gathered row sums of sweep t-1
        double* out = c.ipc ? nullptr : const_cast<double*>(gathered_rows(c, ch, t));
        // one rank: the sweep writes its row sums straight into the "gathered" slot
        double* rows = c.coll ? (double*)(c.w + cw.rows) : out;
        int rc = RC_OK;
        if (c.xsweep) {
            // ONE launch per iteration: the reducer of every sub-quantiser pushes its row sums, the next sweep's prologue
            // (or, ranks sharing a device / no sweep to follow, a flag-wait kernel) waits for the peers'
            const sk_xchg x = xchg_of(c, ch, t);
            const size_t swb = rc_sk_ws_bytes(c.B > 0 ? c.B : 1, mc, RC_K);
            if (c.B == 0) {
                hipLaunchKernelGGL(xchg_rowpush_kernel, dim3(mc), dim3(RC_K), 0, c.st[ch], (const double*)(c.w + cw.rows), mc, t, x,
                                   c.flags);
                RC_LAUNCH_CHECK(c.h);
            } else if (t == 0 && c.fuse_centre) {
                rc = rc_sk_sweep0_centre(c.h, dc, c.minmax + L.m0[ch], c.minmax + c.M + L.m0[ch], (double*)(c.w + cw.g),
                                         (double*)(c.w + cw.colsum), nullptr, c.B, mc, c.eps, c.flags, c.w + cw.sweep, swb, c.st[ch], &x);
            } else {
                rc = rc_sk_sweep_x(c.h, dc, prev, c.G, (double*)(c.w + cw.f2), (double*)(c.w + cw.g), (double*)(c.w + cw.colsum),
                                   nullptr, c.B, mc, c.eps, t, c.flags, c.w + cw.sweep, swb, c.st[ch], &x);
            }
            if (rc != RC_OK) return rc;
            if (!c.inwait || c.B == 0 || t == c.iters - 1) {
                hipLaunchKernelGGL(xchg_flagwait_kernel, dim3((mc * c.G + 255) / 256), dim3(256), 0, c.st[ch],
                                   reinterpret_cast<const unsigned long long*>(c.h->ipc.mine + x.push_flag_off), mc, c.G, x.seq_base, t,
                                   c.flags, x.status, x.timeout_ticks);
                RC_LAUNCH_CHECK(c.h);
            }
            continue;
        }
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
        if (c.ipc) {
            const char* region = nullptr;
            if ((rc = ipc_exchange(c.h, ch, c.ipc_base[ch] + (unsigned long long)t, rows, (size_t)mc * RC_K * sizeof(double),
                                   &region, c.flags, c.st[ch])) != RC_OK) return rc;
        } else if (c.coll) {
            RC_NCCL_CHECK(c.h, c.n->AllGather(rows, out, (size_t)mc * RC_K, ncclDouble, (ncclComm_t)c.h->comm[ch], c.st[ch]));
        }
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
    const bool ipc = h->ipc.on != 0;
    const bool coll = G > 1 || ((h->comm[0] || ipc) && rc_env_int("RC_DIST_FORCE_COLL", 0) != 0);
    nccl_api* n = (coll && !ipc) ? nccl() : nullptr;
    if (coll && !ipc && (!n || !h->comm[0])) return RC_ECOMM;
    if (G == 1 && B == 1) {      // a global batch of one row: exact K-way tie, the reference returns code 0
        if (codes_u8) RC_HIP_CHECK(h, hipMemsetAsync(codes_u8, 0, (size_t)M, s0));
        if (codes_i64) RC_HIP_CHECK(h, hipMemsetAsync(codes_i64, 0, (size_t)M * sizeof(int64_t), s0));
        return RC_OK;
    }
    const int64_t Bw = B > 0 ? B : 1;                        // workspace geometry of an empty rank
    const dist_ws L = dist_layout(Bw, M, G, want_split(h, G));
    if (!ws || ws_bytes < L.total + 256) return RC_EWORKSPACE;
    // the IPC transport moves one chain's [mc, K] fp64 row sums per exchange through a fixed slot: refuse BEFORE anything is
    // enqueued (MCQ_M > 128 in one chain, > 256 in two; RC_COMM=rccl has no such limit)
    if (coll && ipc)
        for (int c = 0; c < L.nch; ++c)
            if ((size_t)L.mc[c] * RC_K * sizeof(double) > IPC_SLOT || L.mc[c] > IPC_XMAX_M) return RC_ESHAPE;
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
    if (coll && ipc) {   // modeling_repconc.py:79-80 as one all-gather of the [2M] ranges + a local max / min
        const char* region = nullptr;
        if ((rc = ipc_ch2_begin(h, s0)) != RC_OK) return rc;
        if ((rc = ipc_exchange(h, 2, h->ipc.seq[2]++, minmax, (size_t)2 * M * sizeof(float), &region, own_flags, s0)) != RC_OK) return rc;
        hipLaunchKernelGGL(ipc_minmax_kernel, dim3((2 * M + 63) / 64), dim3(64), 0, s0, (const float*)region, G, M, minmax);
        RC_LAUNCH_CHECK(h);
        if ((rc = ipc_ch2_end(h, s0)) != RC_OK) return rc;
    } else if (coll) {   // modeling_repconc.py:79-80
        RC_NCCL_CHECK(h, n->AllReduce(minmax, minmax, (size_t)M, ncclFloat, ncclMax, (ncclComm_t)h->comm[0], s0));
        RC_NCCL_CHECK(h, n->AllReduce(minmax + M, minmax + M, (size_t)M, ncclFloat, ncclMin, (ncclComm_t)h->comm[0], s0));
    }
    // centring: fused into the first sweep (one pass over the table less); RC_FUSE_CENTRE=0 keeps the separate kernel
    const bool fuse_centre = rc_env_int("RC_FUSE_CENTRE", 1) != 0;
    if (B > 0 && !fuse_centre && (rc = rc_pq_centre(h, d, minmax, B, M, RC_K, (rc_stream_t)s0)) != RC_OK) return rc;

    const bool use_ipc = coll && ipc;
    const bool xsweep = use_ipc && want_xsweep(h);
    solve_ctx cx = {h, n, &L, w, d, minmax, B, M, G, eps, own_flags, {s0, s0}, fuse_centre, coll, use_ipc,
                    {h->ipc.seq[0], h->ipc.seq[1]}, xsweep, xsweep && want_inwait(h) && L.nch == 1, iters};
    // (prologue waits only with ONE chain: with two, blocks of chain 0's sweep t + 1 spinning for a peer's chain-0 push can
    // hold the CU slots that peer's chain-1 sweep needs to get to ITS push while this rank's chain-1 blocks queue behind the
    // spinners — a cross-chain stall that only RC_IPC_TIMEOUT_MS ends.  Two chains use the flag-wait kernels.)
    if (use_ipc) {   // this solve's exchanges are numbered base .. base + iters - 1 on each channel it uses
        h->ipc.seq[0] += (unsigned long long)iters;
        if (L.nch == 2) h->ipc.seq[1] += (unsigned long long)iters;
    }
    if (xsweep) {    // the sweeps (eager or replayed) read their exchange numbers relative to these words
        hipLaunchKernelGGL(xchg_set_seq_kernel, dim3(1), dim3(1), 0, s0,
                           reinterpret_cast<unsigned long long*>(h->ipc.mine + ipc_seq_off(G, 0)), cx.ipc_base[0],
                           reinterpret_cast<unsigned long long*>(h->ipc.mine + ipc_seq_off(G, 1)), cx.ipc_base[1]);
        RC_LAUNCH_CHECK(h);
    }
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
    const int variant = (rc_env_int("RC_SK_V1", 0) | (rc_env_int("RC_SK_FKLDS", 1) << 1) | (rc_env_int("RC_SK_NB", 0) << 2) |
                         (rc_env_int("RC_SK_CPB", 0) << 14) | ((int)coll << 24) | ((int)use_ipc << 25) |
                         ((int)(cx.ipc_base[0] & 1) << 26) | ((int)(cx.ipc_base[1] & 1) << 27) | ((int)cx.xsweep << 28) |
                         ((int)cx.inwait << 29)) ^
                        (int)((unsigned)(rc_env_int("RC_SK_PRIO", -1) + 1) * 0x10000001u);   // (a captured launch keeps its priority setting)
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
            const double* gath = gathered_rows(cx, c, iters - 1);
            if ((rc = rc_sk_argmax_strided(h, d + (size_t)L.m0[c] * B * RC_K, gath, G, (const double*)(w + cw.f2), B, mc, eps,
                                           iters, M, L.m0[c], codes_u8, codes_i64, own_flags, cx.st[c])) != RC_OK) return rc;
        }
    if ((rc = join()) != RC_OK) return rc;
    hipLaunchKernelGGL(solve_merge_flags_kernel, dim3(1), dim3(1), 0, s0, (const int*)own_flags, flags);
    RC_LAUNCH_CHECK(h);
    return RC_OK;
}

// number of independent chains (launches per sweep) the solve uses for `world` ranks and M sub-quantisers
extern "C" int rc_solve_num_chains(int world, int M) { return (want_split(nullptr, world) && M >= 2) ? 2 : 1; }
// ... on THIS handle's transport (the IPC transport with the exchange fused into the sweeps runs one chain)
extern "C" int rc_solve_num_chains_on(rc_handle_t h, int world, int M) { return (want_split(h, world) && M >= 2) ? 2 : 1; }

extern "C" size_t rc_pq_assign_sinkhorn_dist_ws_bytes(int64_t B_local, int M, int K, int world) {
    if (B_local < 0 || M <= 0 || K != RC_K || world < 1) return 0;
    return rc_solve_ws_bytes(B_local, M, world);
}

extern "C" int rc_pq_assign_sinkhorn_dist(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B, int D,
                                          int M, int K, double eps, int iters, uint8_t* codes_u8, int64_t* codes_i64,
                                          int* flags, void* ws, size_t ws_bytes, rc_stream_t stream) {
    rc_device_guard device_guard_(h);
    if (!h || !(h->comm[0] || h->ipc.on) || !C || !flags || B < 0 || (B > 0 && !x) || M <= 0 || iters < 1 || !(eps > 0.0) ||
        (B > 0 && !codes_u8 && !codes_i64))
        return RC_EINVAL;
    if (K != RC_K || D % M != 0) return RC_ESHAPE;             // any width: rc_pq_dist_table picks the kernel
    return rc_solve_chains(h, x, ldx, C, B, D, M, eps, iters, h->comm_world, codes_u8, codes_i64, flags, ws, ws_bytes,
                           (hipStream_t)stream);
}

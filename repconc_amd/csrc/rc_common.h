// Shared declarations for librepconc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/repconc_hip.h"

#include <vector>

// kernel classes that can be bracketed with HIP events (rc_profile_*)
enum { RC_PROF_SK_PASS = 0, RC_PROF_ADC_SCAN = 1, RC_PROF_ASSIGN_NEAREST = 2, RC_PROF_DIST_TABLE = 3, RC_PROF_NSLOT = 4 };

struct rc_handle_s {
    int device;
    int num_cus;
    int last_hip_error;
    int profile_on;
    std::vector<hipEvent_t> prof_ev[RC_PROF_NSLOT];  // start, stop, start, stop, ...
    std::vector<int> prof_n[RC_PROF_NSLOT];          // launches covered by each (start, stop) pair
    double* exp2_tab[3];                             // device tables 2^(j/N): [0] N=256, [1] N=2048, [2] N=4096
    // RCCL state of rc_comm_init (comm.hip): two communicators so two independent chains of collectives can be
    // in flight on two streams
    void* comm[2];
    int comm_rank, comm_world;
    hipStream_t side_stream;
    hipEvent_t ev_fork, ev_join;
    // hipGraph cache of the iteration chain of rc_solve_chains (comm.hip): sweeps t = 2 .. T-1 (+ all-gathers) are
    // captured once per (workspace, shape, eps, T, world, variant) and replayed
    struct solve_graph {
        void* ws; int64_t B; int M, iters, world, nch, variant; double eps;
        hipGraph_t graph; hipGraphExec_t exec; unsigned long long stamp;
    };
    solve_graph graphs[4];
    unsigned long long graph_stamp;
    // IPC transport (comm.hip, rc_comm_ipc_*): peer-mapped receive buffers instead of RCCL communicators
    struct ipc_state {
        int on;                                       // rc_comm_ipc_connect succeeded
        int exported;                                 // rc_comm_ipc_export done, waiting for connect
        int shared_device;                            // some peer lives on my device (one-GPU test boxes)
        char* mine;                                   // my receive buffer
        char* peer[RC_IPC_MAX_WORLD];                 // peer[r] = rank r's buffer as mapped here (peer[rank] = mine)
        unsigned long long seq[3];                    // exchanges done per channel (0/1: Sinkhorn chains, 2: everything else)
        hipEvent_t ev_ch2;                            // end of the last channel-2 exchange: the next one (on ANY stream) waits for it
        int ev_ch2_set;
    } ipc;
    void* scratch;                                    // handle-owned device scratch (rc_scratch), grown on demand
    size_t scratch_bytes;
    std::vector<void*> scratch_retired;               // outgrown scratch blocks, kept until rc_destroy (see rc_scratch)
    int graph_broken;                                 // capture failed once on this handle: stay eager
    int capturing;                                    // inside stream capture: no event marks
    void* km_ctl;                                     // kmeans.hip: device control block of the fixed-point statistics (hint, per-piece words)
};

// Fused exchange of the Sinkhorn row sums (sinkhorn.hip: sk_sweep2_kernel<.., XCHG = true>; set up by comm.hip on the IPC
// transport).  Passed by value in the kernel arguments.
struct sk_xchg {
    int push;                                 // the reducer of sub-quantiser m stores its [K] sums + flag (m, rank) at every peer
    int wait;                                 // the prologue waits for the `world` flags of m of the PREVIOUS exchange
    int rank, world;
    char* const* peers;                       // device array [world]: every rank's receive buffer as mapped here
    size_t push_data_off, push_flag_off;      // byte offsets (the same in every rank's buffer) of THIS sweep's exchange
    const unsigned long long* wait_flags;     // my flags of the previous exchange, u64 [M][RC_IPC_MAX_WORLD]
    const unsigned long long* seq_base;       // device word: number of the solve's first exchange on this channel
    int* status;                              // the transport's status word (RC_FLAG_COMM once broken)
    long long timeout_ticks;                  // of the 100 MHz clock
};

// comm.hip: the full constrained assignment as one or two chains of sub-quantisers (world == 1: no RCCL)
size_t rc_solve_ws_bytes(int64_t B, int M, int world);
int rc_solve_chains(rc_handle_t h, const float* x, int64_t ldx, const float* C, int64_t B, int D, int M, double eps,
                    int iters, int world, uint8_t* codes_u8, int64_t* codes_i64, int* flags, void* ws, size_t ws_bytes,
                    hipStream_t s0);
int rc_sk_sweep0_centre(rc_handle_t h, float* d, const float* mx, const float* mn, double* g, double* colsum,
                        double* rows_out, int64_t B, int M, double eps, int* flags, void* ws, size_t ws_bytes,
                        hipStream_t s, const sk_xchg* xc = nullptr);
// rc_sk_sweep with the row sums leaving / arriving through `xc` (version-2 sweep only; nullptr = rc_sk_sweep)
int rc_sk_sweep_x(rc_handle_t h, const float* d, const double* rows_prev, int G, double* f2, double* g, double* colsum,
                  double* rows_out, int64_t B, int M, double eps, int t, int* flags, void* ws, size_t ws_bytes,
                  hipStream_t s, const sk_xchg* xc);
bool rc_sk_xchg_capable();                    // the version-2 sweep is selected (RC_SK_V1 unset)
int rc_sk_argmax_strided(rc_handle_t h, const float* d, const double* rows_prev, int G, const double* f2, int64_t B,
                         int M, double eps, int t, int code_stride, int m_offset, uint8_t* codes_u8,
                         int64_t* codes_i64, int* flags, hipStream_t s);

// handle-owned device scratch of at least `bytes` bytes (grown with hipMalloc when too small: the old block is freed
// after a device synchronise); nullptr on allocation failure.  For entry points whose ABI has no workspace argument.
void* rc_scratch(rc_handle_t h, size_t bytes);

// device pointer to the table 2^(j/2^tb), j < 2^tb (tb = 8, 11 or 12), created on first use
const double* rc_exp2_table(rc_handle_t h, int tb);

// Record a start / stop event around one launch of a profiled kernel class (no-ops unless
// rc_profile_enable(h, 1)).  Events are recorded on the stream the kernel is launched on.
void rc_prof_mark(rc_handle_t h, int slot, hipStream_t s);
// Profile mode 2 ("bracket"): ONE (start, stop) pair around a whole run of `launches` back-to-back launches of a kernel
// class (the sweeps of a solve, eager or replayed from the hipGraph): two event records per solve instead of two per
// launch.  Call with open = true before the run and open = false after it.  No-op unless rc_profile_enable(h, 2).
void rc_prof_bracket(rc_handle_t h, int slot, hipStream_t s, bool open, int launches);

#define RC_K 256  // centroids per sub-quantiser (reference asserts MCQ_K == 256)

#define RC_HIP_CHECK(h, expr)                                   \
    do {                                                        \
        hipError_t _e = (expr);                                 \
        if (_e != hipSuccess) {                                 \
            if (h) (h)->last_hip_error = (int)_e;               \
            return RC_EHIP;                                     \
        }                                                       \
    } while (0)

#define RC_LAUNCH_CHECK(h) RC_HIP_CHECK(h, hipGetLastError())

static inline bool rc_dsub_supported(int dsub) {
    return dsub == 8 || dsub == 12 || dsub == 16 || dsub == 24 || dsub == 32 || dsub == 48 ||
           dsub == 64 || dsub == 96;
}

static inline size_t rc_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Development / test switches (RC_SK_FKLDS, RC_SK_CPB, RC_FUSE_CENTRE, RC_DIST_SPLIT, RC_ADC_*) are read from the
// environment on EVERY call, never cached: a test that sets one between two calls gets what it asked for.
#include <stdlib.h>
static inline int rc_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
static inline bool rc_env_set(const char* name) {
    const char* e = getenv(name);
    return e && *e;
}

// dispatch a template<int DSUB> callable over the supported sub-vector widths
#define RC_DISPATCH_DSUB(dsub, ...)             \
    switch (dsub) {                               \
        case 8:  { constexpr int DSUB = 8;  __VA_ARGS__; } break;  \
        case 12: { constexpr int DSUB = 12; __VA_ARGS__; } break;  \
        case 16: { constexpr int DSUB = 16; __VA_ARGS__; } break;  \
        case 24: { constexpr int DSUB = 24; __VA_ARGS__; } break;  \
        case 32: { constexpr int DSUB = 32; __VA_ARGS__; } break;  \
        case 48: { constexpr int DSUB = 48; __VA_ARGS__; } break;  \
        case 64: { constexpr int DSUB = 64; __VA_ARGS__; } break;  \
        case 96: { constexpr int DSUB = 96; __VA_ARGS__; } break;  \
        default: return RC_ESHAPE;                \
    }

// Every entry point runs on its handle's device whatever the calling thread's current device is (in-process multi-GPU
// indexes drive several handles from several host threads) and restores the caller's device on return.
struct rc_device_guard {
    int prev;
    bool switched;
    explicit rc_device_guard(rc_handle_t h) : prev(-1), switched(false) {
        if (h && hipGetDevice(&prev) == hipSuccess && prev != h->device) switched = hipSetDevice(h->device) == hipSuccess;
    }
    ~rc_device_guard() {
        if (switched) (void)hipSetDevice(prev);
    }
    rc_device_guard(const rc_device_guard&) = delete;
    rc_device_guard& operator=(const rc_device_guard&) = delete;
};

// ---- wave64 DPP helpers -------------------------------------------------------------------
// rotate right by N lanes inside each row of 16 lanes (DPP row_ror:N, ctrl 0x120+N)
template <int N>
__device__ __forceinline__ int rc_dpp_row_ror(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xF, 0xF, false);
}
template <int N>
__device__ __forceinline__ double rc_dpp_row_ror(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = rc_dpp_row_ror<N>(lo);
    hi = rc_dpp_row_ror<N>(hi);
    return __hiloint2double(hi, lo);
}
// all-reduce (sum) of a double over each row of 16 lanes.  With power-of-two rotations lane i
// computes ((x_i+x_{i+8})+(x_{i+4}+x_{i+12}))+…: the same balanced tree on every lane up to
// commutation of each addition, so all 16 lanes end with identical bits.
__device__ __forceinline__ double rc_row16_allreduce_sum(double v) {
    v += rc_dpp_row_ror<8>(v);
    v += rc_dpp_row_ror<4>(v);
    v += rc_dpp_row_ror<2>(v);
    v += rc_dpp_row_ror<1>(v);
    return v;
}

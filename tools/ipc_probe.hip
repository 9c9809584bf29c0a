// Feasibility probe for the IPC exchange layer (rc_comm_init_ipc): G PROCESSES sharing ONE GPU (or one per GPU when
// several are visible) map each other's receive buffers with hipIpcGetMemHandle / hipIpcOpenMemHandle and run T rounds
// of "store my slice into every peer's slot, signal, wait for all peers" with
//   mode A: signal = device atomic on the peer's counter from the copy kernel, wait = hipStreamWaitValue64(GTE) + reset
//   mode B: signal = hipStreamWriteValue64 to a per-sender flag word, wait = one hipStreamBatchMemOp of G-1 waits
// eagerly and (if the runtime lets them be captured) from a hipGraph.  Prints per-round latency and data checks.
//
//   hipcc --offload-arch=gfx950 -O2 -o ipc_probe tools/ipc_probe.hip && ./ipc_probe 2
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[r%d] %s:%d %s -> %s\n", g_rank, __FILE__, __LINE__, #x, hipGetErrorString(e_)); fflush(stdout); return 1; } } while (0)
static int g_rank = -1;

constexpr int MAXG = 8;
struct shm_block {
    std::atomic<int> arrive, sense;
    hipIpcMemHandle_t handle[MAXG];
    int fail[MAXG];
};

static void barrier(shm_block* s, int G) {
    const int my = s->sense.load();
    if (s->arrive.fetch_add(1) == G - 1) { s->arrive.store(0); s->sense.store(my ^ 1); }
    else { int spins = 0; while (s->sense.load() == my) { if (++spins > 1000) usleep(50); } }
}

// layout of a rank's receive buffer: [2 parities][G slots][PAY doubles] | counters[2] (u64) | flags[2][G] (u64)
constexpr int PAY = 24 * 256;   // one chain's [M/2, K] fp64 row sums at M = 48: 49 152 B
struct peers { double* buf[MAXG]; };

__global__ void fill_kernel(double* src, int rank, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < PAY) src[i] = (double)(rank + 1) * 1000.0 + t + i * 1e-6;
}
// block p stores this rank's slice into peer p's slot `rank`; SIGNAL: then bumps peer p's counter[par]
template <bool SIGNAL>
__global__ void exchange_kernel(const double* __restrict__ src, peers P, int G, int rank, int par) {
    const int p = blockIdx.x;
    double* dst = P.buf[p] + ((size_t)par * G + rank) * PAY;
    for (int i = threadIdx.x; i < PAY; i += blockDim.x) __builtin_nontemporal_store(src[i], dst + i);
    if (SIGNAL) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long* cnt = (unsigned long long*)(P.buf[p] + (size_t)2 * G * PAY) + par;
            __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// mode C: the wait is an ordinary one-thread kernel (capturable like any kernel node): spin until all G signals of
// this parity have arrived, then re-arm the counter
__global__ void wait_kernel(unsigned long long* cnt, unsigned long long want) {
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) __builtin_amdgcn_s_sleep(2);
    __hip_atomic_store(cnt, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void check_kernel(const double* mine, int G, int par, int t, int* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PAY) return;
    for (int r = 0; r < G; ++r) {
        const double want = (double)(r + 1) * 1000.0 + t + i * 1e-6;
        if (mine[((size_t)par * G + r) * PAY + i] != want) atomicAdd(bad, 1);
    }
}

static int run_rank(int rank, int G, const char* shm_name) {
    g_rank = rank;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    int fd = shm_open(shm_name, O_RDWR, 0600);
    if (fd < 0) { perror("shm_open"); return 1; }
    shm_block* S = (shm_block*)mmap(nullptr, sizeof(shm_block), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    const int dev = ndev >= G ? rank : 0;
    CK(hipSetDevice(dev));
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, dev);
    const size_t bytes = (size_t)2 * G * PAY * sizeof(double) + 4096;
    double* mine = nullptr;
    const char* kind = getenv("PROBE_ALLOC");
    if (kind && !strcmp(kind, "uncached")) CK(hipExtMallocWithFlags((void**)&mine, bytes, hipDeviceMallocUncached));
    else if (kind && !strcmp(kind, "fine")) CK(hipExtMallocWithFlags((void**)&mine, bytes, hipDeviceMallocFinegrained));
    else CK(hipMalloc((void**)&mine, bytes));
    CK(hipMemset(mine, 0, bytes));
    CK(hipDeviceSynchronize());
    CK(hipIpcGetMemHandle(&S->handle[rank], mine));
    barrier(S, G);
    peers P;
    for (int p = 0; p < G; ++p) {
        if (p == rank) { P.buf[p] = mine; continue; }
        void* q = nullptr;
        CK(hipIpcOpenMemHandle(&q, S->handle[p], hipIpcMemLazyEnablePeerAccess));
        P.buf[p] = (double*)q;
    }
    if (rank == 0) printf("G=%d devices=%d can_wait_value=%d alloc=%s: IPC handles mapped\n", G, ndev, can_wait, kind ? kind : "plain");
    barrier(S, G);

    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double* src; int* bad;
    CK(hipMalloc((void**)&src, PAY * sizeof(double)));
    CK(hipMalloc((void**)&bad, sizeof(int)));
    CK(hipMemset(bad, 0, sizeof(int)));
    unsigned long long* cnt = (unsigned long long*)(mine + (size_t)2 * G * PAY);
    unsigned long long* flg = cnt + 2;   // [2][G]
    const int T = 200;
    int h_bad = 0;

    auto round_A = [&](int t) -> int {
        const int par = t & 1;
        hipLaunchKernelGGL(fill_kernel, dim3((PAY + 255) / 256), dim3(256), 0, st, src, rank, t);
        hipLaunchKernelGGL(exchange_kernel<true>, dim3(G), dim3(256), 0, st, src, P, G, rank, par);
        CK(hipStreamWaitValue64(st, cnt + par, (uint64_t)G, hipStreamWaitValueGte, ~0ull));
        CK(hipStreamWriteValue64(st, cnt + par, 0, 0));
        hipLaunchKernelGGL(check_kernel, dim3((PAY + 255) / 256), dim3(256), 0, st, mine, G, par, t, bad);
        return 0;
    };
    auto round_C = [&](int t) -> int {
        const int par = t & 1;
        hipLaunchKernelGGL(fill_kernel, dim3((PAY + 255) / 256), dim3(256), 0, st, src, rank, t);
        hipLaunchKernelGGL(exchange_kernel<true>, dim3(G), dim3(256), 0, st, src, P, G, rank, par);
        hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(1), 0, st, cnt + par, (unsigned long long)G);
        hipLaunchKernelGGL(check_kernel, dim3((PAY + 255) / 256), dim3(256), 0, st, mine, G, par, t, bad);
        return 0;
    };
    auto round_B = [&](int t) -> int {
        const int par = t & 1;
        hipLaunchKernelGGL(fill_kernel, dim3((PAY + 255) / 256), dim3(256), 0, st, src, rank, t);
        hipLaunchKernelGGL(exchange_kernel<false>, dim3(G), dim3(256), 0, st, src, P, G, rank, par);
        hipStreamBatchMemOpParams ops[2 * MAXG];
        int n = 0;
        for (int p = 0; p < G; ++p) {      // tell every peer (and myself) that my slice of round t is there
            memset(&ops[n], 0, sizeof(ops[n]));
            ops[n].writeValue.operation = hipStreamMemOpWriteValue64;
            ops[n].writeValue.address = (hipDeviceptr_t)((unsigned long long*)(P.buf[p] + (size_t)2 * G * PAY) + 2 + (size_t)par * G + rank);
            ops[n].writeValue.value64 = (uint64_t)(t + 1);
            ++n;
        }
        CK(hipStreamBatchMemOp(st, n, ops, 0));
        n = 0;
        for (int p = 0; p < G; ++p) {
            memset(&ops[n], 0, sizeof(ops[n]));
            ops[n].waitValue.operation = hipStreamMemOpWaitValue64;
            ops[n].waitValue.address = (hipDeviceptr_t)(flg + (size_t)par * G + p);
            ops[n].waitValue.value64 = (uint64_t)(t + 1);
            ops[n].waitValue.flags = hipStreamWaitValueGte;
            ++n;
        }
        CK(hipStreamBatchMemOp(st, n, ops, 0));
        hipLaunchKernelGGL(check_kernel, dim3((PAY + 255) / 256), dim3(256), 0, st, mine, G, par, t, bad);
        return 0;
    };

    const char* modes = getenv("PROBE_MODES") ? getenv("PROBE_MODES") : "AB";
    const bool try_graph = getenv("PROBE_GRAPH") && atoi(getenv("PROBE_GRAPH"));
    for (const char* m = modes; *m; ++m) {
        // ---- eager
        CK(hipMemset(cnt, 0, 4096 - 0));
        CK(hipDeviceSynchronize());
        barrier(S, G);
        auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < T; ++t) { if ((*m == 'A' ? round_A(t) : *m == 'C' ? round_C(t) : round_B(t)) != 0) return 1; }
        CK(hipStreamSynchronize(st));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipMemcpy(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost));
        barrier(S, G);
        printf("[r%d] mode %c eager : %7.2f us per round, bad=%d\n", rank, *m, std::chrono::duration<double, std::micro>(t1 - t0).count() / T, h_bad);
        fflush(stdout);
        if (!try_graph) continue;
        // ---- captured: rounds 0 .. T-1 as one graph (mode B's flag values are absolute, so reset flags before a replay)
        if (rank == 0) printf("mode %c: capture ...\n", *m);
        CK(hipMemset(cnt, 0, 4096));
        CK(hipDeviceSynchronize());
        barrier(S, G);
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
        int crc = 0;
        if (ok) {
            for (int t = 0; t < T && crc == 0; ++t) crc = (*m == 'A' ? round_A(t) : *m == 'C' ? round_C(t) : round_B(t));
            if (rank == 0) printf("mode %c: enqueued into the capture (crc=%d), ending capture\n", *m, crc);
            hipError_t e = hipStreamEndCapture(st, &graph);
            ok = crc == 0 && e == hipSuccess && graph;
            if (!ok) printf("[r%d] mode %c capture failed: crc=%d end=%s\n", rank, *m, crc, hipGetErrorString(e));
        }
        if (rank == 0) printf("mode %c: capture ok=%d, instantiating\n", *m, (int)ok);
        if (ok) { hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0); ok = e == hipSuccess; if (!ok) printf("[r%d] instantiate: %s\n", rank, hipGetErrorString(e)); }
        (void)hipGetLastError();
        S->fail[rank] = ok ? 0 : 1;
        barrier(S, G);
        bool all_ok = true;
        for (int p = 0; p < G; ++p) all_ok = all_ok && !S->fail[p];
        if (all_ok) {
            for (int rep = 0; rep < 3; ++rep) {
                if (*m == 'B') { CK(hipMemset(cnt, 0, 4096)); CK(hipDeviceSynchronize()); }
                barrier(S, G);
                if (rank == 0) printf("mode %c: graph launch %d\n", *m, rep);
                t0 = std::chrono::steady_clock::now();
                CK(hipGraphLaunch(exec, st));
                CK(hipStreamSynchronize(st));
                t1 = std::chrono::steady_clock::now();
                CK(hipMemcpy(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost));
                barrier(S, G);
                if (rep) printf("[r%d] mode %c graph : %7.2f us per round, bad=%d\n", rank, *m, std::chrono::duration<double, std::micro>(t1 - t0).count() / T, h_bad);
            }
        } else if (rank == 0) printf("mode %c: graph path unavailable\n", *m);
        fflush(stdout);
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        barrier(S, G);
    }
    for (int p = 0; p < G; ++p) if (p != rank) (void)hipIpcCloseMemHandle(P.buf[p]);
    barrier(S, G);
    (void)hipFree(mine);
    return h_bad != 0;
}

int main(int argc, char** argv) {
    if (argc >= 4) return run_rank(atoi(argv[1]), atoi(argv[2]), argv[3]);
    const int G = argc >= 2 ? atoi(argv[1]) : 2;
    if (G < 1 || G > MAXG) return 2;
    char name[64];
    snprintf(name, sizeof name, "/rc_ipc_probe_%d", (int)getpid());
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(shm_block)) != 0) { perror("shm"); return 2; }
    shm_block* S = (shm_block*)mmap(nullptr, sizeof(shm_block), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    memset((void*)S, 0, sizeof(shm_block));
    pid_t pids[MAXG];
    for (int r = 0; r < G; ++r) {
        pids[r] = fork();
        if (pids[r] == 0) {
            char a[16], b[16];
            snprintf(a, sizeof a, "%d", r); snprintf(b, sizeof b, "%d", G);
            execl(argv[0], argv[0], a, b, name, (char*)nullptr);
            _exit(127);
        }
    }
    const int limit_s = getenv("PROBE_TIMEOUT") ? atoi(getenv("PROBE_TIMEOUT")) : 60;
    int rc = 0, live = G;
    for (int waited = 0; live > 0 && waited < limit_s * 10; ++waited) {
        for (int r = 0; r < G; ++r) {
            if (pids[r] <= 0) continue;
            int st = 0;
            if (waitpid(pids[r], &st, WNOHANG) == pids[r]) {
                if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { rc = 1; printf("rank %d exited abnormally (status %d)\n", r, st); }
                pids[r] = -1; --live;
            }
        }
        if (rc) break;
        usleep(100000);
    }
    if (live > 0) {
        printf("TIMEOUT or failure: killing %d ranks\n", live);
        for (int r = 0; r < G; ++r) if (pids[r] > 0) { kill(pids[r], SIGKILL); waitpid(pids[r], nullptr, 0); }
        rc = 1;
    }
    shm_unlink(name);
    printf("probe G=%d %s\n", G, rc ? "FAILED" : "OK");
    return rc;
}

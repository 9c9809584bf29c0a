"""Tensor-level wrappers over the C ABI (include/repconc_hip.h).

Every function takes CUDA(=HIP) torch tensors, launches on `torch.cuda.current_stream()` of the
tensor's device and returns without synchronising.  torch is used for memory and streams only;
all arithmetic happens in librepconc_hip.so.  CPU tensors are rejected: there is no fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib

K = 256
SUPPORTED_DSUB = (8, 12, 16, 24, 32, 48, 64, 96)


def _need_cuda(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.RepconcHipError(
                "repconc_amd runs on the GPU only (got a CPU tensor); the package has no CPU fallback")


def _ctx(t: torch.Tensor):
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    return _lib.load(), _lib.handle(dev), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), dev


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _rows_f32(x: torch.Tensor) -> torch.Tensor:
    """fp32 [B, D] with unit inner stride and 16-byte aligned rows (fp16/bf16 inputs are promoted
    exactly like the reference's fp32 centroids promote them)."""
    if x.dim() != 2:
        raise ValueError("expected a [B, D] matrix")
    if x.dtype != torch.float32:
        x = x.float()
    if x.stride(1) != 1 or x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0:
        x = x.contiguous()
    return x


def _centroids(c: torch.Tensor) -> torch.Tensor:
    if c.dim() != 3 or c.shape[1] != K:
        raise ValueError("centroids must be [M, 256, dsub]")
    c = c.detach()
    if c.dtype != torch.float32 or not c.is_contiguous():
        c = c.float().contiguous()
    return c


def _shape(x: torch.Tensor, c: torch.Tensor):
    B, D = x.shape
    M, _, dsub = c.shape
    if D != M * dsub:
        raise ValueError(f"embedding width {D} != M*dsub = {M}*{dsub}")
    # every width is served: SUPPORTED_DSUB are the widths with specialised kernels (the recipes'), any other divisor goes
    # through the run-time-width kernels with the same arithmetic (csrc/pq_distance.hip, "any width")
    return B, D, M, dsub


def _code_dtype(codes: torch.Tensor) -> int:
    if codes.dtype == torch.uint8:
        return _lib.RC_CODE_U8
    if codes.dtype == torch.int64:
        return _lib.RC_CODE_I64
    raise ValueError("codes must be uint8 or int64")


# --------------------------------------------------------------------------- nearest codes
def assign_nearest(x: torch.Tensor, centroids: torch.Tensor, dtype=torch.int64, method: str = "auto",
                   stats: dict | None = None) -> torch.Tensor:
    """argmin_k ||x_m - C[m,k]||^2 -> codes [B, M].  modeling_repconc.py:49-52,66.

    method: "exact" = every distance in the reference's fp32 order (rc_pq_assign_nearest); "mfma" = matrix-core screen
    + exact recomputation of the doubtful pairs (rc_pq_assign_nearest_fast) — the same codes, bit for bit; "auto" =
    mfma when its alignment preconditions hold.  stats (optional dict) receives {"method", "doubtful"}.
    """
    _need_cuda(x, centroids)
    x, c = _rows_f32(x), _centroids(centroids)
    B, D, M, _ = _shape(x, c)
    lib, h, s, _ = _ctx(x)
    codes = torch.empty((B, M), dtype=dtype, device=x.device)
    u8 = codes if dtype == torch.uint8 else None
    i64 = codes if dtype == torch.int64 else None
    if u8 is None and i64 is None:
        raise ValueError("dtype must be torch.uint8 or torch.int64")
    if method not in ("auto", "exact", "mfma"):
        raise ValueError("method must be auto|exact|mfma")
    if B == 0:
        return codes
    fast_ok = x.data_ptr() % 16 == 0 and x.stride(0) % 4 == 0 and B * M < 2 ** 32
    if method == "mfma" and not fast_ok:
        raise _lib.RepconcHipError("mfma assignment needs 16-byte aligned rows, ldx % 4 == 0 and B*M < 2^32")
    if method != "exact" and fast_ok:
        n = lib.rc_pq_assign_nearest_fast_ws_bytes(B, M)
        ws = torch.empty(n, dtype=torch.uint8, device=x.device)
        _lib.check(lib.rc_pq_assign_nearest_fast(h, _p(x), x.stride(0), _p(c), B, D, M, K, _p(u8), _p(i64), _p(ws), n, s),
                   "rc_pq_assign_nearest_fast", h)
        if stats is not None:                               # statistics only: this (and nothing else) synchronises
            doubtful = C.c_int(0)
            over = lib.rc_pq_assign_nearest_fast_overflow(h, _p(ws), B, M, C.byref(doubtful))
            if over < 0:
                _lib.check(over, "rc_pq_assign_nearest_fast_overflow", h)
            stats.update(method="mfma", doubtful=int(doubtful.value), overflow=bool(over))
        return codes
    _lib.check(lib.rc_pq_assign_nearest(h, _p(x), x.stride(0), _p(c), B, D, M, K, _p(u8), _p(i64), s),
               "rc_pq_assign_nearest", h)
    if stats is not None:
        stats.setdefault("doubtful", 0)
        stats["method"] = "exact"
    return codes


# --------------------------------------------------------------------------- distance table
def dist_table(x: torch.Tensor, centroids: torch.Tensor, with_minmax: bool = True):
    """d [M,B,K] fp32 (+ minmax [2M]: max then min per m).  modeling_repconc.py:50,76-77."""
    _need_cuda(x, centroids)
    x, c = _rows_f32(x), _centroids(centroids)
    B, D, M, _ = _shape(x, c)
    lib, h, s, _ = _ctx(x)
    d = torch.empty((M, B, K), dtype=torch.float32, device=x.device)
    mm = torch.empty((2 * M,), dtype=torch.float32, device=x.device) if with_minmax else None
    wsb = lib.rc_pq_dist_table_ws_bytes(B, M)
    ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=x.device)
    _lib.check(lib.rc_pq_dist_table(h, _p(x), x.stride(0), _p(c), B, D, M, K, _p(d), _p(mm), _p(ws), wsb, s),
               "rc_pq_dist_table", h)
    return d, mm


def centre_(d: torch.Tensor, minmax: torch.Tensor) -> torch.Tensor:
    """In-place center_distance_for_constraint, modeling_repconc.py:81-84."""
    _need_cuda(d, minmax)
    M, B, _ = d.shape
    lib, h, s, _ = _ctx(d)
    _lib.check(lib.rc_pq_centre(h, _p(d), _p(minmax), B, M, K, s), "rc_pq_centre", h)
    return d


# --------------------------------------------------------------------------- Sinkhorn stages
def sinkhorn_potentials_f64(out: torch.Tensor, eps: float, iters: int, allgather=None) -> torch.Tensor:
    """Row potentials f [M,K] (fp64) of `iters` Sinkhorn iterations on ANY fp64 cost tensor out [M,K,B] (this rank's columns) —
    rc_sk64_rows / rc_sk64_cols, modeling_repconc.py:137-165.  `allgather(t[M,K]) -> [G,M,K]` moves the row values between
    ranks (None: one rank).  The plan is softmax_k(out / eps + f)."""
    _need_cuda(out)
    if out.dtype != torch.float64 or out.dim() != 3 or out.shape[1] != K:
        raise _lib.RepconcHipError(f"out must be fp64 [M, {K}, B], got {out.dtype} {tuple(out.shape)}")
    out = out.contiguous()
    M, _, B = out.shape
    lib, h, s, _ = _ctx(out)
    dev = out.device
    lse = torch.empty((M, K), dtype=torch.float64, device=dev)
    f = torch.empty((M, K), dtype=torch.float64, device=dev)
    g = torch.empty((M, max(B, 1)), dtype=torch.float64, device=dev)
    gather = allgather if allgather is not None else (lambda t: t.unsqueeze(0))

    def rows(gp):
        _lib.check(lib.rc_sk64_rows(h, _p(out), _p(gp), B, M, K, float(eps), _p(lse), s), "rc_sk64_rows", h)

    def cols(want_g):
        lg = gather(lse).contiguous()
        _lib.check(lib.rc_sk64_cols(h, _p(out), _p(lg), lg.shape[0], B, M, K, float(eps), _p(f), _p(g if want_g else None), s),
                   "rc_sk64_cols", h)
    rows(None)
    for _ in range(1, int(iters)):
        cols(True)
        rows(g)
    cols(False)
    return f


class SinkhornState:
    """Device buffers of one rank's Sinkhorn solve over a centred table d [M,B,K] (staged C ABI).

        rows = st.sweep(eps, 0, None)                    # sweep 0
        for t in 1 .. T-1:  rows = st.sweep(eps, t, gathered(rows))
        codes = st.argmax(eps, T, gathered(rows))

    `gathered(rows)` is the [G,M,K] stack of every rank's row sums (G = 1: rows.unsqueeze(0))."""

    def __init__(self, d: torch.Tensor):
        _need_cuda(d)
        self.d = d
        self.M, self.B, _ = d.shape
        dev = d.device
        self.f2 = torch.empty((2, self.M, K), dtype=torch.float64, device=dev)
        self.g = torch.empty((self.M, self.B), dtype=torch.float64, device=dev)
        self.colsum = torch.empty((self.M, self.B), dtype=torch.float64, device=dev)
        self._rows = torch.empty((2, self.M, K), dtype=torch.float64, device=dev)
        self.flags = torch.zeros((1,), dtype=torch.int32, device=dev)
        lib = _lib.load()
        self._wsb = lib.rc_sk_ws_bytes(self.B, self.M, K)
        self._ws = torch.empty((max(self._wsb, 1),), dtype=torch.uint8, device=dev)

    @staticmethod
    def _gathered(rows_prev):
        if rows_prev is None:
            return None, 0
        if rows_prev.dim() == 2:
            rows_prev = rows_prev.unsqueeze(0)
        return rows_prev.contiguous(), rows_prev.shape[0]

    def sweep(self, eps: float, t: int, rows_prev: Optional[torch.Tensor]) -> torch.Tensor:
        """Sweep t; returns this rank's row sums [M,K] (a buffer that stays valid until sweep t+2)."""
        lib, h, s, _ = _ctx(self.d)
        rp, G = self._gathered(rows_prev)
        out = self._rows[t & 1]
        _lib.check(lib.rc_sk_sweep(h, _p(self.d), _p(rp), G, _p(self.f2), _p(self.g), _p(self.colsum), _p(out),
                                   self.B, self.M, K, float(eps), int(t), _p(self.flags), _p(self._ws), self._wsb, s),
                   "rc_sk_sweep", h)
        return out

    def argmax(self, eps: float, t: int, rows_prev: torch.Tensor, dtype=torch.int64) -> torch.Tensor:
        lib, h, s, _ = _ctx(self.d)
        rp, G = self._gathered(rows_prev)
        codes = torch.empty((self.B, self.M), dtype=dtype, device=self.d.device)
        u8 = codes if dtype == torch.uint8 else None
        i64 = codes if dtype == torch.int64 else None
        _lib.check(lib.rc_sk_argmax(h, _p(self.d), _p(rp), G, _p(self.f2), self.B, self.M, K, float(eps), int(t),
                                    _p(u8), _p(i64), _p(self.flags), s), "rc_sk_argmax", h)
        return codes

    def potentials(self, t: int, rows_prev: torch.Tensor) -> torch.Tensor:
        """f [M,K] after t row normalisations (what rc_sk_argmax(t) uses); bookkeeping on [M,K] only."""
        rp, _ = self._gathered(rows_prev)
        tot = rp[0].clone()
        for r in range(1, rp.shape[0]):
            tot = tot + rp[r]
        prev = torch.zeros_like(tot) if t == 1 else self.f2[(t - 1) & 1]
        return prev - torch.log(tot)


def assign_sinkhorn(x: torch.Tensor, centroids: torch.Tensor, eps: float, iters: int,
                    dtype=torch.int64) -> Tuple[torch.Tensor, torch.Tensor]:
    """Single-rank constrained codes [B,M] + flags (int32[1]).  modeling_repconc.py:47-67 with
    dist.is_initialized()==False.  One C call: distance table, centring, `iters` Sinkhorn
    iterations, argmax."""
    _need_cuda(x, centroids)
    x, c = _rows_f32(x), _centroids(centroids)
    B, D, M, _ = _shape(x, c)
    lib, h, s, _ = _ctx(x)
    codes = torch.empty((B, M), dtype=dtype, device=x.device)
    flags = torch.zeros((1,), dtype=torch.int32, device=x.device)
    if B == 0:
        return codes, flags
    wsb = lib.rc_pq_assign_sinkhorn_ws_bytes(B, M, K)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
    u8 = codes if dtype == torch.uint8 else None
    i64 = codes if dtype == torch.int64 else None
    _lib.check(lib.rc_pq_assign_sinkhorn(h, _p(x), x.stride(0), _p(c), B, D, M, K, float(eps), int(iters),
                                         _p(u8), _p(i64), _p(flags), _p(ws), wsb, s), "rc_pq_assign_sinkhorn", h)
    return codes, flags


# --------------------------------------------------------------------------- N ranks, native loop
_comm_ready = {}
_dist_ws = {}          # device index -> persistent workspace of assign_sinkhorn_dist


def comm_transport() -> str:
    """RC_COMM=ipc|rccl|auto (default auto): how the ranks exchange the Sinkhorn row sums and k-means statistics.
    ipc = hand-written peer stores into IPC-mapped receive buffers (csrc/comm.hip; works between processes on ONE GPU
    too; all ranks on one node, at most 16), rccl = two RCCL communicators (one GPU per rank; any number of nodes).
    auto = ipc where it can work (one node, world <= 16), else rccl; a failed IPC connect falls back to rccl on ALL ranks."""
    import os
    v = os.environ.get("RC_COMM", "auto").lower()
    if v not in ("ipc", "rccl", "auto"):
        raise ValueError("RC_COMM must be ipc, rccl or auto")
    return v


RC_IPC_MAX_WORLD = 16
IPC_MAX_M = 128        # csrc/comm.hip: IPC_XMAX_M (sub-quantisers per chain on the IPC transport)


def _all_ranks_ok(ok: bool, group) -> bool:
    """Collective AND over the group (any backend): a set-up step either worked on every rank or is undone on every rank."""
    import torch.distributed as dist
    flags = [None] * dist.get_world_size(group)
    dist.all_gather_object(flags, bool(ok), group=group)
    return all(flags)


def _same_node(group) -> bool:
    import socket
    import torch.distributed as dist
    names = [None] * dist.get_world_size(group)
    dist.all_gather_object(names, socket.gethostname(), group=group)
    return len(set(names)) == 1


def _connect_ipc(lib, h, rank, world, group) -> bool:
    """Export + connect on every rank; False (with the handle released again) on every rank if any rank failed."""
    import torch.distributed as dist
    blob = (C.c_char * _lib.RC_IPC_BLOB_BYTES)()
    ok = lib.rc_comm_ipc_export(h, rank, world, C.cast(blob, C.c_void_p)) == _lib.RC_OK
    blobs = [None] * world
    dist.all_gather_object(blobs, bytes(blob.raw) if ok else None, group=group)
    if all(b is not None for b in blobs):
        raw = (C.c_char * (_lib.RC_IPC_BLOB_BYTES * world)).from_buffer_copy(b"".join(blobs))
        ok = lib.rc_comm_ipc_connect(h, C.cast(raw, C.c_void_p)) == _lib.RC_OK
    else:
        ok = False
    ok = _all_ranks_ok(ok, group)                           # also the barrier: every rank has mapped every buffer
    if not ok:
        lib.rc_comm_destroy(h)                              # releases an exported-but-unconnected buffer too
    return ok


def _connect_rccl(lib, h, dev, rank, world, group) -> bool:
    import torch.distributed as dist
    ids = torch.zeros(256, dtype=torch.uint8)
    if rank == 0:
        buf = (C.c_char * 256)()
        _lib.check(lib.rc_comm_unique_ids(C.cast(buf, C.c_void_p)), "rc_comm_unique_ids")
        ids = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    if world > 1:
        backend = dist.get_backend(group)
        t = ids.to(torch.device("cuda", dev)) if backend == "nccl" else ids
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ids = t.cpu()
    raw = (C.c_char * 256).from_buffer_copy(bytes(ids.numpy().tobytes()))
    ok = lib.rc_comm_init(h, C.cast(raw, C.c_void_p), rank, world) == _lib.RC_OK
    ok = _all_ranks_ok(ok, group) if world > 1 else ok
    if not ok:
        lib.rc_comm_destroy(h)
    return ok


def comm_init(group=None, transport: Optional[str] = None) -> str:
    """Set up the exchange layer of this process's handle (once per process group); returns the transport that runs.
    Only the set-up handshake goes through torch.distributed (any backend): the IPC descriptors (128 bytes per rank,
    all-gathered) or the RCCL unique ids (256 bytes, broadcast from rank 0); everything after that happens inside
    librepconc_hip.so.  COLLECTIVE: every rank of the group calls it, every step is agreed on by all ranks (a connect that
    fails on one rank is undone on all of them), and all ranks end on the same transport or all raise — no rank is left
    spinning in IPC waits while another has fallen back.  transport = "auto" (default, RC_COMM): ipc when all ranks share
    a host and world <= 16, else rccl; ipc that fails to connect falls back to rccl."""
    import torch.distributed as dist
    dev = torch.cuda.current_device()
    want = (transport or comm_transport()).lower()
    key0 = id(group) if group is not None else 0
    ready = _comm_ready.get(dev)
    if ready is not None and ready[0] == key0 and (want == "auto" or ready[1] == want):
        return ready[1]
    lib, h = _lib.load(), _lib.handle(dev)
    if dev in _comm_ready:                                  # another process group / transport: start over
        comm_destroy(group_barrier=False)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    import logging
    log = logging.getLogger(__name__)
    tried = []
    if want in ("ipc", "auto"):
        ipc_possible = world <= RC_IPC_MAX_WORLD and _same_node(group)
        if ipc_possible and _connect_ipc(lib, h, rank, world, group):
            _comm_ready[dev] = (key0, "ipc")
            log.info("comm_init: rank %d / %d on cuda:%d exchanges over the IPC transport", rank, world, dev)
            return "ipc"
        tried.append("ipc (world > 16 or ranks on several hosts)" if not ipc_possible else "ipc (export / connect failed on a rank)")
        # an EXPLICIT request for ipc (argument or RC_COMM=ipc) is strict: ranks that share one GPU cannot run RCCL at all
        # (its ncclCommInitRank would hang, not fail), so only "auto" falls back.  RC_COMM_STRICT=0 restores the fall-back.
        if want == "ipc" and rc_env_strict(default=True):
            raise _lib.RepconcHipError("comm_init: the IPC transport was requested and is unavailable: " + tried[-1] +
                                       " (RC_COMM=auto or RC_COMM_STRICT=0 fall back to rccl)")
        log.warning("comm_init: %s; falling back to rccl", tried[-1])
    if _connect_rccl(lib, h, dev, rank, world, group):
        _comm_ready[dev] = (key0, "rccl")
        log.info("comm_init: rank %d / %d on cuda:%d exchanges over RCCL", rank, world, dev)
        return "rccl"
    tried.append("rccl (rc_comm_init failed on a rank)")
    raise _lib.RepconcHipError("comm_init: no exchange transport could be set up on all ranks: " + "; ".join(tried))


def rc_env_strict(default: bool = False) -> bool:
    """RC_COMM_STRICT=1 / 0: whether an explicitly requested transport that is unavailable raises instead of falling back."""
    import os
    v = os.environ.get("RC_COMM_STRICT", "")
    return default if v == "" else v == "1"


_comm_flags = {}       # device index -> int32 flags word passed to every rc_comm_allgather of that device


def comm_check(device=None) -> None:
    """Raise if an exchange of this device's handle has timed out since comm_init() (RC_FLAG_COMM: a peer rank is missing or
    more than RC_IPC_TIMEOUT_MS behind; the transport is broken from then on and every later gather is flagged).
    Synchronises — call where the results of the gathers are consumed (the multi-rank paths of the package do)."""
    if not _comm_flags:                                     # no native gather has run (CPU / torch.distributed paths)
        return
    if isinstance(device, torch.device):
        if device.type != "cuda":
            return
        device = device.index
    dev = torch.cuda.current_device() if device is None else int(device)
    fl = _comm_flags.get(dev)
    if fl is not None and int(fl.item()) & _lib.RC_FLAG_COMM:
        raise _lib.RepconcHipError("an inter-rank exchange timed out (RC_FLAG_COMM): a peer rank is missing or more than "
                                   "RC_IPC_TIMEOUT_MS behind; the gathered data of this and every later exchange is invalid")


def comm_destroy(group=None, group_barrier: bool = True):
    """Release the exchange layer of the current device's handle.  With the IPC transport no peer may still be storing
    into this rank's buffer: `group_barrier` synchronises the device and the process group first."""
    dev = torch.cuda.current_device()
    if dev not in _comm_ready:
        return
    if group_barrier:
        import torch.distributed as dist
        torch.cuda.synchronize(dev)
        if dist.is_initialized():
            dist.barrier(group=group)
    lib, h = _lib.load(), _lib.handle(dev)
    _lib.check(lib.rc_comm_destroy(h), "rc_comm_destroy", h)
    del _comm_ready[dev]
    _comm_flags.pop(dev, None)


def comm_allgather(t: torch.Tensor, flags: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[world, *t.shape] stack of every rank's `t` (same shape and dtype on all ranks) through the handle's exchange
    layer (rc_comm_allgather; comm_init() first), on the current stream."""
    _need_cuda(t)
    t = t.contiguous()
    lib, h, s, _ = _ctx(t)
    world = lib.rc_comm_world(h)
    if world < 1:
        raise _lib.RepconcHipError("comm_allgather: call ops.comm_init() first")
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if flags is None:                                       # the device's own flags word: comm_check() reads it
        dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
        flags = _comm_flags.get(dev)
        if flags is None:
            flags = _comm_flags[dev] = torch.zeros((1,), dtype=torch.int32, device=t.device)
    _lib.check(lib.rc_comm_allgather(h, _p(t), _p(out), t.numel() * t.element_size(), _p(flags), s), "rc_comm_allgather", h)
    return out


def all_gather(t: torch.Tensor, group=None) -> torch.Tensor:
    """[world, *t.shape] stack of every rank's `t`: through the handle's own exchange layer when comm_init() has been
    called for this process group on t's device (IPC peer stores or RCCL, from C), otherwise through torch.distributed.
    The ONE all-gather the multi-rank code paths (k-means statistics run_warmup.py:102-113, row-sharded / replicated
    search results evaluate_repconc.py:121-135) are written against."""
    import torch.distributed as dist
    if t.is_cuda:
        dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
        ready = _comm_ready.get(dev)
        if ready is not None and ready[0] == (id(group) if group is not None else 0):
            return comm_allgather(t)
    G = dist.get_world_size(group)
    out = torch.empty((G,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=group)
    return out


def assign_sinkhorn_dist(x: torch.Tensor, centroids: torch.Tensor, eps: float, iters: int, dtype=torch.int64):
    """This rank's codes of the batch-sharded constrained assignment, the whole solve in one C call
    (rc_pq_assign_sinkhorn_dist; comm_init() first).  modeling_repconc.py:47-67 with dist.is_initialized()."""
    _need_cuda(x, centroids)
    x, c = _rows_f32(x), _centroids(centroids)
    B, D, M, _ = _shape(x, c)
    lib, h, s, _ = _ctx(x)
    world = lib.rc_comm_world(h)
    if world < 1:
        raise _lib.RepconcHipError("assign_sinkhorn_dist: call ops.comm_init() first")
    if lib.rc_comm_kind(h) == 2 and M > IPC_MAX_M * lib.rc_solve_num_chains_on(h, world, M):
        raise _lib.RepconcHipError(f"assign_sinkhorn_dist: MCQ_M = {M} exceeds what the IPC transport exchanges per chain "
                                   f"({IPC_MAX_M} sub-quantisers: [M, 256] fp64 row sums through a 256 KiB slot); "
                                   "set RC_COMM=rccl for this model")
    codes = torch.empty((B, M), dtype=dtype, device=x.device)
    flags = torch.zeros((1,), dtype=torch.int32, device=x.device)
    # a rank without rows (ragged last batch) still takes part in every collective of the solve
    wsb = lib.rc_pq_assign_sinkhorn_dist_ws_bytes(B, M, K, world)
    # The iteration graph (csrc/comm.hip) is cached per workspace ADDRESS: keep one block per device and reuse it while
    # it is large enough, instead of asking the caching allocator for a (possibly different) block on every call.  The
    # block is only touched by this function, on the caller's stream, so consecutive solves are ordered by the stream.
    dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
    ws = _dist_ws.get(dev)
    if ws is None or ws.numel() < wsb:
        _dist_ws.pop(dev, None)
        ws = _dist_ws[dev] = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
    u8 = codes if dtype == torch.uint8 else None
    i64 = codes if dtype == torch.int64 else None
    _lib.check(lib.rc_pq_assign_sinkhorn_dist(h, _p(x), x.stride(0), _p(c), B, D, M, K, float(eps), int(iters),
                                              _p(u8), _p(i64), _p(flags), _p(ws), wsb, s), "rc_pq_assign_sinkhorn_dist", h)
    return codes, flags


# --------------------------------------------------------------------------- decode
def decode_raw(codes: torch.Tensor, centroids: torch.Tensor) -> torch.Tensor:
    _need_cuda(codes, centroids)
    c = _centroids(centroids)
    M, _, dsub = c.shape
    if codes.dim() != 2 or codes.shape[1] != M:
        raise ValueError("codes must be [n, M]")
    codes = codes.contiguous()
    n = codes.shape[0]
    out = torch.empty((n, M * dsub), dtype=torch.float32, device=codes.device)
    lib, h, s, _ = _ctx(codes)
    _lib.check(lib.rc_pq_decode(h, _p(codes), _code_dtype(codes), _p(c), n, M, K, dsub, _p(out), s),
               "rc_pq_decode", h)
    return out


class _DecodeFn(torch.autograd.Function):
    """decode with the scatter-add gradient into the centroids (autograd of the gather at
    modeling_repconc.py:175)."""

    @staticmethod
    def forward(ctx, codes, centroids):
        ctx.save_for_backward(codes)
        ctx.cshape = tuple(centroids.shape)
        ctx.cdtype = centroids.dtype
        return decode_raw(codes, centroids)

    @staticmethod
    def backward(ctx, grad_out):
        (codes,) = ctx.saved_tensors
        M, _, dsub = ctx.cshape
        go = grad_out.float().contiguous()
        gC = torch.zeros(ctx.cshape, dtype=torch.float32, device=go.device)
        codes = codes.contiguous()
        lib, h, s, _ = _ctx(go)
        _lib.check(lib.rc_pq_decode_bwd(h, _p(codes), _code_dtype(codes), _p(go), codes.shape[0], M, K, dsub,
                                        _p(gC), s), "rc_pq_decode_bwd", h)
        return None, gC.to(ctx.cdtype)


def decode(codes: torch.Tensor, centroids: torch.Tensor) -> torch.Tensor:
    if centroids.requires_grad and torch.is_grad_enabled():
        return _DecodeFn.apply(codes, centroids)
    return decode_raw(codes, centroids)


# --------------------------------------------------------------------------- small ops
def normalize_centroids_(centroids: torch.Tensor) -> torch.Tensor:
    _need_cuda(centroids)
    if centroids.dtype != torch.float32 or not centroids.is_contiguous():
        raise ValueError("centroids must be contiguous fp32")
    M, Kc, dsub = centroids.shape
    lib, h, s, _ = _ctx(centroids)
    _lib.check(lib.rc_normalize_centroids(h, _p(centroids), M, Kc, dsub, s), "rc_normalize_centroids", h)
    return centroids


def code_hist(codes: torch.Tensor) -> torch.Tensor:
    """hist [M, 256] int32.  finetune_repconc.py:588-592 for every sub-quantiser."""
    _need_cuda(codes)
    codes = codes.contiguous()
    n, M = codes.shape
    hist = torch.empty((M, K), dtype=torch.int32, device=codes.device)
    lib, h, s, _ = _ctx(codes)
    _lib.check(lib.rc_code_hist(h, _p(codes), _code_dtype(codes), n, M, K, _p(hist), s), "rc_code_hist", h)
    return hist


def kmeans_stats(x: torch.Tensor, codes: torch.Tensor, sums: Optional[torch.Tensor] = None,
                 counts: Optional[torch.Tensor] = None):
    """Accumulate Lloyd sufficient statistics (sums [M,K,dsub] fp64, counts [M,K] int64)."""
    _need_cuda(x, codes)
    x = _rows_f32(x)
    n, D = x.shape
    M = codes.shape[1]
    if codes.dtype != torch.uint8:
        codes = codes.to(torch.uint8)
    codes = codes.contiguous()
    dsub = D // M
    if sums is None:
        sums = torch.zeros((M, K, dsub), dtype=torch.float64, device=x.device)
    if counts is None:
        counts = torch.zeros((M, K), dtype=torch.int64, device=x.device)
    lib, h, s, _ = _ctx(x)
    _lib.check(lib.rc_kmeans_stats(h, _p(x), x.stride(0), _p(codes), n, D, M, K, _p(sums), _p(counts), s),
               "rc_kmeans_stats", h)
    return sums, counts


def kmeans_update_(sums: torch.Tensor, counts: torch.Tensor, centroids: torch.Tensor) -> torch.Tensor:
    _need_cuda(sums, counts, centroids)
    M, Kc, dsub = centroids.shape
    lib, h, s, _ = _ctx(centroids)
    _lib.check(lib.rc_kmeans_update(h, _p(sums), _p(counts), _p(centroids), M, Kc, dsub, s), "rc_kmeans_update", h)
    return centroids


def kmeans_split_empty_(centroids: torch.Tensor, counts: torch.Tensor, nsplit: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Faiss's empty-cluster rule (`split_clusters`) applied in place on the device; see include/repconc_hip.h."""
    _need_cuda(centroids, counts, nsplit)
    if centroids.dtype != torch.float32 or not centroids.is_contiguous() or counts.dtype != torch.int64 or not counts.is_contiguous():
        raise ValueError("centroids must be contiguous fp32 [M,256,dsub], counts contiguous int64 [M,256]")
    M, Kc, dsub = centroids.shape
    lib, h, s, _ = _ctx(centroids)
    _lib.check(lib.rc_kmeans_split_empty(h, _p(centroids), _p(counts), M, Kc, dsub, _p(nsplit), s), "rc_kmeans_split_empty", h)
    return centroids


# --------------------------------------------------------------------------- ADC search
def adc_lut(centroids: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    _need_cuda(centroids, q)
    c, q = _centroids(centroids), _rows_f32(q).contiguous()
    M, _, dsub = c.shape
    nq, D = q.shape
    lut = torch.empty((nq, M, K), dtype=torch.float32, device=q.device)
    lib, h, s, _ = _ctx(q)
    _lib.check(lib.rc_adc_lut(h, _p(c), _p(q), nq, D, M, K, _p(lut), s), "rc_adc_lut", h)
    return lut


def adc_image_row_bytes(M: int) -> int:
    """Bytes per row of the permuted code image the ADC screen of this M streams (rc_adc_scan_image); 0 = none."""
    return int(_lib.load().rc_adc_scan_image_bytes(1 << 20, int(M))) >> 20


def adc_image_bytes(N: int, M: int) -> int:
    """Bytes of the FLAT-SEARCH image of an N-row index (rc_adc_scan_image_bytes): N * row bytes, rounded up to whole
    storage tiles for the M whose image is tile-blocked (M = 96: 32768-row tiles, phase-major inside a tile)."""
    return int(_lib.load().rc_adc_scan_image_bytes(int(N), int(M)))


def adc_image_supported(M: int) -> bool:
    return adc_image_row_bytes(M) > 0


def adc_image_rows_bytes(N: int, M: int) -> int:
    """Bytes of the IVF search's image (layout "rows") of an N-row index: whole chunks of 16 rows."""
    return int(_lib.load().rc_adc_scan_image_rows_bytes(int(N), int(M)))


def adc_image_rows_at(M: int, n: int, m: int) -> int:
    """Byte offset of codes[n][m] in the IVF search's image (host-side description, no GPU involved)."""
    return int(_lib.load().rc_adc_scan_image_rows_at(int(M), int(n), int(m)))


def adc_image_rows16_at(M: int, n: int, m: int) -> int:
    """Byte offset of codes[n][m] in the image of the 16-query IVF screen (layout "rows16"; host-side description)."""
    return int(_lib.load().rc_adc_scan_image_rows16_at(int(M), int(n), int(m)))


def adc_scan_image_(codes: torch.Tensor, image: torch.Tensor, n0: int = 0, n: Optional[int] = None,
                    layout: str = "flat") -> torch.Tensor:
    """(Re)build rows [n0, n0+n) of the permuted code image of an index: codes uint8 [>=n0+n, M] contiguous; image a
    contiguous uint8 buffer of at least adc_image_bytes(n0+n, M) bytes (layout "flat": what `adc_search` takes — row-major
    for M in {16,32,48,64}, tile-blocked for M = 96) or adc_image_rows_bytes(n0+n, M) bytes (layout "rows": what the
    list-centric IVF search takes, blocked by chunks of 16 rows; `adc_image_rows_at(M, n, m)` is the byte offset of
    codes[n][m] in it; layout "rows16": the image of its 16-query screen, same size, `adc_image_rows16_at`).  See include/repconc_hip.h rc_adc_scan_image."""
    _need_cuda(codes, image)
    if codes.dtype != torch.uint8 or image.dtype != torch.uint8 or not codes.is_contiguous() or not image.is_contiguous():
        raise ValueError("codes and image must be contiguous uint8")
    if layout not in ("flat", "rows", "rows16"):
        raise ValueError("layout must be flat|rows|rows16")
    M = codes.shape[1]
    if n is None:
        n = codes.shape[0] - n0
    need = adc_image_bytes(n0 + n, M) if layout == "flat" else adc_image_rows_bytes(n0 + n, M)   # rows16: the same size as rows
    if n0 < 0 or n < 0 or n0 + n > codes.shape[0] or adc_image_row_bytes(M) == 0 or image.numel() < need:
        raise ValueError("row range outside the code / image buffers")
    lib, h, s, _ = _ctx(codes)
    fn = {"flat": lib.rc_adc_scan_image, "rows": lib.rc_adc_scan_image_rows, "rows16": lib.rc_adc_scan_image_rows16}[layout]
    _lib.check(fn(h, _p(codes), int(n0), int(n), M, _p(image), s), "rc_adc_scan_image", h)
    return image


SEARCH_SCREEN_M = frozenset((8, 12, 16, 24, 32, 48, 64, 96))      # widths of the screened flat search (csrc/adc_search.hip)
# Default head-room of the sampled candidate threshold: the threshold is the r-th best score of the 32 768-row sample,
# r = floor(mu + slack sqrt(mu + 1) + 4) + 1 with mu = k S / N expected top-k rows in the sample (rc_adc_search_q).  A query whose
# sample holds MORE than r - 1 of the true top-k keeps too few candidates and is repeated alone (~1 ms, PendingSearch).  Round 6:
# 6 -> 3.  At k = 1000 over 8.84 M rows (mu = 3.7, r = 22 -> 15) the chance of that is 8e-6 per query instead of 1e-10 — one repeated
# query per ~100 batches of 1200 — and the screen keeps 6.3 k rows per query instead of 8.7 k, the rescoring 4.0 k instead of 5.7 k:
# 147.4 -> 154.2 k queries/s; slack 2 (1e-4 per query: 3 of 28 800 repeated) is not faster (profiles/r06m_adc_slack.txt).
ADC_SEL_SLACK = 3.0
_warned_slow_m = set()


class PendingSearch:
    """A search that has been enqueued but whose status word has not been read yet (`adc_search(..., defer=True)`).
    `result()` synchronises and reads the status.  In the rare case that the sampled threshold admitted too few or too
    many candidates for SOME queries, only those queries are repeated with an adjusted slack (the others' results stand),
    and queries that still fail after `max_retries` repetitions are answered by the exact path (`adc_search_exact`),
    which terminates for any index content — like Faiss's `index.search` (evaluate_repconc.py:180-185) this never raises
    on degenerate data (thousands of identical codes, all rows tied).  Lets a caller enqueue every query batch before the
    first host synchronisation."""

    def __init__(self, rerun, scores, ids, status, qstatus, slack, max_retries, stream=None):
        self._rerun, self._scores, self._ids, self._status, self._qstatus = rerun, scores, ids, status, qstatus
        self._slack, self._left, self._done = slack, max_retries, status is None
        self._stream = stream
        self.stats = {"retried_queries": 0, "exact_queries": 0}

    def result(self):
        if self._done:
            return self._scores, self._ids
        # retries run on the stream the search was enqueued on, whatever stream is current when result() is called
        with torch.cuda.stream(self._stream) if self._stream is not None else _nullcontext():
            if int(self._status.item()) != 0:
                bad = torch.nonzero(self._qstatus).flatten()
                bits = self._qstatus[bad]
                slack_few, slack_many = self._slack, self._slack
                while bad.numel() and self._left > 0:
                    self._left -= 1
                    self.stats["retried_queries"] += int(bad.numel())
                    slack_few = max(slack_few, 0.0) * 3.0 + 2.0          # bit0: too few candidates -> wider
                    slack_many = max(slack_many / 3.0, 0.0)              # bit1 only: a list overflowed -> tighter
                    still, still_bits = [], []
                    for sel, slack in (((bits & 1) != 0, slack_few), ((bits & 1) == 0, slack_many)):
                        idx = bad[sel]
                        if idx.numel() == 0:
                            continue
                        s, i, qs = self._rerun(idx, slack, False)
                        ok = qs == 0
                        self._scores[idx[ok]] = s[ok]
                        self._ids[idx[ok]] = i[ok]
                        still.append(idx[~ok])
                        still_bits.append(qs[~ok])
                    bad, bits = torch.cat(still), torch.cat(still_bits)
                if bad.numel():
                    self.stats["exact_queries"] = int(bad.numel())
                    s, i, _ = self._rerun(bad, 0.0, True)
                    self._scores[bad] = s
                    self._ids[bad] = i
        self._done = True
        self._rerun = None
        return self._scores, self._ids


_retry_ops_warm = set()


def _warm_retry_ops(dev):
    """The torch operators of the retry path (`PendingSearch.result`, `rerun`: nonzero, mask / index selection, index_put,
    cat) once per device, on eight elements.  A framework operator's first use on a device loads its code object — measured: the
    FIRST repeated query of a process cost 300 ms, every later one ~1 ms (profiles/r06m_adc_retry_cost.txt).  With the threshold's
    head-room at 3 standard deviations a 6 980-query evaluation repeats a query with a probability of 5 %: that load belongs to
    the first search of the process, not to a random batch in the middle of an evaluation."""
    key = (dev.type, dev.index)
    if key in _retry_ops_warm:
        return
    _retry_ops_warm.add(key)
    qs = torch.tensor([0, 1, 0, 2, 0, 0, 1, 0], dtype=torch.int32, device=dev)
    sc = torch.zeros((8, 4), dtype=torch.float32, device=dev)
    ii = torch.zeros((8, 4), dtype=torch.int64, device=dev)
    bad = torch.nonzero(qs).flatten()
    bits = qs[bad]
    for sel in ((bits & 1) != 0, (bits & 1) == 0):
        idx = bad[sel]
        part = sc[idx].contiguous()
        ok = qs[idx] == 1
        sc[idx[ok]] = part[ok]
        ii[idx[ok]] = ii[idx][ok]
        torch.cat([idx[~ok], idx[ok]])
    torch.cat([bad, bad])


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def adc_search_exact(codes: torch.Tensor, centroids: torch.Tensor, q: torch.Tensor, k: int, id_offset: int = 0):
    """The same answer as `adc_search` by the path that cannot fail (rc_adc_search_exact): exact scores of every row and
    a radix select of the min(k, N) best keys.  For the queries the sampled-threshold path gives up on, and a slow but
    independent cross-check in the tests."""
    _need_cuda(codes, centroids, q)
    if codes.dtype != torch.uint8 or not codes.is_contiguous():
        raise ValueError("index codes must be contiguous uint8 [N, M]")
    c, q = _centroids(centroids), _rows_f32(q).contiguous()
    N, M = codes.shape
    nq, D = q.shape
    lib, h, s, _ = _ctx(q)
    scores = torch.empty((nq, k), dtype=torch.float32, device=q.device)
    ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
    if nq == 0 or N == 0:
        scores.fill_(float("-inf"))
        ids.fill_(-1)
        return scores, ids
    wsb = lib.rc_adc_search_exact_ws_bytes(N, M, K, nq, k)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=q.device)
    _lib.check(lib.rc_adc_search_exact(h, _p(codes), N, M, K, _p(c), D, _p(q), nq, int(k), int(id_offset), _p(scores), _p(ids),
                                       _p(ws), wsb, s), "rc_adc_search_exact", h)
    return scores, ids


def adc_search(codes: torch.Tensor, centroids: torch.Tensor, q: torch.Tensor, k: int, id_offset: int = 0,
               sel_slack: Optional[float] = None, max_retries: int = 2, scan_image: Optional[torch.Tensor] = None,
               defer: bool = False, stats: Optional[dict] = None):
    """Top-k inner-product ADC search of `q` [nq,D] against uint8 `codes` [N,M].
    Returns (scores [nq,k] fp32, ids [nq,k] int64), sorted (score desc, id asc).
    evaluate_repconc.py:180-185 / finetune_jpq.py:176.
    scan_image: the index's permuted code image (adc_scan_image_), kept by PQIndex; None = rebuilt per call.
    defer: return a `PendingSearch` instead (no host synchronisation here; `.result()` gives the pair).
    stats (measurement): receives "survivors" / "candidates" = int32 [nq] device tensors, the rows of every query that passed
    the 8-bit screen / that the exact rescoring kept in the FIRST pass (rc_adc_search_ws_counts).
    Never raises on degenerate index content: see `PendingSearch`."""
    _need_cuda(codes, centroids, q, scan_image)
    if sel_slack is None:
        sel_slack = ADC_SEL_SLACK
    if codes.dtype != torch.uint8 or not codes.is_contiguous():
        raise ValueError("index codes must be contiguous uint8 [N, M]")
    if scan_image is not None and (scan_image.dtype != torch.uint8 or not scan_image.is_contiguous()
                                   or adc_image_row_bytes(codes.shape[1]) == 0
                                   or scan_image.numel() < adc_image_bytes(codes.shape[0], codes.shape[1])):
        raise ValueError("scan_image must be a contiguous uint8 buffer of >= adc_image_bytes(N, M) bytes (adc_scan_image_)")
    c, q = _centroids(centroids), _rows_f32(q).contiguous()
    N, M = codes.shape
    nq, D = q.shape
    if D != M * c.shape[2]:
        raise ValueError("query width does not match the centroid table")
    lib, h, s, dev = _ctx(q)
    if M not in SEARCH_SCREEN_M and nq and N:
        # a width without a screening kernel (the reference's IndexPQ takes any M, evaluate_repconc.py:81): the exact scan with a
        # run-time width answers — correct and complete, an order of magnitude slower than the screened search
        global _warned_slow_m
        if M not in _warned_slow_m:
            _warned_slow_m.add(M)
            import logging
            logging.getLogger(__name__).warning("adc_search: MCQ_M = %d has no screening kernel (%s have); using the exact scan",
                                                M, sorted(SEARCH_SCREEN_M))
        got = adc_search_exact(codes, c, q, k, id_offset)
        return PendingSearch(None, got[0], got[1], None, None, 0.0, 0) if defer else got
    scores = torch.empty((nq, k), dtype=torch.float32, device=q.device)
    ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
    if nq == 0 or N == 0:
        if nq and N == 0:
            scores.fill_(float("-inf"))
            ids.fill_(-1)
        return PendingSearch(None, scores, ids, None, None, 0.0, 0) if defer else (scores, ids)
    ws_fn = lib.rc_adc_search_img_ws_bytes if scan_image is not None else lib.rc_adc_search_ws_bytes

    def launch(qq, slack, out_s, out_i):
        # the workspace is released when this returns: the caching allocator hands it out again in stream order only
        n = qq.shape[0]
        wsb = ws_fn(N, M, K, n, k)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=q.device)
        status = torch.zeros((1,), dtype=torch.int32, device=q.device)
        qstatus = torch.zeros((n,), dtype=torch.int32, device=q.device)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.rc_adc_search_q(h, _p(codes), _p(scan_image), N, M, K, _p(c), D, _p(qq), n, int(k), int(id_offset),
                                       float(slack), _p(out_s), _p(out_i), _p(status), _p(qstatus), _p(ws), wsb, st),
                   "rc_adc_search_q", h)
        if stats is not None and "candidates" not in stats:
            so, co = C.c_size_t(0), C.c_size_t(0)
            _lib.check(lib.rc_adc_search_ws_counts(N, M, K, n, C.byref(so), C.byref(co)), "rc_adc_search_ws_counts", h)
            if so.value:
                stats["survivors"] = ws[so.value:so.value + 4 * n].view(torch.int32).clone()
            stats["candidates"] = ws[co.value:co.value + 4 * n].view(torch.int32).clone()
        return status, qstatus

    def rerun(idx, slack, exact):
        qq = q[idx].contiguous()
        if exact:
            s_, i_ = adc_search_exact(codes, c, qq, k, id_offset)
            return s_, i_, None
        s_ = torch.empty((qq.shape[0], k), dtype=torch.float32, device=q.device)
        i_ = torch.empty((qq.shape[0], k), dtype=torch.int64, device=q.device)
        _, qs = launch(qq, slack, s_, i_)
        return s_, i_, qs

    _warm_retry_ops(q.device)
    status, qstatus = launch(q, float(sel_slack), scores, ids)
    pending = PendingSearch(rerun, scores, ids, status, qstatus, float(sel_slack), max_retries,
                            stream=torch.cuda.current_stream(dev))
    return pending if defer else pending.result()

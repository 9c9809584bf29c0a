"""Search that cannot fail (evaluate_repconc.py:180-185: Faiss's index.search returns for any index content).

The fast path places a candidate threshold from a sample of the rows; degenerate indexes (every row identical, a few distinct
codes, k = N, thousands of copies of one passage) defeat any slack.  Then: per-query status bits, only the queries concerned
are repeated, and what still fails goes through rc_adc_search_exact — never a RepconcHipError, and the answer is the
oracle's (score desc, id asc) top-k."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import c_oracle, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _case(M, N, nq, seed):
    C = synth.gaussian(seed + 1, (M, 256, 768 // M))
    codes = synth.uniform_codes(seed + 2, N, M)
    q = synth.gaussian(seed + 3, (nq, 768))
    return C, codes, q


def _same(scores, ids, ws, wi):
    assert np.array_equal(ids.cpu().numpy(), wi)
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), ws.view(np.uint32))


@pytest.mark.parametrize("M,N,nq,k", [(48, 70000, 5, 100), (96, 30000, 3, 1000), (8, 3000, 4, 3000), (24, 300000, 9, 10),
                                      (48, 500, 2, 1000), (64, 20001, 3, 1)])
def test_exact_path_equals_the_oracle(M, N, nq, k):
    from repconc_amd import ops
    C, codes, q = _case(M, N, nq, seed=7 * M + N)
    s, i = ops.adc_search_exact(_t(codes), _t(C), _t(q), k)
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    _same(s, i, ws, wi)


@pytest.mark.parametrize("N,k", [(50000, 1000), (50000, 10), (300000, 200)])
def test_index_of_identical_rows(N, k):
    """Every row has the same code: every score ties, every list overflows whatever the slack.  The answer is rows
    0 .. k-1 (ties by id) with that one score."""
    from repconc_amd import ops
    C, codes, q = _case(48, 1, 4, seed=1)
    codes = np.repeat(codes, N, 0)
    pend = ops.adc_search(_t(codes), _t(C), _t(q), k, defer=True)
    s, i = pend.result()
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    _same(s, i, ws, wi)
    assert np.array_equal(wi, np.tile(np.arange(k), (4, 1)))
    assert pend.stats["exact_queries"] == 4


@pytest.mark.parametrize("N", [60000, 400000])
def test_index_with_three_distinct_codes(N):
    """One of 3 codes per row (N >= 2^18 goes through the 8-bit screen)."""
    from repconc_amd import ops
    from repconc_amd.index import PQIndex
    C, base, q = _case(48, 3, 6, seed=2)
    codes = base[np.random.default_rng(5).integers(0, 3, N)]
    idx = PQIndex(768, 48)
    idx.set_centroids(_t(C))
    idx.add_codes(_t(codes))
    s, i = idx.search(_t(q), 500)
    ws, wi = c_oracle.adc_search(codes, C, q, 500)
    _same(s, i, ws, wi)


def test_k_equals_n():
    from repconc_amd import ops
    for M, N in ((48, 2000), (16, 8000)):
        C, codes, q = _case(M, N, 3, seed=N)
        s, i = ops.adc_search(_t(codes), _t(C), _t(q), N)
        ws, wi = c_oracle.adc_search(codes, C, q, N)
        _same(s, i, ws, wi)


def test_one_degenerate_query_does_not_tax_the_others():
    """300 k random rows + 20 000 copies of one row.  For the queries whose top-k reaches the copies' score the candidate
    list overflows: only THOSE queries are repeated / answered exactly, every query gets the oracle's answer."""
    from repconc_amd import ops
    M, N, nq, k = 48, 300000, 24, 1000
    C, codes, q = _case(M, N, nq, seed=11)
    dup = codes[123].copy()
    codes = np.concatenate([codes, np.repeat(dup[None], 20000, 0)], 0)
    # half of the queries point at the duplicated passage: its 20 001 copies fill their top-k
    from oracle import pq_oracle
    recon = pq_oracle.decode(dup[None].astype(np.int64), C)[0]
    q[: nq // 2] = (q[: nq // 2] * 0.1 + recon * 3.0).astype(np.float32)
    pend = ops.adc_search(_t(codes), _t(C), _t(q), k, defer=True)
    s, i = pend.result()
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    _same(s, i, ws, wi)
    assert 1 <= pend.stats["exact_queries"] <= nq // 2
    assert pend.stats["retried_queries"] <= 2 * (nq // 2)


def test_c_index_never_returns_eselect():
    """rc_index_search on an index of identical rows: RC_OK and the oracle's answer (was RC_ESELECT)."""
    from repconc_amd import _lib
    lib, h = _lib.load(), _lib.handle(0)
    M, N, nq, k = 48, 40000, 3, 100
    C, codes, q = _case(M, 1, nq, seed=3)
    codes = np.repeat(codes, N, 0)
    # one ordinary query batch on an ordinary index through the same handle afterwards: workspace regrowth is harmless
    idx = ctypes.c_void_p()
    assert lib.rc_index_create(h, 768, M, 256, ctypes.byref(idx)) == 0
    try:
        dq, dC, dcodes = _t(q), _t(C), _t(codes)
        sc = torch.empty((nq, k), dtype=torch.float32, device=DEV)
        ids = torch.empty((nq, k), dtype=torch.int64, device=DEV)
        assert lib.rc_index_set_centroids(idx, dC.data_ptr(), None) == 0
        assert lib.rc_index_add_codes(idx, dcodes.data_ptr(), N, None) == 0
        assert lib.rc_index_search(idx, dq.data_ptr(), nq, k, sc.data_ptr(), ids.data_ptr(), None) == 0
        ws, wi = c_oracle.adc_search(codes, C, q, k)
        _same(sc, ids, ws, wi)
    finally:
        assert lib.rc_index_destroy(idx) == 0


@pytest.mark.parametrize("mode", ["replicated", "sharded"])
def test_in_place_centroid_write_reaches_every_part_of_a_multi_index(mode):
    """finetune_jpq.py:211-213 writes the centroids IN PLACE through `faiss.copy_array_to_vector(c, index.pq.centroids)`.
    On a replicated / sharded index every part has its own resident table: the write must reach all of them (it used to
    update part 0 only, and the parts then scored with different codebooks)."""
    from repconc_amd import faiss_compat as faiss
    from repconc_amd.index import PQIndex
    from repconc_amd.models.repconc.evaluate_repconc import load_index_to_gpu
    C, codes, q = _case(48, 30000, 9, seed=21)
    idx = PQIndex(768, 48)
    idx.set_centroids(_t(C))
    idx.add_codes(_t(codes))
    ndev = torch.cuda.device_count()
    multi = load_index_to_gpu(idx, None, shard=(mode == "sharded"), devices=list(range(ndev)) if ndev >= 2 else [0, 0, 0])
    C2 = (C * np.float32(0.5) + np.float32(0.125)).astype(np.float32)
    faiss.copy_array_to_vector(C2.ravel(), multi.pq.centroids)
    for part in multi.parts:
        assert np.array_equal(part.pq.centroids.cpu().numpy(), C2)
    s, i = multi.search(_t(q), 50)
    ws, wi = c_oracle.adc_search(codes, C2, q, 50)
    _same(s, i, ws, wi)
    assert np.array_equal(faiss.vector_to_array(multi.pq.centroids), C2.ravel())


@pytest.mark.parametrize("method", ["lists", "lists8", "lists16", "scan"])
def test_ivf_search_never_raises_on_degenerate_cells(method):
    """IVF cells full of identical rows (duplicated passages): the list-centric screen and the per-query scan both hand
    over to the exact path instead of raising; answers equal the brute-force oracle (score desc, corpus id asc)."""
    from oracle import pq_oracle
    from repconc_amd.ivf import IVFPQIndex
    M, nlist, N, nq, k = 48, 16, 60000, 5, 200
    C, base, q = _case(M, 3, nq, seed=31)
    rng = np.random.default_rng(32)
    codes = base[rng.integers(0, 3, N)]                           # three distinct rows only
    cells = rng.integers(0, nlist, N)
    coarse = synth.gaussian(33, (nlist, 768))
    ivf = IVFPQIndex(768, M, nlist, device=DEV)
    ivf.set_centroids(_t(C))
    ivf.coarse = _t(coarse)
    ivf.set_lists(_t(codes), _t(cells))
    for nprobe in (4, nlist):
        s, i = ivf.search(_t(q), k, nprobe, method=method)
        ws, wi = pq_oracle.ivf_search(q, C, codes, cells, coarse, k, nprobe)
        assert np.array_equal(i.cpu().numpy(), wi), (method, nprobe)
        assert np.array_equal(s.cpu().numpy().view(np.uint32), ws.view(np.uint32))


@pytest.mark.parametrize("M", [32, 48, 96])
def test_ivf_pipelined_screen_on_cells_of_several_rounds(M):
    """Cells far larger than the 2048 rows a block screens per round (several rounds per task, ragged last chunk, first row
    of a cell in the middle of a chunk), cells smaller than one chunk, empty cells, fewer queries than a task holds and
    queries that keep every row: the list-centric search (persistent pipelined screen, survivor streams, bucket pass)
    equals the oracle's brute force over the probed cells."""
    from oracle import pq_oracle
    from repconc_amd.ivf import IVFPQIndex
    nlist, N, k = 12, 70001, 300
    rng = np.random.default_rng(4100 + M)
    codes = synth.uniform_codes(4101 + M, N, M)
    cells = rng.choice(nlist, N, p=[0.45, 0.3, 0.15, 0.05, 0.03, 0.0199, 0.0001, 0, 0, 0, 0, 0])   # 31 k-row cell ... 7 rows, empty
    C = synth.gaussian(4102 + M, (M, 256, 768 // M))
    coarse = synth.gaussian(4103 + M, (nlist, 768))
    ivf = IVFPQIndex(768, M, nlist, device=DEV)
    ivf.set_centroids(_t(C))
    ivf.coarse = _t(coarse)
    ivf.set_lists(_t(codes), _t(cells))
    for nq, nprobe in ((3, 2), (19, 5), (70, nlist)):
        q = synth.gaussian(4104 + M + nq, (nq, 768))
        ws, wi = pq_oracle.ivf_search(q, C, codes, cells, coarse, k, nprobe)
        for method in ("lists8", "lists16"):                 # both widths of the screen (round 6)
            s, i = ivf.search(_t(q), k, nprobe, method=method)
            assert np.array_equal(i.cpu().numpy(), wi), (M, nq, nprobe, method)
            assert np.array_equal(s.cpu().numpy().view(np.uint32), ws.view(np.uint32))


def test_ivf_survivor_stream_overflow_is_answered_not_raised(monkeypatch):
    """A wave's survivor stream too small for what it keeps (RC_IVF_STREAM_CAP: a test knob of the workspace layout): the
    screen drops the overflow and raises status bit 1, the search narrows the slack and finally takes the per-query scan —
    the answer is still the oracle's."""
    from oracle import pq_oracle
    from repconc_amd.ivf import IVFPQIndex
    M, nlist, N, nq, k, nprobe = 48, 40, 120000, 24, 500, 10
    rng = np.random.default_rng(77)
    codes = synth.uniform_codes(78, N, M)
    cells = rng.integers(0, nlist, N)
    C = synth.gaussian(79, (M, 256, 768 // M))
    coarse = synth.gaussian(80, (nlist, 768))
    q = synth.gaussian(81, (nq, 768))
    ivf = IVFPQIndex(768, M, nlist, device=DEV)
    ivf.set_centroids(_t(C))
    ivf.coarse = _t(coarse)
    ivf.set_lists(_t(codes), _t(cells))
    monkeypatch.setenv("RC_IVF_STREAM_CAP", "16")
    ws, wi = pq_oracle.ivf_search(q, C, codes, cells, coarse, k, nprobe)
    for method in ("lists8", "lists16"):
        s, i = ivf.search(_t(q), k, nprobe, method=method)
        assert np.array_equal(i.cpu().numpy(), wi) and np.array_equal(s.cpu().numpy().view(np.uint32), ws.view(np.uint32)), method


def test_ivf_list_search_equals_the_per_query_scan_in_a_fresh_process(tmp_path):
    """The list-centric search (pipelined 8-bit screen) against the per-query exact scan — two independent code paths of the
    library — in a fresh process: same ids and score bits.  (Round 3 compared it with the round-2 IVF screen, RC_IVF_PIPE=0,
    which round 4 removed together with the other superseded screen generations.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from oracle import synth\n"
        "from repconc_amd.ivf import IVFPQIndex\n"
        "M, nlist, N, nq, k = 96, 50, 90000, 40, 200\n"
        "rng = np.random.default_rng(5)\n"
        "ivf = IVFPQIndex(768, M, nlist, device='cuda')\n"
        "ivf.set_centroids(torch.from_numpy(synth.gaussian(1, (M, 256, 768 // M))).cuda())\n"
        "ivf.coarse = torch.from_numpy(synth.gaussian(2, (nlist, 768))).cuda()\n"
        "ivf.set_lists(torch.from_numpy(synth.uniform_codes(3, N, M)).cuda(), torch.from_numpy(rng.integers(0, nlist, N)).cuda())\n"
        "s, i = ivf.search(torch.from_numpy(synth.gaussian(4, (nq, 768))).cuda(), k, 12, method=sys.argv[3])\n"
        "np.save(sys.argv[1], i.cpu().numpy()); np.save(sys.argv[2], s.cpu().numpy())\n")
    got = {}
    for method in ("lists", "scan"):
        fi, fs = str(tmp_path / f"i{method}.npy"), str(tmp_path / f"s{method}.npy")
        r = subprocess.run([sys.executable, "-c", code, fi, fs, method], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[method] = (np.load(fi), np.load(fs))
    assert np.array_equal(got["lists"][0], got["scan"][0])
    assert np.array_equal(got["lists"][1].view(np.uint32), got["scan"][1].view(np.uint32))


def test_ivf_one_degenerate_query_is_answered_alone(monkeypatch):
    """IVF batch in which ONE query probes a cell full of identical rows (its threshold admits thousands of equal scores):
    rc_ivf_search_probes_q flags that query only; the others keep the list-centric answer (no per-query scan for them), all
    results equal the oracle."""
    from oracle import pq_oracle
    from repconc_amd.ivf import IVFPQIndex
    M, nlist, N, nq, k, nprobe = 48, 24, 150000, 33, 300, 3
    rng = np.random.default_rng(611)
    codes = synth.uniform_codes(612, N, M)
    cells = rng.integers(1, nlist, N)
    cells[:40000] = 0
    codes[:40000] = codes[0]                                  # cell 0: 40 000 copies of one row
    C = synth.gaussian(613, (M, 256, 768 // M))
    coarse = synth.gaussian(614, (nlist, 768))
    q = synth.gaussian(615, (nq, 768))
    coarse[0] = q[7] * 4.0                                    # query 7 (and almost only it) probes cell 0
    ivf = IVFPQIndex(768, M, nlist, device=DEV)
    ivf.set_centroids(_t(C))
    ivf.coarse = _t(coarse)
    ivf.set_lists(_t(codes), _t(cells))
    calls = []
    orig = IVFPQIndex.search

    def spy(self, x, kk, npb, method="auto"):
        calls.append((int(x.shape[0]), method))
        return orig(self, x, kk, npb, method)
    monkeypatch.setattr(IVFPQIndex, "search", spy)
    s, i = ivf.search(_t(q), k, nprobe, method="lists")
    ws, wi = pq_oracle.ivf_search(q, C, codes, cells, coarse, k, nprobe)
    assert np.array_equal(i.cpu().numpy(), wi) and np.array_equal(s.cpu().numpy().view(np.uint32), ws.view(np.uint32))
    scans = [c for c in calls if c[1] == "scan"]
    assert len(scans) <= 1 and all(c[0] < nq // 2 for c in scans), calls      # only the flagged few went to the exact scan

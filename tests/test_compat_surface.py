"""The `repconc.*` import surface (compat/repconc) and the Faiss-idiom helpers — CPU-side checks (no kernels run)."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT


def test_repconc_namespace_aliases_repconc_amd_modules():
    """A caller-shaped import block (evaluate/run_repconc_eval.py:16-24, train/run_warmup.py, finetune_jpq.py) in a fresh
    interpreter with compat/ on the path: every name resolves, and to the very objects repconc_amd defines."""
    code = r'''
import sys
from repconc.models.repconc import RepCONC, QuantizeOutput, sinkhorn_algorithm, decode
from repconc.models.repconc.evaluate_repconc import (ModelArguments, EvalArguments, RepCONCEvaluater, initialize_index,
    add_docs, from_pq_to_ivfpq, load_index_to_gpu, encode_corpus, encode_query, search, batch_search)
from repconc.models.dense import AutoDense, BertDense, RobertaDense, DistilBertDense
from repconc.train.run_warmup import warmup_from_embeds
from repconc.models.jpq.finetune_jpq import JPQ
from repconc.utils.eval_utils import load_corpus, load_queries, TextDataset, get_collator_func
import repconc_amd.models.repconc.modeling_repconc as real
import repconc.models.repconc.modeling_repconc as alias
assert alias is real and RepCONC is real.RepCONC
assert sys.modules["repconc.models.repconc.evaluate_repconc"] is sys.modules["repconc_amd.models.repconc.evaluate_repconc"]
m = ModelArguments(model_name_or_path="x")
assert m.doc_encoder_path == m.query_encoder_path == "x"
assert "faiss" not in sys.modules
print("ok")
'''
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "compat"), ROOT]))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]


def test_eval_utils_collator_dataset_and_metrics(tmp_path):
    from repconc_amd.utils.eval_utils import (TextDataset, get_collator_func, load_corpus, load_queries, mrr_at_k,
                                              recall_at_k)

    class Tok:
        def __call__(self, texts, padding, truncation, max_length):
            ids = [[1] + [3] * min(len(t.split()), max_length - 2) + [2] for t in texts]
            L = max(map(len, ids))
            return {"input_ids": [i + [0] * (L - len(i)) for i in ids],
                    "attention_mask": [[1] * len(i) + [0] * (L - len(i)) for i in ids]}

    ds = TextDataset(["a b c", "d"], text_ids=[7, 9])
    batch = get_collator_func(Tok(), 8, "doc")([ds[0], ds[1]])
    assert batch["input_ids"].shape == (2, 5) and batch["text_ids"].tolist() == [7, 9]
    assert batch["attention_mask"].sum().item() == 8
    assert "text_ids" not in get_collator_func(Tok(), 8, "query")(["x y"])
    p = tmp_path / "c.tsv"
    p.write_text("d1\ttitle\tbody text\nd2\tonly body\n")
    assert load_corpus(str(p), " [SEP] ") == {"d1": "title [SEP] body text", "d2": "only body"}
    q = tmp_path / "q.tsv"
    q.write_text("5\twhat is x\n")
    assert load_queries(str(q)) == {"5": "what is x\n"}
    qrels = {"q1": {"a": 1}, "q2": {"b": 1, "c": 0}, "q3": {}}
    run = [["x", "a", "y"], ["c", "z", "w"], ["a"]]
    assert mrr_at_k(run, qrels, ["q1", "q2", "q3"], 10) == 0.25            # (1/2 + 0) / 2 scored queries
    assert recall_at_k(run, qrels, ["q1", "q2", "q3"], 2) == 0.5


def test_faiss_idiom_helpers_on_cpu_tensors():
    import torch
    from repconc_amd import faiss_compat as faiss
    vec = torch.zeros(4, 256, 2)
    arr = np.arange(4 * 256 * 2, dtype=np.float32)
    faiss.copy_array_to_vector(arr, vec)
    assert np.array_equal(faiss.vector_to_array(vec), arr) and vec[1, 0, 1].item() == 513.0
    codes = torch.arange(12, dtype=torch.uint8).reshape(3, 4)
    assert faiss.vector_to_array(codes).tolist() == list(range(12))
    faiss.omp_set_num_threads(32)
    assert faiss.METRIC_INNER_PRODUCT == 0


def test_indexpq_reader_parses_the_hand_assembled_golden_file_header():
    """tests/golden/ixpq_d8_m2_n5.faissindex is assembled by tests/golden/make_ixpq.py in Faiss 1.7.x writer order,
    independently of faiss_io.  CPU part: the byte layout agrees field by field (the GPU test reads it into a PQIndex)."""
    import struct
    from conftest import GOLDEN
    raw = open(os.path.join(GOLDEN, "ixpq_d8_m2_n5.faissindex"), "rb").read()
    exp = np.load(os.path.join(GOLDEN, "ixpq_d8_m2_n5_expected.npz"))
    assert raw[:4] == b"IxPq" and len(raw) == 4 + 33 + 24 + 8 + 4 * 2 * 256 * 4 + 8 + 10 + 9
    d, ntotal, d1, d2, trained, metric = struct.unpack_from("<iqqqBi", raw, 4)
    assert (d, ntotal, d1, d2, trained, metric) == (8, 5, 1 << 20, 1 << 20, 1, 0)
    assert struct.unpack_from("<QQQQ", raw, 37) == (8, 2, 8, 2 * 256 * 4)
    assert np.array_equal(np.frombuffer(raw, "<f4", 2048, 69).reshape(2, 256, 4), exp["centroids"])


def test_gradcache_two_pass_gradients_equal_direct_backward_cpu():
    """repconc_amd.gradcache on a small CPU model with dropout: forward_no_grad + build_cache + a second pass under the
    recorded RNG contexts reproduces the gradients of one big-batch backward (same dropout masks)."""
    import torch
    from repconc_amd.gradcache import GradCache
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Dropout(0.3), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    xq, xd = torch.randn(10, 6), torch.randn(10, 6)
    loss_fn = lambda q, d, scale: torch.nn.functional.cross_entropy(q @ d.T * scale, torch.arange(q.shape[0]))
    gc = GradCache([net], 4, loss_fn)
    chunks = lambda x: [{"input": c} for c in x.split(4)]
    net.zero_grad()
    rq, sq = gc.forward_no_grad(net, chunks(xq))
    rd, sd = gc.forward_no_grad(net, chunks(xd))
    assert not rq.requires_grad and len(sq) == 3
    (gq, gd), loss = gc.build_cache(rq, rd, scale=2.0)
    for x, g, states in ((xq, gq, sq), (xd, gd, sd)):
        off = 0
        for c, st in zip(x.split(4), states):
            with st:
                out = net(c)
            torch.dot(g[off:off + len(c)].flatten(), out.flatten()).backward()
            off += len(c)
    got = [p.grad.clone() for p in net.parameters()]
    # direct: the same masks come from replaying the per-chunk RNG states
    net.zero_grad()
    outs = []
    for x, states in ((xq, sq), (xd, sd)):
        parts = []
        for c, st in zip(x.split(4), states):
            with st:
                parts.append(net(c))
        outs.append(torch.cat(parts))
    direct = loss_fn(outs[0], outs[1], scale=2.0)
    direct.backward()
    assert abs(float(direct) - float(loss)) < 1e-6
    for a, p in zip(got, net.parameters()):
        assert torch.allclose(a, p.grad, atol=1e-6)


def test_procrustes_rotation_equals_the_svd_solution():
    """train/run_warmup.py (OPQ step): the Newton-Schulz polar factor is U V^T of the SVD; singular input falls back."""
    import torch
    from repconc_amd.train.run_warmup import procrustes_rotation
    g = torch.Generator().manual_seed(4)
    for D, cond in ((96, 1e2), (96, 1e6), (33, 1e3)):
        U = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
        V = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
        S = torch.logspace(0, -float(torch.log10(torch.tensor(cond))), D, dtype=torch.float64) * 12.5
        R = procrustes_rotation((U * S) @ V.T)
        assert float((R - U @ V.T).abs().max()) < 1e-9
        assert float((R.T @ R - torch.eye(D, dtype=torch.float64)).abs().max()) < 1e-12
    P = torch.zeros(8, 8, dtype=torch.float64)
    P[0, 0] = 1.0                                                   # rank 1: no unique polar factor -> SVD branch
    R = procrustes_rotation(P)
    assert float((R.T @ R - torch.eye(8, dtype=torch.float64)).abs().max()) < 1e-12
    assert float((R.T @ R - torch.eye(8, dtype=torch.float64)).abs().max()) < 1e-12 and abs(float(R[0, 0])) > 0.999


def test_procrustes_fixed_schedule_and_its_deferred_check():
    """The optimally scaled Newton-Schulz schedule is a host-side constant (no look at the matrix): its lower bound grows
    to 1, its last steps are the plain iteration; a matrix beyond the schedule's range is REPORTED by the deferred check
    (what train_opq reads once, after the last round) and handled by the SVD in the checked mode."""
    import torch
    from repconc_amd.train.run_warmup import _polar_schedule, procrustes_rotation
    sched = _polar_schedule(1e-12)
    assert 30 <= len(sched) <= 40 and abs(sched[0][0] - 2.598076) < 1e-5 and sched[-1] == (1.5, -0.5)
    ell = 1e-12
    for a, b in sched[:-2]:
        assert a + b <= 1.0 + 1e-12 and a > 1.0                    # p(1) = l' <= 1: nothing above 1 is ever produced
        ell = a * ell + b * ell ** 3
    assert 1.0 - ell < 1e-15
    g = torch.Generator().manual_seed(9)
    D = 48
    U = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
    V = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
    good = (U * torch.logspace(0, -8, D, dtype=torch.float64)) @ V.T
    X, err = procrustes_rotation(good, defer=True)
    assert float(err) < 1e-12 and float((X - U @ V.T).abs().max()) < 1e-7
    bad = (U * torch.logspace(0, -15, D, dtype=torch.float64)) @ V.T       # cond 1e15: outside [1e-12, 1]
    _, err_bad = procrustes_rotation(bad, defer=True)
    assert not (float(err_bad) < 1e-9)
    R = procrustes_rotation(bad)                                            # checked mode: SVD fall-back, orthogonal
    assert float((R.T @ R - torch.eye(D, dtype=torch.float64)).abs().max()) < 1e-12


def test_procrustes_matrix_as_a_batch_of_row_slices():
    """run_warmup._xt_y (x^T y for the Procrustes step: sixteen row slices as one batched product, partial products added in
    fp64): the fp64 product to fp32 accuracy for any number of rows (fewer than 16, a ragged tail, non-contiguous rows), and the
    static-buffer form of the Procrustes iteration equals the allocating one bit for bit."""
    import torch
    from repconc_amd.train.run_warmup import _polar_schedule, _procrustes_static, _xt_y, procrustes_rotation
    g = torch.Generator().manual_seed(4)
    for n in (1, 15, 16, 17, 100, 1000, 4099):
        x, y = torch.randn(n, 24, generator=g), torch.randn(n, 24, generator=g)
        want = x.double().T @ y.double()
        assert torch.allclose(_xt_y(x, y), want, atol=1e-4 * max(n, 16) ** 0.5), n
        xs = torch.randn(n, 48, generator=g)[:, ::2]
        assert torch.allclose(_xt_y(xs, y), xs.double().T @ y.double(), atol=1e-4 * max(n, 16) ** 0.5), n
    D = 32
    P = torch.randn(D, D, generator=g, dtype=torch.float64)
    X = [torch.empty((D, D), dtype=torch.float64) for _ in range(2)]
    cur, err = _procrustes_static(P, X, torch.empty((D, D), dtype=torch.float64), torch.eye(D, dtype=torch.float64),
                                  list(_polar_schedule(1e-12)))
    ref, err_ref = procrustes_rotation(P, defer=True)
    assert torch.equal(cur, ref) and float(err) == float(err_ref)


def test_faiss_shim_serves_the_scripts_faiss_idioms():
    """compat/faiss: with compat/ first on the path the reference's entry scripts' `import faiss` lines resolve (no edit at
    all): the names they touch exist, `import faiss.contrib.torch_utils` works, and what they do not need is absent."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import faiss, faiss.contrib.torch_utils\n"
            "import repconc_amd.faiss_compat as fc\n"
            "assert faiss.read_index is fc.read_index and faiss.write_index is fc.write_index\n"
            "assert faiss.copy_array_to_vector is fc.copy_array_to_vector and faiss.vector_to_array is fc.vector_to_array\n"
            "assert callable(faiss.omp_set_num_threads) and faiss.METRIC_INNER_PRODUCT == 0\n"
            "def f(index: faiss.IndexPQ, other: faiss.IndexIVFPQ): pass\n"
            "assert not hasattr(faiss, 'index_factory')\n"
            "import repconc.models.repconc as m; import repconc_amd.models.repconc as r; assert m is r\n"
            "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "compat") + os.pathsep + root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_indexpq_file_reader_is_strict(tmp_path):
    """The Faiss IndexPQ layout was restated from the published writer and has never met a Faiss-written file: the reader
    must reject whatever does not fit it — another fourcc, mismatching sizes, truncation, trailing bytes — loudly."""
    import pytest
    from repconc_amd.faiss_io import parse_index_file
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ixpq_d8_m2_n5.faissindex")
    raw = open(golden, "rb").read()
    p = parse_index_file(golden)
    assert p["d"] == 8 and p["M"] == 2 and p["ntotal"] == 5 and p["centroids"].shape == (2, 256, 4) and p["codes"].shape == (5, 2)

    def check(data, fragment):
        f = tmp_path / "x.faissindex"
        f.write_bytes(data)
        with pytest.raises(ValueError, match=fragment):
            parse_index_file(str(f))
    check(b"IxFI" + raw[4:], "fourcc")                                  # another index type
    check(raw[:-3], "truncated")                                        # short file
    check(raw + b"\\x00", "after the last")                              # trailing bytes
    bad_nt = bytearray(raw); bad_nt[8:16] = (6).to_bytes(8, "little")    # ntotal says 6, the code vector holds 5 rows
    check(bytes(bad_nt), "code vector")
    bad_m = bytearray(raw); off = 4 + 4 + 8 * 3 + 1 + 4 + 8              # M field of the ProductQuantizer header
    bad_m[off:off + 8] = (3).to_bytes(8, "little")
    check(bytes(bad_m), "ProductQuantizer")


def test_alias_import_keeps_the_real_modules_spec():
    """`import repconc.X` hands out the repconc_amd.X module object; its __spec__ / __name__ / __package__ must stay the
    real ones (module_from_spec would otherwise leave the alias spec behind: relative imports inside the module then warn
    about __package__ != __spec__.parent and importlib.reload breaks)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import warnings; warnings.simplefilter('error')\n"
            "import repconc.models.repconc.modeling_repconc as a\n"
            "import repconc_amd.models.repconc.modeling_repconc as b\n"
            "assert a is b and b.__spec__.name == 'repconc_amd.models.repconc.modeling_repconc', b.__spec__\n"
            "assert b.__name__ == 'repconc_amd.models.repconc.modeling_repconc' and b.__package__ == 'repconc_amd.models.repconc'\n"
            "import importlib; importlib.reload(b)\n"
            "import repconc.sharded_search as s; import repconc_amd.sharded_search as t; assert s is t and t.__spec__.name == 'repconc_amd.sharded_search'\n"
            "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "compat") + os.pathsep + root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]

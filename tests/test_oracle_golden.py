"""The oracle pinned against the reference's outputs (tests/golden, made by oracle/gen_golden.py).
CPU only.  numpy restatement on the small cases, C restatement on every case."""
import os
import zlib

import numpy as np
import pytest

from conftest import golden_cases, load_case
from oracle import c_oracle, pq_oracle, synth

EPS, ITERS = 0.003, 100
SMALL = [c for c in golden_cases() if "b6144" not in c]


@pytest.mark.parametrize("name", golden_cases())
def test_c_oracle_codes_bit_exact(name):
    g, x, C = load_case(name)
    near, _ = c_oracle.quantize(x, C, False)
    assert np.array_equal(near, g["codes_nearest"])
    con, flags = c_oracle.quantize(x, C, True, EPS, ITERS)
    assert flags == 0
    assert np.array_equal(con, g["codes_constrained"])


@pytest.mark.parametrize("name", golden_cases())
def test_c_oracle_fp32_stage_bitwise(name):
    g, x, C = load_case(name)
    d = c_oracle.dist_table(x, C)
    idx = g["samp_idx"]
    assert np.array_equal(d[idx[:, 0], idx[:, 1], idx[:, 2]].view(np.uint32), g["samp_dist_bits"])
    mm = c_oracle.minmax(d)
    M = d.shape[0]
    assert np.array_equal(mm[:M].view(np.uint32), g["mx_bits"])
    assert np.array_equal(mm[M:].view(np.uint32), g["mn_bits"])
    c_oracle.centre_(d, mm)
    assert np.array_equal(d[idx[:, 0], idx[:, 1], idx[:, 2]].view(np.uint32), g["samp_centred_bits"])


@pytest.mark.parametrize("kind,m0", [("sample", 0), ("lloyd", 44)])
def test_c_oracle_on_the_headline_batch_equals_the_reference(kind, m0):
    """BASELINE configs[1] itself — 49 152 x 768, M = 48 — as the REFERENCE computed it (oracle/gen_golden.py --headline: four
    column slices of twelve sub-quantisers; every reduction of quantize is per sub-quantiser).  The CPU suite checks the C
    restatement on four of the 48 sub-quantisers per centroid kind (the same slicing argument; all 48 took 93 s per kind when
    the fixtures were made: 0 mismatches); the GPU suite compares the HIP path with all 2 x 2 359 296 codes."""
    from conftest import load_headline
    x, C, con, near = load_headline(kind)
    xs = np.ascontiguousarray(x[:, m0 * 16:(m0 + 4) * 16])
    Cs = np.ascontiguousarray(C[m0:m0 + 4])
    got, fl = c_oracle.quantize(xs, Cs, True, EPS, ITERS)
    assert fl == 0 and np.array_equal(got, con[:, m0:m0 + 4])
    assert np.array_equal(c_oracle.quantize(xs, Cs, False)[0], near[:, m0:m0 + 4])


@pytest.mark.parametrize("name", ["m8_b300_gauss", "m48_b1024_sample"])
def test_numpy_oracle_plan_equals_the_references_plan(name):
    """The transport plan itself, not only its argmax: tests/golden/plan_<case>.npz holds 4096 sampled entries of what the
    reference's sinkhorn_algorithm returned on the case's centred table (gen_golden.py --plan).  oracle/pq_oracle.sinkhorn_q is
    the same sequence of fp64 operations: relative difference below 1e-12 (numpy and torch reduce sums in different orders),
    identical column argmax."""
    g, x, C = load_case(name)
    p = np.load(os.path.join(os.path.dirname(__file__), "golden", f"plan_{name}.npz"))
    d = pq_oracle.dist_table(x, C)
    dc = pq_oracle.centre(d, *pq_oracle.minmax_per_m(d))
    Q = pq_oracle.sinkhorn_q([-(dc.astype(np.float64)).transpose(0, 2, 1)], EPS, ITERS)[0]
    assert np.array_equal(Q.argmax(1).astype(np.uint8), p["argmax"])
    got, want = Q.reshape(-1)[p["sample_index"]], p["sample_q"]
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-300)
    assert np.array_equal(p["argmax"].T, g["codes_constrained"])          # the plan's argmax IS the fixture's codes


@pytest.mark.parametrize("name", ["m6_b384_eps003", "m3_b1000_eps05"])
def test_numpy_oracle_plan_on_a_general_fp64_cost_tensor_equals_the_references(name):
    """tests/golden/plan64_<case>.npz (gen_golden.py --plan64): the reference's `sinkhorn_algorithm` on out = uniform(-1, 1) in
    fp64 — NOT fp32-representable, the generality of modeling_repconc.py:137-141 — against pq_oracle.sinkhorn_q (same sequence of
    fp64 operations: 1e-12 relative) and, as two column shards, its rank-ordered restatement of :149-157."""
    import zlib
    p = np.load(os.path.join(os.path.dirname(__file__), "golden", f"plan64_{name}.npz"))
    M, B, eps, iters = int(p["M"]), int(p["B"]), float(p["eps"]), int(p["iters"])
    out = np.random.default_rng(int(p["seed"])).uniform(-1.0, 1.0, (M, 256, B))
    assert zlib.crc32(out.tobytes()) == int(p["out_crc"])
    Q = pq_oracle.sinkhorn_q([out.copy()], eps, iters)[0]
    assert np.array_equal(Q.argmax(1).astype(np.uint8), p["argmax"])
    np.testing.assert_allclose(Q.reshape(-1)[p["sample_index"]], p["sample_q"], rtol=1e-12, atol=1e-300)
    Q2 = np.concatenate(pq_oracle.sinkhorn_q([out[:, :, :B // 2].copy(), out[:, :, B // 2:].copy()], eps, iters), axis=2)
    np.testing.assert_allclose(Q2, Q, rtol=1e-11, atol=1e-300)


def test_numpy_oracle_decode_gradient_and_normalisation_equal_the_references():
    """tests/golden/aux_m48_b1024.npz (gen_golden.py --aux): the gradient the reference's autograd sends to the centroids
    through `decode` (modeling_repconc.py:168-175) and its `normalize_centrodis` (:112-116) — pq_oracle.decode_bwd /
    normalize_centroids restate them (fp32 sums of <= a few dozen terms in another order: 1e-5; one division: 1e-6)."""
    g, x, C = load_case("m48_b1024_sample")
    a = np.load(os.path.join(os.path.dirname(__file__), "golden", "aux_m48_b1024.npz"))
    go = synth.gaussian(5, (1024, 768))
    np.testing.assert_allclose(pq_oracle.decode_bwd(g["codes_constrained"], go, 48, 256), a["decode_grad"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pq_oracle.normalize_centroids(synth.gaussian(3, (48, 256, 16))), a["normalized"], rtol=1e-6, atol=1e-7)


def test_reference_on_eight_ranks_returns_its_one_process_codes():
    """tests/golden/recipe8_b49152_m48_sample.npz (gen_golden.py --recipe8): the reference's distributed branch on eight gloo
    ranks of 6 144 rows gave the codes of its one-process run of the same 49 152-row batch in all 2 359 296 places — "sharded ==
    unsharded" is the reference's own behaviour at the recipe shape, which is what the multi-rank tests assert of this build."""
    from conftest import GOLDEN, load_headline
    r8 = np.load(os.path.join(GOLDEN, "recipe8_b49152_m48_sample.npz"))
    _, _, con, _ = load_headline("sample")
    assert int(r8["world"]) == 8 and r8["codes_xor_one_process"].shape == con.shape and not r8["codes_xor_one_process"].any()
    assert zlib.crc32(con.tobytes()) == int(r8["codes_crc"])


def test_c_oracle_on_config0_equals_the_reference():
    """BASELINE configs[0] at its exact inputs (oracle/gen_golden.py --config0 ran the reference whole): the C
    restatement returns the reference's constrained and nearest codes of all 10 000 x 8."""
    import zlib
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "config0_b10000_m8.npz"))
    N, M = int(g["B"]), int(g["M"])
    x = np.random.default_rng(20220).standard_normal((N, 768), dtype=np.float32)
    C = np.ascontiguousarray(x[np.random.default_rng(20221).permutation(N)[:256]].reshape(256, M, 768 // M).transpose(1, 0, 2))
    assert zlib.crc32(x.tobytes()) == int(g["x_crc"]) and zlib.crc32(C.tobytes()) == int(g["centroids_crc"])
    got, fl = c_oracle.quantize(x, C, True, EPS, ITERS)
    assert fl == 0 and np.array_equal(got, g["codes_constrained"])
    assert np.array_equal(c_oracle.quantize(x, C, False)[0], g["codes_nearest"])


@pytest.mark.parametrize("M,m0,nm", [(24, 11, 2), (96, 70, 4)])
def test_c_oracle_on_the_other_full_batches_equals_the_reference(M, m0, nm):
    """BASELINE c5 / c4 (49 152 x 768, M = 24 / 96) as the reference computed them (gen_golden.py --headline sample
    --headline-m M): the C restatement on a few sub-quantisers (dsub 32 / 8), constrained and nearest."""
    from conftest import load_headline
    x, C, con, near = load_headline("sample", M)
    ds = 768 // M
    xs = np.ascontiguousarray(x[:, m0 * ds:(m0 + nm) * ds])
    Cs = np.ascontiguousarray(C[m0:m0 + nm])
    got, fl = c_oracle.quantize(xs, Cs, True, EPS, ITERS)
    assert fl == 0 and np.array_equal(got, con[:, m0:m0 + nm])
    assert np.array_equal(c_oracle.quantize(xs, Cs, False)[0], near[:, m0:m0 + nm])


@pytest.mark.parametrize("name", ["m48_b1024_sample", "m8_b2048_sample", "m96_b512_blend", "m48_b1000_ragged"])
def test_torch_port_matches_reference(name):
    """oracle/torch_port.py — bench.py's torch-CPU baseline (the reference's algorithmic shape on the host's
    cores) — returns the reference's codes on the golden fixtures."""
    import torch
    from oracle import torch_port
    g, x, C = load_case(name)
    xt, Ct = torch.from_numpy(x), torch.from_numpy(C)
    assert np.array_equal(torch_port.quantize(xt, Ct, False).numpy().astype(np.uint8), g["codes_nearest"])
    assert np.array_equal(torch_port.quantize(xt, Ct, True, EPS, ITERS).numpy().astype(np.uint8), g["codes_constrained"])


def test_tiled_cpu_search_equals_the_query_parallel_one():
    """oracle/pq_oracle.c: the cache-blocked ADC search (bench.py's many-core CPU baseline) returns the ids and score bits of
    the Faiss-style one (every query streams the whole index), ragged tile sizes included."""
    rng = np.random.default_rng(5)
    N, M = 50003, 24
    codes = rng.integers(0, 256, (N, M), dtype=np.uint8)
    codes[1000:1400] = codes[999]                                             # a run of tied rows
    C = rng.standard_normal((M, 256, 32)).astype(np.float32)
    q = rng.standard_normal((9, 768)).astype(np.float32)
    s0, i0 = c_oracle.adc_search(codes, C, q, 300)
    for tile in (0, 4 * 1001, 4, 1 << 20):
        s1, i1 = c_oracle.adc_search(codes, C, q, 300, tile=tile)
        assert np.array_equal(s0.view(np.uint32), s1.view(np.uint32)) and np.array_equal(i0, i1), tile


@pytest.mark.parametrize("name", ["m8_b300_gauss", "m64_b512_sample", "m96_b512_blend", "m48_b1000_ragged"])
def test_numpy_oracle_matches_reference(name):
    g, x, C = load_case(name)
    codes, im = pq_oracle.quantize(x, C, True, EPS, ITERS, return_intermediates=True)
    assert np.array_equal(codes.astype(np.uint8), g["codes_constrained"])
    assert im["flags"] == 0
    idx = g["samp_idx"]
    assert np.array_equal(im["dist"][idx[:, 0], idx[:, 1], idx[:, 2]].view(np.uint32), g["samp_dist_bits"])
    assert np.array_equal(im["centred"][idx[:, 0], idx[:, 1], idx[:, 2]].view(np.uint32), g["samp_centred_bits"])
    assert np.array_equal(im["mx"].view(np.uint32), g["mx_bits"])
    assert np.array_equal(pq_oracle.quantize(x, C, False).astype(np.uint8), g["codes_nearest"])
    dec = pq_oracle.decode(g["codes_constrained"], C)
    assert zlib.crc32(np.ascontiguousarray(dec).tobytes()) == int(g["decode_crc"])
    assert zlib.crc32(c_oracle.decode(g["codes_constrained"], C).tobytes()) == int(g["decode_crc"])


@pytest.mark.parametrize("name,shards", [("m8_b2048_sample", 2), ("m8_b2048_sample", 4)])
def test_numpy_oracle_sharded_equals_unsharded(name, shards):
    """The reference's own gloo run (shard{2,4}_equal in the fixture) says sharded == unsharded;
    the oracle's simulated ranks must say the same."""
    g, x, C = load_case(name)
    assert int(g[f"shard{shards}_equal"]) == 1
    codes = pq_oracle.quantize(x, C, True, EPS, ITERS, shards=shards)
    assert np.array_equal(codes.astype(np.uint8), g["codes_constrained"])


def test_logdomain_restatement_gives_reference_codes():
    """The potential form the HIP kernels use is algebraically the reference's matrix form."""
    g, x, C = load_case("m8_b300_gauss")
    d = pq_oracle.dist_table(x, C)
    dc = pq_oracle.centre(d, *pq_oracle.minmax_per_m(d))
    codes = pq_oracle.sinkhorn_codes_logdomain(dc, EPS, ITERS)
    assert np.array_equal(codes.astype(np.uint8), g["codes_constrained"])


def test_sinkhorn_uniform_cost_gives_uniform_plan():
    out = [np.zeros((2, 256, 512))]
    Q = pq_oracle.sinkhorn_q(out, 0.05, 3)[0]
    assert np.allclose(Q, 1.0 / 256) and np.allclose(Q.sum(1), 1.0)


def test_balance_of_constrained_codes():
    g, _, _ = load_case("m48_b6144_sample")
    h = pq_oracle.code_histogram(g["codes_constrained"])
    assert h.sum() == 6144 * 48
    assert h.min() >= 14 and h.max() <= 34          # ideal 24 per centroid
    hn = pq_oracle.code_histogram(g["codes_nearest"])
    assert hn.max() > 100                           # unconstrained codes are badly unbalanced


def test_adc_identity_against_reference_decode():
    """sum_m LUT[m][code] == <q, decode(code)> (decode pinned by the reference): the brute-force
    fixture built on the reference's decode must be reproduced by the LUT oracle."""
    import os
    from conftest import GOLDEN
    from oracle import synth
    g = np.load(os.path.join(GOLDEN, "adc_m48_n20000.npz"))
    M, N, nq, k = int(g["M"]), int(g["N"]), int(g["nq"]), int(g["k"])
    C = synth.gaussian(777, (M, 256, 768 // M))
    codes = synth.uniform_codes(778, N, M)
    q = synth.gaussian(779, (nq, 768))
    assert zlib.crc32(np.ascontiguousarray(pq_oracle.decode(codes, C)).tobytes()) == int(g["recon_crc"])
    scores, ids = pq_oracle.adc_search(q, C, codes, k)
    np.testing.assert_allclose(scores, g["top_scores"], rtol=0, atol=2e-4)
    # identical sets except for swaps among near-equal scores at the boundary
    for r in range(nq):
        assert len(set(ids[r].tolist()) & set(g["top_ids"][r].tolist())) >= k - 2
    cs, ci = c_oracle.adc_search(codes, C, q, k)
    assert np.array_equal(ci, ids) and np.array_equal(cs.view(np.uint32), scores.view(np.uint32))


@pytest.mark.parametrize("M,N", [(24, 100000), (64, 100000), (96, 120000)])
def test_adc_fixtures_from_the_reference_decode_other_widths(M, N):
    """Round-3 fixtures (oracle/gen_golden.py --extra): exact fp64 <q, decode(codes)> on the reference's decode, k = 1000,
    N >= 1e5, M in {24, 64, 96}.  The C restatement's LUT sums agree to fp32 summation error; ids differ only where two
    scores are closer than that."""
    import os
    from conftest import GOLDEN
    from oracle import synth
    g = np.load(os.path.join(GOLDEN, f"adc_m{M}_n{N}.npz"))
    nq, k, seed = int(g["nq"]), int(g["k"]), int(g["seed"])
    C = synth.gaussian(seed, (M, 256, 768 // M))
    codes = synth.uniform_codes(seed + 1, N, M)
    q = synth.gaussian(seed + 2, (nq, 768))
    cs, ci = c_oracle.adc_search(codes, C, q, k)
    np.testing.assert_allclose(cs, g["top_scores"], rtol=0, atol=1e-3)
    same = ci == g["top_ids"].astype(np.int64)
    assert same.mean() > 0.98
    for qi, r in zip(*np.nonzero(~same)):
        assert abs(float(cs[qi, r]) - float(g["top_scores_f64"][qi, r])) < 1e-3
    for r in range(nq):
        assert len(set(ci[r].tolist()) & set(g["top_ids"][r].tolist())) >= k - 3


def test_forward_fixture_m96_codes_from_the_reference():
    """forward() of the reference at M = 96 (dsub 8): on the reference's own continuous embeddings both C / numpy
    restatements give the reference's codes, nearest and constrained."""
    import os
    from conftest import GOLDEN
    from oracle import synth
    g = np.load(os.path.join(GOLDEN, "forward_m96_b512.npz"))
    seed = int(g["seed"])
    table = synth.clustered_embeddings(seed, 512)
    C = synth.sample_centroids(seed + 1, table, 96)
    near, _ = c_oracle.quantize(g["ip_continuous"], C, False)
    assert np.array_equal(near, g["ip_codes"])
    con, _ = c_oracle.quantize(g["ip_continuous"], C, True, 0.003, 100)
    assert np.array_equal(con, g["ip_codes_constrained"])
    assert zlib.crc32(np.ascontiguousarray(pq_oracle.decode(g["ip_codes"].astype(np.int64), C)).tobytes()) == int(g["ip_quantized_crc"])


def test_mrr_at_k():
    ranked = np.array([[5, 3, 9], [1, 2, 3], [7, 8, 9]])
    assert pq_oracle.mrr_at_k(ranked, [{3}, {1}, {4}], 10) == round((0.5 + 1.0 + 0.0) / 3, 5)
    assert pq_oracle.mrr_at_k(ranked, [{9}, {1}, {4}], 2) == round((0.0 + 1.0 + 0.0) / 3, 5)


def _correlated(seed, n, D):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((32, D), dtype=np.float32) * 2.0
    x = cent[rng.integers(0, 32, n)] + rng.standard_normal((n, D), dtype=np.float32)
    mix = rng.standard_normal((D, D), dtype=np.float32) / np.sqrt(D)
    return np.ascontiguousarray((x @ mix).astype(np.float32))


def test_warmup_procedure_oracle_lowers_the_error_and_keeps_the_rotation_orthogonal():
    """oracle train_pq / train_opq (train/run_warmup.py:92-113 restated): the C restatement of the assignment plugs into
    the numpy procedure (same codes as the numpy one), Lloyd lowers the reconstruction error, OPQ lowers it further on
    correlated data and returns an orthogonal matrix."""
    D, M, n = 64, 8, 4096
    x = _correlated(11, n, D)
    cq = lambda xx, cc: c_oracle.quantize(xx, cc, False)[0]
    C0, mse0 = pq_oracle.train_pq(x, M, 0, quantize_fn=cq)
    C3, mse3 = pq_oracle.train_pq(x, M, 3, quantize_fn=cq)
    C3n, mse3n = pq_oracle.train_pq(x[:1024], M, 1)                      # numpy assignment == C assignment
    C3c, mse3c = pq_oracle.train_pq(x[:1024], M, 1, quantize_fn=cq)
    assert np.array_equal(C3n, C3c) and mse3n == mse3c
    assert mse3 < mse0
    R0 = np.linalg.qr(np.random.default_rng(3).standard_normal((D, D)))[0].astype(np.float32)
    R, hist = pq_oracle.train_opq(x, M, R0, 4, 3, 2, quantize_fn=cq)
    assert np.abs(R @ R.T - np.eye(D)).max() < 1e-5
    assert hist[-1] < hist[0]
    # empty clusters: Faiss's split_clusters — a seeded, size-proportional donor (not the arg-max), +-1/1024 perturbation
    C = C3.copy()
    cnt = np.full((M, 256), 16, np.int64)
    cnt[2, 7] = 0
    cnt[2, 9] = 100
    assert pq_oracle.reseed_empty(C, cnt) == 1
    changed = [k for k in range(256) if not np.array_equal(C[2, k], C3[2, k])]
    assert len(changed) == 2 and 7 in changed
    donor = [k for k in changed if k != 7][0]
    assert np.allclose(C[2, 7] + C[2, donor], 2 * C3[2, donor], rtol=1e-6) and not np.array_equal(C[2, 7], C[2, donor])
    # the walk is the published one: std::mt19937(1234), accept cluster j with probability (size_j - 1) / (n - k)
    rs = np.random.RandomState(1234)
    n, j = int(cnt[2].sum()), 0
    while not (np.float32(rs._bit_generator.random_raw()) / np.float32(4294967295.0)
               < np.float32((float(cnt[2, j]) - 1.0) / float(np.float32(n - 256)))):
        j = (j + 1) % 256
    assert j == donor
    assert np.array_equal(C[[0, 1, 3]], C3[[0, 1, 3]])


@pytest.mark.parametrize("M", [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 768])
def test_distance_order_equals_torch_cpu_for_every_divisor_of_768(M):
    """The reference's distance expression (modeling_repconc.py:50) evaluated by torch-CPU — the machine the fixtures come
    from — against both restatements, every bit, for every MCQ_M the reference accepts: 8-wide vectors with four ILP
    accumulators, the 16-round cascade (dsub = 768) and the scalar path below 8 floats (dsub = 6)."""
    import torch
    from oracle import c_oracle, pq_oracle
    B, dsub = 48, 768 // M
    g = torch.Generator().manual_seed(900 + M)
    x = torch.randn(B, 768, generator=g)
    C = torch.randn(M, 256, dsub, generator=g)
    d = ((x.reshape(B, M, 1, -1).transpose(0, 1) - C.unsqueeze(1)) ** 2).sum(-1).numpy()
    assert np.array_equal(pq_oracle.dist_table(x.numpy(), C.numpy()).view(np.uint32), d.view(np.uint32))
    assert np.array_equal(c_oracle.dist_table(x.numpy(), C.numpy()).view(np.uint32), d.view(np.uint32))

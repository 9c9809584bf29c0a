"""Parity of the HIP path (through the C ABI) with the reference's golden vectors and the oracle.
Bit-exact for codes and for the fp32 distance / centring stage."""
import ctypes
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_cases, load_case
from oracle import c_oracle, pq_oracle, synth

pytestmark = pytest.mark.gpu
EPS, ITERS = 0.003, 100
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("name", golden_cases())
def test_nearest_codes_bit_exact(name):
    from repconc_amd import ops
    g, x, C = load_case(name)
    for dt in (torch.uint8, torch.int64):
        for method in ("exact", "mfma"):
            codes = ops.assign_nearest(_t(x), _t(C), dt, method=method)
            assert codes.dtype == dt
            assert np.array_equal(codes.cpu().numpy().astype(np.uint8), g["codes_nearest"]), method


@pytest.mark.parametrize("name", golden_cases())
def test_constrained_codes_bit_exact(name):
    from repconc_amd import ops
    g, x, C = load_case(name)
    codes, flags = ops.assign_sinkhorn(_t(x), _t(C), EPS, ITERS)
    assert int(flags.item()) == 0
    got = codes.cpu().numpy().astype(np.uint8)
    assert int((got != g["codes_constrained"]).sum()) == 0


@pytest.mark.parametrize("name", golden_cases())
def test_fp32_stage_bitwise(name):
    from repconc_amd import ops
    g, x, C = load_case(name)
    d, mm = ops.dist_table(_t(x), _t(C))
    M = d.shape[0]
    idx = g["samp_idx"]
    dn = d.cpu().numpy()
    assert np.array_equal(dn[idx[:, 0], idx[:, 1], idx[:, 2]].view(np.uint32), g["samp_dist_bits"])
    mmn = mm.cpu().numpy()
    assert np.array_equal(mmn[:M].view(np.uint32), g["mx_bits"])
    assert np.array_equal(mmn[M:].view(np.uint32), g["mn_bits"])
    ops.centre_(d, mm)
    dn = d.cpu().numpy()
    assert np.array_equal(dn[idx[:, 0], idx[:, 1], idx[:, 2]].view(np.uint32), g["samp_centred_bits"])
    # whole table against the C oracle, every bit
    ref = c_oracle.dist_table(x, C)
    c_oracle.centre_(ref, c_oracle.minmax(ref))
    assert np.array_equal(dn.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("name,shards", [("m8_b2048_sample", 2), ("m8_b2048_sample", 4), ("m48_b1024_sample", 2),
                                         ("m48_b1024_sample", 4), ("m48_b6144_sample", 8)])
def test_virtual_shards_equal_unsharded(name, shards):
    """The collective choreography (rank-ordered sum of all-gathered row sums) on G virtual ranks."""
    from repconc_amd.sharded import assign_sinkhorn_virtual
    g, x, C = load_case(name)
    B = x.shape[0]
    bl = B // shards
    xs = [_t(x[r * bl:(r + 1) * bl]) for r in range(shards)]
    for split in (False, True):      # split: the two halves of M solved separately, as the multi-rank driver does
        codes, flags = assign_sinkhorn_virtual(xs, _t(C), EPS, ITERS, split=split)
        got = torch.cat(codes, 0).cpu().numpy().astype(np.uint8)
        assert all(int(f.item()) == 0 for f in flags)
        assert np.array_equal(got, g["codes_constrained"])


def test_strided_and_half_inputs():
    from repconc_amd import ops
    g, x, C = load_case("m48_b1024_sample")
    wide = torch.zeros((1024, 1024), dtype=torch.float32, device=DEV)
    wide[:, :768] = _t(x)
    codes = ops.assign_nearest(wide[:, :768], _t(C), torch.uint8)           # ldx = 1024
    assert np.array_equal(codes.cpu().numpy(), g["codes_nearest"])
    xh = _t(x).half()                                                       # promoted to fp32 like the reference
    want = pq_oracle.quantize(xh.float().cpu().numpy(), C, False).astype(np.uint8)
    assert np.array_equal(ops.assign_nearest(xh, _t(C), torch.uint8).cpu().numpy(), want)


def test_empty_and_tiny_batches():
    from repconc_amd import ops
    _, x, C = load_case("m48_b1024_sample")
    assert ops.assign_nearest(_t(x[:0]), _t(C)).shape == (0, 48)
    codes, _ = ops.assign_sinkhorn(_t(x[:0]), _t(C), EPS, ITERS)
    assert codes.shape == (0, 48)
    # B == 1: exact K-way tie in the reference -> code 0 everywhere
    got, _ = ops.assign_sinkhorn(_t(x[:1]), _t(C), EPS, ITERS, torch.uint8)
    assert not got.cpu().numpy().any()
    for B in (3, 100, 255):     # B < K: the reference's plan is full of fp64-level ties (see oracle)
        want, im = pq_oracle.quantize(x[:B], C, True, EPS, ITERS, return_intermediates=True)
        got, fl = ops.assign_sinkhorn(_t(x[:B]), _t(C), EPS, ITERS, torch.uint8)
        ok, n = pq_oracle.codes_equal_up_to_fp64_ties(got.cpu().numpy(), want, im["Q"])
        assert ok and int(fl.item()) == 0
        assert np.array_equal(ops.assign_nearest(_t(x[:B]), _t(C), torch.uint8).cpu().numpy(),
                              c_oracle.quantize(x[:B], C, False)[0])


@pytest.mark.parametrize("iters", [1, 2, 7, 30])
def test_iteration_count_is_part_of_the_spec(iters):
    """Sinkhorn is not run to convergence: T iterations means exactly T.
    With few iterations and B ~ K the reference's plan has exact fp64 ties, hence the tie-aware check;
    at B >> K the codes must be identical."""
    from repconc_amd import ops
    _, x, C = load_case("m8_b300_gauss")
    want, im = pq_oracle.quantize(x, C, True, EPS, iters, return_intermediates=True)
    got, _ = ops.assign_sinkhorn(_t(x), _t(C), EPS, iters, torch.uint8)
    ok, n = pq_oracle.codes_equal_up_to_fp64_ties(got.cpu().numpy(), want, im["Q"])
    assert ok
    full = pq_oracle.quantize(x, C, True, EPS, ITERS)
    assert (got.cpu().numpy() != full).mean() > 0.001 or iters >= 30     # fewer iterations = different codes
    _, x2, C2 = load_case("m8_b2048_sample")
    want2, _ = c_oracle.quantize(x2, C2, True, EPS, iters)
    got2, _ = ops.assign_sinkhorn(_t(x2), _t(C2), EPS, iters, torch.uint8)
    assert np.array_equal(got2.cpu().numpy(), want2)


def test_other_epsilon():
    from repconc_amd import ops
    _, x, C = load_case("m24_b1024_sample")
    for eps in (0.01, 0.05):
        want, _ = c_oracle.quantize(x, C, True, eps, 30)
        got, _ = ops.assign_sinkhorn(_t(x), _t(C), eps, 30, torch.uint8)
        assert np.array_equal(got.cpu().numpy(), want)


def test_decode_forward_backward():
    from repconc_amd import ops
    g, x, C = load_case("m48_b1024_sample")
    codes = _t(g["codes_constrained"])
    out = ops.decode(codes, _t(C))
    assert zlib.crc32(out.cpu().numpy().tobytes()) == int(g["decode_crc"])
    out64 = ops.decode(codes.long(), _t(C))
    assert torch.equal(out, out64)
    Cp = _t(C).clone().requires_grad_(True)
    go = _t(synth.gaussian(5, (1024, 768)))
    ops.decode(codes.long(), Cp).backward(go)
    want = pq_oracle.decode_bwd(g["codes_constrained"], go.cpu().numpy(), 48, 256)
    np.testing.assert_allclose(Cp.grad.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    # ... and against the gradient the REFERENCE's autograd produced for these inputs (gen_golden.py --aux); fp32 atomic
    # scatter-add in any order against a sequential index_put: 1e-5
    aux = np.load(os.path.join(os.path.dirname(__file__), "golden", "aux_m48_b1024.npz"))
    np.testing.assert_allclose(Cp.grad.cpu().numpy(), aux["decode_grad"], rtol=1e-5, atol=1e-5)
    # numpy variant of the module-level decode
    from repconc_amd.models.repconc import decode
    assert zlib.crc32(np.ascontiguousarray(decode(g["codes_constrained"], C)).tobytes()) == int(g["decode_crc"])


def test_histogram_and_balance():
    from repconc_amd import ops
    g, _, _ = load_case("m48_b6144_sample")
    for arr in (g["codes_constrained"], g["codes_nearest"]):
        h = ops.code_hist(_t(arr)).cpu().numpy()
        assert np.array_equal(h, pq_oracle.code_histogram(arr))
        h64 = ops.code_hist(_t(arr).long()).cpu().numpy()
        assert np.array_equal(h64, h)


def test_kmeans_stats_and_update():
    from repconc_amd import ops
    for name in ("m48_b1024_sample", "m8_b2048_sample", "m96_b512_blend"):
        g, x, C = load_case(name)
        M = C.shape[0]
        codes = g["codes_nearest"]
        sums, cnt = ops.kmeans_stats(_t(x), _t(codes))
        ws, wc = pq_oracle.kmeans_stats(x, codes, M)
        assert np.array_equal(cnt.cpu().numpy(), wc)
        np.testing.assert_allclose(sums.cpu().numpy(), ws, rtol=1e-12, atol=1e-12)
        newC = ops.kmeans_update_(sums, cnt, _t(C).clone()).cpu().numpy()
        want = pq_oracle.kmeans_update(ws, wc, C)
        np.testing.assert_allclose(newC, want, rtol=1e-6, atol=1e-7)
        # shard-sum of statistics == statistics of the whole (the all-gather path of §8e)
        h = x.shape[0] // 2
        s2, c2 = ops.kmeans_stats(_t(x[:h]), _t(codes[:h]))
        s2, c2 = ops.kmeans_stats(_t(x[h:]), _t(codes[h:]), s2, c2)
        assert torch.equal(c2, cnt)
        np.testing.assert_allclose(s2.cpu().numpy(), ws, rtol=1e-12, atol=1e-12)


def test_normalize_centroids():
    from repconc_amd import ops
    C = synth.gaussian(3, (48, 256, 16))
    got = ops.normalize_centroids_(_t(C).clone()).cpu().numpy()
    np.testing.assert_allclose(got, pq_oracle.normalize_centroids(C), rtol=1e-6, atol=1e-7)
    aux = np.load(os.path.join(os.path.dirname(__file__), "golden", "aux_m48_b1024.npz"))      # the reference's normalize_centrodis
    np.testing.assert_allclose(got, aux["normalized"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------- ADC
def _adc_case(M, N, nq, seed=0):
    C = synth.gaussian(seed + 1, (M, 256, 768 // M))
    codes = synth.uniform_codes(seed + 2, N, M)
    q = synth.gaussian(seed + 3, (nq, 768))
    return C, codes, q


def test_adc_lut_bitwise():
    from repconc_amd import ops
    C, _, q = _adc_case(48, 10, 7)
    lut = ops.adc_lut(_t(C), _t(q)).cpu().numpy()
    assert np.array_equal(lut.view(np.uint32), pq_oracle.adc_lut(q, C).view(np.uint32))


@pytest.mark.parametrize("M,N,nq,k", [(48, 20000, 16, 100), (48, 50000, 5, 1000), (96, 40000, 3, 10),
                                      (24, 70000, 9, 200), (8, 3000, 4, 50), (64, 33000, 2, 10),
                                      (48, 500, 3, 1000), (12, 100000, 3, 10), (16, 5000, 1, 1), (32, 17000, 6, 64)])
def test_adc_search_matches_oracle_exactly(M, N, nq, k):
    from repconc_amd import ops
    C, codes, q = _adc_case(M, N, nq, seed=M + N)
    scores, ids = ops.adc_search(_t(codes), _t(C), _t(q), k)
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    assert np.array_equal(ids.cpu().numpy(), wi)
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), ws.view(np.uint32))


def test_adc_golden_fixture_and_duplicates():
    from repconc_amd import ops
    g = np.load(os.path.join(GOLDEN, "adc_m48_n20000.npz"))
    C = synth.gaussian(777, (48, 256, 16))
    codes = synth.uniform_codes(778, 20000, 48)
    q = synth.gaussian(779, (16, 768))
    scores, ids = ops.adc_search(_t(codes), _t(C), _t(q), 100)
    np.testing.assert_allclose(scores.cpu().numpy(), g["top_scores"], rtol=0, atol=2e-4)
    # ids against the fixture too (its ranking comes from <q, decode(codes)> in a different summation order, so two rows
    # whose scores differ by less than the rounding of that GEMM may swap): every id of the fixture's top-100 is either
    # returned at the same rank, or sits within 4e-4 of the score we return at that rank
    got_i, want_i, want_s = ids.cpu().numpy(), g["top_ids"].astype(np.int64), g["top_scores"]
    same = got_i == want_i
    assert same.mean() > 0.97, float(same.mean())
    lut_h = pq_oracle.adc_lut(q, C)
    for qi, r in zip(*np.nonzero(~same)):
        mine = pq_oracle.adc_scores(lut_h[qi:qi + 1], codes[[got_i[qi, r], want_i[qi, r]]])[0]
        assert abs(float(mine[0]) - float(mine[1])) < 4e-4 and abs(float(mine[1]) - float(want_s[qi, r])) < 4e-4
    # duplicated rows: equal scores must come back in ascending id order
    dup = np.concatenate([codes[:5000]] * 4, 0)
    s2, i2 = ops.adc_search(_t(dup), _t(C), _t(q), 64, id_offset=1000)
    ws, wi = c_oracle.adc_search(dup, C, q, 64)
    assert np.array_equal(i2.cpu().numpy(), wi + 1000)


@pytest.mark.parametrize("M,N", [(24, 100000), (64, 100000), (96, 120000)])
def test_adc_fixtures_from_the_reference_decode_other_widths(M, N):
    """oracle/gen_golden.py --extra: exact <q, decode(codes)> on the REFERENCE's decode, k = 1000, M in {24, 64, 96}.  The
    HIP search equals the C oracle bit for bit and the reference-derived scores to fp32 summation error."""
    from repconc_amd import ops
    g = np.load(os.path.join(GOLDEN, f"adc_m{M}_n{N}.npz"))
    nq, k, seed = int(g["nq"]), int(g["k"]), int(g["seed"])
    C = synth.gaussian(seed, (M, 256, 768 // M))
    codes = synth.uniform_codes(seed + 1, N, M)
    q = synth.gaussian(seed + 2, (nq, 768))
    scores, ids = ops.adc_search(_t(codes), _t(C), _t(q), k)
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    assert np.array_equal(ids.cpu().numpy(), wi) and np.array_equal(scores.cpu().numpy().view(np.uint32), ws.view(np.uint32))
    np.testing.assert_allclose(scores.cpu().numpy(), g["top_scores"], rtol=0, atol=1e-3)
    assert (ids.cpu().numpy() == g["top_ids"].astype(np.int64)).mean() > 0.98


def test_forward_fixture_m96_from_the_reference():
    """The reference's forward() at M = 96: module output codes on the reference's continuous embeddings, nearest and
    constrained, and decode CRC."""
    from types import SimpleNamespace
    from repconc_amd.models.repconc import RepCONC
    g = np.load(os.path.join(GOLDEN, "forward_m96_b512.npz"))
    seed = int(g["seed"])
    table = torch.from_numpy(synth.clustered_embeddings(seed, 512))
    C = synth.sample_centroids(seed + 1, table.numpy(), 96)
    cfg = SimpleNamespace(MCQ_M=96, MCQ_K=256, hidden_size=768, similarity_metric="METRIC_IP")
    model = RepCONC(cfg, _TableEncoder(table), False, EPS, ITERS).to(DEV)
    rot = torch.linalg.qr(torch.from_numpy(synth.gaussian(int(g["rotation_seed"]), (768, 768))))[0].contiguous()
    with torch.no_grad():
        model.centroids.copy_(_t(C))
        model.rotation.copy_(rot.to(DEV))
    ids = torch.arange(512, device=DEV)[:, None].repeat(1, 4)
    out = model(ids, torch.ones_like(ids), return_code=True, return_quantized_embedding=True)
    np.testing.assert_allclose(out.continuous_embeds.cpu().numpy(), g["ip_continuous"], rtol=1e-4, atol=1e-4)
    cont = _t(g["ip_continuous"])
    assert np.array_equal(model.quantize(cont).cpu().numpy().astype(np.uint8), g["ip_codes"])
    model.use_constraint = True
    assert np.array_equal(model.quantize(cont).cpu().numpy().astype(np.uint8), g["ip_codes_constrained"])
    assert zlib.crc32(model.decode(_t(g["ip_codes"]).long()).detach().cpu().numpy().tobytes()) == int(g["ip_quantized_crc"])


@pytest.mark.parametrize("M", [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 768])
def test_every_divisor_of_the_hidden_size_is_a_valid_mcq_m(M):
    """modeling_repconc.py:41 asserts only hidden_size % MCQ_M == 0.  RepCONC(config) constructs for every divisor of 768;
    nearest and constrained codes and decode() equal the oracle (whose distance order is checked against torch-CPU for all
    18 widths in tests/test_oracle_golden.py) — the recipes' widths on the specialised kernels, the others on the
    run-time-width kernels."""
    from types import SimpleNamespace
    from repconc_amd.models.repconc import RepCONC
    B = 512
    x = synth.clustered_embeddings(5000 + M, B)
    C = synth.sample_centroids(6000 + M, x, M)
    cfg = SimpleNamespace(MCQ_M=M, MCQ_K=256, hidden_size=768, similarity_metric="METRIC_IP")
    model = RepCONC(cfg, _TableEncoder(torch.zeros(1, 768)), False, EPS, ITERS).to(DEV)
    with torch.no_grad():
        model.centroids.copy_(_t(C))
    near = model.quantize(_t(x)).cpu().numpy()
    assert near.shape == (B, M) and np.array_equal(near.astype(np.uint8), c_oracle.quantize(x, C, False)[0])
    model.use_constraint = True
    got = model.quantize(_t(x)).cpu().numpy().astype(np.uint8)
    want = c_oracle.quantize(x, C, True, EPS, ITERS)[0]
    if not np.array_equal(got, want):              # narrow sub-vectors: a plan column tied to the last ulp of exp may differ
        wn, inter = pq_oracle.quantize(x, C, True, EPS, ITERS, return_intermediates=True)
        assert pq_oracle.codes_equal_up_to_fp64_ties(got, wn.astype(np.uint8), inter["Q"])
    dec = model.decode(_t(near)).detach().cpu().numpy()
    assert np.array_equal(dec, c_oracle.decode(near.astype(np.uint8), C))


def test_quantize_logs_and_returns_when_the_solve_flags_a_range_problem(caplog):
    """modeling_repconc.py:64-65: the reference logs "Sinkhorn Algorithm returns nan/inf values." and returns its codes.  An
    sk_epsilon below the kernels' range (where the reference's exp(1/eps) has long overflowed) does the same here — no raise."""
    import logging
    from types import SimpleNamespace
    from repconc_amd.models.repconc import RepCONC
    g, x, C = load_case("m8_b300_gauss")
    cfg = SimpleNamespace(MCQ_M=8, MCQ_K=256, hidden_size=768, similarity_metric="METRIC_IP")
    model = RepCONC(cfg, _TableEncoder(torch.zeros(1, 768)), True, 1e-5, 5).to(DEV)
    with torch.no_grad():
        model.centroids.copy_(_t(C))
    with caplog.at_level(logging.WARNING):
        codes = model.quantize(_t(x))
    assert codes.shape == (300, 8) and codes.dtype == torch.int64
    assert int(codes.min()) >= 0 and int(codes.max()) < 256
    assert any("nan/inf" in r.getMessage() for r in caplog.records)


def test_config0_exact_shape_10000_m8_against_the_oracle_and_the_reference():
    """BASELINE configs[0] at its exact shape: x [10 000, 768] seed 20220, M = 8, centroids = rows of x at
    rng(20221).permutation(N)[:256], ONE batch of 10 000 rows, eps 0.003, 100 iterations — constrained and nearest codes
    against the C oracle."""
    from repconc_amd import ops
    N, M = 10000, 8
    x = np.random.default_rng(20220).standard_normal((N, 768), dtype=np.float32)
    C = np.ascontiguousarray(x[np.random.default_rng(20221).permutation(N)[:256]].reshape(256, M, 768 // M).transpose(1, 0, 2))
    got, flags = ops.assign_sinkhorn(_t(x), _t(C), EPS, ITERS, torch.uint8)
    want, _ = c_oracle.quantize(x, C, True, EPS, ITERS)
    assert int(flags.item()) == 0 and np.array_equal(got.cpu().numpy(), want)
    near = ops.assign_nearest(_t(x), _t(C), torch.uint8)
    assert np.array_equal(near.cpu().numpy(), c_oracle.quantize(x, C, False)[0])
    hist = np.bincount(got.cpu().numpy()[:, 0], minlength=256)
    assert 30 <= hist.min() and hist.max() <= 48              # 35 .. 43 around the ideal 39.06
    # ... and against the REFERENCE itself on these inputs (oracle/gen_golden.py --config0)
    import zlib
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "config0_b10000_m8.npz"))
    assert zlib.crc32(x.tobytes()) == int(g["x_crc"]) and zlib.crc32(C.tobytes()) == int(g["centroids_crc"])
    assert np.array_equal(got.cpu().numpy(), g["codes_constrained"]) and np.array_equal(near.cpu().numpy(), g["codes_nearest"])


def test_adc_large_index_properties():
    """BASELINE-size index (8.84M x 48): top-k must be sorted, contain planted winners, and agree
    with an exact rescoring of the returned ids."""
    from repconc_amd import ops
    N, M, nq, k = 8841823, 48, 8, 1000
    C = _t(synth.gaussian(11, (M, 256, 16)))
    gen = torch.Generator(device=DEV).manual_seed(12)
    codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device=DEV, generator=gen)
    q = _t(synth.gaussian(13, (nq, 768)))
    lut = ops.adc_lut(C, q)                                   # [nq, M, 256]
    best = lut.argmax(dim=2).to(torch.uint8)                  # per query the best code per m
    plant = torch.tensor([17, 4_000_000, N - 1], device=DEV)
    for r in range(3):
        codes[plant[r]] = best[r]
    scores, ids = ops.adc_search(codes, C, q, k)
    assert bool((scores[:, :-1] >= scores[:, 1:]).all())
    for r in range(3):
        assert int(ids[r, 0]) == int(plant[r])
    # exact rescoring of what came back
    rows = codes[ids.reshape(-1)].long().reshape(nq, k, M)
    re = torch.zeros((nq, k), dtype=torch.float32, device=DEV)
    for m in range(M):
        re = re + torch.gather(lut[:, m, :], 1, rows[:, :, m])
    assert torch.equal(re, scores)
    # nothing outside the result beats the k-th score (checked on a 1M-row slice)
    sl = codes[1_000_000:2_000_000].long()
    s = torch.zeros((nq, sl.shape[0]), dtype=torch.float32, device=DEV)
    for m in range(M):
        s = s + lut[:, m, :][:, sl[:, m]]
    kth = scores[:, -1:]
    inside = ((ids >= 1_000_000) & (ids < 2_000_000)).sum(1)
    assert torch.equal((s >= kth).sum(1) >= inside, torch.ones(nq, dtype=torch.bool, device=DEV))
    assert bool(((s > kth).sum(1) <= inside).all())


@pytest.mark.parametrize("M", [48, 96])
def test_adc_full_size_index_against_the_oracle(M):
    """The flat search at the BASELINE index size (8 841 823 rows, k = 1000; M = 48: headline 2, M = 96: configs[3]'s flat
    leg) against the C restatement of Faiss's IndexPQ search over the WHOLE index: ids and score bits of 8 queries
    (the full-size tests compared properties only; the oracle comparison lived in bench.py)."""
    from repconc_amd import ops
    N, nq, k = 8841823, 8, 1000
    C = synth.gaussian(3100 + M, (M, 256, 768 // M))
    gen = torch.Generator(device=DEV).manual_seed(3200 + M)
    codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device=DEV, generator=gen)
    q = synth.gaussian(3300 + M, (nq, 768))
    scores, ids = ops.adc_search(codes, _t(C), _t(q), k)
    want_s, want_i = c_oracle.adc_search(codes.cpu().numpy(), C, q, k)
    assert np.array_equal(ids.cpu().numpy(), want_i)
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), want_s.view(np.uint32))


def test_ivf_baseline_size_properties():
    """BASELINE configs[3] at its full size (8.84 M x 96 B in 5000 cells, nprobe 128, k = 1000): the list-centric search
    (pipelined 8-bit screen, streams, bucket pass, exact rescoring) equals the per-query exact scan — an independent
    kernel chain — in ids and score bits; scores are sorted, ids lie in probed cells, a planted best row is found."""
    from repconc_amd import ops
    from repconc_amd.ivf import IVFPQIndex
    N, M, nlist, nq, k, nprobe = 8841823, 96, 5000, 40, 1000, 128
    gen = torch.Generator(device=DEV).manual_seed(4242)
    codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device=DEV, generator=gen)
    cells = torch.randint(0, nlist, (N,), device=DEV, generator=gen)
    C = _t(synth.gaussian(4243, (M, 256, 768 // M)))
    ivf = IVFPQIndex(768, M, nlist, device=DEV)
    ivf.set_centroids(C)
    ivf.coarse = _t(synth.gaussian(4244, (nlist, 768)))
    q = _t(synth.gaussian(4245, (nq, 768)))
    probes = ivf.probe(q, nprobe, ordered=False)
    # plant: the best possible row for query 0 inside one of ITS probed cells
    best = ops.adc_lut(C, q[:1]).argmax(dim=2).to(torch.uint8)[0]
    victim = int(torch.nonzero(cells == probes[0, 5])[3])
    codes[victim] = best
    ivf.set_lists(codes, cells)
    s1, i1 = ivf.search(q, k, nprobe, method="lists")
    s2, i2 = ivf.search(q, k, nprobe, method="scan")
    assert torch.equal(i1, i2) and torch.equal(s1, s2)
    assert bool((s1[:, :-1] >= s1[:, 1:]).all()) and int(i1[0, 0]) == victim
    hit_cells = cells[i1.reshape(-1)].reshape(nq, k)
    assert bool((hit_cells.unsqueeze(2) == probes.long().unsqueeze(1)).any(2).all())
    # ... and against the ORACLE at this size (brute-force restatement: its own probe selection, flat ADC arithmetic over the
    # rows of the probed cells, (score desc, id asc)): ids and score bits of 4 queries
    want_s, want_i = pq_oracle.ivf_search(q[:4].cpu().numpy(), C.cpu().numpy(), codes.cpu().numpy(), cells.cpu().numpy(),
                                          ivf.coarse.cpu().numpy(), k, nprobe)
    assert np.array_equal(i1[:4].cpu().numpy(), want_i)
    assert np.array_equal(s1[:4].cpu().numpy().view(np.uint32), want_s.view(np.uint32))
    # both widths of the screen at this size
    for method in ("lists8", "lists16"):
        sw, iw = ivf.search(q, k, nprobe, method=method)
        assert torch.equal(iw, i2) and torch.equal(sw, s2), method


# ------------------------------------------------------------------------------------------- model API
class _TableEncoder(torch.nn.Module):
    def __init__(self, table):
        super().__init__()
        self.register_buffer("table", table)
        from types import SimpleNamespace
        self.config = SimpleNamespace(hidden_size=table.shape[1])

    def forward(self, input_ids, attention_mask):
        return self.table[input_ids[:, 0]]


def test_repconc_module_forward_and_quantize():
    from types import SimpleNamespace
    from repconc_amd.models.repconc import RepCONC
    g = np.load(os.path.join(GOLDEN, "forward_m48_b256.npz"))
    table = torch.from_numpy(synth.clustered_embeddings(4242, 256))
    C = synth.sample_centroids(4243, table.numpy(), 48)
    cfg = SimpleNamespace(MCQ_M=48, MCQ_K=256, hidden_size=768, similarity_metric="METRIC_IP")
    model = RepCONC(cfg, _TableEncoder(table), False, EPS, ITERS).to(DEV)
    assert set(k for k in model.state_dict() if not k.startswith("dense_encoder")) == {"rotation", "centroids"}
    with torch.no_grad():
        model.centroids.copy_(_t(C))
        model.rotation.copy_(_t(g["rotation"]))
    ids = torch.arange(256, device=DEV)[:, None].repeat(1, 4)
    out = model(ids, torch.ones_like(ids), return_code=True, return_quantized_embedding=True)
    np.testing.assert_allclose(out.continuous_embeds.cpu().numpy(), g["ip_continuous"], rtol=1e-4, atol=1e-4)
    assert out.discrete_codes.dtype == torch.int64 and out.discrete_codes.shape == (256, 48)
    assert torch.equal(out.quantized_embeds, model.decode(out.discrete_codes))
    # on the reference's exact continuous embeddings the codes are the reference's codes
    cont = _t(g["ip_continuous"])
    assert np.array_equal(model.quantize(cont).cpu().numpy().astype(np.uint8), g["ip_codes"])
    assert zlib.crc32(model.decode(_t(g["ip_codes"]).long()).detach().cpu().numpy().tobytes()) == int(g["ip_quantized_crc"])
    # passing discrete codes skips quantisation; gradient reaches the centroids
    out2 = model(ids, torch.ones_like(ids), discrete_codes=out.discrete_codes, return_quantized_embedding=True)
    out2.quantized_embeds.sum().backward()
    assert model.centroids.grad is not None and float(model.centroids.grad.abs().sum()) > 0
    # constrained mode toggled at run time (finetune_repconc.py:603-612)
    model.use_constraint = True
    cc = model.quantize(cont)
    want, _ = c_oracle.quantize(g["ip_continuous"], C, True, EPS, ITERS)
    assert np.array_equal(cc.cpu().numpy().astype(np.uint8), want)
    # COS variant: per-sub-vector normalisation happens before quantisation
    cfg2 = SimpleNamespace(MCQ_M=48, MCQ_K=256, hidden_size=768, similarity_metric="METRIC_CENTROID_COS")
    m2 = RepCONC(cfg2, _TableEncoder(table), False, EPS, ITERS).to(DEV)
    n = m2.centroids.detach().norm(dim=-1)
    assert torch.allclose(n, torch.ones_like(n), atol=1e-5)
    with torch.no_grad():
        m2.centroids.copy_(_t(C))
    codes = m2.quantize(_t(g["cos_continuous"]))
    assert np.array_equal(codes.cpu().numpy().astype(np.uint8), g["cos_codes"])


def test_sinkhorn_algorithm_and_centring_api():
    from repconc_amd.models.repconc import RepCONC, sinkhorn_algorithm
    _, x, C = load_case("m8_b300_gauss")
    d = pq_oracle.dist_table(x, C)
    dc = RepCONC.center_distance_for_constraint(_t(d))
    want = pq_oracle.centre(d, *pq_oracle.minmax_per_m(d))
    assert np.array_equal(dc.cpu().numpy().view(np.uint32), want.view(np.uint32))
    Q = sinkhorn_algorithm(-dc.double().transpose(1, 2), 0.003, 100, False)          # [M,K,B]
    ref = pq_oracle.sinkhorn_q([-(want.astype(np.float64)).transpose(0, 2, 1)], 0.003, 100)[0]
    assert Q.shape == ref.shape
    assert np.array_equal(Q.argmax(1).cpu().numpy(), ref.argmax(1))
    np.testing.assert_allclose(Q.sum(1).cpu().numpy(), 1.0, rtol=1e-12)
    np.testing.assert_allclose(Q.cpu().numpy(), ref, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("name", ["m8_b300_gauss", "m48_b1024_sample"])
def test_sinkhorn_algorithm_plan_against_the_references_plan(name):
    """The floating-point stage on its own output: `sinkhorn_algorithm` (the reference's name and arguments) returns the
    transport plan; 4096 sampled entries and every column's argmax are compared with what the REFERENCE's sinkhorn_algorithm
    returned on the same centred table (tests/golden/plan_<case>.npz, gen_golden.py --plan).  Tolerance, stated here: the
    product carries potentials and rebuilds Q = softmax_k(out / eps + f) instead of normalising Q in place 100 times, so entries
    agree to 1e-9 relative (+ 1e-30 absolute for entries that underflow towards zero); the argmax is exact."""
    from repconc_amd.models.repconc import RepCONC, sinkhorn_algorithm
    g, x, C = load_case(name)
    p = np.load(os.path.join(os.path.dirname(__file__), "golden", f"plan_{name}.npz"))
    from repconc_amd import ops
    d, _ = ops.dist_table(_t(x), _t(C))
    dc = RepCONC.center_distance_for_constraint(d)
    Q = sinkhorn_algorithm(-dc.double().transpose(1, 2), EPS, ITERS, False)          # [M,K,B]
    assert tuple(Q.shape) == (int(p["M"]), 256, int(p["B"]))
    Qn = Q.cpu().numpy()
    assert np.array_equal(Qn.argmax(1).astype(np.uint8), p["argmax"])
    np.testing.assert_allclose(Qn.reshape(-1)[p["sample_index"]], p["sample_q"], rtol=1e-9, atol=1e-30)
    np.testing.assert_allclose(Qn.sum(1), 1.0, rtol=1e-12)


@pytest.mark.parametrize("name", ["m6_b384_eps003", "m3_b1000_eps05"])
def test_sinkhorn_algorithm_on_a_cost_tensor_that_is_not_fp32_representable(name):
    """`sinkhorn_algorithm` takes ANY fp64 tensor (modeling_repconc.py:137-141); round 5 raised NotImplementedError unless the
    values were fp32-representable.  out = uniform(-1, 1) in fp64 goes through rc_sk64_rows / rc_sk64_cols; the plan is compared
    with what the REFERENCE's sinkhorn_algorithm returned on the same tensor (tests/golden/plan64_<case>.npz, gen_golden.py
    --plan64): every column's argmax exact, 4096 sampled entries to 1e-9 relative (+ 1e-300 absolute) — log-domain potentials
    against the reference's in-place normalisations —, columns sum to 1.  Then the same solve as TWO ranks (column shards, the
    [2,M,K] row values exchanged by hand between the raw C entries in lockstep): potentials equal the one-rank ones to 1e-12."""
    from repconc_amd import _lib, ops
    from repconc_amd.models.repconc import sinkhorn_algorithm
    p = np.load(os.path.join(os.path.dirname(__file__), "golden", f"plan64_{name}.npz"))
    M, B, eps, iters = int(p["M"]), int(p["B"]), float(p["eps"]), int(p["iters"])
    out = np.random.default_rng(int(p["seed"])).uniform(-1.0, 1.0, (M, 256, B))
    assert zlib.crc32(out.tobytes()) == int(p["out_crc"])
    ot = torch.from_numpy(out).to(DEV)
    Q = sinkhorn_algorithm(ot, eps, iters, False)
    Qn = Q.cpu().numpy()
    assert Q.dtype == torch.float64 and Qn.shape == (M, 256, B)
    assert np.array_equal(Qn.argmax(1).astype(np.uint8), p["argmax"])
    np.testing.assert_allclose(Qn.reshape(-1)[p["sample_index"]], p["sample_q"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(Qn.sum(1), 1.0, rtol=1e-12)
    # two ranks in lockstep through the raw entries
    f1 = ops.sinkhorn_potentials_f64(ot, eps, iters)
    lib, h = _lib.load(), _lib.handle(0)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    cut = B // 3
    sh = [ot[:, :, :cut].contiguous(), ot[:, :, cut:].contiguous()]
    lse = torch.empty((2, M, 256), dtype=torch.float64, device=DEV)
    f = [torch.empty((M, 256), dtype=torch.float64, device=DEV) for _ in range(2)]
    g = [torch.empty((M, x.shape[2]), dtype=torch.float64, device=DEV) for x in sh]
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    def rows(gs):
        for r in range(2):
            _lib.check(lib.rc_sk64_rows(h, P(sh[r]), P(gs[r]) if gs else None, sh[r].shape[2], M, 256, eps, P(lse[r]), s), "rows", h)
    def cols(want_g):
        for r in range(2):
            _lib.check(lib.rc_sk64_cols(h, P(sh[r]), P(lse), 2, sh[r].shape[2], M, 256, eps, P(f[r]), P(g[r]) if want_g else None, s),
                       "cols", h)
    rows(None)
    for _ in range(1, iters):
        cols(True)
        rows(g)
    cols(False)
    assert torch.equal(f[0], f[1])
    np.testing.assert_allclose(f[0].cpu().numpy(), f1.cpu().numpy(), rtol=1e-12, atol=1e-12)
    # a tensor that IS fp32-representable still takes the streaming sweep and agrees with the general kernels
    o32 = ot.float().double()
    np.testing.assert_allclose(sinkhorn_algorithm(o32, eps, iters, False).cpu().numpy(),
                               torch.softmax(o32 / eps + ops.sinkhorn_potentials_f64(o32, eps, iters)[:, :, None], dim=1).cpu().numpy(),
                               rtol=1e-9, atol=1e-300)


def test_index_build_and_search_api():
    from types import SimpleNamespace
    from repconc_amd.models.repconc import evaluate_repconc as ev
    C, codes, q = _adc_case(48, 30000, 12, seed=5)
    model = SimpleNamespace(config=SimpleNamespace(hidden_size=768, MCQ_M=48, MCQ_K=256),
                            centroids=torch.nn.Parameter(_t(C)))
    index = ev.initialize_index(model)
    ev.add_docs(index, codes[:10000])
    ev.add_docs(index, codes[10000:])
    assert index.ntotal == 30000 and index.pq.code_size == 48 and index.pq.M == 48
    assert np.array_equal(index.codes.cpu().numpy(), codes)
    index = ev.load_index_to_gpu(ev.from_pq_to_ivfpq(index))
    corpus_ids = np.arange(30000)[::-1].copy() + 7
    qids = np.arange(12)
    s, ids = ev.batch_search(qids, q, corpus_ids, index, 10, batch_size=5)
    ws, wi = c_oracle.adc_search(codes, C, q, 10)
    assert np.array_equal(ids, corpus_ids[wi]) and np.array_equal(s.view(np.uint32), ws.view(np.uint32))
    # torch-in/torch-out search (JPQ path) and centroid refresh without touching the codes
    s2, i2 = index.search(_t(q), 10)
    assert s2.is_cuda and np.array_equal(i2.cpu().numpy(), wi)
    index.set_centroids(C * 2)
    s3, _ = index.search(q, 10)
    np.testing.assert_allclose(s3, ws * 2, rtol=1e-6)


def test_mrr_parity_planted_relevance():
    """MRR@10 of the HIP search == MRR@10 of the brute-force oracle on a planted-relevance task."""
    from repconc_amd import ops
    M, N, nq = 48, 60000, 64
    x = synth.clustered_embeddings(21, N)
    C = synth.sample_centroids(22, x[:4096], M)
    codes = ops.assign_nearest(_t(x), _t(C), torch.uint8)
    assert np.array_equal(codes.cpu().numpy()[:2048], c_oracle.quantize(x[:2048], C, False)[0])
    rng = np.random.default_rng(23)
    pos = rng.integers(0, N, nq)
    q = x[pos] + 0.3 * synth.gaussian(24, (nq, 768))
    _, ids = ops.adc_search(codes, _t(C), _t(q), 10)
    _, wi = c_oracle.adc_search(codes.cpu().numpy(), C, q, 10)
    positives = [{int(p)} for p in pos]
    got, want = pq_oracle.mrr_at_k(ids.cpu().numpy(), positives), pq_oracle.mrr_at_k(wi, positives)
    assert abs(got - want) <= 0.001 and got == want


def test_diagnostics_match_oracle():
    """eval_balance / test_quantize (finetune_repconc.py:580-613) against the oracle's restatement."""
    from types import SimpleNamespace
    from repconc_amd.diagnostics import eval_balance, test_quantize as hip_test_quantize
    from repconc_amd.models.repconc import RepCONC
    g, x, C = load_case("m48_b1024_sample")
    for arr in (g["codes_constrained"], g["codes_nearest"]):
        for blk in (0, 17):
            assert eval_balance(_t(arr), False, blk) == pq_oracle.eval_balance(arr, blk)
    cfg = SimpleNamespace(MCQ_M=48, MCQ_K=256, hidden_size=768, similarity_metric="METRIC_IP")
    model = RepCONC(cfg, _TableEncoder(torch.zeros(1, 768)), True, EPS, ITERS).to(DEV)
    with torch.no_grad():
        model.centroids.copy_(_t(C))
    got = hip_test_quantize(_t(x), model, -1)
    want = pq_oracle.test_quantize(x, C, EPS, ITERS)
    assert got == want and model.use_constraint is True


def test_warmup_opq_pq():
    """train/run_warmup.py:85-132 restated: PQ k-means lowers the reconstruction error monotonically-ish, OPQ
    does not make it worse, the returned index holds the nearest codes of the rotated corpus."""
    from types import SimpleNamespace
    from repconc_amd.models.repconc import RepCONC
    from repconc_amd.train.run_warmup import train_pq, warmup_from_embeds
    rng = np.random.default_rng(5)
    N, M = 20000, 48
    mix = rng.standard_normal((768, 768), dtype=np.float32) / np.sqrt(768)
    x = (synth.clustered_embeddings(6, N) @ mix).astype(np.float32)          # correlated dimensions: OPQ helps
    xt = _t(x)
    _, mse0 = train_pq(xt, M, 0)
    _, mse5 = train_pq(xt, M, 5)
    _, mse15 = train_pq(xt, M, 15)
    assert mse15 <= mse5 < mse0
    # one Lloyd step against the oracle's restatement
    C0 = synth.sample_centroids(7, x, M)
    codes = pq_oracle.quantize(x[:4096], C0, False)
    s, c = pq_oracle.kmeans_stats(x[:4096], codes, M)
    want = pq_oracle.kmeans_update(s, c, C0)
    sums, cnt = ops_kmeans(xt[:4096], _t(codes.astype(np.uint8)))
    from repconc_amd import ops
    got = ops.kmeans_update_(sums, cnt, _t(C0).clone()).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7)
    cfg = SimpleNamespace(MCQ_M=M, MCQ_K=256, hidden_size=768, similarity_metric="METRIC_IP")
    model = RepCONC(cfg, _TableEncoder(torch.zeros(1, 768)), False, None, None).to(DEV)
    model, index = warmup_from_embeds(x, model, opq_iters=6, pq_iters=10)
    R = model.rotation.detach()
    assert torch.allclose(R @ R.T, torch.eye(768, device=DEV), atol=1e-4), float((R @ R.T - torch.eye(768, device=DEV)).abs().max())
    assert index.index.ntotal == N
    xr = (xt @ R.T).contiguous()
    codes = index.index.codes
    assert torch.equal(codes, ops.assign_nearest(xr, model.centroids, torch.uint8))
    mse_opq = float(((ops.decode_raw(codes, model.centroids) - xr) ** 2).sum(-1).mean())
    assert mse_opq < mse15 * 1.02
    # rotated search == plain ADC on rotated queries
    q = x[:5]
    s1, i1 = index.search(q, 10)
    s2, i2 = index.index.search((xt[:5] @ R.T).cpu().numpy(), 10)
    assert np.array_equal(i1, i2)
    assert all(int(i1[r, 0]) == r or s1[r, 0] >= s1[r, 1] for r in range(5))


def ops_kmeans(x, codes):
    from repconc_amd import ops
    return ops.kmeans_stats(x, codes)


# ------------------------------------------------------------------------------------------- BASELINE sizes
def test_full_batch_properties_49152_m48():
    """BASELINE configs[1]/[2] shape (one 49 152 x 768 training batch, M = 48): size-independent properties.
    (a) 8 virtual ranks of 6144 rows (the 8-GPU recipe) give the codes of the single-rank solve,
    (b) every centroid of every sub-quantiser receives its share of the batch (the constraint),
    (c) the result is identical run to run, (d) decode(quantize) beats random codes by a wide margin but is
    worse than the unconstrained nearest codes (the MSE ordering the reference logs in test_quantize)."""
    from repconc_amd import ops
    from repconc_amd.sharded import assign_sinkhorn_virtual
    B, M = 49152, 48
    x = _t(synth.gaussian(20220, (B, 768)))
    C = x[torch.from_numpy(np.random.default_rng(20221).permutation(B)[:256].copy()).to(DEV)]
    C = C.reshape(256, M, 16).transpose(0, 1).contiguous()
    codes, flags = ops.assign_sinkhorn(x, C, EPS, ITERS, torch.uint8)
    assert int(flags.item()) == 0
    codes2, _ = ops.assign_sinkhorn(x, C, EPS, ITERS, torch.uint8)
    assert torch.equal(codes, codes2)                                               # (c)
    hist = ops.code_hist(codes).float()
    assert float((hist / (B / 256) - 1).abs().max()) < 0.15                         # (b) ideal 192 per centroid
    near = ops.assign_nearest(x, C, torch.uint8)
    hn = ops.code_hist(near).float()
    assert float((hn / (B / 256) - 1).abs().max()) > 1.0                            # nearest codes are unbalanced
    shards, fl = assign_sinkhorn_virtual([x[r * 6144:(r + 1) * 6144] for r in range(8)], C, EPS, ITERS,
                                         dtype=torch.uint8)
    assert torch.equal(torch.cat(shards, 0), codes)                                 # (a)
    mse_c = float(((ops.decode_raw(codes, C) - x) ** 2).sum(-1).mean())
    mse_n = float(((ops.decode_raw(near, C) - x) ** 2).sum(-1).mean())
    rnd = torch.randint(0, 256, (B, M), dtype=torch.uint8, device=DEV)
    mse_r = float(((ops.decode_raw(rnd, C) - x) ** 2).sum(-1).mean())
    assert mse_n < mse_c < 0.8 * mse_r                                              # (d)
    # a slice of the batch against the C oracle's nearest codes (exact) — the constrained codes of a slice are a
    # different problem, so only the unconstrained path can be sliced
    assert np.array_equal(near[:2048].cpu().numpy(), c_oracle.quantize(x[:2048].cpu().numpy(), C.cpu().numpy(), False)[0])


@pytest.mark.parametrize("M", [24, 96, 8])
def test_other_config_shapes_constrained(M):
    """BASELINE configs[3] (M = 96), [4] (M = 24) and [0] (M = 8) sub-vector widths on a 4096-row batch vs the oracle."""
    from repconc_amd import ops
    B = 4096
    x = synth.clustered_embeddings(300 + M, B)
    C = synth.sample_centroids(301 + M, x, M)
    want, fl = c_oracle.quantize(x, C, True, EPS, ITERS)
    got, flags = ops.assign_sinkhorn(_t(x), _t(C), EPS, ITERS, torch.uint8)
    assert fl == 0 and int(flags.item()) == 0
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(ops.assign_nearest(_t(x), _t(C), torch.uint8).cpu().numpy(), c_oracle.quantize(x, C, False)[0])


def test_row_sharded_index_search_equals_whole_index():
    from repconc_amd.index import PQIndex
    from repconc_amd.sharded_search import search_virtual_shards
    C, codes, q = _adc_case(48, 90001, 7, seed=31)
    codes[50000:50040] = codes[10:50]                      # duplicates straddling shard boundaries -> score ties
    whole = PQIndex(768, 48, device=DEV)
    whole.set_centroids(C)
    whole.add_codes(codes)
    bounds = [0, 30000, 30001, 65536, 90001]
    shards = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        sh = PQIndex(768, 48, device=DEV)
        sh.set_centroids(C)
        sh.add_codes(codes[a:b])
        sh.id_offset = a
        shards.append(sh)
    for k in (10, 300):
        ws, wi = whole.search(_t(q), k)
        gs, gi = search_virtual_shards(shards, _t(q), k)
        assert torch.equal(wi, gi) and torch.equal(ws, gs)


def test_faiss_indexpq_file_round_trip(tmp_path):
    from repconc_amd.faiss_io import read_index, write_index
    from repconc_amd.index import PQIndex
    C, codes, q = _adc_case(24, 12345, 3, seed=77)
    idx = PQIndex(768, 24, device=DEV)
    idx.set_centroids(C)
    idx.add_codes(codes)
    p = str(tmp_path / "index")
    write_index(idx, p)
    raw = open(p, "rb").read()
    assert raw[:4] == b"IxPq" and len(raw) == 4 + 33 + 24 + 8 + 24 * 256 * 32 * 4 + 8 + 12345 * 24 + 9
    back = read_index(p, device=DEV)
    assert back.ntotal == 12345 and back.pq.M == 24 and torch.equal(back.codes, idx.codes)
    assert torch.equal(back.pq.centroids, idx.pq.centroids)
    s1, i1 = idx.search(q, 20)
    s2, i2 = back.search(q, 20)
    assert np.array_equal(i1, i2) and np.array_equal(s1, s2)


@pytest.mark.parametrize("M,N,nq,k", [(48, 300000, 6, 1000), (96, 270000, 3, 10), (8, 400000, 9, 100),
                                      (64, 262144, 4, 50), (24, 500000, 11, 200), (12, 262145, 2, 1)])
def test_adc_integer_screening_path_is_exact(M, N, nq, k):
    """N >= 2^18 takes the 8-bit screening + exact rescoring path; ids and score bits must still equal the
    brute-force oracle (the integer threshold is a rigorous bound)."""
    from repconc_amd import ops
    C, codes, q = _adc_case(M, N, nq, seed=M * 7 + N)
    codes[N // 2: N // 2 + 300] = codes[:300]          # duplicated rows: ties across the candidate boundary
    scores, ids = ops.adc_search(_t(codes), _t(C), _t(q), k)
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    assert np.array_equal(ids.cpu().numpy(), wi)
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), ws.view(np.uint32))


def test_adc_screening_with_skewed_tables():
    """Queries with one dominant sub-space and a constant (zero-range) sub-space: the common quantisation step is
    set by the widest table, narrow tables collapse to a few levels, the bound must still hold."""
    from repconc_amd import ops
    M, N, nq, k = 48, 280000, 5, 100
    C, codes, q = _adc_case(M, N, nq, seed=99)
    q = q.copy()
    q[:, :16] *= 40.0            # sub-space 0 dominates the score range
    q[:, 16:32] = 0.0            # sub-space 1 contributes a constant 0
    scores, ids = ops.adc_search(_t(codes), _t(C), _t(q), k)
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    assert np.array_equal(ids.cpu().numpy(), wi)
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), ws.view(np.uint32))


def test_native_rccl_solve_single_rank(monkeypatch):
    """csrc/comm.hip: the distributed solve driven from C over RCCL, on a one-rank communicator (all this box can
    host): one chain and the two-chain/two-stream split must both reproduce the reference codes."""
    import socket
    import torch.distributed as dist
    from repconc_amd import ops
    from repconc_amd.sharded import TorchDistComm, assign_sinkhorn_sharded
    created = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device(DEV))
        created = True
    try:
        for name in ("m48_b1024_sample", "m8_b2048_sample", "m48_b1000_ragged"):
            g, x, C = load_case(name)
            for split in ("0", "1"):
                monkeypatch.setenv("RC_DIST_SPLIT", split)
                codes, flags = assign_sinkhorn_sharded(_t(x), _t(C), EPS, ITERS, TorchDistComm(), dtype=torch.uint8)
                torch.cuda.synchronize()
                assert int(flags.item()) == 0
                assert np.array_equal(codes.cpu().numpy(), g["codes_constrained"]), (name, split)
            c64, _ = ops.assign_sinkhorn_dist(_t(x), _t(C), EPS, ITERS, torch.int64)
            assert np.array_equal(c64.cpu().numpy().astype(np.uint8), g["codes_constrained"])
    finally:
        if created:
            dist.destroy_process_group()


def test_ivf_extension_exact_and_equals_flat_when_all_lists_probed():
    """repconc_amd.ivf (nlist > 1: a build-side extension, the reference only has nlist = 1): (a) probing every cell
    == the flat search, bit for bit; (b) any nprobe == the brute-force oracle on the same cells; (c) recall grows
    with nprobe."""
    from repconc_amd import ops
    from repconc_amd.index import PQIndex
    from repconc_amd.ivf import IVFPQIndex
    N, M, nlist, nq = 120000, 48, 64, 9
    x = synth.clustered_embeddings(41, N)
    C = synth.sample_centroids(42, x[:8192], M)
    q = x[np.random.default_rng(43).integers(0, N, nq)] + 0.2 * synth.gaussian(44, (nq, 768))
    ivf = IVFPQIndex(768, M, nlist, device=DEV)
    ivf.set_centroids(C)
    ivf.train(x[:30000], iters=6)
    ivf.add(x)
    assert ivf.ntotal == N and int(ivf.list_off[-1]) == N
    codes = ops.assign_nearest(_t(x), _t(C), torch.uint8)
    list_ids = torch.empty(N, dtype=torch.int64, device=DEV)
    lens = (ivf.list_off[1:] - ivf.list_off[:-1])
    list_ids[ivf.ids] = torch.repeat_interleave(torch.arange(nlist, device=DEV), lens)
    assert torch.equal(ivf.codes, codes[ivf.ids])
    flat = PQIndex(768, M, device=DEV)
    flat.set_centroids(C)
    flat.add_codes(codes)
    for k in (10, 200):
        fs, fi = flat.search(_t(q), k)
        for method in ("lists", "scan"):
            s, i = ivf.search(_t(q), k, nprobe=nlist, method=method)
            assert torch.equal(i, fi) and torch.equal(s, fs), method             # (a)
    recalls = []
    for nprobe in (1, 4, 16):
        s, i = ivf.search(q, 10, nprobe)
        ws, wi = pq_oracle.ivf_search(q, C, codes.cpu().numpy(), list_ids.cpu().numpy(), ivf.coarse.cpu().numpy(), 10, nprobe)
        assert np.array_equal(i, wi)                                             # (b)
        assert np.array_equal(s.view(np.uint32), ws.view(np.uint32))
        s_, i_ = ivf.search(q, 10, nprobe, method="scan")
        assert np.array_equal(i_, wi) and np.array_equal(s_.view(np.uint32), ws.view(np.uint32))
        fi10 = flat.search(q, 10)[1]
        recalls.append(np.mean([len(set(i[r]) & set(fi10[r])) / 10 for r in range(nq)]))
    assert recalls[0] <= recalls[1] <= recalls[2] and recalls[2] > 0.8          # (c)
    # a flat index re-organised without embeddings (cells from the reconstructions)
    ivf2 = IVFPQIndex.from_flat(flat, 32, iters=4)
    s2, i2 = ivf2.search(_t(q), 10, nprobe=32)
    assert torch.equal(i2, flat.search(_t(q), 10)[1])
    # tiny cells: fewer than k rows probed -> padded with -1 / -inf
    s3, i3 = ivf.search(q[:2], 5000, nprobe=1)
    n_rows = int((i3[0] >= 0).sum())
    assert n_rows < 5000 and np.all(i3[0][n_rows:] == -1) and np.all(np.isinf(s3[0][n_rows:]))


def test_encode_corpus_to_index_matches_forward_codes():
    from types import SimpleNamespace
    from repconc_amd.encode import encode_corpus_to_index
    from repconc_amd.models.repconc import RepCONC
    table = torch.from_numpy(synth.clustered_embeddings(4242, 1000))
    C = synth.sample_centroids(4243, table.numpy(), 48)
    cfg = SimpleNamespace(MCQ_M=48, MCQ_K=256, hidden_size=768, similarity_metric="METRIC_IP")
    model = RepCONC(cfg, _TableEncoder(table), False, None, None).to(DEV)
    with torch.no_grad():
        model.centroids.copy_(_t(C))
    ids = torch.arange(1000, device=DEV)[:, None].repeat(1, 3)
    batches = [(ids[i:i + 128], torch.ones_like(ids[i:i + 128])) for i in range(0, 1000, 128)]
    index = encode_corpus_to_index(model, batches, id_offset=5000)
    want = model(ids, torch.ones_like(ids), return_code=True).discrete_codes.to(torch.uint8)
    assert index.ntotal == 1000 and torch.equal(index.codes, want) and index.id_offset == 5000
    s, i = index.search(table[:4].numpy(), 3)
    assert (i >= 5000).all()


def test_stage1_training_step_gradients_match_direct_autograd():
    """The two-pass cached-gradient step (finetune_repconc.py:245-396 restated in repconc_amd/train/stage1.py) gives
    the gradients of the one-pass objective  L(q, sg(decode)+ste) + w*mse ; the constrained quantiser and the decode
    backward kernel sit inside the step."""
    from transformers import BertConfig
    from repconc_amd.models.dense import BertDense
    from repconc_amd.models.repconc import RepCONC
    from repconc_amd.train.stage1 import Stage1Config, contrastive_loss, make_optimizer, stage1_training_step
    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=128, vocab_size=500,
                     max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg.MCQ_M, cfg.MCQ_K, cfg.similarity_metric, cfg.pooling = 48, 256, "METRIC_IP", "mean"
    enc = BertDense(cfg)
    model = RepCONC(cfg, enc, True, 0.003, 20).to(DEV)
    with torch.no_grad():
        model.centroids.mul_(0.05)
    nq, nneg, L = 24, 72, 12
    mk = lambda n: {"input_ids": torch.randint(1, 500, (n, L), device=DEV), "attention_mask": torch.ones((n, L), dtype=torch.long, device=DEV)}
    qin, pin, nin = mk(nq), mk(nq), mk(nneg)
    qids = torch.arange(nq, device=DEV)
    pos_ids = torch.arange(1000, 1000 + nq, device=DEV)
    neg_ids = torch.arange(2000, 2000 + nneg, device=DEV)
    neg_ids[5] = pos_ids[3]                                         # a duplicate and a false negative
    qrels = {i: {1000 + i} for i in range(nq)}
    qrels[2].add(int(neg_ids[7]))
    scfg = Stage1Config(cache_chunk_size=10, mse_loss_weight=1e-2, dynamic_topk_hard_negative=11)
    model.zero_grad()
    loss = stage1_training_step(model, qin, pin, qids, pos_ids, qrels, scfg, nin, neg_ids)
    got = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert np.isfinite(loss) and "centroids" in got and float(got["centroids"].abs().sum()) > 0
    # one-pass reference objective with the same codes
    model.zero_grad()
    q = model(**qin).continuous_embeds
    p = model(**pin).continuous_embeds
    n = model(**nin).continuous_embeds
    docs = torch.cat([p, n], 0)
    with torch.no_grad():
        codes = model.quantize(docs)
    quant = model.decode(codes)
    ste = quant.detach() + (docs - docs.detach()) + (quant - quant.detach())     # value = quant, grads to both
    direct = contrastive_loss(q, ste, qids, torch.cat([pos_ids, neg_ids]), qrels, scfg, "METRIC_IP", 48)
    mse = (((quant[:nq] - p) ** 2).sum(-1).mean() + ((quant[nq:] - n) ** 2).sum(-1).mean()) * scfg.mse_loss_weight
    # chunk means: the step averages the MSE per chunk of 10, the reference does too (:374) — mirror it
    mse = 0
    for rep, qt in ((p, quant[:nq]), (n, quant[nq:])):
        for a in range(0, rep.shape[0], 10):
            mse = mse + ((qt[a:a + 10] - rep[a:a + 10]) ** 2).sum(-1).mean() * scfg.mse_loss_weight
    (direct + mse).backward()
    assert abs(float(direct.detach()) - loss) < 1e-4
    for name, grad in got.items():
        ref = dict(model.named_parameters())[name].grad
        assert torch.allclose(grad, ref, rtol=2e-3, atol=2e-5), name
    opt = make_optimizer(model)
    opt.step()
    assert len(opt.param_groups) == 3 and opt.param_groups[2]["lr"] == 5e-4


# ---------------------------------------------------------------------------------------------------------------------
# matrix-core screen + exact rescoring (csrc/pq_assign_mfma.hip): the same codes as the exact-order kernel
@pytest.mark.parametrize("M", [96, 64, 48, 32, 24, 16, 12, 8])
def test_mfma_screened_assignment_equals_exact(M):
    from repconc_amd import ops
    rng = np.random.default_rng(1000 + M)
    B = 1000 + 7 * M                                   # not a multiple of the 128-row block
    x = rng.standard_normal((B, 768)).astype(np.float32)
    C = rng.standard_normal((M, 256, 768 // M)).astype(np.float32)
    st = {}
    fast = ops.assign_nearest(_t(x), _t(C), torch.uint8, method="mfma", stats=st)
    exact = ops.assign_nearest(_t(x), _t(C), torch.uint8, method="exact")
    assert st["method"] == "mfma" and not st["overflow"]
    assert torch.equal(fast, exact)
    assert np.array_equal(fast.cpu().numpy(), c_oracle.quantize(x, C, False)[0])
    assert st["doubtful"] < B * M // 20               # the screen decides almost everything


def test_mfma_screened_assignment_ties_and_near_ties():
    """Duplicated centroids (exact ties -> the FIRST index must win), centroids differing in the last bit, rows that
    sit exactly between two centroids, and wildly scaled rows: everything doubtful must reach the exact judge."""
    from repconc_amd import ops
    rng = np.random.default_rng(77)
    M, dsub, B = 48, 16, 2048
    C = rng.standard_normal((M, 256, dsub)).astype(np.float32)
    C[:, 200] = C[:, 3]                                # exact duplicates
    C[:, 201] = np.nextafter(C[:, 5], np.float32(np.inf))   # one ulp away
    x = rng.standard_normal((B, 768)).astype(np.float32)
    xs = x.reshape(B, M, dsub)
    xs[:256, :] = C[:, 3][None]                        # on a duplicated centroid
    xs[256:512, :] = C[:, 5][None] + 1e-7
    xs[512:768, :] = 0.5 * (C[:, 7] + C[:, 9])[None]   # equidistant in exact arithmetic
    xs[768:900] *= 1e4
    xs[900:1024] *= 1e-4
    x = xs.reshape(B, 768)
    st = {}
    fast = ops.assign_nearest(_t(x), _t(C), torch.int64, method="mfma", stats=st)
    exact = ops.assign_nearest(_t(x), _t(C), torch.int64, method="exact")
    assert torch.equal(fast, exact)
    assert np.array_equal(fast.cpu().numpy(), c_oracle.quantize(x, C, False)[0])
    assert st["doubtful"] >= 512 * M            # the duplicate and one-ulp rows for certain


def test_mfma_screened_assignment_non_finite_rows():
    """NaN / Inf / overflow-scale rows: the margin is non-finite, so the pair is doubtful and the exact judge decides
    exactly like the exact kernel (code 0 when every distance is NaN or +inf)."""
    from repconc_amd import ops
    rng = np.random.default_rng(78)
    M, B = 48, 512
    C = rng.standard_normal((M, 256, 16)).astype(np.float32)
    x = rng.standard_normal((B, 768)).astype(np.float32)
    x[3, 100] = np.nan
    x[7, :] = np.inf
    x[11, 5] = -np.inf
    x[13, :] = 1e30                      # squares overflow fp32
    x[17, :] = 3e19                      # ||x||^2 overflows only in the sum
    x[19:40] *= 1e-22                    # squares are denormal / underflow
    fast = ops.assign_nearest(_t(x), _t(C), torch.uint8, method="mfma")
    exact = ops.assign_nearest(_t(x), _t(C), torch.uint8, method="exact")
    assert torch.equal(fast, exact)
    tiny = (rng.standard_normal((B, 768)) * 1e-20).astype(np.float32)      # everything below the 1e-30 scale floor
    Ct = (C * np.float32(1e-20)).astype(np.float32)
    st = {}
    fast = ops.assign_nearest(_t(tiny), _t(Ct), torch.uint8, method="mfma", stats=st)
    assert torch.equal(fast, ops.assign_nearest(_t(tiny), _t(Ct), torch.uint8, method="exact"))
    assert st["doubtful"] == B * M or st["overflow"]


def test_mfma_assignment_overflow_falls_back_and_unaligned_rows():
    """All centroids identical: EVERY pair is doubtful — the doubt list holds one slot per pair, so nothing overflows,
    the exact judge re-decides all of them (first minimum: code 0) without any host round trip.  Unaligned rows: ops
    realigns, the raw C call refuses."""
    from repconc_amd import ops, _lib
    rng = np.random.default_rng(5)
    M, B = 48, 8192
    C = np.repeat(rng.standard_normal((M, 1, 16)).astype(np.float32), 256, axis=1)
    x = rng.standard_normal((B, 768)).astype(np.float32)
    st = {}
    codes = ops.assign_nearest(_t(x), _t(C), torch.uint8, stats=st)
    assert st["method"] == "mfma" and not st["overflow"] and st["doubtful"] == B * M
    assert int(codes.max()) == 0
    assert int(ops.assign_nearest(_t(x), _t(C), torch.uint8).max()) == 0          # the path without statistics
    wide = torch.zeros((64, 771), device=DEV)
    wide[:, 1:769] = _t(x[:64])
    st = {}
    C2 = rng.standard_normal((M, 256, 16)).astype(np.float32)
    got = ops.assign_nearest(wide[:, 1:769], _t(C2), torch.uint8, stats=st)   # ops realigns the rows (one copy)
    assert st["method"] == "mfma"
    assert np.array_equal(got.cpu().numpy(), c_oracle.quantize(x[:64], C2, False)[0])
    # the C entry point itself rejects what it cannot load as float4
    lib, h = _lib.load(), _lib.handle(0)
    v = wide[:, 1:769]
    out = torch.empty((64, M), dtype=torch.uint8, device=DEV)
    n = lib.rc_pq_assign_nearest_fast_ws_bytes(64, M)
    ws = torch.empty(n, dtype=torch.uint8, device=DEV)
    rc = lib.rc_pq_assign_nearest_fast(h, v.data_ptr(), v.stride(0), _t(C2).data_ptr(), 64, 768, M, 256,
                                       out.data_ptr(), None, ws.data_ptr(), n, None)
    assert rc == _lib.RC_EINVAL


def test_adc_large_k_uses_exact_scan():
    """k > 2048 on a large index (round 1 bypassed the integer screen there; it is screened now) must still be exact."""
    from repconc_amd import ops
    M, N, nq, k = 48, 300000, 3, 3000
    C, codes, q = _adc_case(M, N, nq, seed=4242)
    scores, ids = ops.adc_search(_t(codes), _t(C), _t(q), k)
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    assert np.array_equal(ids.cpu().numpy(), wi)
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), ws.view(np.uint32))


def test_c_index_handle_api_matches_oracle():
    """The stateful C index (csrc/index.hip: create / set_centroids / add_codes in chunks / search / reset) driven with
    raw pointers, as a host without torch tensors would: results equal the brute-force oracle bit for bit; growth keeps
    earlier rows; an empty index answers -inf / -1; set_centroids takes effect in place (JPQ step)."""
    import ctypes
    from repconc_amd import _lib
    lib, h = _lib.load(), _lib.handle(0)
    M, N, nq, k = 48, 70000, 5, 50
    C, codes, q = _adc_case(M, N, nq, seed=31337)
    idx = ctypes.c_void_p()
    assert lib.rc_index_create(h, 768, M, 256, ctypes.byref(idx)) == 0
    try:
        dq, dC, dcodes = _t(q), _t(C), _t(codes)
        sc = torch.empty((nq, k), dtype=torch.float32, device=DEV)
        ids = torch.empty((nq, k), dtype=torch.int64, device=DEV)
        assert lib.rc_index_search(idx, dq.data_ptr(), nq, k, sc.data_ptr(), ids.data_ptr(), None) == _lib.RC_EINVAL  # no centroids
        assert lib.rc_index_set_centroids(idx, dC.data_ptr(), None) == 0
        assert lib.rc_index_search(idx, dq.data_ptr(), nq, k, sc.data_ptr(), ids.data_ptr(), None) == 0
        assert bool((ids == -1).all()) and bool(torch.isinf(sc).all())
        for a, b in ((0, 1000), (1000, 1001), (1001, 40000), (40000, N)):          # forces several reallocations
            assert lib.rc_index_add_codes(idx, dcodes[a:b].data_ptr(), b - a, None) == 0
        assert lib.rc_index_ntotal(idx) == N
        torch.cuda.synchronize()
        assert lib.rc_index_search(idx, dq.data_ptr(), nq, k, sc.data_ptr(), ids.data_ptr(), None) == 0
        ws, wi = c_oracle.adc_search(codes, C, q, k)
        assert np.array_equal(ids.cpu().numpy(), wi)
        assert np.array_equal(sc.cpu().numpy().view(np.uint32), ws.view(np.uint32))
        C2 = (C * np.float32(0.5) + np.float32(0.25)).astype(np.float32)
        assert lib.rc_index_set_centroids(idx, _t(C2).data_ptr(), None) == 0
        assert lib.rc_index_search(idx, dq.data_ptr(), nq, k, sc.data_ptr(), ids.data_ptr(), None) == 0
        ws, wi = c_oracle.adc_search(codes, C2, q, k)
        assert np.array_equal(ids.cpu().numpy(), wi)
        assert lib.rc_index_reset(idx) == 0 and lib.rc_index_ntotal(idx) == 0
    finally:
        assert lib.rc_index_destroy(idx) == 0


@pytest.mark.parametrize("M", [128, 6, 1])
def test_search_at_a_width_without_a_screening_kernel(M):
    """a model warmed up with MCQ_M = 128, 6 or 1 (the constructor and the training kernels take every divisor of the
    hidden size, as the reference does: modeling_repconc.py:41) must also evaluate.  The screened search is compiled for
    eight widths; every other one is answered by the exact scan with a run-time width — through ops.adc_search, PQIndex and the
    stateful C index alike: ids and score bits of the brute-force oracle."""
    import ctypes
    from repconc_amd import _lib, ops
    from repconc_amd.index import PQIndex
    N, nq, k = 30011, 5, 40
    C, codes, q = _adc_case(M, N, nq, seed=4100 + M)
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    gs, gi = ops.adc_search(_t(codes), _t(C), _t(q), k)
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gs.cpu().numpy().view(np.uint32), ws.view(np.uint32))
    index = PQIndex(768, M, device=torch.device(DEV))
    index.set_centroids(_t(C))
    index.add_codes(_t(codes[:20000]))
    index.add_codes(_t(codes[20000:]))
    ps, pi = index.search(q, k)                                     # numpy in, numpy out (evaluate_repconc.py:182)
    assert np.array_equal(pi, wi) and np.array_equal(ps.view(np.uint32), ws.view(np.uint32))
    lib, h = _lib.load(), _lib.handle(0)
    idx = ctypes.c_void_p()
    assert lib.rc_index_create(h, 768, M, 256, ctypes.byref(idx)) == 0
    try:
        dq, dC, dcodes = _t(q), _t(C), _t(codes)
        sc = torch.empty((nq, k), dtype=torch.float32, device=DEV)
        ids = torch.empty((nq, k), dtype=torch.int64, device=DEV)
        assert lib.rc_index_set_centroids(idx, dC.data_ptr(), None) == 0
        assert lib.rc_index_add_codes(idx, dcodes.data_ptr(), N, None) == 0
        assert lib.rc_index_search(idx, dq.data_ptr(), nq, k, sc.data_ptr(), ids.data_ptr(), None) == 0
        assert np.array_equal(ids.cpu().numpy(), wi) and np.array_equal(sc.cpu().numpy().view(np.uint32), ws.view(np.uint32))
    finally:
        assert lib.rc_index_destroy(idx) == 0


def test_index_build_scale_properties():
    """Index-build path at scale (2 M x 768, M = 48, a quarter of the BASELINE corpus per pass): the MFMA-screened codes
    equal the exact-order kernel's, and coding is idempotent — re-assigning decode(codes) returns the same codes
    (a reconstructed vector sits ON its centroids, distance 0 in every sub-space, first minimum wins)."""
    from repconc_amd import ops
    N, M = 1 << 21, 48
    g = torch.Generator(device=DEV).manual_seed(2024)
    x = torch.randn((N, 768), device=DEV, generator=g)
    C = x[torch.randperm(N, device=DEV, generator=g)[:256]].reshape(256, M, 16).transpose(0, 1).contiguous()
    st = {}
    fast = ops.assign_nearest(x, C, torch.uint8, method="mfma", stats=st)
    exact = ops.assign_nearest(x, C, torch.uint8, method="exact")
    assert torch.equal(fast, exact)
    assert st["method"] == "mfma" and 0 < st["doubtful"] < N * M // 100
    del x
    rec = ops.decode_raw(fast, C)
    again = ops.assign_nearest(rec, C, torch.uint8, method="mfma")
    assert torch.equal(again, fast)
    hist = ops.code_hist(fast)
    assert int(hist.sum()) == N * M and hist.shape == (M, 256)


def test_jpq_module_step_matches_plain_autograd_and_keeps_index_in_sync():
    """Stage 2 (N1): JPQ.forward over the resident index — loss and gradients (query table, centroids) equal a plain
    torch restatement of the same step; after an optimiser step + jpq_step_end the index scores with the NEW centroids
    while its codes never moved."""
    import random
    from types import SimpleNamespace
    from repconc_amd.index import PQIndex
    from repconc_amd.models.jpq import JPQ, jpq_step_end
    from repconc_amd.models.repconc import RepCONC
    torch.manual_seed(7)
    M, N, nq, k = 48, 20000, 12, 50
    docs = torch.from_numpy(synth.clustered_embeddings(515, N)).to(DEV)
    C = _t(synth.sample_centroids(516, docs[:4096].cpu().numpy(), M))
    cfg = SimpleNamespace(MCQ_M=M, MCQ_K=256, hidden_size=768, similarity_metric="METRIC_IP")

    class _Enc(torch.nn.Module):
        def __init__(self, table):
            super().__init__()
            self.table = torch.nn.Parameter(table.clone())
            self.config = SimpleNamespace(hidden_size=768)

        def forward(self, input_ids, attention_mask):
            return self.table[input_ids[:, 0]]

    qtable = docs[torch.randperm(N, device=DEV)[:64]] + 0.05 * torch.randn(64, 768, device=DEV)
    model = RepCONC(cfg, _Enc(qtable), False, EPS, ITERS).to(DEV)
    with torch.no_grad():
        model.centroids.copy_(C)
    index = PQIndex(768, M, device=DEV)
    index.set_centroids(C)
    index.add(docs)
    codes_ptr = index.codes.data_ptr()
    qrels = {q: [int(3 * q), int(3 * q + 1)] for q in range(64)}
    jpq = JPQ(model, index, qrels, neg_top_k=k, temperature=1.0)
    qids = torch.arange(nq, device=DEV)
    ids = qids[:, None].repeat(1, 4)
    random.seed(99)
    loss = jpq(ids, torch.ones_like(ids), qids)["loss"]
    loss.backward()
    g_tab, g_cent = model.dense_encoder.table.grad.clone(), model.centroids.grad.clone()
    # plain torch restatement with the same retrieved negatives / sampled positives
    random.seed(99)
    with torch.no_grad():
        qe = qtable[:nq] @ model.rotation.T
        neg = index.search(qe.contiguous(), k)[1]
    pos = torch.tensor([random.choice(qrels[int(q)]) for q in qids.tolist()], device=DEV)
    tab2 = qtable.clone().requires_grad_(True)
    C2 = C.clone().requires_grad_(True)
    codes_l = index.codes.long()

    def dec(p):
        rows = codes_l[p.reshape(-1)]
        return torch.cat([C2[m, rows[:, m]] for m in range(M)], dim=1)

    q2 = tab2[:nq] @ model.rotation.T
    sn = (q2.unsqueeze(1) * dec(neg).reshape(nq, k, -1)).sum(-1)
    sp = (q2 * dec(pos)).sum(-1, keepdim=True)
    want = torch.nn.functional.cross_entropy(torch.hstack((sp, sn)), torch.zeros(nq, dtype=torch.long, device=DEV))
    want.backward()
    assert abs(float(loss.detach()) - float(want.detach())) < 1e-4 * max(1.0, abs(float(want.detach())))
    torch.testing.assert_close(g_tab, tab2.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(g_cent, C2.grad, rtol=1e-4, atol=1e-5)
    # optimiser step, then the callback work: the index must score with the new centroids, codes untouched
    opt = torch.optim.SGD([model.centroids], lr=0.05)
    opt.step()
    jpq_step_end(jpq)
    assert index.codes.data_ptr() == codes_ptr
    sc, idn = index.search(qe.contiguous(), 10)
    brute = qe @ ops_decode_all(index, model.centroids.detach()).T
    ws, wi = brute.topk(10, dim=1)
    torch.testing.assert_close(sc, ws, rtol=1e-4, atol=1e-4)
    assert float((idn == wi).float().mean()) > 0.95          # ties / last-ulp order may differ from the dense GEMM


def ops_decode_all(index, centroids):
    from repconc_amd import ops
    return ops.decode_raw(index.codes.contiguous(), centroids.contiguous())


# ------------------------------------------------------------------------------------------- round 2
SK_VARIANTS = [  # (RC_SK_V1, RC_SK_FKLDS, RC_SK_NB, RC_SK_CPB, RC_FUSE_CENTRE, RC_SK_PRIO)
    ("0", "1", "", "", "1", ""),      # default: version-2 sweep, potentials from LDS, 4 blocks per CU, rotating wave priority
    ("0", "1", "", "", "1", "0"),     # ... without the rotating priority (round 4: a scheduling matter, never a result)
    ("0", "1", "", "", "1", "3"),     # ... rotating every 40 ns
    ("0", "0", "", "", "1"),      # potentials in registers
    ("0", "1", "", "", "0"),      # separate centring kernel
    ("0", "1", "7", "", "1"),     # few blocks: every block straddles several sub-quantisers
    ("0", "1", "333", "", "0"),   # odd grid
    ("0", "0", "2048", "", "1"),  # as many blocks as the workspace allows
    ("1", "1", "", "64", "1"),    # round-1 kernel, LDS potentials, short blocks
    ("1", "0", "", "192", "0"),   # round-1 kernel, register potentials
    ("1", "1", "", "512", "1"),   # round-1 kernel as timed in round 1
]


@pytest.mark.parametrize("variant", SK_VARIANTS, ids=lambda v: "v1=%s,fklds=%s,nb=%s,cpb=%s,fuse=%s" % v[:5] + (",prio=%s" % v[5] if len(v) > 5 else ""))
@pytest.mark.parametrize("name", ["m48_b6144_sample", "m48_b1000_ragged", "m96_b512_blend", "m8_b2048_sample",
                                  "m24_b1024_sample"])
def test_every_sweep_variant_reproduces_the_golden_codes(name, variant, monkeypatch):
    """The environment switches are read on every call (csrc/rc_common.h rc_env_int), so each kernel variant — among
    them the one bench.py times — runs against the reference's own codes (fixtures generated by importing the reference,
    oracle/gen_golden.py)."""
    from repconc_amd import ops
    variant = tuple(variant) + ("",) * (6 - len(variant))
    for key, val in zip(("RC_SK_V1", "RC_SK_FKLDS", "RC_SK_NB", "RC_SK_CPB", "RC_FUSE_CENTRE", "RC_SK_PRIO"), variant):
        if val == "":
            monkeypatch.delenv(key, raising=False)
        else:
            monkeypatch.setenv(key, val)
    g, x, C = load_case(name)
    codes, flags = ops.assign_sinkhorn(_t(x), _t(C), EPS, ITERS, torch.uint8)
    assert int(flags.item()) == 0
    assert int((codes.cpu().numpy() != g["codes_constrained"]).sum()) == 0


@pytest.mark.parametrize("kind", ["sample", "lloyd"])
def test_full_training_batch_against_the_reference_49152_m48(kind):
    """BASELINE configs[1], the shape bench.py times, pinned to the REFERENCE directly: tests/golden/headline_b49152_m48_*.npz
    hold what /root/reference's RepCONC.quantize returned for this 49 152 x 768 batch at M = 48 (constrained, eps 0.003,
    T = 100, and nearest), computed as four column slices of twelve sub-quantisers (oracle/gen_golden.py --headline: the
    reference cannot hold its [48, 49152, 256, 16] scratch in 62 GB, and every reduction of quantize is per sub-quantiser).
    All 2 359 296 constrained codes and all nearest codes of the HIP path equal them, for sampled and Lloyd-refined centroids."""
    from conftest import load_headline
    from repconc_amd import ops
    x, C, con, near = load_headline(kind)
    xt, Ct = _t(x), _t(C)
    got, flags = ops.assign_sinkhorn(xt, Ct, EPS, ITERS, torch.uint8)
    assert int(flags.item()) == 0
    assert int((got.cpu().numpy() != con).sum()) == 0
    assert np.array_equal(ops.assign_nearest(xt, Ct, torch.uint8).cpu().numpy(), near)
    g64, _ = ops.assign_sinkhorn(xt, Ct, EPS, ITERS, torch.int64)          # the reference's own dtype
    assert np.array_equal(g64.cpu().numpy().astype(np.uint8), con)


@pytest.mark.parametrize("kind", ["sampled", "lloyd"])
def test_full_training_batch_against_the_oracle_49152_m48(kind):
    """BASELINE configs[1]: ONE whole 49 152 x 768 training batch, M = 48, eps 0.003, T = 100 — every one of the
    2 359 296 constrained codes (and the nearest codes) against the C restatement of the reference (~35 s of host time on
    the GPU box per case).  Runs the default sweep at the grid bench.py times."""
    from repconc_amd import ops
    B, M = 49152, 48
    x = synth.clustered_embeddings(777, B, n_clusters=512)
    C = synth.sample_centroids(778, x, M)
    xt = _t(x)
    if kind == "lloyd":
        Ct = _t(C)
        for _ in range(5):                                      # Lloyd refinement on the GPU kernels
            codes = ops.assign_nearest(xt, Ct, torch.uint8)
            sums, counts = ops.kmeans_stats(xt, codes)
            ops.kmeans_update_(sums, counts, Ct)
        C = Ct.cpu().numpy()
    want, fl = c_oracle.quantize(x, C, True, EPS, ITERS)
    got, flags = ops.assign_sinkhorn(xt, _t(C), EPS, ITERS, torch.uint8)
    assert fl == 0 and int(flags.item()) == 0
    got = got.cpu().numpy()
    assert int((got != want).sum()) == 0
    hist = np.stack([np.bincount(got[:, m], minlength=256) for m in range(M)])
    assert hist.min() > 0.6 * B / 256 and hist.max() < 1.4 * B / 256      # the constraint holds (nearest codes: 1 .. 1441)
    near = ops.assign_nearest(xt, _t(C), torch.uint8).cpu().numpy()
    assert np.array_equal(near, c_oracle.quantize(x, C, False)[0])


@pytest.mark.parametrize("M", [24, 96])
def test_full_training_batch_other_widths_against_the_reference(M):
    """BASELINE c5 / c4 (49 152 x 768 at M = 24, dsub 32, and M = 96, dsub 8) pinned to the REFERENCE like the headline:
    tests/golden/headline_b49152_m<M>_sample.npz = RepCONC.quantize of /root/reference on four column slices of 192 columns
    (oracle/gen_golden.py --headline sample --headline-m M).  Every constrained and every nearest code of the HIP path."""
    from conftest import load_headline
    from repconc_amd import ops
    x, C, con, near = load_headline("sample", M)
    xt, Ct = _t(x), _t(C)
    got, flags = ops.assign_sinkhorn(xt, Ct, EPS, ITERS, torch.uint8)
    assert int(flags.item()) == 0
    assert int((got.cpu().numpy() != con).sum()) == 0
    assert np.array_equal(ops.assign_nearest(xt, Ct, torch.uint8).cpu().numpy(), near)


def test_per_rank_shape_6144_against_the_oracle():
    """BASELINE configs[2] per-rank shape (6144 x 768, M = 48) as a stand-alone batch vs the oracle, default sweep and
    the register-potential variant (the grid of this shape is the one the 8-GPU recipe runs per rank)."""
    from repconc_amd import ops
    x = synth.clustered_embeddings(4242, 6144)
    C = synth.sample_centroids(4243, x, 48)
    want, _ = c_oracle.quantize(x, C, True, EPS, ITERS)
    got, flags = ops.assign_sinkhorn(_t(x), _t(C), EPS, ITERS, torch.uint8)
    assert int(flags.item()) == 0 and np.array_equal(got.cpu().numpy(), want)


def test_epsilon_outside_the_supported_range_is_reported():
    """eps far below anything the reference itself can run (its exp(1/eps) overflows at eps < 1.4e-3): the range flag."""
    from repconc_amd import _lib, ops
    g, x, C = load_case("m8_b300_gauss")
    codes, flags = ops.assign_sinkhorn(_t(x), _t(C), 1e-5, 5, torch.uint8)
    assert int(flags.item()) & _lib.RC_FLAG_RANGE


@pytest.mark.parametrize("M,N,nq,k", [(48, 300017, 9, 1000), (96, 270001, 5, 10), (64, 262144, 4, 50),
                                      (32, 400003, 3, 200), (16, 300000, 11, 100), (48, 1_000_003, 17, 3000)])
def test_adc_conflict_free_screen_equals_oracle_and_old_screen(M, N, nq, k, monkeypatch):
    """The conflict-free screen (permuted code image, rotated sub-quantiser order, 16x16x64 i8 MFMA): ids and score
    bits equal the brute-force oracle — with the image rebuilt per call (ops.adc_search), with the image kept by an index
    that was filled in ragged chunks (PQIndex.add_codes), and they equal the round-1 screen (RC_ADC_OLD_SCREEN=1)."""
    from repconc_amd import ops
    from repconc_amd.index import PQIndex
    C, codes, q = _adc_case(M, N, nq, seed=M * 13 + N)
    codes[N // 3: N // 3 + 500] = codes[:500]          # duplicated rows: ties across the candidate boundary
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    ct = _t(codes)
    s1, i1 = ops.adc_search(ct, _t(C), _t(q), k)
    assert np.array_equal(i1.cpu().numpy(), wi)
    assert np.array_equal(s1.cpu().numpy().view(np.uint32), ws.view(np.uint32))
    idx = PQIndex(768, M)
    idx.set_centroids(C)
    cuts = [0, 7, 7 + 16 * 1000 + 3, N // 2 + 5, N]
    for a, b in zip(cuts[:-1], cuts[1:]):
        idx.add_codes(ct[a:b])
    assert idx._image is not None and torch.equal(idx.codes, ct)
    s2, i2 = idx.search(_t(q), k)
    assert torch.equal(i2, i1) and torch.equal(s2, s1)
    monkeypatch.setenv("RC_ADC_OLD_SCREEN", "1")
    s3, i3 = ops.adc_search(ct, _t(C), _t(q), k)
    assert torch.equal(i3, i1) and torch.equal(s3, s1)


def test_adc_scan_image_is_a_row_permutation():
    """Every row of the image holds exactly the bytes of the canonical row, permuted by a rule that depends on
    n mod 16 only; converting a row range leaves the other rows of the image alone."""
    from repconc_amd import ops
    for M in (16, 32, 48, 64, 96):
        base = synth.uniform_codes(5 + M, 16, M)
        codes = _t(np.concatenate([base] * 40 + [synth.uniform_codes(6 + M, 3, M)], 0))        # 643 rows, period 16
        # the IVF search's image ("rows"): blocked by chunks of 16 rows; the library's host-side description gives the byte
        # offset of codes[n][m] — image[at(n, m)] == codes[n][m] is the whole contract
        N = codes.shape[0]
        nb = ops.adc_image_rows_bytes(N, M)
        assert nb == (N + 15) // 16 * 16 * M and ops.adc_image_rows_at(M, N, 0) >= 0 and ops.adc_image_rows_at(M, 0, M) == -1
        at = np.array([[ops.adc_image_rows_at(M, n, m) for m in range(M)] for n in range(N)], dtype=np.int64)
        assert at.min() == 0 and at.max() < nb and len(np.unique(at)) == N * M            # a bijection onto the chunks' bytes
        assert np.array_equal(at[16:32] - at[:16], np.full((16, M), 16 * M))              # the rule depends on n mod 16 only
        blank = torch.full((nb,), 255, dtype=torch.uint8, device=DEV)
        img = ops.adc_scan_image_(codes, blank.clone(), layout="rows")
        assert np.array_equal(img.cpu().numpy()[at], codes.cpu().numpy())
        part = blank.clone()
        ops.adc_scan_image_(codes, part, 100, 37, layout="rows")
        pn = part.cpu().numpy()
        assert np.array_equal(pn[at[100:137]], codes.cpu().numpy()[100:137])
        untouched = np.ones(nb, dtype=bool)
        untouched[at[100:137].ravel()] = False
        assert bool((pn[untouched] == 255).all())
        # the flat-search image: row-major for the one-phase 8-query screen; tile-blocked and phase-major for the two-phase
        # one (M = 96 without the 16-query screen) and for the 16-query screen (rc_adc_q16_describe says which M use it)
        import ctypes
        from repconc_amd import _lib
        lib = _lib.load()
        flat = torch.full((ops.adc_image_bytes(codes.shape[0], M),), 255, dtype=torch.uint8, device=DEV)
        ops.adc_scan_image_(codes, flat)
        slot = ctypes.c_int(0)
        uses_q16 = lib.rc_adc_q16_describe(M, 0, 0, ctypes.byref(slot))
        n, T = codes.shape[0], 32768
        if uses_q16 == 1:
            # [phase][round of 2048 rows][wave 16][lane = r + 16 g][chunk c 8][step j 4], row = 2048 round + 128 wave + 16 c + r
            assert flat.numel() == T * M
            tile = flat.view(M // 16, T // 2048, 16, 64, 8, 4).cpu().numpy()
            hc = codes.cpu().numpy()
            seen = np.zeros_like(tile, dtype=bool)
            for rr in range(16):
                for gq in range(4):
                    for j in range(4):
                        assert lib.rc_adc_q16_describe(M, rr + 16 * gq, j, ctypes.byref(slot)) == 1
                        rows = np.arange(rr, n, 16)
                        rd, wv_, cc = rows // 2048, (rows % 2048) // 128, (rows % 128) // 16
                        for ph in range(M // 16):
                            assert np.array_equal(tile[ph, rd, wv_, rr + 16 * gq, cc, j], hc[rows, 16 * ph + slot.value])
                            seen[ph, rd, wv_, rr + 16 * gq, cc, j] = True
            assert bool((tile[~seen] == 255).all())
        elif M != 96:
            assert flat.numel() == img.numel() and torch.equal(flat.view_as(img), img)
        else:
            assert flat.numel() == T * 96
            tile = flat.view(2, T, 48)
            assert torch.equal(tile[0, :n], img[:, :48]) and torch.equal(tile[1, :n], img[:, 48:])
            assert bool((tile[:, n:] == 255).all())


def test_adc_q16_layout_is_conflict_free_and_covers_every_sub_quantiser():
    """Host-side check of the 16-query screen's gather pattern (rc_adc_q16_describe): in every step the 16 lanes of each
    ds_read_b128 service group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, and +32) read 16 different 16-byte slots, and the
    four lanes of a row visit each of the 16 sub-quantisers of a phase exactly once."""
    import ctypes
    from repconc_amd import _lib
    lib = _lib.load()
    slot = ctypes.c_int(0)

    def sl(lane, j):
        assert lib.rc_adc_q16_describe(48, lane, j, ctypes.byref(slot)) >= 0
        return slot.value
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in grp] for grp in groups]
    for j in range(4):
        for grp in groups:
            assert sorted(sl(x, j) for x in grp) == list(range(16))
    for r in range(16):
        assert sorted(sl(r + 16 * gq, j) for gq in range(4) for j in range(4)) == list(range(16))


@pytest.mark.parametrize("name", ["m48_b1024_sample", "m48_b1000_ragged", "m8_b2048_sample"])
def test_iteration_graph_equals_eager_loop(name, monkeypatch):
    """csrc/comm.hip: sweeps t >= 2 are replayed from a captured hipGraph (cached per workspace / shape); the codes
    must equal the eager loop's and the goldens, on the first (capture) and on later (replay) calls, one and two chains."""
    from repconc_amd import ops
    g, x, C = load_case(name)
    xt, Ct = _t(x), _t(C)
    for split in ("0", "1"):
        monkeypatch.setenv("RC_DIST_SPLIT", split)
        monkeypatch.setenv("RC_GRAPH", "0")
        eager, fl0 = ops.assign_sinkhorn(xt, Ct, EPS, ITERS, torch.uint8)
        monkeypatch.setenv("RC_GRAPH", "1")
        for _ in range(3):
            got, fl = ops.assign_sinkhorn(xt, Ct, EPS, ITERS, torch.uint8)
            assert int(fl.item()) == 0 and torch.equal(got, eager)
        assert np.array_equal(eager.cpu().numpy(), g["codes_constrained"])
    # non-finite input: the flag raised inside the captured kernels reaches the caller's flag word
    bad = xt.clone()
    bad[3, 5] = float("nan")
    _, fl = ops.assign_sinkhorn(bad, Ct, EPS, ITERS, torch.uint8)
    assert int(fl.item()) != 0


def test_native_rccl_collectives_inside_the_graph_single_rank(monkeypatch):
    """RC_DIST_FORCE_COLL=1 makes the one-rank communicator issue every RCCL call of the multi-rank loop (range
    all-reduces, one all-gather per sweep and chain), eagerly and from inside the captured graph."""
    import socket
    import torch.distributed as dist
    from repconc_amd import ops
    created = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device(DEV))
        created = True
    try:
        ops.comm_init(transport="rccl")                              # this test is about the RCCL calls (auto would pick ipc)
        monkeypatch.setenv("RC_DIST_FORCE_COLL", "1")
        g, x, C = load_case("m48_b1024_sample")
        for split in ("0", "1"):
            for graph in ("0", "1", "1"):
                monkeypatch.setenv("RC_DIST_SPLIT", split)
                monkeypatch.setenv("RC_GRAPH", graph)
                codes, flags = ops.assign_sinkhorn_dist(_t(x), _t(C), EPS, ITERS, torch.uint8)
                torch.cuda.synchronize()
                assert int(flags.item()) == 0
                assert np.array_equal(codes.cpu().numpy(), g["codes_constrained"]), (split, graph)
    finally:
        if created:
            dist.destroy_process_group()


def test_fused_exchange_on_a_one_rank_ipc_transport(monkeypatch):
    """RC_DIST_FORCE_COLL=1 on a one-rank IPC transport (the proxy bench.py and DESIGN.md §5 time): every iteration's exchange
    runs — the sweep's reducer stores the row sums and the flags into the rank's OWN receive buffer, the next sweep's prologue
    waits for them (no peer shares the device, so the wait is inside the sweep) — eager, captured, replayed, one chain and two,
    and the round-3/4 push + wait kernels; golden codes every time, at a fixture and at the per-rank shape of the 8-GPU recipe."""
    import socket
    import torch.distributed as dist
    from repconc_amd import _lib, ops
    created = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        created = True
    try:
        assert ops.comm_init(transport="ipc") == "ipc"
        lib, h = _lib.load(), _lib.handle(torch.cuda.current_device())
        monkeypatch.setenv("RC_DIST_FORCE_COLL", "1")
        monkeypatch.setenv("RC_IPC_TIMEOUT_MS", "20000")
        assert lib.rc_solve_num_chains_on(h, 8, 48) == 1 and lib.rc_solve_num_chains(8, 48) == 2
        g, x, C = load_case("m48_b1024_sample")
        x6 = synth.clustered_embeddings(4242, 6144)
        C6 = synth.sample_centroids(4243, x6, 48)
        want6 = None
        for xsweep, inwait, split in (("1", "1", "0"), ("1", "0", "0"), ("1", "1", "1"), ("0", "0", "1"), ("0", "0", "0")):
            monkeypatch.setenv("RC_IPC_XSWEEP", xsweep)
            monkeypatch.setenv("RC_IPC_INWAIT", inwait)
            monkeypatch.setenv("RC_DIST_SPLIT", split)
            for graph in ("0", "1", "1"):
                monkeypatch.setenv("RC_GRAPH", graph)
                codes, flags = ops.assign_sinkhorn_dist(_t(x), _t(C), EPS, ITERS, torch.uint8)
                torch.cuda.synchronize()
                assert int(flags.item()) == 0
                assert np.array_equal(codes.cpu().numpy(), g["codes_constrained"]), (xsweep, inwait, split, graph)
            codes, flags = ops.assign_sinkhorn_dist(_t(x6), _t(C6), EPS, ITERS, torch.uint8)
            if want6 is None:
                want6, _ = c_oracle.quantize(x6, C6, True, EPS, ITERS)
            assert int(flags.item()) == 0 and np.array_equal(codes.cpu().numpy(), want6), (xsweep, inwait, split)
        # a chain moves [M, 256] fp64 through a 256 KiB slot — MCQ_M = 192 in ONE chain is refused before anything is
        # enqueued (with the way out in the message); as two chains of 96 it runs and gives the oracle's codes
        xw = synth.clustered_embeddings(4250, 1024)
        Cw = synth.sample_centroids(4251, xw, 192)
        monkeypatch.setenv("RC_IPC_XSWEEP", "1")
        monkeypatch.setenv("RC_IPC_INWAIT", "1")
        monkeypatch.setenv("RC_DIST_SPLIT", "0")
        with pytest.raises(_lib.RepconcHipError, match="RC_COMM=rccl"):
            ops.assign_sinkhorn_dist(_t(xw), _t(Cw), EPS, ITERS, torch.uint8)
        monkeypatch.setenv("RC_DIST_SPLIT", "1")
        codes, flags = ops.assign_sinkhorn_dist(_t(xw), _t(Cw), EPS, ITERS, torch.uint8)
        assert int(flags.item()) == 0 and np.array_equal(codes.cpu().numpy(), c_oracle.quantize(xw, Cw, True, EPS, ITERS)[0])
        ops.comm_check()
    finally:
        try:
            ops.comm_destroy(group_barrier=False)
        except Exception:
            pass
        if created:
            dist.destroy_process_group()


def _two_rank_worker(rank, world, port, name, ret):
    import torch.distributed as dist
    from repconc_amd import ops
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    try:
        g, x, C = load_case(name)
        B = x.shape[0]
        cuts = [0, B // 2, B] if name != "m48_b1000_ragged" else [0, B, B]      # ragged: rank 1 holds no rows
        xl = torch.from_numpy(x[cuts[rank]:cuts[rank + 1]]).to(dev)
        ops.comm_init(transport="rccl")                              # the IPC transport has its own multi-process tests
        out = []
        for graph in ("0", "1", "1"):
            os.environ["RC_GRAPH"] = graph
            codes, flags = ops.assign_sinkhorn_dist(xl, torch.from_numpy(C).to(dev), EPS, ITERS, torch.uint8)
            torch.cuda.synchronize()
            out.append((codes.cpu().numpy(), int(flags.item())))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("name", ["m48_b1024_sample", "m8_b2048_sample", "m48_b1000_ragged"])
def test_native_rccl_solve_on_two_real_ranks(name):
    """rc_pq_assign_sinkhorn_dist on two processes / two GPUs (the reference's dist.is_initialized() branch,
    modeling_repconc.py:78-80,149-157): the concatenated shard codes equal the unsharded reference codes, with the eager
    loop and with the captured graph; a rank without rows takes part in every collective instead of hanging the others."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_two_rank_worker, args=(2, port, name, ret), nprocs=2, join=True)
    g, _, _ = load_case(name)
    for i in range(3):
        got = np.concatenate([ret[0][i][0], ret[1][i][0]], 0)
        assert ret[0][i][1] == 0 and ret[1][i][1] == 0
        assert np.array_equal(got, g["codes_constrained"]), (name, i)


@pytest.mark.parametrize("mode", ["replicated", "sharded"])
def test_load_index_to_all_gpus_in_process(mode):
    """load_index_to_gpu(index, None) (evaluate_repconc.py:131-134): replicas with the query batch split, or row shards
    with merged top-k lists — on every visible GPU, and as two virtual parts on one device when only one is visible.
    Results (ids and score bits) equal the single-device search and the oracle."""
    from repconc_amd.index import PQIndex
    from repconc_amd.models.repconc.evaluate_repconc import load_index_to_gpu, batch_search
    C, codes, q = _adc_case(48, 300011, 13, seed=4711)
    idx = PQIndex(768, 48)
    idx.set_centroids(C)
    idx.add_codes(codes)
    want_s, want_i = idx.search(_t(q), 100)
    ndev = torch.cuda.device_count()
    devs = list(range(ndev)) if ndev >= 2 else [0, 0, 0]
    multi = load_index_to_gpu(idx, None, shard=(mode == "sharded"), devices=devs)
    assert type(multi).__name__ == ("ShardedPQIndex" if mode == "sharded" else "ReplicatedPQIndex")
    assert multi.ntotal == idx.ntotal and multi.pq.M == 48
    s, i = multi.search(_t(q), 100)
    assert torch.equal(i.to(DEV), want_i) and torch.equal(s.to(DEV), want_s)
    sn, inn = multi.search(q, 100)                                     # numpy in -> numpy out (evaluate_repconc.py:182)
    assert np.array_equal(inn, want_i.cpu().numpy())
    ws, wi = c_oracle.adc_search(codes, C, q, 100)
    assert np.array_equal(inn, wi) and np.array_equal(sn.view(np.uint32), ws.view(np.uint32))
    corpus_ids = np.arange(codes.shape[0]) * 3 + 7
    bs, bi = batch_search(np.arange(13), q, corpus_ids, multi, 100, 5)
    assert np.array_equal(bi, corpus_ids[wi])
    assert load_index_to_gpu(idx, None, devices=[0]) is idx


class _HashTokenizer:
    """Whitespace tokenizer over a 500-word vocabulary (ids by hash); pads to the longest text of the batch."""
    sep_token = "[SEP]"

    def __call__(self, texts, padding=True, truncation=True, max_length=32):
        rows = [[1] + [3 + (zlib.crc32(w.encode()) % 490) for w in t.split()][: max_length - 2] + [2] for t in texts]
        L = max(map(len, rows))
        return {"input_ids": [r + [0] * (L - len(r)) for r in rows],
                "attention_mask": [[1] * len(r) + [0] * (L - len(r)) for r in rows]}


def test_evaluation_pipeline_with_the_reference_call_sequence(tmp_path):
    """The call sequence of evaluate/run_repconc_eval.py (load_or_encode_corpus :36-60, search :88-110) against this
    package's names: encode_corpus -> index file + corpus_ids.npy -> read back -> from_pq_to_ivfpq -> load_index_to_gpu
    -> encode_query -> batch_search.  Checked against per-text model calls and a brute-force score of the decoded index."""
    from transformers import BertConfig
    from repconc_amd import faiss_compat as faiss
    from repconc_amd.faiss_io import load_index_dir, save_index_dir
    from repconc_amd.models.dense import BertDense
    from repconc_amd.models.repconc import RepCONC
    from repconc_amd.models.repconc.evaluate_repconc import (EvalArguments, ModelArguments, batch_search, encode_corpus,
                                                             encode_query, from_pq_to_ivfpq, load_index_to_gpu)
    torch.manual_seed(1)
    cfg = BertConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=128, vocab_size=500,
                     max_position_embeddings=40, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg.MCQ_M, cfg.MCQ_K, cfg.similarity_metric, cfg.pooling = 48, 256, "METRIC_IP", "mean"
    model = RepCONC(cfg, BertDense(cfg), False, None, None).to(DEV)
    with torch.no_grad():
        model.centroids.mul_(0.05)
        model.rotation.copy_(torch.linalg.qr(torch.randn(768, 768, device=DEV))[0])
    rng = np.random.default_rng(5)
    words = [f"w{i}" for i in range(300)]
    corpus = {f"d{i}": " ".join(rng.choice(words, rng.integers(3, 25))) for i in range(257)}
    queries = {100 + i: " ".join(rng.choice(words, rng.integers(2, 8))) for i in range(11)}
    tok = _HashTokenizer()
    margs = ModelArguments(model_name_or_path="unused", max_seq_length=32)
    eargs = EvalArguments(output_dir=str(tmp_path / "out"), per_device_eval_batch_size=50, report_to=[])
    assert eargs.topk == 1000 and eargs.search_batch == 1200 and margs.doc_encoder_path == "unused"
    index, corpus_ids = encode_corpus(corpus, model, tok, margs.max_seq_length, eargs)
    assert index.ntotal == 257 and len(corpus[corpus_ids[0]]) >= len(corpus[corpus_ids[-1]])     # longest first
    # codes: what a per-document forward(return_code=True) gives (the reference's prediction_step, :51-75)
    col = lambda t: {k: torch.tensor(v, device=DEV) for k, v in tok([t]).items()}
    for row in (0, 100, 256):
        want = model(return_code=True, **col(corpus[corpus_ids[row]])).discrete_codes.to(torch.uint8)
        assert torch.equal(index.codes[row:row + 1], want)
    # ... and against the ORACLE (not only the HIP path against itself): nearest codes of the model's own continuous
    # embeddings by the numpy restatement of the reference's arithmetic
    sample_rows = [0, 1, 57, 100, 255, 256]
    cont = torch.cat([model(**col(corpus[corpus_ids[r]])).continuous_embeds for r in sample_rows]).detach().float().cpu().numpy()
    want_codes = pq_oracle.quantize(cont, model.centroids.detach().cpu().numpy(), False).astype(np.uint8)
    assert np.array_equal(index.codes[sample_rows].cpu().numpy(), want_codes)
    save_index_dir(index, corpus_ids, str(tmp_path / "corpus"))
    index2, ids2 = load_index_dir(str(tmp_path / "corpus"))
    assert np.array_equal(ids2, corpus_ids) and torch.equal(index2.codes, index.codes)
    faiss.copy_array_to_vector(model.centroids.detach().cpu().numpy().ravel(), index2.pq.centroids)   # run_repconc_eval.py:123-127
    index2 = load_index_to_gpu(from_pq_to_ivfpq(index2), 0)
    qemb, qids = encode_query(queries, model, tok, 16, eargs)
    assert qemb.shape == (11, 768) and qids.tolist() == sorted(queries)
    scores, ids = batch_search(qids, qemb, corpus_ids, index2, 10, batch_size=4)
    recon = index2.reconstruct_n(0, 257)
    brute = torch.from_numpy(qemb).to(DEV) @ recon.T
    top = torch.topk(brute, 10, dim=1)
    np.testing.assert_allclose(scores, top.values.cpu().numpy(), rtol=0, atol=2e-4)
    assert (ids == corpus_ids[top.indices.cpu().numpy()]).mean() > 0.99
    assert np.array_equal(faiss.vector_to_array(index2.codes).reshape(257, 48), index.codes.cpu().numpy())


def test_indexpq_reader_on_the_hand_assembled_golden_file():
    from repconc_amd.faiss_io import read_index
    exp = np.load(os.path.join(GOLDEN, "ixpq_d8_m2_n5_expected.npz"))
    idx = read_index(os.path.join(GOLDEN, "ixpq_d8_m2_n5.faissindex"), device=DEV)
    assert (idx.pq.d, idx.pq.M, idx.ntotal, idx.metric_type, idx.is_trained) == (8, 2, 5, 0, True)
    assert np.array_equal(idx.pq.centroids.cpu().numpy(), exp["centroids"])
    assert np.array_equal(idx.codes.cpu().numpy(), exp["codes"])


class _PtHashTokenizer(_HashTokenizer):
    def __call__(self, texts, padding=True, truncation=True, max_length=32, return_tensors=None, **_unused):
        enc = super().__call__(texts, padding, truncation, max_length)
        return {k: torch.tensor(v, dtype=torch.long) for k, v in enc.items()} if return_tensors == "pt" else enc


def test_repconc_finetuner_on_transformers5_gradients_and_train_loop(tmp_path):
    """`RepCONCFinetuner` (models/repconc/finetune_repconc.py of this package) driven like the reference's trainer
    (finetune_repconc.py:225-344): built from `RepCONCFinetuneArguments` + `FinetuneCollator`, one `training_step` on a
    batch with explicit hard negatives (GradCache `forward_no_grad` -> constrained quantize -> `build_cache` ->
    `_forward_backward`) gives the gradients of the one-pass objective; `create_optimizer` makes the three groups; and
    `train()` runs on the installed transformers."""
    from transformers import BertConfig
    from repconc_amd.models.dense import BertDense
    from repconc_amd.models.repconc import RepCONC
    from repconc_amd.models.repconc.finetune_repconc import FinetuneCollator, RepCONCFinetuneArguments, RepCONCFinetuner
    torch.manual_seed(0)
    random_state = np.random.default_rng(3)
    cfg = BertConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=128, vocab_size=500,
                     max_position_embeddings=40, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.0)
    cfg.MCQ_M, cfg.MCQ_K, cfg.similarity_metric, cfg.pooling = 48, 256, "METRIC_IP", "mean"
    model = RepCONC(cfg, BertDense(cfg), True, 0.003, 20).to(DEV)
    with torch.no_grad():
        model.centroids.mul_(0.05)
    words = [f"w{i}" for i in range(200)]
    text = lambda lo, hi: " ".join(random_state.choice(words, random_state.integers(lo, hi)))
    nq, npq = 16, 3
    feats = [{"query": text(2, 6), "pos_doc": text(5, 20), "qid": i, "pos_docid": 1000 + i,
              "neg_docs": [text(5, 20) for _ in range(npq)], "neg_docids": [2000 + npq * i + j for j in range(npq)]}
             for i in range(nq)]
    feats[5]["neg_docids"][0] = feats[3]["pos_docid"]                 # a duplicate ...
    qrels = {i: [1000 + i] for i in range(nq)}
    qrels[2].append(feats[7]["neg_docids"][1])                        # ... and a false negative
    args = RepCONCFinetuneArguments(output_dir=str(tmp_path / "o"), per_device_train_batch_size=nq, cache_chunk_size=6,
                                    mse_loss_weight=1e-2, dynamic_topk_hard_negative=7, centroid_learning_rate=5e-4,
                                    learning_rate=2e-5, max_steps=2, logging_steps=1, save_strategy="no", report_to=[],
                                    dataloader_drop_last=True, seed=2022)
    trainer = RepCONCFinetuner(qrels=qrels, model=model, args=args, train_dataset=feats,
                               data_collator=FinetuneCollator(_PtHashTokenizer(), 8, 24))
    batch = trainer.data_collator(feats)
    model.zero_grad()
    loss = trainer.training_step(model, batch)
    got = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert torch.isfinite(loss) and float(got["centroids"].abs().sum()) > 0
    # one-pass objective on the same codes; dropout masks differ, so compare in eval-mode-free terms: dropout 0.1 on
    # a 1-layer encoder would make the passes differ — replay is what RandContext is for, so re-run with it disabled
    cfg.hidden_dropout_prob = 0.0
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.zero_grad()
    loss = trainer.training_step(model, batch)
    got = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    dev_batch = {k: {kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV) for k, v in batch.items()}
    q = model(**dev_batch["query_input"]).continuous_embeds
    p = model(**dev_batch["pos_doc_input"]).continuous_embeds
    n = model(**dev_batch["neg_doc_input"]).continuous_embeds
    docs = torch.cat([p, n], 0)
    with torch.no_grad():
        codes = model.quantize(docs)
    quant = model.decode(codes)
    ste = quant.detach() + (docs - docs.detach()) + (quant - quant.detach())
    direct = trainer.compute_contrastive_loss(q, ste, dev_batch["qids"], torch.cat([dev_batch["pos_docids"], dev_batch["neg_docids"]]))
    mse = 0
    for rep, qt in ((p, quant[:nq]), (n, quant[nq:])):
        for a in range(0, rep.shape[0], 6):
            mse = mse + ((qt[a:a + 6] - rep[a:a + 6]) ** 2).sum(-1).mean() * args.mse_loss_weight
    (direct + mse).backward()
    assert abs(float(direct.detach()) - float(loss)) < 1e-4
    for name, grad in got.items():
        assert torch.allclose(grad, dict(model.named_parameters())[name].grad, rtol=2e-3, atol=2e-5), name
    opt = trainer.create_optimizer()
    assert len(opt.param_groups) == 3 and opt.param_groups[2]["lr"] == 5e-4 and len(opt.param_groups[2]["params"]) == 1
    before = model.centroids.detach().clone()
    out = trainer.train()
    assert out.global_step == 2 and np.isfinite(out.training_loss)
    assert not torch.equal(before, model.centroids.detach())


def test_ivf_coarse_assignment_kernel_is_the_nearest_cell():
    """rc_ivf_coarse_assign (fp32 matrix cores, fused argmin): every document goes to a cell whose fp64 distance equals
    the fp64 minimum up to fp32 rounding of the GEMM form; exact ties (duplicated centroids) take the lower cell;
    ragged sizes (B, nlist not multiples of the 128-tile)."""
    from repconc_amd.ivf import coarse_assign
    rng = np.random.default_rng(9)
    for B, nlist in ((1000, 300), (4097, 5000), (77, 129)):
        x = rng.standard_normal((B, 768)).astype(np.float32)
        cent = rng.standard_normal((nlist, 768)).astype(np.float32)
        cent[nlist // 2] = cent[3]                                  # an exact duplicate: cell 3 must win over nlist/2
        x[:5] = cent[3] + 1e-3 * rng.standard_normal((5, 768)).astype(np.float32)
        got = coarse_assign(_t(x), _t(cent)).cpu().numpy()
        d = ((x.astype(np.float64)[:, None, :] - cent.astype(np.float64)[None, :, :]) ** 2).sum(-1) if B * nlist < 2e6 else None
        if d is None:
            xx, cc = torch.from_numpy(x).double().to(DEV), torch.from_numpy(cent).double().to(DEV)
            d = (torch.cdist(xx, cc) ** 2).cpu().numpy()
        best = d.min(1)
        mine = d[np.arange(B), got]
        assert np.all(mine <= best + 2e-4 * (1.0 + best)), (B, nlist)
        assert np.all(got[:5] == 3)
        assert (got == d.argmin(1)).mean() > 0.999


def test_ivf_m96_nlist5000_against_the_oracle():
    """BASELINE configs[3] shape: M = 96 (dsub 8), IVF with nlist = 5000 cells (HIP coarse assignment), 300 k rows.
    (a) every nprobe: ids and score bits equal the brute-force oracle on the same cells; (b) nprobe = nlist equals the
    flat M = 96 search; (c) the cells are the nearest coarse centroids."""
    from repconc_amd import ops
    from repconc_amd.index import PQIndex
    from repconc_amd.ivf import IVFPQIndex
    N, M, nlist, nq = 300000, 96, 5000, 7
    x = synth.clustered_embeddings(51, N, n_clusters=256)
    C = synth.sample_centroids(52, x[:8192], M)
    q = x[np.random.default_rng(53).integers(0, N, nq)] + 0.2 * synth.gaussian(54, (nq, 768))
    ivf = IVFPQIndex(768, M, nlist, device=DEV)
    ivf.set_centroids(C)
    ivf.train(x[:60000], iters=3)
    ivf.add(x)
    codes = ops.assign_nearest(_t(x), _t(C), torch.uint8)
    lens = (ivf.list_off[1:] - ivf.list_off[:-1])
    list_ids = torch.empty(N, dtype=torch.int64, device=DEV)
    list_ids[ivf.ids] = torch.repeat_interleave(torch.arange(nlist, device=DEV), lens)
    assert torch.equal(ivf.codes, codes[ivf.ids]) and int(lens.sum()) == N
    xs, cs = torch.from_numpy(x[:2000]).double().to(DEV), ivf.coarse.double()
    d = torch.cdist(xs, cs) ** 2
    assert float((d.gather(1, list_ids[:2000, None])[:, 0] <= d.min(1).values * (1 + 1e-5) + 1e-4).float().mean()) == 1.0   # (c)
    cn, ln, co = codes.cpu().numpy(), list_ids.cpu().numpy(), ivf.coarse.cpu().numpy()
    for nprobe, k in ((1, 10), (16, 100), (128, 1000), (700, 1000)):
        ws, wi = pq_oracle.ivf_search(q, C, cn, ln, co, k, nprobe)
        for method in ("lists", "scan"):                                           # list-centric screen / per-query scan
            s, i = ivf.search(q, k, nprobe, method=method)
            assert np.array_equal(i, wi), (nprobe, method)                         # (a)
            assert np.array_equal(s.view(np.uint32), ws.view(np.uint32)), (nprobe, method)
    flat = PQIndex(768, M, device=DEV)
    flat.set_centroids(C)
    flat.add_codes(codes)
    fs, fi = flat.search(_t(q), 200)
    s, i = ivf.search(_t(q), 200, nprobe=nlist)
    assert torch.equal(i, fi) and torch.equal(s, fs)                              # (b)


def _warmup_worker(rank, world, port, ret):
    import torch.distributed as dist
    from types import SimpleNamespace
    from repconc_amd.models.repconc import RepCONC
    from repconc_amd.train.run_warmup import warmup_from_embeds
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    try:
        x = synth.clustered_embeddings(91, 6000)
        cfg = SimpleNamespace(hidden_size=768, MCQ_M=48, MCQ_K=256, similarity_metric="METRIC_IP")
        enc = _TableEncoder(torch.zeros(4, 768))
        model = RepCONC(cfg, enc, False, None, None).to(dev)
        model, index = warmup_from_embeds(x[rank * 3000:(rank + 1) * 3000], model, opq_iters=3, pq_iters=3)
        ret[rank] = (model.rotation.cpu().numpy(), model.centroids.detach().cpu().numpy(), index.index.id_offset,
                     index.index.ntotal)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_multi_rank_warmup_gives_every_rank_the_same_rotation_and_centroids():
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_warmup_worker, args=(2, port, ret), nprocs=2, join=True)
    assert np.array_equal(ret[0][0], ret[1][0]) and np.array_equal(ret[0][1], ret[1][1])
    assert (ret[0][2], ret[1][2]) == (0, 3000) and ret[0][3] == ret[1][3] == 3000


def test_kmeans_statistics_are_deterministic_and_match_the_oracle():
    """rc_kmeans_stats: fixed-order two-stage reduction — identical bits run to run, sums equal to the fp64 oracle to
    rounding, counts exact, accumulation into non-zero buffers, every supported sub-vector width and a ragged size."""
    from repconc_amd import ops
    for M, n in ((48, 70001), (8, 5000), (64, 9000), (96, 3001)):
        x = synth.gaussian(60 + M, (n, 768))
        codes = synth.uniform_codes(61 + M, n, M)
        xt, ct = _t(x), _t(codes)
        s1, c1 = ops.kmeans_stats(xt, ct)
        s2, c2 = ops.kmeans_stats(xt, ct)
        assert torch.equal(s1, s2) and torch.equal(c1, c2)
        ws, wc = pq_oracle.kmeans_stats(x, codes, M)
        assert np.array_equal(c1.cpu().numpy(), wc)
        np.testing.assert_allclose(s1.cpu().numpy(), ws, rtol=1e-12, atol=1e-9)
        s3, c3 = ops.kmeans_stats(xt, ct, s1.clone(), c1.clone())                    # accumulates
        np.testing.assert_allclose(s3.cpu().numpy(), 2 * ws, rtol=1e-12, atol=1e-9)
        assert np.array_equal(c3.cpu().numpy(), 2 * wc)
    # the default path sums exact fixed-point parts with integer atomics; the fixed-order strip kernels (forced here) must
    # agree to fp64 rounding, and take over by themselves when the input is not finite
    M, n = 48, 20000
    x = synth.gaussian(70, (n, 768)) * np.float32(37.5)
    x[:, 5] *= np.float32(1e-6)                                                    # a column 2^-20 below the rest
    codes = synth.uniform_codes(71, n, M)
    s_fx, c_fx = ops.kmeans_stats(_t(x), _t(codes))
    os.environ["RC_KMEANS_STRIPS"] = "1"
    try:
        s_st, c_st = ops.kmeans_stats(_t(x), _t(codes))
    finally:
        del os.environ["RC_KMEANS_STRIPS"]
    ws, wc = pq_oracle.kmeans_stats(x, codes, M)
    assert torch.equal(c_fx, c_st) and np.array_equal(c_fx.cpu().numpy(), wc)
    np.testing.assert_allclose(s_fx.cpu().numpy(), ws, rtol=1e-13, atol=1e-10)
    np.testing.assert_allclose(s_st.cpu().numpy(), ws, rtol=1e-12, atol=1e-9)
    xbad = x.copy()
    xbad[123, 40] = np.inf
    xbad[456, 700] = np.nan
    s_bad, c_bad = ops.kmeans_stats(_t(xbad), _t(codes))
    sb = s_bad.cpu().numpy()
    assert np.isinf(sb[40 // 16, codes[123, 40 // 16], 40 % 16]) and np.isnan(sb[700 // 16, codes[456, 700 // 16], 700 % 16])
    good = np.isfinite(sb)
    assert good.sum() == sb.size - 2 and np.array_equal(c_bad.cpu().numpy(), wc)
    np.testing.assert_allclose(sb[good], ws[good], rtol=1e-12, atol=1e-9)


def _fsum_stats(x, codes, M):
    """sums[m, k, j] = the correctly rounded real sum (math.fsum) of the rows with code k — what exact integer parts give."""
    import math
    n, D = x.shape
    dsub = D // M
    out = np.zeros((M, 256, dsub))
    xd = x.astype(np.float64).reshape(n, M, dsub)
    for m in range(M):
        order = np.argsort(codes[:, m], kind="stable")
        ks, starts = np.unique(codes[order, m], return_index=True)
        ends = list(starts[1:]) + [n]
        for k, a, b in zip(ks, starts, ends):
            blk = xd[order[a:b], m, :]
            for j in range(dsub):
                out[m, k, j] = math.fsum(blk[:, j])
    return out


def test_kmeans_statistics_are_the_correctly_rounded_sums_whatever_was_called_before():
    """Round 4: the fixed-point statistics take their scale from a per-handle hint (no max|x| pass) and repeat the work on
    the device when the hint was too small or a part had to be rounded.  While all parts are exact the 128-bit integer total
    is rounded once, so the answer is math.fsum of the members bit for bit — independent of the hint (call history), of the
    strip geometry and of the data's magnitude."""
    from repconc_amd import ops
    M, n = 48, 4000
    x = synth.gaussian(170, (n, 768))
    x[:, 7] *= np.float32(3e-5)                                       # small values: their low parts are not empty
    codes = synth.uniform_codes(171, n, M)
    want = _fsum_stats(x, codes, M)
    xt, ct = _t(x), _t(codes)
    s_a, c_a = ops.kmeans_stats(xt, ct)
    assert np.array_equal(s_a.cpu().numpy(), want), "not the correctly rounded sums"
    big = _t(x * np.float32(4096.0))                                  # exceeds any hint left by the call above: second pass
    s_b, _ = ops.kmeans_stats(big, ct)
    assert np.array_equal(s_b.cpu().numpy(), want * 4096.0)
    s_c, c_c = ops.kmeans_stats(xt, ct)                               # now the hint is 2^12 too loose: still exact
    assert torch.equal(s_c, s_a) and torch.equal(c_c, c_a)
    tiny = _t(x * np.float32(2.0 ** -40))
    s_d, _ = ops.kmeans_stats(tiny, ct)                               # far below the hint: low parts rounded -> second pass
    assert np.array_equal(s_d.cpu().numpy(), want * 2.0 ** -40)
    s_e, _ = ops.kmeans_stats(xt, ct)
    assert torch.equal(s_e, s_a)
    # a range of 2^70 inside one call: the tight-bound pass rounds what cannot be held; result within fp64 rounding
    wide = x.copy()
    wide[::2, 3] *= np.float32(2.0 ** 40)
    wide[1::2, 3] *= np.float32(2.0 ** -30)
    s_w, _ = ops.kmeans_stats(_t(wide), ct)
    np.testing.assert_allclose(s_w.cpu().numpy(), _fsum_stats(wide, codes, M), rtol=1e-15, atol=1e-10)   # n 2^-45
    # other widths / several column passes / ragged sizes
    for M2, n2 in ((8, 1500), (64, 777), (24, 2049), (96, 300)):
        x2 = synth.gaussian(180 + M2, (n2, 768)) * np.float32(0.37)
        c2 = synth.uniform_codes(181 + M2, n2, M2)
        s2, cnt2 = ops.kmeans_stats(_t(x2), _t(c2))
        assert np.array_equal(s2.cpu().numpy(), _fsum_stats(x2, c2, M2)), M2
        assert np.array_equal(cnt2.cpu().numpy(), pq_oracle.kmeans_stats(x2, c2, M2)[1])


def test_ivf_coarse_update_is_the_fp64_mean_in_row_order_and_deterministic():
    """rc_ivf_coarse_update (Lloyd step of the coarse quantiser, BASELINE configs[3]): every cell's centroid is the fp32 of
    the fp64 mean of its rows added in a fixed order (four interleaved lanes, each ascending; a numpy restatement of exactly that), the same bits on every
    call; empty cells take a row of x; out-of-range assignments are ignored; coarse_kmeans runs on it end to end."""
    import ctypes as C
    from repconc_amd import _lib, ops
    from repconc_amd.ivf import coarse_assign, coarse_kmeans
    n, D, nlist = 30011, 768, 700
    x = synth.gaussian(808, (n, D))
    rng = np.random.default_rng(809)
    assign = rng.integers(0, nlist - 40, n).astype(np.int32)              # the last 40 cells stay empty
    assign[rng.integers(0, n, 50)] = -1                                    # ignored rows
    assign[:5000] = 3                                                      # one big cell (several tiles)
    xt, at = _t(x), _t(assign)
    lib, h, s, _ = ops._ctx(xt)
    wsb = lib.rc_ivf_coarse_update_ws_bytes(n, nlist)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=DEV)
    outs = []
    for _ in range(2):
        cent = torch.zeros((nlist, D), dtype=torch.float32, device=DEV)
        cnt = torch.zeros((nlist,), dtype=torch.int32, device=DEV)
        _lib.check(lib.rc_ivf_coarse_update(h, C.c_void_p(xt.data_ptr()), xt.stride(0), C.c_void_p(at.data_ptr()), n, D, nlist,
                                            C.c_void_p(cent.data_ptr()), C.c_void_p(cnt.data_ptr()), 1234, 7,
                                            C.c_void_p(ws.data_ptr()), wsb, s), "rc_ivf_coarse_update", h)
        outs.append((cent.cpu().numpy(), cnt.cpu().numpy()))
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    got, gcnt = outs[0]
    want_cnt = np.bincount(assign[assign >= 0], minlength=nlist)
    assert np.array_equal(gcnt, want_cnt)
    x64 = x.astype(np.float64)
    for c in list(range(0, 12)) + [3, nlist - 41]:
        rows = np.nonzero(assign == c)[0]                                  # ascending
        parts = []
        for lane in range(4):                                              # four interleaved row lanes, each ascending
            acc = np.zeros(D)
            for r in rows[lane::4]:
                acc = acc + x64[r]
            parts.append(acc)
        acc = ((parts[0] + parts[1]) + parts[2]) + parts[3]
        assert np.array_equal(got[c].view(np.uint32), (acc / len(rows)).astype(np.float32).view(np.uint32)), c
    xs = {row.tobytes() for row in x}
    for c in range(nlist - 40, nlist):                                     # empty cells: some row of x, not zeros
        assert got[c].tobytes() in xs
    # end to end: the k-means built on it improves the quantisation error and is reproducible
    x2 = synth.clustered_embeddings(810, 20000)
    c_a = coarse_kmeans(_t(x2), 256, iters=5)
    c_b = coarse_kmeans(_t(x2), 256, iters=5)
    assert torch.equal(c_a, c_b)
    def err(cc):
        a = coarse_assign(_t(x2), cc)
        return float(((_t(x2) - cc[a]) ** 2).sum(1).mean())
    assert err(c_a) < 0.8 * err(coarse_kmeans(_t(x2), 256, iters=0))


@pytest.mark.parametrize("M", [16, 32, 48, 64])
def test_ivf_list_centric_search_equals_per_query_scan(M):
    """rc_ivf_search_lists (cells scanned once per group of up to 8 probing queries, 8-bit screen + exact rescoring) against
    rc_ivf_search (per-query exact scan) on skewed cells (a few very large ones: thresholds from the sample; many tiny or
    empty ones: every row a candidate), for every M the screen supports in one table phase."""
    from repconc_amd.ivf import IVFPQIndex
    N, nlist, nq = 400000, 300, 37
    rng = np.random.default_rng(700 + M)
    codes = synth.uniform_codes(701 + M, N, M)
    cells = np.minimum((rng.pareto(1.2, N) * 3).astype(np.int64), nlist - 1)      # cell 0..: heavy head, empty tail cells
    C = synth.gaussian(702 + M, (M, 256, 768 // M))
    ivf = IVFPQIndex(768, M, nlist, device=DEV)
    ivf.set_centroids(C)
    ivf.coarse = _t(synth.gaussian(703 + M, (nlist, 768)))
    ivf.set_lists(_t(codes), _t(cells))
    q = _t(synth.gaussian(704 + M, (nq, 768)))
    for nprobe, k in ((3, 10), (40, 1000), (nlist, 200)):
        s1, i1 = ivf.search(q, k, nprobe, method="lists")                 # plan made on the device (rc_ivf_search_probes)
        s2, i2 = ivf.search(q, k, nprobe, method="scan")
        s3, i3 = ivf.search(q, k, nprobe, method="lists_host_plan")       # plan spelled out in torch (rc_ivf_search_lists)
        assert torch.equal(i1, i2) and torch.equal(s1, s2), (M, nprobe)
        assert torch.equal(i3, i2) and torch.equal(s3, s2), (M, nprobe)


@pytest.mark.parametrize("M", [16, 32, 48, 64, 96])
def test_ivf_sixteen_query_screen_equals_the_scan_and_the_oracle(M):
    """Round 6: the 16-QUERY list-centric screen (rc_ivf_search_probes_q16, ivfs_screen16.h: tasks of up to 16 queries, tables
    [code][16 slots][16 queries] in phases of 16 sub-quantisers, ds_read_b128 gathers, its own image) against the per-query
    exact scan and the 8-query screen — ids and score bits — on skewed cells (cells of several rounds, cells smaller than a
    chunk, empty cells), with more queries per cell than a task holds (several tasks per cell, a ragged last one) and fewer
    (empty columns), for every M the screen supports; and against the oracle's brute force on the first queries.  The image
    itself is checked against the library's host-side description (image[at(n, m)] == codes[n][m])."""
    from repconc_amd import ops
    from repconc_amd.ivf import IVFPQIndex
    N, nlist, nq = 150000, 40, 83
    rng = np.random.default_rng(9100 + M)
    codes = synth.uniform_codes(9101 + M, N, M)
    cells = np.minimum((rng.pareto(1.1, N) * 2).astype(np.int64), nlist - 1)      # heavy head (tens of thousands of rows), empty tail
    cells[rng.integers(0, N, 5)] = nlist - 2                                       # a cell of a handful of rows
    C = synth.gaussian(9102 + M, (M, 256, 768 // M))
    coarse = synth.gaussian(9103 + M, (nlist, 768))
    ivf = IVFPQIndex(768, M, nlist, device=DEV)
    ivf.set_centroids(_t(C))
    ivf.coarse = _t(coarse)
    ivf.set_lists(_t(codes), _t(cells))
    img = ivf._wide_image().cpu().numpy()
    lm = ivf.codes.cpu().numpy()
    rows = np.concatenate([np.arange(0, 64), rng.integers(0, N, 200), np.arange(N - 33, N)])
    at = np.array([[ops.adc_image_rows16_at(M, int(n), m) for m in range(M)] for n in rows], dtype=np.int64)
    assert np.array_equal(img[at], lm[rows])
    q = synth.gaussian(9104 + M, (nq, 768))
    for nprobe, k in ((2, 10), (9, 300), (nlist, 1000)):
        s16, i16 = ivf.search(_t(q), k, nprobe, method="lists16")
        s8, i8 = ivf.search(_t(q), k, nprobe, method="lists8")
        sc, ic = ivf.search(_t(q), k, nprobe, method="scan")
        assert torch.equal(i16, ic) and torch.equal(s16, sc), (M, nprobe)
        assert torch.equal(i8, ic) and torch.equal(s8, sc), (M, nprobe)
    ws, wi = pq_oracle.ivf_search(q[:6], C, codes, cells, coarse, 300, 9)
    s16, i16 = ivf.search(_t(q[:6]), 300, 9, method="lists16")               # 6 queries: ten empty columns in every task
    assert np.array_equal(i16.cpu().numpy(), wi) and np.array_equal(s16.cpu().numpy().view(np.uint32), ws.view(np.uint32))
    # "lists" picks the width by the call's queries per probed cell
    assert 8 < ivf.WIDE_MIN_SHARE < 18
    s_auto, i_auto = ivf.search(_t(q), 300, 9, method="lists")               # 83 x 9 / 40 = 18.7 queries per cell: the 16-query screen
    assert torch.equal(i_auto, ivf.search(_t(q), 300, 9, method="scan")[1])


def test_batch_search_hands_a_list_centric_index_every_query_at_once():
    """evaluate_repconc.py:188-206 with an IVF index (round 5): `batch_search` passes the WHOLE query set in one call
    (`IVFPQIndex.whole_query_set`; `index.nprobe` is Faiss's attribute) — the answer is the per-batch one, and the oracle's."""
    from repconc_amd.ivf import IVFPQIndex
    from repconc_amd.models.repconc.evaluate_repconc import batch_search
    M, N, nlist, nq, k = 48, 300000, 97, 301, 20
    rng = np.random.default_rng(811)
    codes = synth.uniform_codes(812, N, M)
    cells = rng.integers(0, nlist, N)
    C = synth.gaussian(813, (M, 256, 16))
    coarse = synth.gaussian(814, (nlist, 768))
    ivf = IVFPQIndex(768, M, nlist, device=DEV)
    ivf.set_centroids(C)
    ivf.coarse = _t(coarse)
    ivf.set_lists(_t(codes), _t(cells))
    ivf.nprobe = 11
    q = synth.gaussian(815, (nq, 768))
    corpus_ids = np.arange(N, dtype=np.int64) * 3 + 7
    qids = np.arange(nq)
    sc, ids = batch_search(qids, q, corpus_ids, ivf, k, batch_size=64)
    parts = [ivf.search(q[a:a + 64], k) for a in range(0, nq, 64)]
    assert np.array_equal(sc.view(np.uint32), np.concatenate([p_[0] for p_ in parts]).view(np.uint32))
    assert np.array_equal(ids, corpus_ids[np.concatenate([p_[1] for p_ in parts])])
    ws, wi = pq_oracle.ivf_search(q[:16], C, codes, cells, coarse, k, 11)
    assert np.array_equal(ids[:16], corpus_ids[wi]) and np.array_equal(sc[:16].view(np.uint32), ws.view(np.uint32))


def test_search_reports_the_screens_survivor_counts():
    """ops.adc_search(stats=...) (rc_adc_search_ws_counts; bench.py prints them per SURVEY 8d-D): per query, rows that passed the
    8-bit screen >= rows the exact rescoring kept >= k, and far fewer than the index holds."""
    from repconc_amd import ops
    M, N, nq, k = 48, 400000, 33, 100
    C, codes, q = _adc_case(M, N, nq, seed=5151)
    st = {}
    s, i = ops.adc_search(_t(codes), _t(C), _t(q), k, stats=st)
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    assert np.array_equal(i.cpu().numpy(), wi)
    surv, cand = st["survivors"].cpu().numpy(), st["candidates"].cpu().numpy()
    assert surv.shape == (nq,) and (surv >= cand).all() and (cand >= k).all() and surv.max() < N // 20


@pytest.mark.parametrize("graph", ["0", "1"])
def test_opq_training_repeats_with_checked_procrustes_when_a_deferred_check_fails(monkeypatch, graph):
    """train_opq reads the orthogonality checks of all rounds once, after the last; if one is not below 1e-9 (here: forced in
    round 1) it repeats the training with per-round checks (library SVD as their fall-back) — whether the rounds >= 1 are
    replayed from the round's hipGraph (RC_WARMUP_GRAPH=1, the default) or enqueued eagerly.  Rank-deficient training rows
    alone do not need that: rounding noise keeps the Procrustes matrices inside the schedule's range and the result is
    orthogonal."""
    from repconc_amd.train import run_warmup
    monkeypatch.setenv("RC_WARMUP_GRAPH", graph)
    rng = np.random.default_rng(8)
    basis = rng.standard_normal((3, 128)).astype(np.float32)
    x = (rng.standard_normal((4096, 3)).astype(np.float32) @ basis)                  # rows in a 3-dimensional subspace
    eye = torch.eye(128, device=DEV)
    R = run_warmup.train_opq(_t(x), 8, n_outer=3, n_pq_first=3, n_pq=2)
    assert float((R @ R.T - eye).abs().max()) < 1e-4 and bool(torch.isfinite(R).all())
    modes = []
    orig, orig_static = run_warmup.procrustes_rotation, run_warmup._procrustes_static

    def spy(P, lower=1e-12, defer=False):
        modes.append(bool(defer))
        out = orig(P, lower, defer)
        if defer and len(modes) == 2:                          # round 1 of the deferred pass "fails" (eager rounds)
            return out[0], out[1] + 1.0
        return out

    def spy_static(*a):                                        # ... and so does the round the graph is made of
        modes.append("graph")
        cur, err = orig_static(*a)
        return cur, err + 1.0
    monkeypatch.setattr(run_warmup, "procrustes_rotation", spy)
    monkeypatch.setattr(run_warmup, "_procrustes_static", spy_static)
    R2 = run_warmup.train_opq(_t(x), 8, n_outer=3, n_pq_first=3, n_pq=2)
    # (the first call orthogonalises the seeded Gaussian matrix the rotation starts from; the repeat starts from the same R0)
    if graph == "0":
        assert modes == [True] * 4 + [False] * 3, modes
    else:      # round 0 eager and deferred, round 1 = the body run eagerly, then its capture (round 2 replays it), then the repeat
        assert modes == [True, True, "graph", "graph"] + [False] * 3, modes
    assert float((R2 @ R2.T - eye).abs().max()) < 1e-4


def test_opq_rounds_replayed_from_the_round_graph_equal_the_eager_rounds(monkeypatch):
    """Rounds >= 1 of a single-rank OPQ training are one hipGraph each (run_warmup._RoundGraph: rotate, Lloyd block, error,
    x^T x_rec, Procrustes iteration, R update, log row): the rotation and every round's error are bit-identical to the
    eagerly enqueued rounds (RC_WARMUP_GRAPH=0), and a second training on the same process replays a fresh capture."""
    from repconc_amd.train import run_warmup
    x = _t(synth.gaussian(4242, (8192, 128)) * np.linspace(1.0, 0.05, 128, dtype=np.float32))
    out = {}
    for graph in ("0", "1", "1"):
        monkeypatch.setenv("RC_WARMUP_GRAPH", graph)
        hist = []
        R = run_warmup.train_opq(x, 8, n_outer=6, n_pq_first=5, n_pq=2, history=hist)
        assert len(hist) == 6 and all(np.isfinite(hist))
        if graph in out:
            assert torch.equal(out[graph][0], R) and out[graph][1] == hist
        out[graph] = (R, hist)
    assert torch.equal(out["0"][0], out["1"][0])
    assert out["0"][1] == out["1"][1]


def test_warmup_procedure_follows_the_oracle_round_by_round():
    """a-12, procedure level: `train_pq` / `train_opq` on the HIP kernels against oracle/pq_oracle.py's restatement of
    the same published procedure (same training rows, same initial rotation and centroid sample).  One Lloyd round from a
    given table is exact up to the fp32 rounding of a mean; over several rounds a rare assignment flip (the rotated
    inputs come from different GEMMs) may move single centroids, so the trajectory is compared through the
    reconstruction error of every round and the final rotation."""
    from repconc_amd.train.run_warmup import train_opq, train_pq
    D, M, n = 128, 8, 8192
    rng = np.random.default_rng(21)
    cent = rng.standard_normal((32, D), dtype=np.float32) * 2.0
    x = cent[rng.integers(0, 32, n)] + rng.standard_normal((n, D), dtype=np.float32)
    x = np.ascontiguousarray((x @ (rng.standard_normal((D, D), dtype=np.float32) / np.sqrt(D))).astype(np.float32))
    cq = lambda xx, cc: c_oracle.quantize(xx, cc, False)[0]
    # PQ k-means: same sample initialisation (seed rule of train_pq), 6 rounds
    Cw, mse_w = pq_oracle.train_pq(x, M, 6, quantize_fn=cq)
    Cg, mse_g = train_pq(_t(x), M, 6)
    assert abs(mse_g - mse_w) <= 1e-4 * mse_w, (mse_g, mse_w)
    moved = np.abs(Cg.cpu().numpy() - Cw).max(axis=-1) > 1e-5
    assert moved.mean() < 0.01, float(moved.mean())
    # first round alone: bit-identical codes, means equal to the last bit or two
    C1w, _ = pq_oracle.train_pq(x, M, 1, quantize_fn=cq)
    C1g, _ = train_pq(_t(x), M, 1)
    np.testing.assert_allclose(C1g.cpu().numpy(), C1w, rtol=2e-6, atol=1e-7)
    # OPQ: 4 rounds from the same orthogonal start
    R0 = np.linalg.qr(np.random.default_rng(3).standard_normal((D, D)))[0].astype(np.float32)
    Rw, hist_w = pq_oracle.train_opq(x, M, R0, 4, 5, 2, quantize_fn=cq)
    hist_g = []
    Rg = train_opq(_t(x), M, n_outer=4, n_pq_first=5, n_pq=2, R0=_t(R0), history=hist_g)
    assert len(hist_g) == 4
    np.testing.assert_allclose(hist_g, hist_w, rtol=5e-4)
    R1w, _ = pq_oracle.train_opq(x, M, R0, 1, 5, 2, quantize_fn=cq)     # one round: the rotations still coincide
    R1g = train_opq(_t(x), M, n_outer=1, n_pq_first=5, n_pq=2, R0=_t(R0)).cpu().numpy()
    print("one OPQ round: max |R - R_oracle|", float(np.abs(R1g - R1w).max()))
    assert np.abs(R1g - R1w).max() < 1e-2
    assert hist_g[-1] < hist_g[0]
    Rg = Rg.cpu().numpy()
    assert np.abs(Rg @ Rg.T - np.eye(D)).max() < 1e-5
    print("opq parity: mse", hist_g, hist_w, "max |R - R_oracle|", float(np.abs(Rg - Rw).max()),
          "pq moved", float(moved.mean()), "mse", mse_g, mse_w)
    assert np.abs(Rg - Rw).max() < 0.05


def test_end_to_end_code_flips_from_the_rotation_gemm_are_counted():
    """forward() = encoder -> `@ rotation.T` -> quantize (modeling_repconc.py:87-110).  Codes are bit-exact GIVEN the
    rotated embeddings; the rotation itself is a library GEMM whose summation order differs from the reference's
    (torch-CPU here, cuBLAS on the reference's own box), so a sub-vector sitting within rounding distance of a cell
    boundary may take the neighbouring code.  Count those flips instead of assuming them away: nearest codes from the
    device GEMM against nearest codes from an fp64 rotation rounded to fp32, and against the torch-CPU fp32 GEMM."""
    from repconc_amd import ops
    B, M = 16384, 48
    x = synth.clustered_embeddings(31, B)
    R = np.linalg.qr(np.random.default_rng(32).standard_normal((768, 768)))[0].astype(np.float32)
    C = synth.sample_centroids(33, (x @ R.T).astype(np.float32), M)
    xr_dev = (_t(x) @ _t(R).T).contiguous()
    xr_cpu = (torch.from_numpy(x) @ torch.from_numpy(R).T).contiguous()
    xr_f64 = (x.astype(np.float64) @ R.T.astype(np.float64)).astype(np.float32)
    rel = float((xr_dev.cpu() - torch.from_numpy(xr_f64)).abs().max() / np.abs(xr_f64).max())
    assert rel < 1e-5, rel
    c_dev = ops.assign_nearest(xr_dev, _t(C), torch.uint8).cpu().numpy()
    c_cpu = ops.assign_nearest(xr_cpu.to(DEV), _t(C), torch.uint8).cpu().numpy()
    c_f64 = ops.assign_nearest(_t(xr_f64), _t(C), torch.uint8).cpu().numpy()
    assert np.array_equal(c_f64[:2048], pq_oracle.quantize(xr_f64[:2048], C, False).astype(np.uint8))
    flips_f64 = float((c_dev != c_f64).mean())
    flips_cpu = float((c_dev != c_cpu).mean())
    print(f"rotation GEMM: max rel error {rel:.2e}; code flips vs fp64 rotation {flips_f64:.2e}, vs torch-CPU GEMM {flips_cpu:.2e}")
    assert flips_f64 < 2e-4 and flips_cpu < 2e-4


def test_deferred_search_and_the_retry_path():
    """`adc_search(defer=True)` / `PQIndex.search_async` give the results of the immediate call, and a sampled threshold
    that admits too few candidates (forced with a hugely negative slack: rank 1 of the sample) is repaired by the retry
    inside `result()` — the answer is still the oracle's."""
    from repconc_amd import ops
    from repconc_amd.index import PQIndex
    from repconc_amd.models.repconc.evaluate_repconc import batch_search, search
    N, M, nq, k = 400000, 48, 24, 100
    rng = np.random.default_rng(77)
    codes = rng.integers(0, 256, (N, M), dtype=np.uint8)
    C = rng.standard_normal((M, 256, 16), dtype=np.float32)
    q = rng.standard_normal((nq, 768), dtype=np.float32)
    want_s, want_i = c_oracle.adc_search(codes, C, q, k)
    now_s, now_i = ops.adc_search(_t(codes), _t(C), _t(q), k)
    pend = ops.adc_search(_t(codes), _t(C), _t(q), k, defer=True)
    assert isinstance(pend, ops.PendingSearch)
    later_s, later_i = pend.result()
    assert torch.equal(now_i, later_i) and torch.equal(now_s, later_s)
    assert np.array_equal(now_i.cpu().numpy(), want_i)
    assert np.array_equal(now_s.cpu().numpy().view(np.uint32), want_s.view(np.uint32))
    tight = ops.adc_search(_t(codes), _t(C), _t(q), k, sel_slack=-1e6, defer=True)     # threshold = best sample score
    assert int(tight._status.item()) & 1                                               # too few candidates ...
    ts, ti = tight.result()                                                            # ... repaired by the retry
    assert np.array_equal(ti.cpu().numpy(), want_i)
    # no retries allowed: the queries go straight to the exact path (rc_adc_search_exact) - search never raises
    zero = ops.adc_search(_t(codes), _t(C), _t(q), k, sel_slack=-1e6, max_retries=0, defer=True)
    zs, zi = zero.result()
    assert zero.stats["exact_queries"] == nq and zero.stats["retried_queries"] == 0
    assert np.array_equal(zi.cpu().numpy(), want_i) and np.array_equal(zs.cpu().numpy().view(np.uint32), want_s.view(np.uint32))
    # batch_search: all batches enqueued, then read == batch by batch
    idx = PQIndex(768, M)
    idx.set_centroids(_t(C))
    idx.add_codes(_t(codes))
    qid, cid = np.arange(nq), np.arange(N)[::-1].copy()
    s1, i1 = batch_search(qid, q, cid, idx, k, 7)
    parts = [search(qid[a:a + 6], q[a:a + 6], cid, idx, k) for a in range(0, nq, 6)]
    assert np.array_equal(i1, np.concatenate([p[1] for p in parts])) and np.array_equal(i1, cid[want_i])
    assert np.array_equal(s1, np.concatenate([p[0] for p in parts]))


def test_replicated_ivf_index_splits_the_queries_over_the_replicas():
    """BASELINE configs[3] in one process: a full IVF copy per device (here three virtual replicas on cuda:0), the query
    batch split across them, results identical to the single index for numpy and tensor queries."""
    from repconc_amd.ivf import IVFPQIndex
    from repconc_amd.multi_index import ReplicatedIVFPQIndex
    N, M, nlist, nq = 300000, 96, 200, 41
    codes = synth.uniform_codes(801, N, M)
    cells = np.random.default_rng(802).integers(0, nlist, N)
    ivf = IVFPQIndex(768, M, nlist, device=DEV)
    ivf.set_centroids(synth.gaussian(803, (M, 256, 768 // M)))
    ivf.coarse = _t(synth.gaussian(804, (nlist, 768)))
    ivf.set_lists(_t(codes), _t(cells))
    rep = ReplicatedIVFPQIndex(ivf, devices=[0, 0, 0])
    assert rep.ntotal == N and rep.parts[0] is ivf and rep.parts[1] is not ivf
    assert rep.parts[1].codes.data_ptr() != ivf.codes.data_ptr() and torch.equal(rep.parts[2].image, ivf.image)
    q = synth.gaussian(805, (nq, 768))
    for nprobe, k in ((4, 10), (60, 300)):
        s1, i1 = ivf.search(_t(q), k, nprobe)
        s2, i2 = rep.search(_t(q), k, nprobe)
        s3, i3 = rep.search(q, k, nprobe)
        assert torch.equal(i1, i2) and torch.equal(s1, s2)
        assert np.array_equal(i3, i1.cpu().numpy()) and np.array_equal(s3, s1.cpu().numpy())
    s4, i4 = rep.search(q[:2], 5, 8)                      # fewer queries than replicas
    assert np.array_equal(i4, ivf.search(q[:2], 5, 8)[1])


def test_ivf_probes_entry_through_the_raw_c_abi():
    """rc_ivf_search_probes called with plain pointers (no Python wrapper logic): argument checks, an empty-cell / ragged
    probe list, and the result against the oracle's brute-force IVF search."""
    import ctypes as C
    from repconc_amd import _lib, ops
    lib, h = _lib.load(), _lib.handle(0)
    N, M, nlist, nq, k, nprobe = 120000, 32, 64, 19, 50, 5
    rng = np.random.default_rng(901)
    codes = synth.uniform_codes(902, N, M)
    cells = rng.integers(0, nlist - 8, N)                     # the last 8 cells stay empty
    order = np.argsort(cells, kind="stable")
    list_off = np.concatenate([[0], np.cumsum(np.bincount(cells, minlength=nlist))]).astype(np.int64)
    Cq = synth.gaussian(903, (M, 256, 768 // M))
    q = synth.gaussian(904, (nq, 768))
    probes = np.stack([rng.permutation(nlist)[:nprobe] for _ in range(nq)]).astype(np.int32)
    probes[0, :] = np.arange(nlist - nprobe, nlist)           # a query that probes (almost) only empty cells
    d_codes, d_off, d_ids = _t(codes[order]), _t(list_off), _t(order.astype(np.int64))
    image = torch.empty((ops.adc_image_rows_bytes(N, M),), dtype=torch.uint8, device=DEV)
    ops.adc_scan_image_(d_codes, image, layout="rows")
    lut = ops.adc_lut(_t(Cq), _t(q))
    d_probes = _t(probes)
    sizes = np.sort(np.diff(list_off))[::-1][:nprobe]
    ss = 8
    sstride = int((16 * (sizes // (16 * ss)) + np.minimum(sizes % (16 * ss), 16)).sum())
    wsb = lib.rc_ivf_search_probes_ws_bytes(M, nq, nprobe, nlist, sstride)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=DEV)
    scores = torch.empty((nq, k), dtype=torch.float32, device=DEV)
    ids = torch.empty((nq, k), dtype=torch.int64, device=DEV)
    status = torch.zeros((1,), dtype=torch.int32, device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = [h, p(d_codes), p(image), p(d_off), p(d_ids), N, nlist, M, 256, p(lut), nq, p(d_probes), nprobe, sstride, ss, k,
            6.0, 4096, p(scores), p(ids), p(status), p(ws), wsb, s]
    assert lib.rc_ivf_search_probes(*args) == 0
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    # oracle: brute force over the probed rows
    got_s, got_i = scores.cpu().numpy(), ids.cpu().numpy()
    lut_h = pq_oracle.adc_lut(q, Cq)
    for qi in range(nq):
        rows = np.concatenate([order[list_off[c]:list_off[c + 1]] for c in probes[qi]]) if nprobe else np.empty(0, np.int64)
        sc = pq_oracle.adc_scores(lut_h[qi:qi + 1], codes[rows])[0] if len(rows) else np.empty(0, np.float32)
        top = np.lexsort((rows, -sc.astype(np.float64)))[:k]
        want_i = np.full(k, -1, np.int64)
        want_i[:len(top)] = rows[top]
        assert np.array_equal(got_i[qi], want_i), qi
        assert np.array_equal(got_s[qi][:len(top)].view(np.uint32), sc[top].view(np.uint32)), qi
    # argument checks
    bad = list(args); bad[12] = nlist + 1                      # nprobe > nlist
    assert lib.rc_ivf_search_probes(*bad) == _lib.RC_EINVAL
    bad = list(args); bad[22] = wsb - 1                        # workspace too small
    assert lib.rc_ivf_search_probes(*bad) == _lib.RC_EWORKSPACE
    bad = list(args); bad[7] = 24                              # no screen for this M
    assert lib.rc_ivf_search_probes(*bad) == _lib.RC_ESHAPE


def test_probe_selection_kernel_picks_the_ordered_probe_set():
    """rc_ivf_select_probes: same cells as the ranked selection (torch path), in ascending cell order; exact ties at the
    boundary go to the lower cell ids; nprobe = 1, = nlist and a nlist that is no multiple of anything."""
    from repconc_amd.ivf import IVFPQIndex
    for nlist, nq in ((5000, 67), (257, 5), (64, 3)):
        ivf = IVFPQIndex(768, 48, nlist, device=DEV)
        ivf.coarse = _t(synth.gaussian(950 + nlist, (nlist, 768)))
        q = _t(synth.gaussian(951 + nlist, (nq, 768)))
        for nprobe in (1, 7, 32, nlist // 3 + 1, nlist):
            a = ivf.probe(q, nprobe, ordered=True).cpu().numpy()
            b = ivf.probe(q, nprobe, ordered=False).cpu().numpy()
            assert b.dtype == np.int32 and np.all(np.diff(b, axis=1) > 0)
            assert np.array_equal(np.sort(a, axis=1), b), (nlist, nprobe)
    # exact ties: scores take four values only -> the boundary value is shared by many cells
    from repconc_amd import _lib
    import ctypes as C
    lib, h = _lib.load(), _lib.handle(0)
    nlist, nq, nprobe = 1000, 9, 300
    sc = np.random.default_rng(960).integers(0, 4, (nq, nlist)).astype(np.float32)
    d_sc, out = _t(sc), torch.empty((nq, nprobe), dtype=torch.int32, device=DEV)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.rc_ivf_select_probes(h, C.c_void_p(d_sc.data_ptr()), nq, nlist, nprobe, C.c_void_p(out.data_ptr()), s) == 0
    want = np.sort(np.lexsort((np.broadcast_to(np.arange(nlist), sc.shape), -sc), axis=1)[:, :nprobe], axis=1)
    assert np.array_equal(out.cpu().numpy(), want)
    assert lib.rc_ivf_select_probes(h, C.c_void_p(d_sc.data_ptr()), nq, nlist, nlist + 1, C.c_void_p(out.data_ptr()), s) == _lib.RC_EINVAL


def test_deferred_search_on_empty_inputs():
    """defer=True with nothing to do: no queries, or an empty index (scores -inf, ids -1) — the pending object resolves
    without touching the device; PQIndex.search_async likewise."""
    from repconc_amd import ops
    from repconc_amd.index import PQIndex
    C = _t(synth.gaussian(3, (48, 256, 16)))
    codes = _t(synth.uniform_codes(4, 1000, 48))
    p0 = ops.adc_search(codes, C, torch.empty((0, 768), device=DEV), 5, defer=True)
    s0, i0 = p0.result()
    assert s0.shape == (0, 5) and i0.shape == (0, 5)
    p1 = ops.adc_search(codes[:0], C, _t(synth.gaussian(5, (3, 768))), 4, defer=True)
    s1, i1 = p1.result()
    assert torch.isinf(s1).all() and (s1 < 0).all() and (i1 == -1).all()
    idx = PQIndex(768, 48)
    idx.set_centroids(C)
    s2, i2 = idx.search_async(synth.gaussian(6, (2, 768)), 3)()
    assert s2.shape == (2, 3) and (i2 == -1).all()


def test_empty_cluster_rule_is_the_published_faiss_rule_on_both_sides():
    """train/run_warmup.py:113 runs Faiss's Clustering, whose `split_clusters` re-seeds empty clusters with a seeded,
    size-proportional donor draw.  The product's `_reseed_empty` and the oracle's restatement make the same splits
    (several empties per sub-quantiser, a donor that is split twice, sub-quantisers without empties untouched)."""
    from repconc_amd.train.run_warmup import _reseed_empty
    rng = np.random.default_rng(17)
    M, K, dsub = 6, 256, 16
    C = rng.standard_normal((M, K, dsub)).astype(np.float32)
    cnt = rng.integers(1, 40, (M, K)).astype(np.int64)
    cnt[0, [3, 77, 200]] = 0
    cnt[2, 5] = 0
    cnt[2, 6] = 4000                       # dominant cluster: the likely donor
    cnt[4, :] = 1
    cnt[4, 9] = 0                          # nothing to split by size: falls back to the biggest cluster
    cnt[5, [0, 1, 2, 3, 4, 5, 6, 7]] = 0
    want = C.copy()
    n_want = pq_oracle.reseed_empty(want, cnt)
    Ct, ct = _t(C).clone(), _t(cnt)
    _reseed_empty(Ct, ct)                                   # one kernel, MT19937 on the device
    assert n_want == 3 + 1 + 1 + 8
    from repconc_amd import ops
    ns = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.kmeans_split_empty_(_t(C).clone(), ct, ns)
    assert int(ns.item()) == n_want
    assert np.array_equal(Ct.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(want[[1, 3]], C[[1, 3]])

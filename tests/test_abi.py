"""CPU-side checks of the C-ABI library: it loads, and exports exactly what include/*.h declares."""
import os
import re

from conftest import ROOT


def _declared():
    hdr = open(os.path.join(ROOT, "include", "repconc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", hdr)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from repconc_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/repconc_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototypes out of sync with the header"
    assert lib.rc_version() >= 100
    assert lib.rc_error_string(-2).decode().startswith("unsupported shape")


def test_size_helpers_need_no_gpu():
    from repconc_amd import _lib
    lib = _lib.load()
    B, M = 49152, 48
    d_bytes = M * B * 256 * 4
    ws = lib.rc_pq_assign_sinkhorn_ws_bytes(B, M, 256)
    assert d_bytes < ws < d_bytes * 1.05          # the distance table dominates the workspace
    assert lib.rc_pq_assign_sinkhorn_ws_bytes(B, M, 128) == 0   # K must be 256
    assert lib.rc_adc_search_ws_bytes(8841823, 48, 256, 1200, 1000) > 1200 * 48 * 256 * 4
    assert lib.rc_sk_ws_bytes(6144, 48, 256) % 256 == 0
    # IVF list-centric search: the device-planned entry needs the explicit-plan workspace plus the plan itself
    w_lists = lib.rc_ivf_search_lists_ws_bytes(96, 1200, 8192)
    w_probes = lib.rc_ivf_search_probes_ws_bytes(96, 1200, 32, 5000, 8192)
    pairs = 1200 * 32
    assert w_lists > 0 and w_probes >= w_lists + 4 * (2 * pairs + 4 * 5000 + 3 * 1200)
    assert lib.rc_ivf_search_probes_ws_bytes(24, 1200, 32, 5000, 8192) == 0      # no conflict-free screen for M = 24
    assert lib.rc_ivf_search_probes_ws_bytes(96, 1200, 0, 5000, 8192) == 0


def test_cpu_tensors_are_rejected_loudly():
    import pytest
    import torch
    from repconc_amd import _lib, ops
    with pytest.raises(_lib.RepconcHipError):
        ops.assign_nearest(torch.zeros(4, 768), torch.zeros(48, 256, 16))


def test_adc_conflict_free_layout_properties():
    """Host-side check of the layout the conflict-free ADC screen relies on (csrc/adc_search.hip, rc_adc_cf_describe):
    (a) the four lanes of a row visit every sub-quantiser of a table phase exactly once, (b) in every gather step the 32
    lanes the LDS services together (0-31 and 32-63) read 32 different slots modulo 32 — distinct bank pairs whatever
    the codes — and (c) a slot always belongs to one sub-quantiser (copies of a 16-block are 16 slots apart)."""
    import ctypes as C
    from repconc_amd import _lib
    lib = _lib.load()
    for M in (16, 32, 48, 64, 96):
        slot, m, spc, nph = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        steps = lib.rc_adc_cf_describe(M, 0, 0, C.byref(slot), C.byref(m), C.byref(spc), C.byref(nph))
        PM = M // nph.value
        assert steps == PM // 4 and spc.value % 32 == 0 and spc.value * 8 * 256 <= 160 * 1024
        owner = {}
        for row in range(16):
            seen = []
            for g in range(4):
                for s in range(steps):
                    assert lib.rc_adc_cf_describe(M, row + 16 * g, s, C.byref(slot), C.byref(m), None, None) == steps
                    assert 0 <= slot.value < spc.value and 0 <= m.value < PM
                    assert owner.setdefault(slot.value, m.value) == m.value                     # (c)
                    seen.append(m.value)
            assert sorted(seen) == list(range(PM)), (M, row)                                    # (a)
        for s in range(steps):
            for half in (0, 1):
                banks = set()
                for lane in range(32 * half, 32 * half + 32):
                    lib.rc_adc_cf_describe(M, lane, s, C.byref(slot), C.byref(m), None, None)
                    banks.add(slot.value % 32)
                assert len(banks) == 32, (M, s, half)                                           # (b)
    assert lib.rc_adc_cf_describe(24, 0, 0, C.byref(slot), C.byref(m), None, None) == -2     # RC_ESHAPE: no image for M=24
    # image sizes: N x M bytes, in whole 32768-row tiles for the tile-blocked layouts (M = 96; every M of the 16-query screen)
    assert lib.rc_adc_scan_image_bytes(1000, 24) == 0
    # the IVF search's image: whole chunks of 16 rows; the host-side description is a bijection onto the chunks' bytes
    assert lib.rc_adc_scan_image_rows_bytes(1000, 24) == 0 and lib.rc_adc_scan_image_rows_bytes(1001, 96) == 1008 * 96
    for MM in (16, 32, 48, 64, 96):
        at = sorted(lib.rc_adc_scan_image_rows_at(MM, n, m) for n in range(16, 48) for m in range(MM))
        assert at == list(range(16 * MM, 48 * MM)) and lib.rc_adc_scan_image_rows_at(MM, 0, MM) == -1
        # ... and of the 16-query IVF screen's image (round 6): a bijection onto the chunk's bytes, phase p16 = m / 16 in its own
        # 256-byte block, the four bytes of a lane = the sub-quantisers its four gather steps read (rc_adc_q16_describe)
        at16 = sorted(lib.rc_adc_scan_image_rows16_at(MM, n, m) for n in range(16, 48) for m in range(MM))
        assert at16 == list(range(16 * MM, 48 * MM)) and lib.rc_adc_scan_image_rows16_at(MM, 0, MM) == -1
        slot = C.c_int(0)
        for n in (16, 21, 31):
            for m in range(MM):
                off = lib.rc_adc_scan_image_rows16_at(MM, n, m) - 16 * MM
                p16, lane, j = off // 256, (off % 256) // 4, off % 4
                assert p16 == m // 16 and lane % 16 == n % 16
                lib.rc_adc_q16_describe(48, lane, j, C.byref(slot))
                assert slot.value == m % 16
    assert lib.rc_adc_scan_image_bytes(1000, 96) == 32768 * 96 and lib.rc_adc_scan_image_bytes(40000, 96) == 2 * 32768 * 96
    for MM in (16, 32, 48, 64):
        q16 = lib.rc_adc_q16_describe(MM, 0, 0, C.byref(slot))
        assert q16 in (0, 1) and lib.rc_adc_scan_image_bytes(1000, MM) == (32768 * MM if q16 else 1000 * MM)


def test_numeric_defaults_quoted_in_the_header_match_the_code():
    """The header is the contract: every `RC_<KNOB> (default <n>` it quotes must be the default the sources pass to
    rc_env_int / read for that knob (round 5's header said 30 s for RC_IPC_TIMEOUT_MS while the code used 600000)."""
    import glob
    hdr = open(os.path.join(ROOT, "include", "repconc_hip.h")).read()
    quoted = re.findall(r"\b(RC_[A-Z0-9_]+)\s*\(default\s+(-?\d+)", hdr)
    assert quoted, "the header quotes no numeric default at all: the check below would be empty"
    code = {}
    for f in glob.glob(os.path.join(ROOT, "repconc_amd", "csrc", "*")):
        for name, dflt in re.findall(r'rc_env_int\("([A-Z0-9_]+)",\s*(-?\d+)\)', open(f).read()):
            code.setdefault(name, set()).add(int(dflt))
    for name, dflt in quoted:
        assert name in code, f"{name}: quoted in the header, not read by any source through rc_env_int"
        assert code[name] == {int(dflt)}, f"{name}: header says {dflt}, the sources use {sorted(code[name])}"


def test_search_threshold_defaults_and_the_retry_warm_up_run_without_a_gpu():
    """Round 6: the head-room of the sampled candidate thresholds (flat 3, IVF 4 standard deviations of the sample rank — the
    values the header and INTEGRATION.md quote) and the warm-up of the retry path's framework operators, which must be a no-op
    the second time and run on any device type."""
    import torch
    from repconc_amd import ops
    from repconc_amd.ivf import IVFPQIndex
    assert ops.ADC_SEL_SLACK == 3.0 and IVFPQIndex.SEL_SLACK == 4.0 and IVFPQIndex.SAMPLE_ROWS == 1536
    hdr = open(os.path.join(ROOT, "include", "repconc_hip.h")).read()
    assert "the Python wrapper passes 3 (repconc_amd.ops.ADC_SEL_SLACK" in hdr and "the Python wrapper passes 4: IVFPQIndex.SEL_SLACK" in hdr
    ops._retry_ops_warm.discard(("cpu", None))
    ops._warm_retry_ops(torch.device("cpu"))
    assert ("cpu", None) in ops._retry_ops_warm
    ops._warm_retry_ops(torch.device("cpu"))

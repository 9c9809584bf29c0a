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


def test_cpu_tensors_are_rejected_loudly():
    import pytest
    import torch
    from repconc_amd import _lib, ops
    with pytest.raises(_lib.RepconcHipError):
        ops.assign_nearest(torch.zeros(4, 768), torch.zeros(48, 256, 16))

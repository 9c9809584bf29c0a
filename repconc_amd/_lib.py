"""ctypes binding of librepconc_hip.so (include/repconc_hip.h).

The library is the product: if it is missing or a symbol is absent this module raises — there is
no CPU or PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("REPCONC_HIP_LIB") or os.path.join(_HERE, "lib", "librepconc_hip.so")   # env: A/B builds

RC_OK, RC_EINVAL, RC_ESHAPE, RC_EHIP, RC_EWORKSPACE, RC_ECOMM, RC_ESELECT = 0, -1, -2, -3, -4, -5, -6
RC_CODE_U8, RC_CODE_I64 = 0, 1
RC_FLAG_NONFINITE = 1
RC_FLAG_RANGE = 2
RC_FLAG_COMM = 4
RC_IPC_BLOB_BYTES = 128
PROF_SK_PASS, PROF_ADC_SCAN, PROF_ASSIGN_NEAREST, PROF_DIST_TABLE = 0, 1, 2, 3

_vp, _i, _i64, _sz, _d = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_double

# name -> (restype, argtypes); mirrors include/repconc_hip.h one to one
PROTOTYPES = {
    "rc_version": (_i, []),
    "rc_error_string": (C.c_char_p, [_i]),
    "rc_create": (_i, [C.POINTER(_vp), _i]),
    "rc_destroy": (_i, [_vp]),
    "rc_last_hip_error": (_i, [_vp]),
    "rc_num_cus": (_i, [_vp]),
    "rc_profile_enable": (_i, [_vp, _i]),
    "rc_profile_collect": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_d)]),
    "rc_pq_assign_nearest": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "rc_pq_assign_nearest_fast_ws_bytes": (_sz, [_i64, _i]),
    "rc_pq_assign_nearest_fast": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "rc_pq_assign_nearest_fast_overflow": (_i, [_vp, _vp, _i64, _i, C.POINTER(_i)]),
    "rc_pq_dist_table_ws_bytes": (_sz, [_i64, _i]),
    "rc_pq_dist_table": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "rc_pq_centre": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "rc_sk_ws_bytes": (_sz, [_i64, _i, _i]),
    "rc_sk_sweep": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _i, _d, _i, _vp, _vp, _sz, _vp]),
    "rc_sk_argmax": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _i, _i, _d, _i, _vp, _vp, _vp, _vp]),
    "rc_sk64_rows": (_i, [_vp, _vp, _vp, _i64, _i, _i, _d, _vp, _vp]),
    "rc_sk64_cols": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _d, _vp, _vp, _vp]),
    "rc_pq_assign_sinkhorn_ws_bytes": (_sz, [_i64, _i, _i]),
    "rc_pq_assign_sinkhorn": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _d, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rc_comm_unique_ids": (_i, [_vp]),
    "rc_comm_init": (_i, [_vp, _vp, _i, _i]),
    "rc_comm_destroy": (_i, [_vp]),
    "rc_comm_world": (_i, [_vp]),
    "rc_comm_ipc_export": (_i, [_vp, _i, _i, _vp]),
    "rc_comm_ipc_connect": (_i, [_vp, _vp]),
    "rc_comm_allgather": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "rc_comm_kind": (_i, [_vp]),
    "rc_comm_status": (_i, [_vp]),
    "rc_solve_num_chains": (_i, [_i, _i]),
    "rc_solve_num_chains_on": (_i, [_vp, _i, _i]),
    "rc_pq_assign_sinkhorn_dist_ws_bytes": (_sz, [_i64, _i, _i, _i]),
    "rc_pq_assign_sinkhorn_dist": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _d, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rc_pq_decode": (_i, [_vp, _vp, _i, _vp, _i64, _i, _i, _i, _vp, _vp]),
    "rc_pq_decode_bwd": (_i, [_vp, _vp, _i, _vp, _i64, _i, _i, _i, _vp, _vp]),
    "rc_normalize_centroids": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "rc_code_hist": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _vp]),
    "rc_kmeans_stats": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "rc_kmeans_update": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "rc_kmeans_split_empty": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "rc_adc_search_ws_bytes": (_sz, [_i64, _i, _i, _i, _i]),
    "rc_adc_search": (_i, [_vp, _vp, _i64, _i, _i, _vp, _i, _vp, _i, _i, _i64, _d, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rc_adc_scan_image_bytes": (_sz, [_i64, _i]),
    "rc_adc_cf_describe": (_i, [_i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "rc_adc_scan_image": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _vp]),
    "rc_adc_q16_describe": (_i, [_i, _i, _i, C.POINTER(_i)]),
    "rc_adc_scan_image_rows": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _vp]),
    "rc_adc_scan_image_rows_bytes": (_sz, [_i64, _i]),
    "rc_adc_scan_image_rows_at": (_i64, [_i, _i64, _i]),
    "rc_adc_scan_image_rows16": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _vp]),
    "rc_adc_scan_image_rows16_bytes": (_sz, [_i64, _i]),
    "rc_adc_scan_image_rows16_at": (_i64, [_i, _i64, _i]),
    "rc_adc_search_img_ws_bytes": (_sz, [_i64, _i, _i, _i, _i]),
    "rc_adc_search_ws_counts": (_i, [_i64, _i, _i, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "rc_adc_search_img": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i, _vp, _i, _i, _i64, _d, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rc_adc_search_q": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i, _vp, _i, _i, _i64, _d, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rc_adc_search_exact_ws_bytes": (_sz, [_i64, _i, _i, _i, _i]),
    "rc_adc_search_exact": (_i, [_vp, _vp, _i64, _i, _i, _vp, _i, _vp, _i, _i, _i64, _vp, _vp, _vp, _sz, _vp]),
    "rc_ivf_coarse_assign_ws_bytes": (_sz, [_i]),
    "rc_ivf_coarse_assign": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _vp, _vp, _sz, _vp]),
    "rc_ivf_coarse_update_ws_bytes": (_sz, [_i64, _i]),
    "rc_ivf_coarse_update": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _vp, _vp, C.c_uint64, _i, _vp, _sz, _vp]),
    "rc_ivf_search_lists_ws_bytes": (_sz, [_i, _i, _i64]),
    "rc_ivf_search_lists": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i,
                                 _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rc_ivf_select_probes": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "rc_ivf_search_probes_ws_bytes": (_sz, [_i, _i, _i, _i, _i64]),
    "rc_ivf_search_probes": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _i, _vp, _i, _i64, _i, _i, _d, _i,
                                  _vp, _vp, _vp, _vp, _sz, _vp]),
    "rc_ivf_search_probes_q": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _i, _vp, _i, _i64, _i, _i, _d, _i,
                                    _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rc_ivf_search_probes_q16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _i, _vp, _i, _i64, _i, _i, _d, _i,
                                      _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rc_ivf_search_ws_bytes": (_sz, [_i, _i64]),
    "rc_ivf_search": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rc_adc_lut": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rc_index_create": (_i, [_vp, _i, _i, _i, C.POINTER(_vp)]),
    "rc_index_destroy": (_i, [_vp]),
    "rc_index_set_centroids": (_i, [_vp, _vp, _vp]),
    "rc_index_reserve": (_i, [_vp, _i64, _vp]),
    "rc_index_add_codes": (_i, [_vp, _vp, _i64, _vp]),
    "rc_index_reset": (_i, [_vp]),
    "rc_index_ntotal": (_i64, [_vp]),
    "rc_index_codes": (_vp, [_vp]),
    "rc_index_centroids": (_vp, [_vp]),
    "rc_index_search": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()
_handles = {}


class RepconcHipError(RuntimeError):
    pass


def load():
    """dlopen the library and type every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch bundles its own HIP runtime; it must be the one the process loads first, or two
        # copies of libamdhip64 end up initialised side by side and every HIP call here fails.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RepconcHipError(
                f"{LIB_PATH} not found — build it with `python -m repconc_amd.build` "
                "(hipcc --offload-arch=gfx950).  repconc_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise RepconcHipError(f"librepconc_hip.so does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str, handle=None):
    if rc == RC_OK:
        return
    lib = load()
    msg = lib.rc_error_string(rc).decode()
    if rc == RC_EHIP and handle is not None:
        msg += f" (hipError_t {lib.rc_last_hip_error(handle)})"
    raise RepconcHipError(f"{what}: {msg} [{rc}]")


def handle(device_index: int):
    """One rc_handle per (process, device)."""
    lib = load()
    with _lock:
        h = _handles.get(device_index)
        if h is None:
            out = _vp()
            check(lib.rc_create(C.byref(out), int(device_index)), "rc_create")
            h = _handles[device_index] = out
    return h

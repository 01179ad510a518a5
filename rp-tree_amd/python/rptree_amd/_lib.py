"""ctypes loader for librptree_hip.so (C ABI declared in include/rptree_hip.h).

There is no CPU fallback: if the HIP library is missing, or no gfx950 device is present when
a context is created, the call raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.normpath(os.path.join(_HERE, "..", ".."))          # rp-tree_amd/
# RPTREE_HIP_LIB: another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("RPTREE_HIP_LIB") or os.path.join(PKG_ROOT, "librptree_hip.so")

RPT_F64, RPT_F32, RPT_BF16 = 0, 1, 2
RPT_PROJ_AUTO, RPT_PROJ_EXACT, RPT_PROJ_MFMA = 0, 1, 2
RPT_KNN_KEEP_DUPLICATES, RPT_KNN_DEDUP, RPT_KNN_DEDUP_DISTANCE = 0, 1, 2
RPT_KNN_METRIC_REFERENCE = 1 << 24
RPT_COMM_UID_BYTES = 128

i32, i64, f64 = C.c_int32, C.c_int64, C.c_double
p_i32, p_i64, p_f64 = C.POINTER(i32), C.POINTER(i64), C.POINTER(f64)
vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol of include/rptree_hip.h
SYMBOLS = {
    "rpt_abi_version": (i32, []),
    "rpt_last_error": (C.c_char_p, []),
    "rpt_device_count": (i32, [p_i32]),
    "rpt_ctx_create": (i32, [i32, C.POINTER(vp)]),
    "rpt_ctx_destroy": (i32, [vp]),
    "rpt_ctx_sync": (i32, [vp]),
    "rpt_ctx_trim": (i32, [vp]),
    "rpt_ctx_stream": (i32, [vp, C.POINTER(vp)]),
    "rpt_ctx_set_option": (i32, [vp, C.c_char_p, i64]),
    "rpt_ctx_get_option": (i32, [vp, C.c_char_p, p_i64]),
    "rpt_prof_enable": (i32, [vp, i32]),
    "rpt_prof_reset": (i32, [vp]),
    "rpt_prof_get": (i32, [vp, i32, p_f64, p_i64]),
    "rpt_dataset_dense_host": (i32, [vp, vp, i64, i32, i32, C.POINTER(vp)]),
    "rpt_dataset_dense_dev": (i32, [vp, vp, i64, i32, i32, C.POINTER(vp)]),
    "rpt_dataset_csr_host": (i32, [vp, vp, vp, vp, i64, i32, i32, C.POINTER(vp)]),
    "rpt_dataset_csr_dev": (i32, [vp, vp, vp, vp, i64, i32, i32, i64, C.POINTER(vp)]),
    "rpt_dataset_free": (i32, [vp]),
    "rpt_dataset_info": (i32, [vp, p_i64, p_i32, p_i32, p_i32, p_i64]),
    "rpt_topology": (i32, [i64, i32, i32, vp, i64, p_i64]),
    "rpt_project_host": (i32, [vp, vp, vp, i32, i32, vp]),
    "rpt_project_dev": (i32, [vp, vp, vp, i32, i32, vp]),
    "rpt_forest_build": (i32, [vp, vp, vp, i32, i32, i32, i32, C.POINTER(vp)]),
    "rpt_forest_free": (i32, [vp]),
    "rpt_forest_info": (i32, [vp, p_i64, p_i32, p_i32, p_i32, p_i32]),
    "rpt_forest_get_perm": (i32, [vp, vp]),
    "rpt_forest_get_nodes": (i32, [vp, vp, vp, vp]),
    "rpt_forest_get_proj": (i32, [vp, vp]),
    "rpt_forest_stream_build": (i32, [vp, vp, vp, i32, i32, i32, i64, i32, C.POINTER(vp)]),
    "rpt_forest_get_topology": (i32, [vp, p_i64, vp, vp, vp, p_i64, p_i64]),
    "rpt_forest_import": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, C.POINTER(vp)]),
    "rpt_forest_get_mode": (i32, [vp, p_i32]),
    "rpt_forest_set_mode": (i32, [vp, i32]),
    "rpt_forest_stats": (i32, [vp, p_i64, p_i64]),
    "rpt_split_segments": (i32, [vp, vp, i64, vp, vp, vp, i32, vp]),
    "rpt_candidates": (i32, [vp, vp, vp, vp, vp, i64, p_i64]),
    "rpt_knnh_host": (i32, [vp, vp, vp, vp, i32, vp, vp, vp, i64, p_i64]),
    "rpt_knn_host": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    "rpt_knn_dev": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    "rpt_knn_last_candidates": (i32, [vp, p_i64]),
    "rpt_knn_last_tier": (i32, [vp, p_i32]),
    "rpt_build_last_handed_back": (i32, [vp, p_i64, p_i64]),
    "rpt_knn_last_uncertified": (i32, [vp, p_i64]),
    "rpt_knn_last_retries": (i32, [vp, p_i64]),
    "rpt_knn_merge_dev": (i32, [vp, vp, vp, vp, i32, i64, i32, i32, vp, vp, vp]),
    "rpt_knn_record_layout": (i32, [i64, i32, vp, vp, vp, vp]),
    "rpt_knn_merge_records_dev": (i32, [vp, vp, i64, i32, i64, i32, i32, vp, vp, vp]),
    "rpt_comm_init": (i32, [i32, C.POINTER(vp)]),
    "rpt_comm_unique_id": (i32, [vp]),
    "rpt_comm_init_rank": (i32, [vp, i32, i32, vp, C.POINTER(vp)]),
    "rpt_comm_destroy": (i32, [vp]),
    "rpt_comm_info": (i32, [vp, p_i32, p_i32, p_i32]),
    "rpt_comm_ctx": (i32, [vp, i32, C.POINTER(vp)]),
    "rpt_comm_sync": (i32, [vp]),
    "rpt_forest_build_sharded": (i32, [vp, vp, vp, i32, i32, i32, i32, C.POINTER(vp)]),
    "rpt_sharded_forest_free": (i32, [vp]),
    "rpt_sharded_forest_local": (i32, [vp, i32, C.POINTER(vp), p_i32, p_i32]),
    "rpt_knn_sharded_dev": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    "rpt_knn_sharded": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    "rpt_brute_knn_host": (i32, [vp, vp, vp, i32, vp, vp]),
}


class RPTError(RuntimeError):
    """Non-zero status from the C ABI (the Haskell wrapper would raise next to RPTError,
    Internal.hs:66-72)."""

    def __init__(self, code, msg):
        super().__init__("rptree_hip status %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "librptree_hip.so not found at %s — build it with `make -C %s` "
                "(there is no CPU fallback)" % (LIB_PATH, PKG_ROOT))
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME
        # libamdhip64.so.7 as /opt/rocm's).  If torch is installed, load it FIRST so that this
        # library's DT_NEEDED binds to the runtime torch (and RCCL via torch.distributed) uses;
        # two runtimes in one process cannot both see the GPU.  Without torch (e.g. a Haskell
        # host) the system runtime under /opt/rocm is used.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)            # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.rpt_abi_version() != 1:
            raise ImportError("librptree_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise RPTError(status, lib().rpt_last_error().decode("utf-8", "replace"))

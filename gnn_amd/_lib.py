"""ctypes binding of libmgn_hip.so (the C ABI declared in include/mgn_hip.h).

There is no fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGN_LIB_PATH") or os.path.join(_HERE, "libmgn_hip.so")   # override: kernel experiments
ABI_VERSION = 1
H = 128
MAX_FEAT = 16
MAX_DEC_OUT = 32


class MgnMlp(C.Structure):
    _fields_ = [("w0", C.c_void_p), ("b0", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
                ("ln_w", C.c_void_p), ("ln_b", C.c_void_p)]


class MgnReduceRec(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("out", C.c_void_p), ("vec", C.c_void_p * 3), ("nblocks", C.c_int32),
                ("stride", C.c_int32), ("mcount", C.c_int32), ("ld", C.c_int32), ("nvec", C.c_int32), ("reserved", C.c_int32)]


class MgnWeightSlice(C.Structure):
    _fields_ = [("w", C.c_void_p), ("ld", C.c_int32), ("col0", C.c_int32), ("k", C.c_int32), ("reserved", C.c_int32)]


MAX_PACK_SLICES = 112                   # MGN_MAX_PACK_SLICES
_P, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_MLP = C.POINTER(MgnMlp)

# name -> (restype, argtypes); must list every symbol include/mgn_hip.h declares
SIGNATURES = {
    "mgn_abi_version": (C.c_int, []),
    "mgn_last_error": (C.c_char_p, []),
    "mgn_get_matmul_mode": (C.c_int, []),
    "mgn_set_matmul_mode": (C.c_int, [C.c_int]),
    "mgn_range_check": (C.c_int, [_P, _I32, _P]),
    "mgn_range_poll": (C.c_int, [_P, _P]),
    "mgn_cells_to_edge_index": (C.c_int, [_P, _I64, _P, _P, _I64, _P, _P]),
    "mgn_receiver_csr": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P]),
    "mgn_edge_build_workspace_bytes": (C.c_int64, [_I64]),
    "mgn_cells_to_edge_index_dev": (C.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, _P, _P, _P]),
    "mgn_receiver_csr_workspace_bytes": (C.c_int64, [_I64]),
    "mgn_receiver_csr_dev": (C.c_int, [_P, _P, _I64, _I64, _P, _I64, _P, _P, _P, _P, _P]),
    "mgn_mp_project": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P]),
    "mgn_mp_edge": (C.c_int, [_P, _P, _P, _P, _P, _P, _MLP, _P, _P, _I64, _I64, _P]),
    "mgn_mp_edge_weighted": (C.c_int, [_P, _P, _P, _P, _P, _P, _MLP, _P, _P, _P, _I64, _I64, _P]),
    "mgn_edge_attention": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "mgn_decode_cat": (C.c_int, [_P, _P, _MLP, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "mgn_packed_weight_bytes": (_I64, [_I32]),
    "mgn_pack_weights": (C.c_int, [C.POINTER(MgnWeightSlice), _I32, _P, _P]),
    "mgn_mp_edge_packed": (C.c_int, [_P, _P, _P, _P, _P, _P, _MLP, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "mgn_mp_node_packed": (C.c_int, [_P, _P, _P, _MLP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "mgn_mp_node": (C.c_int, [_P, _P, _P, _MLP, _P, _I64, _P]),
    "mgn_mp_node_project": (C.c_int, [_P, _P, _P, _MLP, _P, _P, _P, _P, _P, _I64, _P]),
    "mgn_node_features": (C.c_int, [_P, _P, _P, _P, _I32, _P, _P, _P, _I64, _P]),
    "mgn_edge_features": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P]),
    "mgn_node_features_sums": (C.c_int, [_P, _P, _P, _P, _I32, _P, _P, _P, _I64, _P, _P]),
    "mgn_edge_features_sums": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "mgn_norm_fold": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _I64, _F, _P, _P]),
    "mgn_feature_sums_scratch_bytes": (_I64, [_I64, _I32]),
    "mgn_feature_sums": (C.c_int, [_P, _I64, _I32, _P, _P, _P]),
    "mgn_feature_sums_static": (C.c_int, [_P, _I64, _I32, _P, _P]),
    "mgn_feature_sums_static_prepare": (C.c_int, []),
    "mgn_norm_update": (C.c_int, [_P, _P, _P, _I32, _I64, _F, _P, _P]),
    "mgn_norm_update_counted": (C.c_int, [_P, _P, _P, _I32, _P, _I64, _F, _P, _P]),
    "mgn_normalize": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P]),
    "mgn_inject_noise": (C.c_int, [_P, _P, _P, _P, _P, _F, _F, _F, _P, _P, _I64, _P]),
    "mgn_encode": (C.c_int, [_P, _I32, _P, _MLP, _P, _I64, _P]),
    "mgn_decode": (C.c_int, [_P, _MLP, _I32, _I32, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "mgn_decode_step": (C.c_int, [_P, _MLP, _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "mgn_gather_rows": (C.c_int, [_P, _P, _P, _I64, _I32, _P]),
    "mgn_scatter_rows": (C.c_int, [_P, _P, _P, _I64, _I32, _P]),
    "mgn_train_scratch_bytes": (_I64, []),
    "mgn_train_partial_bytes": (_I64, [_I32, _I64]),
    "mgn_train_defer_reductions": (C.c_int, [_P, _I32, _P]),
    "mgn_train_reduce_many": (C.c_int, [_P, _I32, _P]),
    "mgn_train_linear_fwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P, _I64, _P]),
    "mgn_train_ln_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P]),
    "mgn_train_ln_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _I64, _P]),
    "mgn_train_linear_bwd": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P, _P, _P, _I32, _P, _P, _I64, _P]),
    "mgn_segment_sum": (C.c_int, [_P, _P, _P, _P, _I64, _P]),
    "mgn_train_pad_features": (C.c_int, [_P, _I32, _P, _P, _I64, _P]),
    "mgn_train_loss_fwd": (C.c_int, [_P, _P, _P, _I32, _I64, _P, _P, _P]),
    "mgn_train_loss_bwd": (C.c_int, [_P, _P, _P, _P, _I32, _I64, _P, _P]),
    "mgn_train_edge_bwd": (C.c_int, [_P] * 19 + [_I64, _I64, _P]),
    "mgn_adam_step": (C.c_int, [_P, _I32, _F, _F, _F, _F, _F, _I64, _P]),
    "mgn_segment_sum_weighted": (C.c_int, [_P, _P, _P, _P, _I64, _P]),
    "mgn_train_attn_edge_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "mgn_train_attn_alpha_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _P]),
    "mgn_train_attn_node_grad": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _I64, _P]),
    "mgn_adam_hyper": (C.c_int, [_F, _F, _F, _F, _F, _I64, _P]),
    "mgn_adam_step_dev": (C.c_int, [_P, _I32, _P, _P]),
    "mgn_halo_pack": (C.c_int, [_P, _P, _P, _I64, _I32, _P]),
    "mgn_halo_unpack_add": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _P]),
    "mgn_halo_stage_state": (C.c_int, [_P, _P, _I64, _P, _I64, _P, _P, _P, _P]),
    "mgn_halo_unpack_state": (C.c_int, [_P, _I64, _P, _P, _P]),
    "mgn_fps": (C.c_int, [_P, _I32, _P, _P, _P, _I32, _I64, _P, _P, _P]),
    "mgn_fps_workspace_floats": (_I64, [_I64]),
    "mgn_fps_single": (C.c_int, [_P, _I32, _I64, _I64, _I64, _P, _P, _P]),
    "mgn_knn_graph": (C.c_int, [_P, _I32, _P, _P, _P, _I32, _I64, _I32, _P, _P, _P]),
}

_lib = None


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python gnn_amd/csrc/build.py` "
                "(hipcc, --offload-arch=gfx950). There is no CPU or PyTorch fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        if l.mgn_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libmgn_hip.so ABI {l.mgn_abi_version()} != binding {ABI_VERSION}")
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().mgn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")

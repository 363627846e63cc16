"""ctypes binding of libhypdiff_hip.so (declared in include/hypdiff.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the
shared object is missing, `load()` raises and every op built on it fails loudly.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# HD_LIB_PATH: another build of the same library (kernel experiments, tools/probes); never a different implementation
LIB_PATH = os.environ.get("HD_LIB_PATH") or os.path.join(_HERE, "libhypdiff_hip.so")


class HdCurv(ctypes.Structure):
    """struct hd_curv of include/hypdiff.h"""
    _fields_ = [("c", c_double), ("sqrt_c", c_double), ("K", c_double), ("sqrt_K", c_double)]


class HdChainLayer(ctypes.Structure):
    """struct hd_chain_layer of include/hypdiff.h"""
    _fields_ = [("wimg", c_void_p), ("bias", c_void_p), ("d_in", c_int32), ("d_out", c_int32), ("epi", c_int32),
                ("add_emb", c_int32)]


class HdMlpChainLayer(ctypes.Structure):
    """struct hd_mlp_chain_layer of include/hypdiff.h"""
    _fields_ = [("W", c_void_p), ("bias", c_void_p), ("out_image", c_void_p), ("out_f32", c_void_p), ("N", c_int64),
                ("K", c_int64), ("ldc", c_int64), ("epi", c_int32), ("z_out", c_void_p), ("aux", c_void_p)]


class HdPackJob(ctypes.Structure):
    """struct hd_pack_job of include/hypdiff.h"""
    _fields_ = [("src", c_void_p), ("image", c_void_p), ("ld", c_int64), ("rows", c_int64), ("cols", c_int64),
                ("transposed", c_int32)]


class HdAdamJob(ctypes.Structure):
    """struct hd_adam_job of include/hypdiff.h"""
    _fields_ = [("param", c_void_p), ("offset", c_int64), ("count", c_int64)]


class HdTnJob(ctypes.Structure):
    """struct hd_tn_job of include/hypdiff.h"""
    _fields_ = [("a_img", c_void_p), ("b_img", c_void_p), ("c", c_void_p), ("colsum_a", c_void_p), ("P", c_int64),
                ("Q", c_int64), ("ldc", c_int64)]


class HdTransposeJob(ctypes.Structure):
    """struct hd_transpose_job of include/hypdiff.h"""
    _fields_ = [("inp", c_void_p), ("out", c_void_p), ("rows", c_int64), ("cols", c_int64), ("ld_in", c_int64),
                ("ld_out", c_int64)]


P = c_void_p
I64 = c_int64
_map1 = [P, P, I64, I64, HdCurv, P]
_map2 = [P, P, P, I64, I64, HdCurv, P]
_at = [P, P, I64, P, P, I64, I64, HdCurv, P]
_act = [P, P, I64, I64, HdCurv, HdCurv, c_int, P]
_agg = [P, P, P, P, P, I64, I64, HdCurv, HdCurv, c_int, c_int, P, I64, P]

# entry points returning an int64 size instead of a status: name -> argtypes
SIZE_QUERIES = {
    "hd_moments_workspace_doubles": [],
    "hd_gemm_tn_workspace_floats": [I64, I64, I64],
    "hd_gemm_splitk_workspace_floats": [I64, I64, I64],
    "hd_gemm_tn_bf16x3_workspace_floats": [I64, I64, I64],
    "hd_gemm_tn_limbs_multi_workspace_floats": [I64, POINTER(HdTnJob), c_int],
    "hd_colsum_workspace_floats": [I64, I64],
    "hd_mse_workspace_doubles": [],
    "hd_hkm_workspace_doubles": [I64, I64, I64],
    "hd_limb_image_bytes": [I64, I64],
    "hd_limb_image_bytes_f16x2": [I64, I64],
    "hd_smallk_image_bytes": [I64],
    "hd_chain_image_bytes": [I64, I64],
    "hd_chain_workspace_bytes": [],
    "hd_mlp_chain_workspace_bytes": [I64, c_int],
}

# name -> argtypes; names ending in "*" exist as _f32 and _f64
SIGNATURES = {
    "hd_pb_proj*": _map1, "hd_pb_expmap0*": _map1, "hd_pb_logmap0*": _map1, "hd_pb_expmap0_proj*": _map1,
    "hd_pb_mobius_add*": _map2, "hd_pb_expmap*": _map2, "hd_pb_logmap*": _map2,
    "hd_pb_logmap_at*": _at, "hd_pb_expmap_at*": _at, "hd_pb_sqdist*": _map2,
    "hd_pb_logmap_at_f32_f64": _at,
    "hd_pb_matvec_epilogue*": [P, P, P, I64, I64, HdCurv, c_int, P],
    "hd_pb_hyp_act*": _act,
    "hd_pb_vae_hnn_tail*": [P, P, P, P, P, I64, I64, HdCurv, c_double, c_int, c_int, P],
    "hd_hb_proj*": _map1, "hd_hb_proj_tan0*": _map1, "hd_hb_expmap0*": _map1, "hd_hb_logmap0*": _map1,
    "hd_hb_expmap0_proj*": _map1, "hd_hb_sqdist*": _map2,
    "hd_hb_linear_epilogue*": [P, P, I64, I64, HdCurv, c_int, P],
    "hd_hb_hyp_act*": _act,
    "hd_rownorm*": [P, P, I64, I64, P],
    "hd_pb_map_bwd*": [c_int, P, P, P, I64, I64, HdCurv, P], "hd_hb_map_bwd*": [c_int, P, P, P, I64, I64, HdCurv, P],
    "hd_pb_map2_bwd*": [c_int, P, P, P, P, P, I64, I64, HdCurv, P],
    "hd_pb_sqdist_bwd*": [P, P, P, P, P, I64, I64, HdCurv, P], "hd_hb_sqdist_bwd*": [P, P, P, P, P, I64, I64, HdCurv, P],
    "hd_pb_linear_tail_bwd*": [P, P, P, P, P, I64, I64, HdCurv, c_int, P],
    "hd_hb_linear_tail_bwd*": [P, P, P, I64, I64, HdCurv, c_int, P],
    "hd_pb_agg_tail*": [P, P, I64, I64, HdCurv, HdCurv, c_int, c_int, P],
    "hd_hb_agg_tail*": [P, P, I64, I64, HdCurv, HdCurv, c_int, c_int, P],
    "hd_pb_agg_tail_bwd*": [P, P, P, I64, I64, HdCurv, HdCurv, c_int, c_int, P],
    "hd_hb_agg_tail_bwd*": [P, P, P, I64, I64, HdCurv, HdCurv, c_int, c_int, P],
    "hd_rownorm_bwd_add*": [P, P, P, P, I64, I64, P],
    "hd_pb_edge_decode*": [P, P, I64, P, I64, I64, HdCurv, c_int, c_double, c_double, P],
    "hd_hb_edge_decode*": [P, P, I64, P, I64, I64, HdCurv, c_int, c_double, c_double, P],
    "hd_pb_edge_decode_bwd*": [P, P, I64, P, P, I64, I64, HdCurv, c_int, c_double, c_double, P],
    "hd_hb_edge_decode_bwd*": [P, P, I64, P, P, I64, I64, HdCurv, c_int, c_double, c_double, P],
    "hd_hb_linear_tail_rows_f32": [P, P, I64, I64, HdCurv, c_int, P],
    "hd_hb_linear_smallk_f32": [P, P, P, I64, I64, I64, HdCurv, c_int, P],
    "hd_smallk_pack_w_f32": [P, I64, I64, I64, P, P],
    "hd_chain_pack_w_f32": [P, I64, I64, I64, P, P],
    "hd_linear_chain_small_f32": [P, I64, I64, POINTER(HdChainLayer), c_int, P, P, I64, P, I64, P, I64, P],
    "hd_chain_failed": [P, P],
    "hd_mlp_chain_limbs_f32": [P, I64, POINTER(HdMlpChainLayer), c_int, P, I64, P],
    "hd_mlp_chain_failed": [P, P],
    "hd_hb_linear_smallk_bf16x3_f32": [P, P, P, I64, I64, I64, HdCurv, c_int, P],
    "hd_csr_from_coo": [P, P, I64, I64, I64, P, P, P, P],
    "hd_adj_count": [P, P, I64, P, P, P],
    "hd_adj_fill": [P, P, P, I64, P, c_int, P, P, P, P],
    "hd_feat_normalize": [P, c_int, I64, I64, P, P],
    "hd_spmm_csr*": [P, P, P, P, P, I64, I64, P, I64, P],
    "hd_spmm_csr_strided*": [P, P, P, P, I64, P, I64, I64, I64, P, I64, I64, P],
    "hd_pb_agg*": _agg, "hd_hb_agg*": _agg,
    "hd_gemm_f32": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, c_int, P],
    "hd_gemm_f64": [P, P, P, P, I64, I64, I64, I64, I64, I64, P],
    "hd_gemm_f32_ex": [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, c_int, P],
    "hd_gemm_f32_bf16x3": [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, c_int, P],
    "hd_gemm_f32_bf16x3_splitk": [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, c_int, P, P],
    "hd_limb_pack_f32": [P, I64, I64, I64, P, P],
    "hd_limb_pack_multi_f32": [POINTER(HdPackJob), c_int, P],
    "hd_limb_unpack_f32": [P, I64, I64, P, I64, P],
    "hd_gemm_limbs_f32": [P, P, P, P, P, P, I64, I64, I64, I64, c_int, P],
    "hd_gemm_limbs_train_f32": [P, P, P, P, P, P, P, I64, I64, I64, I64, c_int, P],
    "hd_mlp_pack_input_bf16x3": [P, c_int, P, c_int, c_double, P, I64, I64, P],
    "hd_row_pack_bf16x3_f32": [P, c_int, HdCurv, P, P, I64, I64, P],
    "hd_limb_pack_f16x2_f32": [P, I64, I64, I64, P, P, P, P],
    "hd_mlp_pack_input_f16x2": [P, c_int, P, c_int, c_double, P, P, P, I64, I64, P],
    "hd_row_pack_f16x2_f32": [P, c_int, HdCurv, P, P, P, P, I64, I64, P],
    "hd_limb_unpack_f16x2_f32": [P, P, I64, I64, P, I64, P],
    "hd_gemm_f16x2_f32": [P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, c_int, c_float, c_float, P],
    "hd_gemm_tn_f32": [P, P, P, I64, I64, I64, I64, I64, I64, P, P],
    "hd_gemm_tn_bf16x3_f32": [P, P, P, P, I64, I64, I64, I64, I64, I64, P, P],
    "hd_gemm_tn_limbs_f32": [P, P, P, P, I64, I64, I64, I64, P, P],
    "hd_gemm_tn_limbs_multi_f32": [I64, POINTER(HdTnJob), c_int, P, P],
    "hd_transpose_f32": [P, P, I64, I64, I64, I64, P],
    "hd_transpose_multi_f32": [POINTER(HdTransposeJob), c_int, P],
    "hd_adam_step_f32": [POINTER(HdAdamJob), c_int, P, P, P, c_double, c_double, c_double, c_double, I64, P, P],
    "hd_colsum_f32": [P, P, I64, I64, I64, P, P],
    "hd_bias_gelu_f32": [P, P, P, P, I64, I64, c_int, P],
    "hd_gelu_bwd_f32": [P, P, P, I64, P],
    "hd_mse_forward_f32": [P, P, P, I64, P, P, P],
    "hd_mse_backward_f32": [P, P, P, I64, P],
    "hd_q_sample": [P, c_int, P, c_int, P, P, P, I64, P, P, I64, P, I64, I64, c_int, c_double, c_double,
                    c_double, P],
    "hd_q_sample_pack_train": [P, P, P, c_int, c_int, P, P, P, P, I64, P, P, I64, P, P, I64, I64, I64, c_double, c_double,
                               c_double, P],
    "hd_p_sample_update": [P, c_int, P, P, P, I64, c_double, c_double, c_double, c_double, I64, I64, P, I64,
                           I64, c_double, c_double, c_double, P],
    "hd_tran_direction": [P, c_int, P, P, I64, I64, I64, P],
    "hd_mlp_pack_input": [P, c_int, P, c_int, c_double, P, I64, I64, I64, P],
    "hd_moments": [P, c_int, I64, P, P, P],
    "hd_moments_noalloc": [P, c_int, I64, P, P],
    "hd_ddpm_noise": [P, P, P, P, I64, P, I64, I64, c_int, c_double, c_double, c_double, P],
    "hd_ddpm_step": [P, P, P, c_double, c_double, c_double, P, I64, P],
    "hd_embedding_add": [P, P, P, I64, I64, I64, P],
    "hd_embedding_grad_f32": [P, P, P, I64, I64, I64, P, P],
    "hd_generate_adj_f32": [P, P, P, I64, I64, P],
    "hd_hkm_to_hyperboloid": [P, P, I64, I64, P],
    "hd_hkm_mean_step": [P, P, P, I64, I64, I64, c_double, P, P],
    "hd_hkm_to_poincare": [P, P, I64, I64, P],
    "hd_hkm_assign": [P, P, P, P, I64, I64, I64, P],
    "hd_standardize": [P, P, c_int, I64, P, c_int, P],
}

_lib = None


def exported_names():
    out = ["hd_version", "hd_last_error"] + list(SIZE_QUERIES)
    for k in SIGNATURES:
        out += [k[:-1] + "_f32", k[:-1] + "_f64"] if k.endswith("*") else [k]
    return out


def load():
    """Load the HIP library once; raise if it is not built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C hypdiff_amd/csrc` "
            "(or __graft_entry__.build()). hypdiff_amd has no CPU/eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.hd_version.restype = c_int
    lib.hd_last_error.argtypes = [c_char_p, c_size_t]
    lib.hd_last_error.restype = c_size_t
    for n, args in SIZE_QUERIES.items():
        fn = getattr(lib, n)
        fn.argtypes = args
        fn.restype = c_int64
    for key, args in SIGNATURES.items():
        names = [key[:-1] + "_f32", key[:-1] + "_f64"] if key.endswith("*") else [key]
        for n in names:
            fn = getattr(lib, n)
            fn.argtypes = args
            fn.restype = c_int
    _lib = lib
    return lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    load().hd_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def call(name, *args):
    """Invoke an entry point; a non-zero status becomes a RuntimeError."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (status {rc}): {last_error()}")

"""ctypes binding of libgrappa_hip.so (the C ABI of include/grappa_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C grappa_amd/csrc`.  There is
no CPU fallback: if the shared object is missing or does not load, importing the product's compute
path raises (tests on a CPU-only box use a test-only backend, see tests/).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRAPPA_HIP_LIB") or os.path.join(_HERE, "libgrappa_hip.so")   # override: kernel A/B builds (tools/)

ABI_VERSION = 11
# grappa_gemm_desc.precision (include/grappa_hip.h GRAPPA_GEMM_*)
GEMM_GROUP4_MAX = 4          # grappa_gemm_f32_group (forward / input-gradient products of the writer heads)
GEMM_GROUP_MAX = 16
GEMM_PRECISIONS = {"f32": 0, "f32_bf16x9": 1, "f32_bf16x6": 2, "bf16x3": 3, "bf16": 4, "f32_f16x3": 5}

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("a_kcontig", C.c_int), ("b_kcontig", C.c_int),
                ("A", C.c_void_p), ("lda", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int), ("C", C.c_void_p), ("ldc", C.c_int),
                ("C2", C.c_void_p), ("ldc2", C.c_int), ("bias", C.c_void_p), ("res", C.c_void_p), ("ldres", C.c_int),
                ("aux", C.c_void_p), ("ldaux", C.c_int), ("pre", C.c_void_p), ("ldpre", C.c_int), ("a_colsum", C.c_void_p), ("act", C.c_int), ("drop_p", C.c_float), ("drop_seed", C.c_uint64),
                ("accumulate", C.c_int), ("precision", C.c_int),
                # plane format (ABI 3)
                ("a_planes", C.c_int), ("b_planes", C.c_int), ("a_plane_stride", C.c_size_t), ("b_plane_stride", C.c_size_t),
                ("Cp", C.c_void_p), ("ldcp", C.c_int), ("cp_plane_stride", C.c_size_t),
                ("resp", C.c_void_p), ("ldresp", C.c_int), ("resp_plane_stride", C.c_size_t),
                ("auxp", C.c_void_p), ("ldauxp", C.c_int), ("auxp_plane_stride", C.c_size_t),
                ("cp_nplanes", C.c_int), ("resp_nplanes", C.c_int), ("auxp_nplanes", C.c_int), ("C1p", C.c_void_p), ("ldc1p", C.c_int),
                ("a_amax", C.c_void_p), ("b_amax", C.c_void_p), ("amax_bcast", C.c_int), ("out_amax", C.c_void_p),
                # ABI 7: residual = LayerNorm(res) recomputed by the epilogue
                ("res_ln_mean", C.c_void_p), ("res_ln_rstd", C.c_void_p), ("res_ln_gamma", C.c_void_p), ("res_ln_beta", C.c_void_p),
                # ABI 8: pair-format operands of the weight-gradient products (token maxima of the operand)
                ("a_rowmax", C.c_void_p), ("b_rowmax", C.c_void_p),
                ("out_amax_parts", C.c_void_p), ("a_amax_nseg", C.c_int),
                # ABI 10: what used to be process-wide setters, per call
                ("plan_cfg", C.c_int), ("plan_nsplit", C.c_int), ("plan_tail", C.c_int), ("splitk_reduce", C.c_int), ("drop_salt", C.c_void_p)]


class LnFwdItem(C.Structure):
    _fields_ = [("M", C.c_int), ("W", C.c_int), ("x", C.c_void_p), ("ldx", C.c_int), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("y", C.c_void_p),
                ("ldy", C.c_int), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("y_amax", C.c_void_p)]


class LnBwdItem(C.Structure):
    _fields_ = [("M", C.c_int), ("W", C.c_int), ("dy", C.c_void_p), ("lddy", C.c_int), ("x", C.c_void_p), ("ldx", C.c_int), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("gamma", C.c_void_p), ("dx", C.c_void_p), ("lddx", C.c_int), ("part", C.c_void_p), ("dx_amax", C.c_void_p)]


class ActDropoutItem(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("dy", C.c_void_p), ("lddy", C.c_int), ("y", C.c_void_p), ("ldy", C.c_int), ("drop_p", C.c_float),
                ("drop_seed", C.c_uint64), ("dz", C.c_void_p), ("lddz", C.c_int), ("dz_amax", C.c_void_p), ("drop_salt", C.c_void_p)]


class SeqAttnItem(C.Structure):
    _fields_ = [("s", C.c_int), ("T", C.c_int), ("nheads", C.c_int), ("dh", C.c_int), ("qkv", C.c_void_p), ("out", C.c_void_p), ("dout", C.c_void_p),
                ("dqkv", C.c_void_p), ("amax", C.c_void_p)]


ROW_BATCH_MAX = 4


class ColsumItem(C.Structure):
    _fields_ = [("part", C.c_void_p), ("nrows", C.c_int), ("n", C.c_int), ("out", C.c_void_p), ("out2", C.c_void_p),
                ("n_first", C.c_int), ("accumulate", C.c_int)]


class MMDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("C", C.c_int), ("B", C.c_int), ("xyz", C.c_void_p), ("T", C.c_int * 4),
                ("idx", C.c_void_p * 4), ("k", C.c_void_p * 4), ("eq", C.c_void_p * 4), ("mol_ptr", C.c_void_p * 4),
                ("n_per", C.c_int * 4), ("offset_torsion", C.c_int), ("inc_ptr", C.c_void_p), ("inc_code", C.c_void_p),
                ("atom_molptr", C.c_void_p)]


class PLossDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("mol_ptr", C.c_void_p * 6), ("p", C.c_void_p * 6), ("ref", C.c_void_p * 6),
                ("width", C.c_int * 6), ("ref_width", C.c_int * 6), ("fac", C.c_float * 6), ("reg", C.c_float * 6),
                ("pw", C.c_void_p), ("inv_B", C.c_float)]


class CollateDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_row", C.c_void_p), ("dst_row", C.c_void_p), ("p0", C.c_void_p),
                ("p1", C.c_void_p), ("c0", C.c_int64), ("width", C.c_int32), ("mode", C.c_int32)]


class WriterLayerDesc(C.Structure):
    """grappa_writer_layer_desc (ABI 11): one fused transformer layer of a writer head"""
    _fields_ = [("s", C.c_int), ("T", C.c_int), ("F", C.c_int), ("nheads", C.c_int), ("dtype", C.c_int), ("x", C.c_void_p), ("out", C.c_void_p),
                ("w_in_pk", C.c_void_p), ("w_o_pk", C.c_void_p), ("w1_pk", C.c_void_p), ("w2_pk", C.c_void_p),
                ("b_in", C.c_void_p), ("b_o", C.c_void_p), ("b1", C.c_void_p), ("b2", C.c_void_p),
                ("n1_gamma", C.c_void_p), ("n1_beta", C.c_void_p), ("nf_gamma", C.c_void_p), ("nf_beta", C.c_void_p),
                ("drop_p", C.c_float), ("seed1", C.c_uint64), ("seed2", C.c_uint64), ("drop_salt", C.c_void_p),
                ("save_mean1", C.c_void_p), ("save_rstd1", C.c_void_p), ("save_meanf", C.c_void_p), ("save_rstdf", C.c_void_p),
                ("save_x1", C.c_void_p), ("save_qkv", C.c_void_p), ("save_att", C.c_void_p), ("save_x2", C.c_void_p), ("save_x3", C.c_void_p),
                ("save_u", C.c_void_p), ("gather_idx", C.c_void_p), ("x1_tab", C.c_void_p), ("qkv_tab", C.c_void_p), ("x2_tiled", C.c_int)]


class WriterLayerBwdDesc(C.Structure):
    """grappa_writer_layer_bwd_desc (ABI 11): the input-gradient chain of one fused transformer layer"""
    _fields_ = [("s", C.c_int), ("T", C.c_int), ("F", C.c_int), ("nheads", C.c_int), ("dtype", C.c_int), ("dout", C.c_void_p),
                ("x", C.c_void_p), ("qkv", C.c_void_p), ("x2", C.c_void_p), ("u", C.c_void_p),
                ("mean1", C.c_void_p), ("rstd1", C.c_void_p), ("meanf", C.c_void_p), ("rstdf", C.c_void_p),
                ("n1_gamma", C.c_void_p), ("nf_gamma", C.c_void_p),
                ("w_in_tpk", C.c_void_p), ("w_o_tpk", C.c_void_p), ("w1_tpk", C.c_void_p), ("w2_tpk", C.c_void_p),
                ("drop_p", C.c_float), ("seed1", C.c_uint64), ("seed2", C.c_uint64), ("drop_salt", C.c_void_p),
                ("dx", C.c_void_p), ("dz2", C.c_void_p), ("dz1", C.c_void_p), ("dzo", C.c_void_p), ("dqkv", C.c_void_p),
                ("ln1_part", C.c_void_p), ("lnf_part", C.c_void_p), ("x2_tiled", C.c_int), ("gather_idx", C.c_void_p)]


WRITER_BF16 = 1

COLLATE_MODES = {"copy": 0, "add": 1, "inv_rows": 2, "inc_code": 3, "conf": 4}

VP4 = C.c_void_p * 4
VP6 = C.c_void_p * 6

# every symbol include/grappa_hip.h declares: name -> (restype, argtypes)
_vp, _i, _f, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64
SIGNATURES = {
    "grappa_abi_version": (_i, []),
    "grappa_build_arch": (C.c_char_p, []),
    "grappa_launch_count": (C.c_longlong, [_i]),
    "grappa_adam_step_dyn_f32": (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _f, _vp, _f]),
    "grappa_split_planes_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _sz, _i]),
    "grappa_split_pairs_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _i]),
    "grappa_split_pairs_f32_batched": (_i, [_vp, _i, _i, _vp]),
    "grappa_amax_f32_workspace_bytes": (_sz, [_i, _i]),
    "grappa_amax_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _sz]),
    "grappa_amax_f32_batched": (_i, [_vp, _i, _vp]),
    "grappa_amax_reduce": (_i, [_vp, _i, C.POINTER(C.c_void_p), C.POINTER(C.c_int), _vp]),
    "grappa_amax_combine": (_i, [_vp, _i, _i, _vp, _vp]),
    "grappa_gemm_f32_workspace_bytes": (_sz, [_i, _i, _i]),
    "grappa_gemm_f32_plan": (_i, [_i, _i, _i, _i, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    "grappa_gemm_f32_plan_desc": (_i, [C.POINTER(GemmDesc), c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    "grappa_gemm_f32_workspace_bytes_desc": (_sz, [C.POINTER(GemmDesc)]),
    "grappa_gemm_f32": (_i, [_vp, C.POINTER(GemmDesc), _vp, _sz]),
    "grappa_gemm_f32_grouped_workspace_bytes": (_sz, [C.POINTER(GemmDesc), _i]),
    "grappa_gemm_f32_grouped": (_i, [_vp, C.POINTER(GemmDesc), _i, _vp, _sz]),
    "grappa_gemm_f32_group_workspace_bytes": (_sz, [C.POINTER(GemmDesc), _i]),
    "grappa_gemm_f32_group": (_i, [_vp, C.POINTER(GemmDesc), _i, _vp, _sz]),
    "grappa_colsum_workspace_bytes": (_sz, [_i, _i]),
    "grappa_colsum_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _sz]),
    "grappa_act_dropout_bwd_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _f, _u64, _vp, _i, _vp]),
    "grappa_act_dropout_bwd_amax_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _f, _u64, _vp, _i, _vp, _vp]),
    "grappa_layernorm_fwd_batched_f32": (_i, [_vp, C.POINTER(LnFwdItem), _i]),
    "grappa_layernorm_bwd_batched_f32": (_i, [_vp, C.POINTER(LnBwdItem), _i]),
    "grappa_act_dropout_bwd_batched_f32": (_i, [_vp, C.POINTER(ActDropoutItem), _i]),
    "grappa_seqattn_fwd_batched_f32": (_i, [_vp, C.POINTER(SeqAttnItem), _i]),
    "grappa_seqattn_bwd_batched_f32": (_i, [_vp, C.POINTER(SeqAttnItem), _i]),
    "grappa_act_dropout_bwd_pairs_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _f, _u64, _vp, _i, _vp, _vp, _i, _vp]),
    "grappa_add_f32": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "grappa_layernorm_fwd_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "grappa_layernorm_fwd_pairs_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i]),
    "grappa_layernorm_fwd_amax_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "grappa_layernorm_bwd_workspace_bytes": (_sz, [_i, _i]),
    "grappa_layernorm_bwd_partial_rows": (_i, [_i]),
    "grappa_colsum_partials_batched": (_i, [_vp, C.POINTER(ColsumItem), _i]),
    "grappa_layernorm_bwd_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _sz]),
    "grappa_layernorm_bwd_amax_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "grappa_layernorm_bwd_drop_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp, _f, _u64, _vp, _i, _vp, _vp]),
    "grappa_gat_fwd_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "grappa_gat_bwd_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "grappa_neighbor_mean_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i]),
    "grappa_neighbor_mean_bf16": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i]),
    "grappa_charge_encoding_f32": (_i, [_vp, _i, _vp, _i, _f, _f, _vp, _i, _i]),
    "grappa_tuple_gather_fwd_f32": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i]),
    "grappa_tuple_gather_bwd_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i]),
    "grappa_seqattn_fwd_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "grappa_seqattn_fwd_pairs_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "grappa_seqattn_bwd_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "grappa_seqattn_fwd_amax_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "grappa_seqattn_bwd_amax_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "grappa_perm_concat_fwd_f32": (_i, [_vp, _i, _i, _i, _i, c_int_p, _vp, _vp]),
    "grappa_perm_concat_bwd_f32": (_i, [_vp, _i, _i, _i, _i, c_int_p, _vp, _vp]),
    "grappa_param_out_fwd_f32": (_i, [_vp, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp]),
    "grappa_param_out_bwd_f32": (_i, [_vp, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp, _vp]),
    "grappa_param_out_stats_workspace_bytes": (_sz, [_i]),
    "grappa_param_out_bwd_stats_f32": (_i, [_vp, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz]),
    "grappa_mm_energy_fwd_f32": (_i, [_vp, C.POINTER(MMDesc), _vp, _vp, C.POINTER(VP4), C.POINTER(VP4)]),
    "grappa_mm_gradient_fwd_f32": (_i, [_vp, C.POINTER(MMDesc), _vp]),
    "grappa_mm_bwd_f32": (_i, [_vp, C.POINTER(MMDesc), _vp, _vp, C.POINTER(VP4), C.POINTER(VP4)]),
    "grappa_loss_ef_fwd_bwd_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp]),
    "grappa_loss_param_fwd_bwd_f32": (_i, [_vp, C.POINTER(PLossDesc), _vp, C.POINTER(VP6)]),
    "grappa_eval_se_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "grappa_collate_batch": (_i, [_vp, _vp, C.POINTER(CollateDesc), _i, _i]),
    "grappa_sumsq_workspace_bytes": (_sz, [_sz]),
    "grappa_sumsq_f32": (_i, [_vp, _sz, _vp, _vp, _i, _vp, _sz]),
    "grappa_adam_step_f32": (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _f, _vp, _f]),
    "grappa_dropout_keep": (_i, [_u64, _u64, _f]),
    # bf16 storage configuration: same argument lists as the *_f32 entry points
    "grappa_layernorm_fwd_bf16": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "grappa_layernorm_bwd_bf16": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _sz]),
    "grappa_act_dropout_bwd_bf16": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _f, _u64, _vp, _i, _vp]),
    "grappa_gat_fwd_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "grappa_gat_bwd_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "grappa_tuple_gather_fwd_bf16": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i]),
    "grappa_tuple_gather_bwd_bf16": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i]),
    "grappa_seqattn_fwd_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "grappa_seqattn_bwd_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "grappa_perm_concat_fwd_bf16": (_i, [_vp, _i, _i, _i, _i, c_int_p, _vp, _vp]),
    "grappa_perm_concat_bwd_bf16": (_i, [_vp, _i, _i, _i, _i, c_int_p, _vp, _vp]),
    "grappa_convert_f32_to_bf16": (_i, [_vp, _i, _i, _vp, _i, _vp, _i]),
    "grappa_convert_bf16_to_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _i]),
    # ABI 11: the fused writer-head layer
    "grappa_writer_head_fwd": (_i, [_vp, C.POINTER(WriterLayerDesc)]),
    "grappa_writer_head_bwd": (_i, [_vp, C.POINTER(WriterLayerBwdDesc)]),
    "grappa_writer_head_tiles": (_i, [_i, _i]),
    "grappa_writer_pack_bytes": (_sz, [_i, _i, _i]),
    "grappa_writer_pack_weight": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp]),
}

_lib = None


def load():
    """dlopen the library and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C grappa_amd/csrc` (there is no CPU fallback for the Grappa HIP path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.grappa_abi_version() != ABI_VERSION:
        raise RuntimeError("libgrappa_hip.so: ABI version mismatch")
    _lib = lib
    return lib

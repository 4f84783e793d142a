"""ctypes binding of liblvae_hip.so (C ABI: include/lvae_hip.h).

The library is the product: if it is missing the package FAILS LOUDLY -- there is no PyTorch/CPU fallback for the
encode/decode hot path.  Build it in-tree with `python lossy-vae_amd/build_native.py` (hipcc, gfx950).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liblvae_hip.so')
ABI_VERSION = 24
_lib = None


class GemmDesc(C.Structure):
    """Mirror of `lvae_gemm_desc` (include/lvae_hip.h)."""
    _fields_ = [
        ('A0', C.c_void_p), ('A1', C.c_void_p),
        ('lda0', C.c_long), ('lda1', C.c_long),
        ('K0', C.c_int), ('K1', C.c_int),
        ('H', C.c_int), ('W', C.c_int),
        ('Wt', C.c_void_p), ('ldw', C.c_long),
        ('bias', C.c_void_p), ('gamma', C.c_void_p),
        ('res', C.c_void_p), ('ldres', C.c_long),
        ('out', C.c_void_p), ('ldo', C.c_long),
        ('M', C.c_int), ('N', C.c_int), ('K', C.c_int),
        ('a_mode', C.c_int), ('epi', C.c_int), ('store', C.c_int), ('r', C.c_int),
        ('a_gelu', C.c_int), ('prec', C.c_int), ('Wt16', C.c_void_p), ('cfg', C.c_int), ('ksplit', C.c_int), ('ws', C.c_void_p), ('a_bf16', C.c_int), ('out_bf16', C.c_int), ('cnt', C.c_void_p),
        ('status', C.c_void_p), ('a_h2', C.c_int), ('out_h2', C.c_int), ('defer_reduce', C.c_int),
    ]


class Op(C.Structure):
    """Mirror of `lvae_op` (include/lvae_hip.h): one entry of a native launch-plan segment (lvae_run_ops)."""
    _fields_ = [('kind', C.c_int), ('side', C.c_int), ('p', C.c_void_p * 8), ('i', C.c_long * 6), ('f', C.c_double * 2)]


class MlpDesc(C.Structure):
    """Mirror of `lvae_mlp_desc` (lvae_mlp_h2f: fc1 -> GELU -> fc2 of a C = 128 / hidden = 192 block as one launch)."""
    _fields_ = [('y', C.c_void_p), ('w1', C.c_void_p), ('b1', C.c_void_p), ('w2', C.c_void_p), ('b2', C.c_void_p), ('gamma', C.c_void_p),
                ('res', C.c_void_p), ('out', C.c_void_p), ('M', C.c_int), ('C', C.c_int), ('hid', C.c_int)]


class MlpSkDesc(C.Structure):
    """Mirror of `lvae_mlp_sk_desc` (lvae_mlp_sk: the MLP of a small-map block whose two GEMMs run split-K, as fused launch + reduce)."""
    _fields_ = [('y', C.c_void_p), ('w1', C.c_void_p), ('b1', C.c_void_p), ('w2', C.c_void_p), ('b2', C.c_void_p), ('gamma', C.c_void_p),
                ('res', C.c_void_p), ('out', C.c_void_p), ('ws', C.c_void_p), ('M', C.c_int), ('C', C.c_int), ('hid', C.c_int),
                ('S1', C.c_int), ('S2', C.c_int)]


class DecBlock(C.Structure):
    """Mirror of `lvae_dec_block`: one latent block of a group's decode (lvae_decode_blocks)."""
    _fields_ = [('ops', C.c_void_p), ('n_ops', C.c_int), ('idx_dev', C.c_void_p), ('idx_host', C.c_void_p), ('sym_host', C.c_void_p),
                ('sym_dev', C.c_void_p), ('per_image', C.c_size_t)]


class EncBlock(C.Structure):
    """Mirror of `lvae_enc_block`: one latent block of a group's encode (lvae_encode_blocks)."""
    _fields_ = [('ops', C.c_void_p), ('n_ops', C.c_int), ('sym_dev', C.c_void_p), ('sym_host', C.c_void_p), ('idx_dev', C.c_void_p),
                ('idx_host', C.c_void_p), ('per_image', C.c_size_t)]


# lvae_op.kind of every entry point a launch plan may hold (enum LVAE_OP_* of the header, in its order)
OP_KINDS = {name: k + 1 for k, name in enumerate([
    'lvae_gemm_f32', 'lvae_dwconv_ln_f32', 'lvae_dwconv_ln_h2', 'lvae_dwconv_ln_bf16', 'lvae_dwconv_ln_q8', 'lvae_stem_f32', 'lvae_stem_bf16',
    'lvae_bias_expand_f32', 'lvae_bias_expand_bf16', 'lvae_prior_index_f32', 'lvae_quantize_f32', 'lvae_dequantize_f32',
    'lvae_gaussian_nll_f32', 'lvae_lossless_params_f32', 'lvae_lossless_output_f32', 'lvae_mlp_h2f', 'lvae_mlp_sk', 'lvae_prior_index_sk_f32', 'lvae_quantize_sk_f32'])}
OP_ORDER = len(OP_KINDS) + 1

TRACE_MAGIC = 1985229328.0       # LVAE_TRACE_MAGIC
A_PLAIN, A_PATCH2, A_CONV3 = 0, 1, 2
STATUS_RANGE, STATUS_NONFINITE_PRIOR, STATUS_NONFINITE_LATENT, STATUS_NONFINITE_IMAGE = 1, 2, 4, 8      # LVAE_STATUS_* (status word)
EPI_BIAS, EPI_BIAS_GELU, EPI_GAMMA_RES, EPI_RES = 0, 1, 2, 3
ST_ROWMAJOR, ST_SHUFFLE, ST_IMAGE = 0, 2, 3

_vp, _i, _l, _f, _d, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_size_t
SIGNATURES = {
    # name: (restype, argtypes)  -- one entry per symbol declared in include/lvae_hip.h
    'lvae_abi_version': (_i, []),
    'lvae_build_info': (C.c_char_p, []),
    'lvae_pmf_to_quantized_cdf': (_i, [_vp, _i, _i, _vp]),
    'lvae_build_gaussian_tables': (_i, [_vp, _i, _d, _i, _vp, _i, _vp, _vp]),
    'lvae_rans_encode_with_indexes': (_l, [_vp, _vp, _sz, _vp, _i, _vp, _vp, _vp, _sz]),
    'lvae_rans_decode_with_indexes': (_i, [_vp, _sz, _vp, _sz, _vp, _i, _vp, _vp, _vp]),
    'lvae_rans_enc_step_selftest': (_i, [C.c_uint64, C.c_uint32, C.c_uint32, _vp, _vp]),
    'lvae_rans_encode_batch': (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i]),
    'lvae_rans_decode_batch': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i]),
    'lvae_gemm_f32': (_i, [C.POINTER(GemmDesc), _vp]),
    'lvae_gemm_num_configs': (_i, []),
    'lvae_mlp_h2f': (_i, [C.POINTER(MlpDesc), _vp]),
    'lvae_mlp_sk': (_i, [C.POINTER(MlpSkDesc), _vp]),
    'lvae_mlp_sk_supported': (_i, [_i, _i, _i, _i]),
    'lvae_gelu_f32': (_i, [_vp, _vp, _l, _vp]),
    'lvae_dwconv_ln_f32': (_i, [_vp] * 8 + [_i] * 5 + [_vp]),
    'lvae_dwconv_ln_h2': (_i, [_vp] * 8 + [_i] * 5 + [_vp]),
    'lvae_dwconv_ln_q8': (_i, [_vp] * 8 + [_i] * 5 + [_vp]),
    'lvae_stem_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp]),
    'lvae_range_flag_f32': (_i, [_vp, _l, _f, _f, _vp, _vp]),
    'lvae_dwconv_ln_bf16': (_i, [_vp] * 8 + [_i] * 5 + [_vp]),
    'lvae_stem_bf16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp]),
    'lvae_bias_expand_bf16': (_i, [_vp, _vp, _l, _i, _vp]),
    'lvae_gemv_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'lvae_prior_index_f32': (_i, [_vp, _vp, _vp, _vp, _i, _f, _i, _i, _i, _vp, _vp]),
    'lvae_prior_index_sk_f32': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _i, _i, _vp, _vp]),
    'lvae_quantize_sk_f32': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'lvae_quantize_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'lvae_dequantize_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'lvae_lossless_params_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _i, _i, _vp, _vp]),
    'lvae_lossless_output_f32': (_i, [_vp, _vp, _vp, C.c_long, _vp, _vp]),
    'lvae_prior_sample_f32': (_i, [_vp, _vp, C.c_long, _i, _i, C.c_float, C.c_ulonglong, C.c_ulonglong, _vp]),
    'lvae_gaussian_nll_f32': (_i, [_vp, _vp, _vp, _f, _i, _i, _i, _i, _vp]),
    'lvae_bias_expand_f32': (_i, [_vp, _vp, _l, _i, _vp]),
    'lvae_event_create': (_vp, []),
    'lvae_event_destroy': (_i, [_vp]),
    'lvae_stream_order': (_i, [_vp, _vp, _vp]),
    'lvae_run_ops': (_i, [_vp, _i, _vp, _vp, _vp]),
    'lvae_decode_blocks': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    'lvae_encode_blocks': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    'lvae_sqerr_sum_f32': (_i, [_vp, _vp, _vp, _i, _l, _vp]),
    'lvae_sqerr_partials_f32': (_i, [_vp, _vp, _vp, _i, _l, _vp]),
}


def lib():
    """Load (once) and return the native library; raises if it is absent or has the wrong ABI."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: the HIP extension is the product path and there is no fallback. '
                f'Build it with `python lossy-vae_amd/build_native.py` (hipcc --offload-arch=gfx950).')
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
            fn.lvae_name = name
        v = L.lvae_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f'liblvae_hip.so ABI {v} != expected {ABI_VERSION}; rebuild')
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what} failed with code {rc} (hipError_t if > 0, argument error if < 0)')

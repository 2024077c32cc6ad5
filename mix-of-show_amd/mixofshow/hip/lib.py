"""ctypes binding of libmos_hip.so (C-ABI declared in include/mos_hip.h).

The library is built in-tree by `mix-of-show_amd/csrc/build.sh` (hipcc --offload-arch=gfx950).
Loading is lazy; a missing library raises MosHipUnavailable — there is NO fallback path.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOS_HIP_LIB: alternative build of the same C-ABI (kernel tuning experiments); default = the in-tree library
LIB_PATH = os.environ.get('MOS_HIP_LIB') or os.path.normpath(os.path.join(_HERE, '..', '..', 'libmos_hip.so'))

MOS_F16 = 0
MOS_BF16 = 1
MOS_LORA_PAD = 16
MOS_MAX_PCOLS = 4
MOS_MAX_SOURCES = 9


class MosHipUnavailable(RuntimeError):
    pass


class MosHipError(RuntimeError):
    pass


class LoraSites(ctypes.Structure):
    _fields_ = [
        ('n_sites', ctypes.c_int),
        ('rank', ctypes.c_int),
        ('K', ctypes.c_int),
        ('N', ctypes.c_int),
        ('down', ctypes.c_void_p * 4),
        ('up', ctypes.c_void_p * 4),
        ('alpha', ctypes.c_float * 4),
        ('n_begin', ctypes.c_int * 4),
        ('n_rows', ctypes.c_int * 4),
    ]


class LoraGroup(ctypes.Structure):
    _fields_ = [('s', LoraSites), ('A16', ctypes.c_void_p), ('A16T', ctypes.c_void_p), ('Bp16', ctypes.c_void_p),
                ('BpT', ctypes.c_void_p)]


class LoraGradOut(ctypes.Structure):
    _fields_ = [
        ('n_sites', ctypes.c_int),
        ('rank', ctypes.c_int),
        ('down_grad', ctypes.c_void_p * 4),
        ('up_grad', ctypes.c_void_p * 4),
        ('alpha', ctypes.c_float * 4),
        ('n_begin', ctypes.c_int * 4),
        ('n_rows', ctypes.c_int * 4),
        ('accumulate_down', ctypes.c_int * 4),
        ('accumulate_up', ctypes.c_int * 4),
    ]


class LoraFinalRec(ctypes.Structure):          # mos_lora_final_rec
    _fields_ = [
        ('partial', ctypes.c_void_p * 2),
        ('C', ctypes.c_int * 2),
        ('cb', ctypes.c_int * 2),
        ('nchunk', ctypes.c_int), ('nj', ctypes.c_int), ('block_begin', ctypes.c_int), ('n_blocks', ctypes.c_int),
        ('out', LoraGradOut),
    ]


class LoraGradJob(ctypes.Structure):           # mos_lora_grad_job
    _fields_ = [
        ('P', ctypes.c_void_p * 2), ('Z', ctypes.c_void_p * 2), ('ldz', ctypes.c_int64 * 2),
        ('C', ctypes.c_int * 2), ('cb', ctypes.c_int * 2), ('partial', ctypes.c_void_p * 2),
        ('M', ctypes.c_int), ('rpc', ctypes.c_int), ('nchunk', ctypes.c_int), ('nj', ctypes.c_int),
        ('block_begin', ctypes.c_int), ('n_blocks', ctypes.c_int),
        ('flops', ctypes.c_double), ('bytes', ctypes.c_double),
    ]


class GemmEpilogue(ctypes.Structure):          # mos_gemm_epilogue
    _fields_ = [('residual', ctypes.c_void_p), ('ldr', ctypes.c_int64)]


class AttnShape(ctypes.Structure):
    _fields_ = [
        ('B', ctypes.c_int), ('H', ctypes.c_int), ('Nq', ctypes.c_int), ('Nkv', ctypes.c_int), ('d', ctypes.c_int),
        ('q_bs', ctypes.c_int64), ('q_rs', ctypes.c_int64),
        ('k_bs', ctypes.c_int64), ('k_rs', ctypes.c_int64),
        ('v_bs', ctypes.c_int64), ('v_rs', ctypes.c_int64),
        ('o_bs', ctypes.c_int64), ('o_rs', ctypes.c_int64),
        ('scale', ctypes.c_float),
        ('causal', ctypes.c_int),
    ]


class AttnGradStrides(ctypes.Structure):
    _fields_ = [
        ('do_bs', ctypes.c_int64), ('do_rs', ctypes.c_int64),
        ('dq_bs', ctypes.c_int64), ('dq_rs', ctypes.c_int64),
        ('dk_bs', ctypes.c_int64), ('dk_rs', ctypes.c_int64),
        ('dv_bs', ctypes.c_int64), ('dv_rs', ctypes.c_int64),
    ]


class RegionDesc(ctypes.Structure):
    _fields_ = [
        ('n_regions', ctypes.c_int),
        ('feat_h', ctypes.c_int), ('feat_w', ctypes.c_int),
        ('box', (ctypes.c_int * 4) * (MOS_MAX_SOURCES - 1)),
        ('src_stride', ctypes.c_int64),
    ]


_vp, _i, _i64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double

# name -> (restype, argtypes); must list EVERY function declared in include/mos_hip.h
SIGNATURES = {
    'mos_version': (_i, []),
    'mos_last_error_string': (ctypes.c_char_p, []),
    'mos_profile_begin': (_i, []),
    'mos_profile_end': (_i, []),
    'mos_profile_get': (_i, [_i, ctypes.c_char_p, _i, ctypes.POINTER(ctypes.c_double),
                             ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double),
                             ctypes.POINTER(ctypes.c_double)]),
    'mos_lora_pack': (_i, [ctypes.POINTER(LoraSites), _i, _vp, _vp, _vp, _vp, _vp]),
    'mos_lora_pack_all': (_i, [_vp, _i, _i, _i, _vp]),
    'mos_lora_linear_fused_fwd': (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _i, _vp]),
    'mos_lora_linear_fwd_ex': (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _i,
                                    ctypes.POINTER(GemmEpilogue), _vp]),
    'mos_lora_linear_fused_bwd': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64,
                                       ctypes.POINTER(LoraGradOut), _vp, _i, _i, _i, _i, _i, _vp]),
    'mos_lora_linear_fused_bwd_deferred': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64,
                                                ctypes.POINTER(LoraGradOut), _vp, _i, _i, _i, _i, _i, _vp,
                                                ctypes.POINTER(LoraFinalRec)]),
    'mos_lora_linear_fused_bwd_deferred_all': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64,
                                                ctypes.POINTER(LoraGradOut), _vp, _i, _i, _i, _i, _i, _vp,
                                                ctypes.POINTER(LoraFinalRec), ctypes.POINTER(LoraGradJob)]),
    'mos_lora_grad_all': (_i, [_vp, _i, _i, _i, _i, ctypes.c_double, ctypes.c_double, _vp]),
    'mos_lora_grad_final_all': (_i, [_vp, _i, _i, _vp]),
    'mos_lora_down': (_i, [_vp, _i64, _vp, _vp, _i, _i, _i, _vp]),
    'mos_lora_linear_fwd': (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    'mos_lora_bwd_workspace_bytes': (_i64, [_i, _i, _i]),
    'mos_lora_linear_bwd': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp,
                                 _i, _i, _i, _i, _i, _vp]),
    'mos_attn_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, ctypes.POINTER(AttnShape), _i, _vp]),
    'mos_attn_bwd_workspace_bytes': (_i64, [ctypes.POINTER(AttnShape)]),
    'mos_attn_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                          ctypes.POINTER(AttnShape), ctypes.POINTER(AttnGradStrides), _i, _vp]),
    'mos_self_attn_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, ctypes.POINTER(AttnShape), _i, _vp]),
    'mos_cross_attn_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, ctypes.POINTER(AttnShape), _i, _vp]),
    'mos_region_cross_attn_fwd': (_i, [_vp, _vp, _vp, _vp, ctypes.POINTER(AttnShape), ctypes.POINTER(RegionDesc),
                                       _i, _vp]),
    'mos_region_cross_attn_fwd_chunk': (_i, [_vp, _vp, _vp, _vp, ctypes.POINTER(AttnShape), ctypes.POINTER(RegionDesc),
                                             _vp, _i, _i, _vp]),
    'mos_attn_probs': (_i, [_vp, _vp, _vp, ctypes.POINTER(AttnShape), _i, _vp]),
    'mos_attn_pv': (_i, [_vp, _vp, _vp, ctypes.POINTER(AttnShape), _i, _vp]),
    'mos_attn_probs_bwd_workspace_bytes': (_i64, [ctypes.POINTER(AttnShape)]),
    'mos_attn_pv_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(AttnShape), ctypes.POINTER(AttnGradStrides), _i, _vp]),
    'mos_attn_probs_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(AttnShape), ctypes.POINTER(AttnGradStrides),
                                _i, _vp]),
    'mos_self_attn_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(AttnShape),
                               ctypes.POINTER(AttnGradStrides), _i, _vp]),
    'mos_cross_attn_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(AttnShape),
                                ctypes.POINTER(AttnGradStrides), _i, _vp]),
    'mos_gram_workspace_bytes': (_i64, [_i64, _i, _i]),
    'mos_gram_accumulate': (_i, [_vp, _i64, _vp, _i64, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'mos_groupnorm_workspace_bytes': (_i64, [_i, _i, _i, _i]),
    'mos_groupnorm_silu_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'mos_groupnorm_silu_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mos_groupnorm_nhwc_workspace_bytes': (_i64, [_i, _i, _i, _i]),
    'mos_groupnorm_silu_fwd_nhwc': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'mos_groupnorm_silu_bwd_nhwc': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mos_groupnorm_silu_bwd_nhwc_res_ps': (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mos_groupnorm_silu_bwd_nhwc_res': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mos_conv3x3_nhwc': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'mos_conv3x3_nhwc_workspace_bytes': (_i64, [_i, _i, _i, _i, _i]),
    'mos_conv3x3_gn_tiles': (_i, [_i, _i, _i, _i, _i]),
    'mos_conv3x3_nhwc_px': (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'mos_conv3x3_nhwc_gn': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'mos_groupnorm_nhwc_reads_twice': (_i, [_i, _i, _i, _i]),
    'mos_groupnorm_silu_fwd_nhwc_pre': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'mos_conv3x3_s2_nhwc': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'mos_conv3x3_nhwc_ws': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'mos_layernorm_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp]),
    'mos_layernorm_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'mos_add_layernorm_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _vp]),
    'mos_add_layernorm_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mos_geglu_fwd': (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    'mos_softmax_rows': (_i, [_vp, _vp, _i, _i, _f, _i, _vp]),
    'mos_geglu_bwd': (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    'mos_quick_gelu_fwd': (_i, [_vp, _vp, _i64, _i, _vp]),
    'mos_quick_gelu_bwd': (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    'mos_lsq_workspace_bytes': (_i64, [_i, _i]),
    'mos_lsq_loss_grad_gram': (_i, [_vp, _vp, _vp, _vp, _d, _i, _i, _vp, _vp, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()




def load():
    """Return the loaded library (ctypes.CDLL) with typed entry points; raise if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MosHipUnavailable(
                f'libmos_hip.so not found at {LIB_PATH}. Build it with `bash mix-of-show_amd/csrc/build.sh` '
                '(or `python -c "import __graft_entry__ as g; g.build()"`). mixofshow has no CPU/PyTorch fallback '
                'for its kernel-backed ops.')
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().mos_last_error_string().decode('utf-8', 'replace')
        raise MosHipError(f'{what} failed (status {rc}): {msg}')

"""ctypes binding of libmts_hip.so (the C ABI declared in include/mts.h).

The product path has NO fallback: if the shared library is missing or fails to load, importing this
module raises -- build it with ``python multimodaltopicsegmentation_amd/build.py`` (hipcc, gfx950).
"""
import ctypes as C
import os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The late-fusion tagger runs its two encoders on two streams;
# once RCCL's own streams exist in the process (any N > 1 run) two of ours share a hardware queue and the encoders run one after
# the other: 13.3 ms per step instead of 7.7 at 64 x 512 (measured with a one-rank RCCL group, profiles/r03_hw_queues.txt).
# Eight queues restore the overlap.  Only effective if set before the HIP runtime initialises, hence here, at import.
_HWQ_PRESET = 'GPU_MAX_HW_QUEUES' in os.environ
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch

if not _HWQ_PRESET and torch.cuda.is_initialized():
    # the HIP runtime read its environment before this import: the default of four hardware queues stays for this process
    import warnings
    warnings.warn('multimodaltopicsegmentation_amd was imported after the GPU runtime had been initialised: GPU_MAX_HW_QUEUES=8 could not '
                  'take effect (the late-fusion tagger\'s two encoder streams may share a hardware queue next to RCCL: 13.3 instead of 7.7 ms '
                  'per step at 64 x 512).  Import the package first, or export GPU_MAX_HW_QUEUES=8.', RuntimeWarning, stacklevel=2)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmts_hip.so')

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} not found: the HIP kernel library has not been built. Run '
        '`python multimodaltopicsegmentation_amd/build.py` (needs hipcc; cross-compiles for gfx950 without a GPU). '
        'There is no CPU/PyTorch fallback for the tagger path.')

lib = C.CDLL(LIB_PATH)

F32, BF16 = 0, 1
LOSS_CE, LOSS_BCE, LOSS_FOCAL = 0, 1, 2
NT, NN, TN, TT = 0, 1, 2, 3
EPI_BIAS, EPI_RESIDUAL, EPI_GELU, EPI_COLSCALE, EPI_ACCUM, EPI_RELU = 1, 2, 4, 8, 16, 32

_vp, _i, _f, _u, _sz = C.c_void_p, C.c_int, C.c_float, C.c_uint, C.c_size_t

# name -> (restype, argtypes): must match include/mts.h exactly (tests/test_abi.py parses the header and checks)
SIGNATURES = {
    'mts_last_error': (C.c_char_p, []),
    'mts_version': (C.c_char_p, []),
    'mts_set_option': (_i, [C.c_char_p, _i]),
    'mts_gemm_last_plan': (_i, [_vp, _vp]),
    'mts_gemm_set_mid_hook': (_i, [_vp]),
    'mts_gemm_plan': (_i, [_i, _i, _i, _i, _i, _i, _u, _sz, _vp, _vp]),
    'mts_async_status': (_i, []),
    'mts_gemm': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _u, _f, _i, _vp, _sz]),
    'mts_wgrad_pair_workspace': (_sz, [_i, _i, _i]),
    'mts_wgrad_pair': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz]),
    'mts_colsum_workspace': (_sz, [_i]),
    'mts_colsum': (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    'mts_cast': (_i, [_vp, _i, _vp, _vp, _sz]),
    'mts_cast_concat': (_i, [_vp, _i, _sz, _i, _i, _vp, _vp, _vp]),
    'mts_embed_layernorm_fwd2': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i]),
    'mts_embed_layernorm_fwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i]),
    'mts_embed_layernorm_fwd_x16': (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i]),
    'mts_layernorm_fwd': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'mts_layernorm_bwd_workspace': (_sz, [_i]),
    'mts_layernorm_bwd': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mts_layernorm_loss_tail_supported': (_i, [_i, _i, _i]),
    'mts_layernorm_loss_tail': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _f, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp]),
    'mts_embed_layernorm_bwd_workspace': (_sz, [_i, _i, _i]),
    'mts_embed_layernorm_bwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _sz]),
    'mts_embed_bwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    'mts_dropout_fwd': (_i, [_vp, _i, _sz, _vp, _vp, _vp, _vp, _f, C.c_uint64]),
    'mts_dropout_bwd': (_i, [_vp, _i, _sz, _vp, _vp, _vp, _f]),
    'mts_gelu_bwd': (_i, [_vp, _i, _sz, _vp, _vp]),
    'mts_relu_bwd': (_i, [_vp, _i, _sz, _vp, _vp]),
    'mts_ffn_supported': (_i, [_i, _i, _i, _i]),
    'mts_ffn_fwd': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    'mts_ffn_bwd_data': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'mts_band_slots': (_i, [_i]),
    'mts_band_attn_fwd': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, C.c_uint64]),
    'mts_band_attn_bwd_workspace': (_sz, [_i, _i, _i]),
    'mts_band_attn_bwd': (_i, [_vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, C.c_uint64]),
    'mts_tagger_loss_workspace': (_sz, [_i, _i]),
    'mts_tagger_loss': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _sz, _vp, _i]),
    'mts_greedy_decode': (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _vp]),
    'mts_head_fwd': (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    'mts_head_bwd_params': (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    'mts_head_bwd_data': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i]),
    'mts_lstm_workspace': (_sz, [_i, _i, _i, _i, _i]),
    'mts_lstm_fwd': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mts_lstm_bwd': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mts_lstm_bwd_recurrence': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mts_lstm_bwd_whh': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'mts_crf_workspace': (_sz, [_i, _i, _i]),
    'mts_crf_nll': (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mts_crf_viterbi': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mts_adam_step': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i, _f, _vp]),
    'mts_sgd_step': (_i, [_vp, _sz, _vp, _vp, _vp, _f, _f, _f, _i, _f, _vp]),
    'mts_scale': (_i, [_vp, _sz, _vp, _f]),
    'mts_collate_pad': (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i]),
}

_missing = []
for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError:
        _missing.append(_name)
        continue
    _fn.restype = _res
    _fn.argtypes = _args
if _missing:
    raise ImportError(f'{LIB_PATH} does not export {_missing}: stale build? re-run python multimodaltopicsegmentation_amd/build.py --force')


class MtsError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = lib.mts_last_error().decode('utf-8', 'replace')
        if rc == 1:
            raise ValueError(msg)                       # MTS_ERR_INVALID  <-> the reference's ValueError / asserts
        if rc == 2:
            raise NotImplementedError(msg)              # MTS_ERR_UNSUPPORTED
        if rc == 5:
            raise MtsError(f'device-side timeout: {msg}')   # MTS_ERR_TIMEOUT (reported by the call AFTER the one that failed)
        raise MtsError(f'mts error {rc}: {msg}')


def check_async():
    """Raise if a launch since the last poll reported a device-side error (no synchronisation; see mts_async_status)."""
    check(lib.mts_async_status())


def stream_ptr():
    """Raw hipStream_t of torch's current stream (kernels are enqueued on it, so torch ordering rules apply)."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise ValueError(f'unsupported activation dtype {dt}')


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('multimodaltopicsegmentation_amd: no GPU visible. The tagger path runs only on its HIP kernels '
                           '(MI355X / gfx950); there is no CPU fallback.')

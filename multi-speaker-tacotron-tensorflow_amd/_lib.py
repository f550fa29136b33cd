"""ctypes binding of libtaco_hip.so (include/taco_abi.h; the test / timing hooks of include/taco_debug.h).  There is NO fallback: if the HIP library
is missing or fails to load, importing the compute path raises."""
import ctypes as C
import os

from .hparams import MODEL_TYPES, ATTENTION_TYPES, NUM_SYMBOLS

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtaco_hip.so")

TACO_ERR_ARG, TACO_ERR_SHAPE, TACO_ERR_UNSUPPORTED, TACO_ERR_HIP, TACO_ERR_STATE = -1, -2, -3, -4, -5


class TacoHParams(C.Structure):
    _fields_ = [
        ("num_symbols", C.c_int32), ("num_mels", C.c_int32), ("num_freq", C.c_int32),
        ("num_speakers", C.c_int32), ("model_type", C.c_int32), ("speaker_embedding_size", C.c_int32),
        ("embedding_size", C.c_int32),
        ("enc_prenet_n", C.c_int32), ("enc_prenet", C.c_int32 * 4),
        ("enc_bank_size", C.c_int32), ("enc_bank_channels", C.c_int32), ("enc_maxpool", C.c_int32),
        ("enc_highway_depth", C.c_int32), ("enc_rnn_size", C.c_int32),
        ("enc_proj_n", C.c_int32), ("enc_proj", C.c_int32 * 4), ("enc_proj_width", C.c_int32),
        ("attention_type", C.c_int32), ("attention_size", C.c_int32), ("attention_state_size", C.c_int32),
        ("dec_layer_num", C.c_int32), ("dec_rnn_size", C.c_int32),
        ("dec_prenet_n", C.c_int32), ("dec_prenet", C.c_int32 * 4),
        ("post_bank_size", C.c_int32), ("post_bank_channels", C.c_int32), ("post_maxpool", C.c_int32),
        ("post_highway_depth", C.c_int32), ("post_rnn_size", C.c_int32),
        ("post_proj_n", C.c_int32), ("post_proj", C.c_int32 * 4), ("post_proj_width", C.c_int32),
        ("reduction_factor", C.c_int32), ("max_iters", C.c_int32),
    ]


class TacoAudioHParams(C.Structure):
    _fields_ = [("num_freq", C.c_int32), ("sample_rate", C.c_int32), ("griffin_lim_iters", C.c_int32),
                ("frame_length_ms", C.c_float), ("frame_shift_ms", C.c_float), ("preemphasis", C.c_float),
                ("min_level_db", C.c_float), ("ref_level_db", C.c_float), ("power", C.c_float)]


class TacoError(Exception):
    """Raised for every non-zero return of the C ABI.  The reference raises bare `Exception`
    for unknown model/attention types and shape mismatches (tacotron.py:88,152,192-194)."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def _list4(vals, what):
    vals = list(vals)
    if not 1 <= len(vals) <= 4:
        raise TacoError(TACO_ERR_ARG, "%s must have 1..4 entries, got %d" % (what, len(vals)))
    arr = (C.c_int32 * 4)(*(vals + [0] * (4 - len(vals))))
    return len(vals), arr


def to_c_hparams(hp, num_speakers):
    """HParams (reference key names) -> taco_hparams POD."""
    if hp.model_type not in MODEL_TYPES:
        raise Exception(" [!] Unkown multi-speaker model type: {}".format(hp.model_type))   # tacotron.py:88
    if hp.attention_type not in ATTENTION_TYPES:
        raise Exception(" [!] Unkown attention type: {}".format(hp.attention_type))         # tacotron.py:152
    c = TacoHParams()
    c.num_symbols = getattr(hp, "num_symbols", NUM_SYMBOLS)
    c.num_mels, c.num_freq = hp.num_mels, hp.num_freq
    c.num_speakers = int(num_speakers)
    c.model_type = MODEL_TYPES[hp.model_type]
    c.speaker_embedding_size = hp.speaker_embedding_size
    c.embedding_size = hp.embedding_size
    c.enc_prenet_n, c.enc_prenet = _list4(hp.enc_prenet_sizes, "enc_prenet_sizes")
    c.enc_bank_size, c.enc_bank_channels = hp.enc_bank_size, hp.enc_bank_channel_size
    c.enc_maxpool, c.enc_highway_depth, c.enc_rnn_size = hp.enc_maxpool_width, hp.enc_highway_depth, hp.enc_rnn_size
    c.enc_proj_n, c.enc_proj = _list4(hp.enc_proj_sizes, "enc_proj_sizes")
    c.enc_proj_width = hp.enc_proj_width
    c.attention_type = ATTENTION_TYPES[hp.attention_type]
    c.attention_size, c.attention_state_size = hp.attention_size, hp.attention_state_size
    c.dec_layer_num, c.dec_rnn_size = hp.dec_layer_num, hp.dec_rnn_size
    c.dec_prenet_n, c.dec_prenet = _list4(hp.dec_prenet_sizes, "dec_prenet_sizes")
    c.post_bank_size, c.post_bank_channels = hp.post_bank_size, hp.post_bank_channel_size
    c.post_maxpool, c.post_highway_depth, c.post_rnn_size = hp.post_maxpool_width, hp.post_highway_depth, hp.post_rnn_size
    c.post_proj_n, c.post_proj = _list4(hp.post_proj_sizes, "post_proj_sizes")
    c.post_proj_width = hp.post_proj_width
    c.reduction_factor, c.max_iters = hp.reduction_factor, hp.max_iters
    return c


_P = C.c_void_p
_I = C.c_int
_S = C.c_size_t
# taco_sync_sum_fn (include/taco_abi.h): void (*)(void* user, float* d_vec, int n)
SYNC_SUM_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int)

# name -> (restype, argtypes); every symbol declared in include/taco_abi.h
PROTOTYPES = {
    "taco_abi_version": (_I, []),
    "taco_last_error": (C.c_char_p, []),
    "taco_model_create": (_I, [C.POINTER(TacoHParams), _I, C.POINTER(_P)]),
    "taco_model_set_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "taco_model_num_weights": (_I, [_P]),
    "taco_model_weight_name": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(C.c_int64), C.POINTER(_I)]),
    "taco_model_finalize": (_I, [_P]),
    "taco_model_destroy": (None, [_P]),
    "taco_workspace_bytes": (_S, [_P, _I, _I, _I]),
    "taco_stage_workspace_bytes": (_S, [_P, _I, _I]),
    "taco_forward_infer": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _S]),
    "taco_plan_create": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _S, C.POINTER(_P)]),
    "taco_plan_launch": (_I, [_P, _P]),
    "taco_plan_num_nodes": (_I, [_P]),
    "taco_plan_whole_chip": (_I, [_P]),
    "taco_plan_destroy": (None, [_P]),
    "taco_encoder_forward": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _S]),
    "taco_decoder_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _S]),
    "taco_postnet_forward": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _S]),
    "taco_conv1d_bn_f32": (_I, [_P, _P, C.c_char_p, _P, _I, _I, _I, _I, _P]),
    "taco_dense_f32": (_I, [_P, _P, C.c_char_p, _P, _I, _I, _P]),
    "taco_highway_f32": (_I, [_P, _P, C.c_char_p, _P, _I, _P]),
    "taco_bigru_f32": (_I, [_P, _P, C.c_char_p, _P, _P, _P, _I, _I, _P, _P, _S]),
    "taco_attention_step_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _S]),
    "taco_gru_cell_f32": (_I, [_P, _P, C.c_char_p, _P, _P, _I, _P, _P, _S]),
    "taco_gl_create": (_I, [C.POINTER(TacoAudioHParams), _I, C.POINTER(_P)]),
    "taco_gl_destroy": (None, [_P]),
    "taco_gl_num_samples": (_I, [_P, _I]),
    "taco_gl_workspace_bytes": (_S, [_P, _I, _I]),
    "taco_gl_inv_spectrogram": (_I, [_P, _P, _P, _P, C.c_ulonglong, _I, _I, _I, _P, _P, _S]),
    "taco_attention_trim": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "taco_loss_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _S]),
    "taco_learning_rate": (C.c_float, [C.c_longlong, C.c_float, _I, _I]),
    "taco_adam_step_f32": (_I, [_P, _P, _P, _P, _P, _S, C.c_longlong, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, _S]),
    "taco_train_create": (_I, [C.POINTER(TacoHParams), _I, C.POINTER(_P)]),
    "taco_train_destroy": (None, [_P]),
    "taco_train_model": (_P, [_P]),
    "taco_train_num_params": (_S, [_P]),
    "taco_train_param_offset": (_I, [_P, C.c_char_p, C.POINTER(_S)]),
    "taco_train_refresh": (_I, [_P, _P, _P]),
    "taco_train_set_sync_bn": (_I, [_P, _P, _P, _I]),
    "taco_train_set_deterministic": (_I, [_P, _I]),
    "taco_train_set_exact_wgrad": (_I, [_P, _I]),
    "taco_train_set_wgrad_planes": (_I, [_P, _I]),
    "taco_train_planes_problems": (_I, [_P]),
    "taco_train_set_bptt_engine": (_I, [_P, _I]),
    "taco_train_set_exact_gemm": (_I, [_P, _I]),
    "taco_train_debug_bigru": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _S]),
    "taco_train_workspace_bytes": (_S, [_P, _I, _I, _I]),
    "taco_train_forward_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _S]),
    "taco_debug_force_gemm_config": (_I, [_P, _I]),
    "taco_debug_set_skip_scans": (_I, [_P, _I]),
    "taco_debug_set_chip_turns": (_I, [_I]),
    "taco_debug_set_front": (_I, [_P, _I, _I]),
    "taco_debug_set_persistent": (_I, [_P, _I]),
    "taco_debug_set_overlap": (_I, [_P, _I]),
    "taco_stop_steps": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "taco_debug_set_fuse_prenet": (_I, [_P, _I]),
    "taco_debug_set_fuse_concat": (_I, [_P, _I]),
    "taco_debug_set_att_split": (_I, [_P, _I]),
    "taco_debug_set_bf3": (_I, [_P, _I, _I]),
    "taco_model_device_errors": (_I, [_P, C.POINTER(_I)]),
    "taco_debug_raise_device_error": (_I, [_P, _I]),
    "taco_debug_set_decoder_persist": (_I, [_P, _I, _I]),
    "taco_debug_decoder_info": (_I, [_P, C.POINTER(_I)]),
    "taco_model_engine_plan": (_I, [_P, _I, _I, _I, _I, C.c_char_p, _I]),
    "taco_model_set_batch_invariant": (_I, [_P, _I]),
    "taco_debug_decoder_trace": (_I, [_P, _I, C.POINTER(C.c_longlong)]),
}

_lib = None


def load_library(path=None):
    """Load libtaco_hip.so and bind every symbol of the header.  Raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("TACO_LIB") or LIB_PATH      # TACO_LIB: an instrumented build (tools/trace_*.py, -DTACO_TRACE)
    if path is None and os.environ.get("TACO_LIB") and os.path.abspath(p) != os.path.abspath(LIB_PATH):
        import warnings
        warnings.warn("TACO_LIB overrides the library: loading %s instead of %s" % (p, LIB_PATH), RuntimeWarning, stacklevel=2)
    if not os.path.exists(p):
        raise ImportError(
            "libtaco_hip.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; g.build()' "
            "or multi-speaker-tacotron-tensorflow_amd/csrc/build.sh).  There is no CPU fallback." % p)
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.taco_abi_version() != 1:
        raise ImportError("libtaco_hip.so ABI version %d, expected 1" % lib.taco_abi_version())
    if path is None:
        _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load_library().taco_last_error().decode("utf-8", "replace")
        if rc == TACO_ERR_UNSUPPORTED and msg.startswith(" [!]"):
            raise Exception(msg)     # same text and type as the reference (tacotron.py:88,152)
        raise TacoError(rc, "libtaco_hip error %d: %s" % (rc, msg))

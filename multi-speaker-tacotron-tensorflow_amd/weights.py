"""Weight pack: canonical tensor names <-> TF variable names, random initialisation with the
reference's initialisers, safetensors save/load.

Replaces the variable creation inside Tacotron.initialize (tf.get_variable / tf.layers defaults,
models/tacotron.py:34-47, models/modules.py:11-15,119) and tf.train.Saver save/restore
(train.py:175,242-244; synthesizer.py:66-67).  The tensor list itself comes from the C library
(taco_model_weight_name), so Python and C cannot drift apart."""
import ctypes as C
import math
import os

import numpy as np

from . import _lib

# canonical scope -> TF 1.x scope under model/inference/ (SURVEY App. B).  Wrapper scope strings are
# TF-version dependent; this is the table a TF1-checkpoint importer would start from.
TF_SCOPE_MAP = {
    "embedding": "embedding",
    "speaker_embedding": "speaker_embedding",
    "spk/before_highway": "dense",
    "spk/encoder_rnn_init": "dense_1",
    "spk/attention_rnn_init": "dense_2",
    "spk/decoder_rnn_init_1": "dense_3",
    "spk/decoder_rnn_init_2": "dense_4",
    "prenet": "prenet",
    "encoder_cbhg": "encoder_cbhg",
    "post_cbhg": "post_cbhg",
    "attention/memory_layer": "memory_layer",
    "attention/query_layer": "decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper/"
                             "concat_output_and_attention_wrapper/attention_wrapper/bahdanau_monotonic_attention/query_layer",
    "decoder/prenet": "decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper/"
                      "concat_output_and_attention_wrapper/attention_wrapper/decoder_prenet_wrapper/decoder_prenet",
    "decoder/attention_gru": "decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper/"
                             "concat_output_and_attention_wrapper/attention_wrapper/decoder_prenet_wrapper/gru_cell",
    "decoder/concat_projection": "decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper",
    "decoder/gru_1": "decoder/output_projection_wrapper/multi_rnn_cell/cell_1/gru_cell",
    "decoder/gru_2": "decoder/output_projection_wrapper/multi_rnn_cell/cell_2/gru_cell",
    "decoder/frame_projection": "decoder/output_projection_wrapper",
    "linear": "dense_5 (dense when single-speaker)",
}


def weight_spec(hp, num_speakers=1):
    """[(name, shape)] required by these hparams, as enumerated by the C library."""
    lib = _lib.load_library()
    chp = _lib.to_c_hparams(hp, num_speakers)
    h = C.c_void_p()
    _lib.check(lib.taco_model_create(C.byref(chp), 0, C.byref(h)))
    try:
        out = []
        buf = C.create_string_buffer(256)
        shp = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(lib.taco_model_num_weights(h)):
            _lib.check(lib.taco_model_weight_name(h, i, buf, 256, shp, C.byref(nd)))
            out.append((buf.value.decode(), tuple(int(shp[d]) for d in range(nd.value))))
        return out
    finally:
        lib.taco_model_destroy(h)


def _glorot(rs, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, size=shape)


def _trunc_normal(rs, shape, std):
    x = rs.normal(0.0, std, size=shape)
    bad = np.abs(x) > 2 * std
    while bad.any():
        x[bad] = rs.normal(0.0, std, size=int(bad.sum()))
        bad = np.abs(x) > 2 * std
    return x


def random_weights(hp, num_speakers=1, seed=0):
    """What tf.global_variables_initializer() would produce (synthesizer.py:65, train.py:187):
    embeddings truncated_normal(0.5) (tacotron.py:36,47), get_embed tables truncated_normal(0.1)
    (modules.py:14), kernels glorot_uniform, biases 0 except GRU gate bias 1 and highway T bias -1
    (modules.py:119), BatchNorm gamma 1 / beta 0 / mean 0 / variance 1, attention_v glorot,
    attention_score_bias 0, attention_g sqrt(1/units), attention_b 0."""
    rs = np.random.RandomState(seed)
    w = {}
    for name, shp in weight_spec(hp, num_speakers):
        leaf = name.rsplit("/", 1)[-1]
        if name in ("embedding", "speaker_embedding"):
            a = _trunc_normal(rs, shp, 0.5)
        elif leaf == "table":
            a = _trunc_normal(rs, shp, 0.1)
        elif leaf == "kernel":
            a = _glorot(rs, shp, shp[0] * shp[1], shp[0] * shp[2]) if len(shp) == 3 else _glorot(rs, shp, shp[0], shp[1])
        elif leaf == "bias":
            a = np.ones(shp) if "/gates/" in name else (-np.ones(shp) if name.endswith("/T/bias") else np.zeros(shp))
        elif leaf in ("gamma", "moving_variance"):
            a = np.ones(shp)
        elif leaf in ("beta", "moving_mean", "attention_score_bias", "attention_b"):
            a = np.zeros(shp)
        elif leaf == "attention_v":
            a = _glorot(rs, shp, shp[0], 1)
        elif leaf == "attention_g":
            a = np.full(shp, math.sqrt(1.0 / hp.attention_size))
        else:
            raise KeyError(name)
        w[name] = np.asarray(a, dtype=np.float32)
    return w


def save_weights(path, weights):
    from safetensors.numpy import save_file
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    # safetensors has no rank-0 support problems, but keep scalars as shape-(1,) for portability
    save_file({k: np.ascontiguousarray(v.reshape(1) if v.ndim == 0 else v, dtype=np.float32)
               for k, v in weights.items()}, path)


def load_weights(path):
    from safetensors.numpy import load_file
    out = {}
    for k, v in load_file(path).items():
        out[k] = v.reshape(()) if k.endswith(("attention_score_bias", "attention_g")) else v
    return out

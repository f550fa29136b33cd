"""CPU oracle for the Tacotron hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED: the reference (GSByeon/multi-speaker-tacotron-tensorflow) delegates every
arithmetic operation to TensorFlow 1.x (requirements.txt:101), which is not vendored under
/root/reference, is not installed here and cannot be installed (no network, no py3.10 wheel).
The reference ships no tests, golden vectors or checkpoints.  This file is therefore a
restatement of (a) the wiring in the reference's own files (cited per function as
`file:line`, relative to the reference root) and (b) the published behaviour of the TF 1.4
ops those lines call (marked TF-sem).  Nothing here was checked against a running TF.
Pinned on the reference's own code, without TensorFlow: (a) -- the wiring -- by the call trace of the reference's model files
(tools/trace_reference_graph.py -> tests/golden/graph_trace.json, tests/test_reference_graph.py), and the trim walk and manual
alignments at the end of this file by reference-run vectors (tests/test_reference_vectors.py).  (b) -- the arithmetic inside TF's
ops -- is what remains unpinned.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / reported CPU baseline.  The product path
(`multi-speaker-tacotron-tensorflow_amd/`) never imports it.

Everything is plain NumPy, dtype-parametric (float64 = checker, float32 = CPU baseline).
Tensors are row-major [batch, time, channels] like the reference's.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Optional, Sequence

import numpy as np

# ----------------------------------------------------------------------------------------
# hyper-parameters: the *model* keys of hparams.py:33-69 plus max_iters (hparams.py:141).
# Defaults are the effective values after the override chain in hparams.py:83-94.
# ----------------------------------------------------------------------------------------

NUM_SYMBOLS = 80  # text/symbols.py:13 -> len(symbols); PAD=0, EOS=1 (text/korean.py:11-21)


@dataclass
class OracleHParams:
    num_symbols: int = NUM_SYMBOLS
    num_mels: int = 80
    num_freq: int = 1025
    model_type: str = "single"            # single | simple | deepvoice   (hparams.py:34)
    speaker_embedding_size: int = 16
    embedding_size: int = 256
    enc_prenet_sizes: List[int] = field(default_factory=lambda: [256, 128])
    enc_bank_size: int = 16
    enc_bank_channel_size: int = 128
    enc_maxpool_width: int = 2
    enc_highway_depth: int = 4
    enc_rnn_size: int = 128
    enc_proj_sizes: List[int] = field(default_factory=lambda: [128, 128])
    enc_proj_width: int = 3
    attention_type: str = "bah_mon"       # bah | bah_norm | bah_mon      (hparams.py:50)
    attention_size: int = 256
    attention_state_size: int = 256
    dec_layer_num: int = 2
    dec_rnn_size: int = 256
    dec_prenet_sizes: List[int] = field(default_factory=lambda: [256, 128])
    post_bank_size: int = 8
    post_bank_channel_size: int = 256
    post_maxpool_width: int = 2
    post_highway_depth: int = 4
    post_rnn_size: int = 256              # overridden at hparams.py:91
    post_proj_sizes: List[int] = field(default_factory=lambda: [256, 80])
    post_proj_width: int = 3
    reduction_factor: int = 4
    max_iters: int = 200
    prioritize_loss: bool = False         # hparams.py:133 (training loss only)
    sample_rate: int = 24000              # hparams.py:28

    def to_dict(self):
        return asdict(self)

    @staticmethod
    def scaled(div: int, **kw) -> "OracleHParams":
        """Shrunken widths, like SCALE_FACTOR in hparams.py:3-6 (num_mels/num_freq given explicitly)."""
        f = lambda n: max(n // div, 4)
        hp = OracleHParams(
            speaker_embedding_size=f(16), embedding_size=f(256),
            enc_prenet_sizes=[f(256), f(128)], enc_bank_channel_size=f(128), enc_rnn_size=f(128),
            enc_proj_sizes=[f(128), f(128)], attention_size=f(256), attention_state_size=f(256),
            dec_rnn_size=f(256), dec_prenet_sizes=[f(256), f(128)], post_bank_channel_size=f(256),
            post_rnn_size=f(256), post_proj_sizes=[f(256), kw.get("num_mels", 80)])
        for k, v in kw.items():
            setattr(hp, k, v)
        hp.post_proj_sizes = [hp.post_proj_sizes[0], hp.num_mels]
        return hp


# ----------------------------------------------------------------------------------------
# weights: canonical names (see DESIGN.md for the TF variable each one stands for)
# ----------------------------------------------------------------------------------------

def _glorot(rs, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))      # TF-sem: glorot_uniform is the tf.layers default
    return rs.uniform(-lim, lim, size=shape)


def _trunc_normal(rs, shape, std):
    x = rs.normal(0.0, std, size=shape)            # TF-sem truncated_normal: resample beyond 2 sigma
    bad = np.abs(x) > 2 * std
    while bad.any():
        x[bad] = rs.normal(0.0, std, size=int(bad.sum()))
        bad = np.abs(x) > 2 * std
    return x


def weight_shapes(hp: OracleHParams, num_speakers: int = 1) -> Dict[str, tuple]:
    """Every tensor of the model and its shape (TF conventions: dense kernel
    [in,out], conv1d kernel [k,in,out], GRU gates/kernel [in+n,2n] (r|u), candidate [in+n,n])."""
    s: Dict[str, tuple] = {}
    E = hp.embedding_size
    s["embedding"] = (hp.num_symbols, E)                                   # tacotron.py:34-36
    multi = num_speakers > 1
    spk = hp.speaker_embedding_size
    if multi:
        if spk != 1:
            s["speaker_embedding"] = (num_speakers, spk)                   # tacotron.py:44-47
        if hp.model_type == "deepvoice":
            dims = [("before_highway", hp.enc_prenet_sizes[-1]),
                    ("encoder_rnn_init", hp.enc_rnn_size * 2),
                    ("attention_rnn_init", hp.attention_state_size)]
            dims += [("decoder_rnn_init_%d" % (i + 1), hp.dec_rnn_size) for i in range(hp.dec_layer_num)]
            for name, d in dims:
                if spk == 1:
                    s["spk/%s/table" % name] = (num_speakers, d)          # tacotron.py:52-66 get_embed
                else:
                    s["spk/%s/kernel" % name] = (spk, d)                   # tacotron.py:68-79
                    s["spk/%s/bias" % name] = (d,)
    simple_spk = spk if (multi and hp.model_type == "simple") else 0

    def dense(name, i, o, bias=True):
        s[name + "/kernel"] = (i, o)
        if bias:
            s[name + "/bias"] = (o,)

    def conv_bn(name, k, i, o):
        s[name + "/kernel"] = (k, i, o)
        s[name + "/bias"] = (o,)
        for p in ("gamma", "beta", "moving_mean", "moving_variance"):
            s[name + "/" + p] = (o,)

    def gru(name, i, n):
        s[name + "/gates/kernel"] = (i + n, 2 * n)
        s[name + "/gates/bias"] = (2 * n,)
        s[name + "/candidate/kernel"] = (i + n, n)
        s[name + "/candidate/bias"] = (n,)

    def cbhg(scope, in_dim, K, C, depth, rnn, projs, pw):
        for k in range(1, K + 1):
            conv_bn("%s/conv_bank/conv1d_%d" % (scope, k), k, in_dim, C)   # modules.py:35-44
        d = K * C
        for i, p in enumerate(projs):
            conv_bn("%s/proj_%d" % (scope, i + 1), pw, d, p)               # modules.py:54-59
            d = p
        if d != rnn:
            dense(scope + "/dense", d, rnn)                                # modules.py:72-73
        for i in range(depth):
            dense("%s/highway_%d/H" % (scope, i + 1), rnn, rnn)            # modules.py:105-120
            dense("%s/highway_%d/T" % (scope, i + 1), rnn, rnn)
        gru(scope + "/bigru/fw", rnn, rnn)                                 # modules.py:88-95
        gru(scope + "/bigru/bw", rnn, rnn)

    d = E
    for i, sz in enumerate(hp.enc_prenet_sizes):                           # modules.py:18-25
        dense("prenet/dense_%d" % (i + 1), d, sz)
        d = sz
    cbhg("encoder_cbhg", d, hp.enc_bank_size, hp.enc_bank_channel_size, hp.enc_highway_depth,
         hp.enc_rnn_size, hp.enc_proj_sizes, hp.enc_proj_width)
    enc_out = 2 * hp.enc_rnn_size
    A = hp.attention_size
    dense("attention/memory_layer", enc_out, A, bias=False)                # TF-sem (BahdanauAttention ctor)
    dense("attention/query_layer", hp.attention_state_size, A, bias=False)
    s["attention/attention_v"] = (A,)
    if hp.attention_type == "bah_mon":
        s["attention/attention_score_bias"] = ()
    if hp.attention_type == "bah_norm":
        s["attention/attention_g"] = ()
        s["attention/attention_b"] = (A,)
    d = hp.num_mels + enc_out                                              # rnn_wrappers.py:249
    for i, sz in enumerate(hp.dec_prenet_sizes):
        dense("decoder/prenet/dense_%d" % (i + 1), d, sz)                  # rnn_wrappers.py:367-370
        d = sz
    gru("decoder/attention_gru", d + simple_spk, hp.attention_state_size)  # tacotron.py:127-130
    dense("decoder/concat_projection", hp.attention_state_size + enc_out + simple_spk, hp.dec_rnn_size)  # :166-170
    for i in range(hp.dec_layer_num):
        gru("decoder/gru_%d" % (i + 1), hp.dec_rnn_size, hp.dec_rnn_size)  # tacotron.py:171-172
    dense("decoder/frame_projection", hp.dec_rnn_size, hp.num_mels * hp.reduction_factor)  # :178-179
    cbhg("post_cbhg", hp.num_mels, hp.post_bank_size, hp.post_bank_channel_size, hp.post_highway_depth,
         hp.post_rnn_size, hp.post_proj_sizes, hp.post_proj_width)         # tacotron.py:219-224
    dense("linear", 2 * hp.post_rnn_size + simple_spk, hp.num_freq)        # tacotron.py:226-235
    return s


def init_weights(hp: OracleHParams, num_speakers: int = 1, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded weights following the reference initialisers: embeddings truncated
    normal 0.5 (tacotron.py:36,47) / 0.1 (modules.py:14); kernels Glorot-uniform; biases 0
    except GRU gate bias 1.0 (TF-sem) and highway T bias -1 (modules.py:119).  BatchNorm
    statistics are randomised (not identity) so the BN epilogue is exercised."""
    rs = np.random.RandomState(seed)
    w: Dict[str, np.ndarray] = {}
    for name, shp in weight_shapes(hp, num_speakers).items():
        leaf = name.rsplit("/", 1)[-1]
        if name in ("embedding", "speaker_embedding"):
            a = _trunc_normal(rs, shp, 0.5)
        elif leaf == "table":
            a = _trunc_normal(rs, shp, 0.1)
        elif leaf == "kernel":
            if len(shp) == 3:
                a = _glorot(rs, shp, shp[0] * shp[1], shp[0] * shp[2])
            else:
                a = _glorot(rs, shp, shp[0], shp[1])
        elif leaf == "bias":
            if "/gates/" in name:
                a = np.ones(shp)
            elif name.endswith("/T/bias"):
                a = -np.ones(shp)
            else:
                a = rs.normal(0.0, 0.05, size=shp)   # non-zero so a dropped bias shows up in parity
        elif leaf == "gamma":
            a = rs.uniform(0.5, 1.5, size=shp)
        elif leaf == "beta":
            a = rs.normal(0.0, 0.1, size=shp)
        elif leaf == "moving_mean":
            a = rs.normal(0.0, 0.1, size=shp)
        elif leaf == "moving_variance":
            a = rs.uniform(0.5, 1.5, size=shp)
        elif leaf == "attention_v":
            a = _glorot(rs, shp, shp[0], 1)
        elif leaf == "attention_score_bias":
            a = np.zeros(shp)
        elif leaf == "attention_g":
            a = np.full(shp, math.sqrt(1.0 / hp.attention_size))
        elif leaf == "attention_b":
            a = rs.normal(0.0, 0.05, size=shp)
        else:
            raise KeyError(name)
        w[name] = np.asarray(a, dtype=np.float32)
    return w


# ----------------------------------------------------------------------------------------
# element ops
# ----------------------------------------------------------------------------------------

def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def relu(x):
    return np.maximum(x, 0)


def softsign(x):
    return x / (1.0 + np.abs(x))


BN_EPS = 1e-3  # TF-sem: tf.layers.batch_normalization default epsilon (modules.py:131)


def dense(x, w, name, act=None, bias=True):
    """TF-sem tf.layers.dense on the last axis: y = x.W + b, then activation."""
    y = x @ w[name + "/kernel"].astype(x.dtype)
    if bias:
        y = y + w[name + "/bias"].astype(x.dtype)
    return act(y) if act is not None else y


def conv1d_same(x, kernel, bias):
    """TF-sem tf.layers.conv1d(padding='same', stride 1): cross-correlation with
    pad_left=(k-1)//2, pad_right=k-1-pad_left (even k pads one more on the right)."""
    B, T, Cin = x.shape
    k, _, Cout = kernel.shape
    pl = (k - 1) // 2
    pr = k - 1 - pl
    xp = np.zeros((B, T + k - 1, Cin), dtype=x.dtype)
    xp[:, pl:pl + T] = x
    y = np.zeros((B, T, Cout), dtype=x.dtype)
    for j in range(k):
        y += xp[:, j:j + T] @ kernel[j].astype(x.dtype)
    return y + bias.astype(x.dtype)


def batch_norm_infer(y, w, name):
    """TF-sem batch_normalization(training=False): gamma*(y-mean)/sqrt(var+eps)+beta."""
    g = w[name + "/gamma"].astype(y.dtype)
    b = w[name + "/beta"].astype(y.dtype)
    m = w[name + "/moving_mean"].astype(y.dtype)
    v = w[name + "/moving_variance"].astype(y.dtype)
    return g * (y - m) / np.sqrt(v + BN_EPS) + b


BN_MOMENTUM = 0.99


def batch_norm_train(y, w, name, updates):
    """TF-sem batch_normalization(training=True) on [B,T,C] (modules.py:131 with is_training): normalise with
    the batch mean and BIASED batch variance over (B,T) -- padded frames included, nothing is masked -- and
    record the moving-average update moving <- moving*0.99 + batch*0.01 that tacotron.py:334 makes the
    optimizer depend on (UPDATE_OPS).  Assumption (unpinned): the non-fused 3-D path, whose
    moving variance also takes the biased batch variance."""
    g = w[name + "/gamma"].astype(y.dtype)
    b = w[name + "/beta"].astype(y.dtype)
    mu = y.mean(axis=(0, 1))
    var = ((y - mu) ** 2).mean(axis=(0, 1))
    if updates is not None:
        updates[name + "/moving_mean"] = w[name + "/moving_mean"].astype(y.dtype) * BN_MOMENTUM + mu * (1 - BN_MOMENTUM)
        updates[name + "/moving_variance"] = w[name + "/moving_variance"].astype(y.dtype) * BN_MOMENTUM + var * (1 - BN_MOMENTUM)
    return g * (y - mu) / np.sqrt(var + BN_EPS) + b


def conv1d_bn(x, w, name, act, bn_updates=None, training=False):
    """modules.py:123-131: conv1d -> activation -> batch_normalization (BN *after* the activation)."""
    y = conv1d_same(x, w[name + "/kernel"], w[name + "/bias"])
    if act is not None:
        y = act(y)
    if training:
        return batch_norm_train(y, w, name, bn_updates)
    return batch_norm_infer(y, w, name)


def maxpool_same_stride1(x, width):
    """TF-sem max_pooling1d(pool=width, strides=1, 'same') (modules.py:47-51): pads
    (width-1)//2 left, rest right, padding never wins the max."""
    if width == 1:
        return x
    B, T, C = x.shape
    pl = (width - 1) // 2
    xp = np.full((B, T + width - 1, C), -np.inf, dtype=x.dtype)
    xp[:, pl:pl + T] = x
    y = xp[:, 0:T].copy()
    for j in range(1, width):
        y = np.maximum(y, xp[:, j:j + T])
    return y


def prenet(x, w, scope, sizes):
    """modules.py:18-25 at inference: relu(dense) per layer, dropout rate 0."""
    for i in range(len(sizes)):
        x = dense(x, w, "%s/dense_%d" % (scope, i + 1), relu)
    return x


def highwaynet(x, w, name):
    """modules.py:105-120: H=relu(dense_H), T=sigmoid(dense_T); H*T + x*(1-T)."""
    H = dense(x, w, name + "/H", relu)
    T = dense(x, w, name + "/T", sigmoid)
    return H * T + x * (1.0 - T)


def gru_cell(x, h, w, name):
    """TF-sem tf.contrib.rnn.GRUCell: [r,u]=sigmoid([x,h].Wg+bg);
    c=tanh([x, r*h].Wc+bc); h'=u*h+(1-u)*c."""
    n = h.shape[-1]
    xh = np.concatenate([x, h], axis=-1)
    g = sigmoid(xh @ w[name + "/gates/kernel"].astype(x.dtype) + w[name + "/gates/bias"].astype(x.dtype))
    r, u = g[..., :n], g[..., n:]
    xrh = np.concatenate([x, r * h], axis=-1)
    c = np.tanh(xrh @ w[name + "/candidate/kernel"].astype(x.dtype) + w[name + "/candidate/bias"].astype(x.dtype))
    return u * h + (1.0 - u) * c


def dynamic_gru(x, lengths, w, name, h0=None):
    """TF-sem dynamic_rnn with sequence_length: for t >= L_b emit zeros and copy
    the state through."""
    B, T, _ = x.shape
    n = w[name + "/candidate/bias"].shape[0]
    h = np.zeros((B, n), dtype=x.dtype) if h0 is None else h0.astype(x.dtype).copy()
    out = np.zeros((B, T, n), dtype=x.dtype)
    for t in range(T):
        hn = gru_cell(x[:, t], h, w, name)
        if lengths is None:
            h = hn
            out[:, t] = hn
        else:
            act = (t < lengths)[:, None]
            h = np.where(act, hn, h)
            out[:, t] = np.where(act, hn, 0)
    return out


def reverse_sequence(x, lengths):
    """TF-sem tf.reverse_sequence: reverse the first L_b steps of row b, leave the tail."""
    y = x.copy()
    for b in range(x.shape[0]):
        L = x.shape[1] if lengths is None else int(lengths[b])
        y[b, :L] = x[b, :L][::-1]
    return y


def bidirectional_gru(x, lengths, w, scope, init_state=None):
    """modules.py:82-96 -> TF-sem bidirectional_dynamic_rnn; concat(fw, bw) on channels."""
    h0f = h0b = None
    if init_state is not None:
        n = init_state.shape[1] // 2
        h0f, h0b = init_state[:, :n], init_state[:, n:]                    # modules.py:83-84 tf.split
    fw = dynamic_gru(x, lengths, w, scope + "/fw", h0f)
    bw = dynamic_gru(reverse_sequence(x, lengths), lengths, w, scope + "/bw", h0b)
    bw = reverse_sequence(bw, lengths)
    return np.concatenate([fw, bw], axis=-1)


def cbhg(x, lengths, w, scope, K, maxpool_width, depth, projs, before_highway=None, rnn_init=None,
         taps=None, training=False, bn_updates=None):
    """modules.py:27-96."""
    bank = np.concatenate([conv1d_bn(x, w, "%s/conv_bank/conv1d_%d" % (scope, k), relu, bn_updates, training)
                           for k in range(1, K + 1)], axis=-1)             # :35-44
    mp = maxpool_same_stride1(bank, maxpool_width)                         # :47-51
    p = mp
    for i in range(len(projs)):
        act = None if i == len(projs) - 1 else relu                        # :55
        p = conv1d_bn(p, w, "%s/proj_%d" % (scope, i + 1), act, bn_updates, training)
    hi = p + x                                                             # :62-69
    if before_highway is not None:
        hi = hi + before_highway[:, None, :]
    if (scope + "/dense/kernel") in w:                                     # :72-73
        hi = dense(hi, w, scope + "/dense")
    for i in range(depth):                                                 # :76-77
        hi = highwaynet(hi, w, "%s/highway_%d" % (scope, i + 1))
    out = bidirectional_gru(hi, lengths, w, scope + "/bigru", rnn_init)    # :82-96
    if taps is not None:
        taps[scope + ":bank"] = bank
        taps[scope + ":maxpool"] = mp
        taps[scope + ":proj"] = p
        taps[scope + ":highway"] = hi
    return out


# ----------------------------------------------------------------------------------------
# attention (TF-sem tf.contrib.seq2seq)
# ----------------------------------------------------------------------------------------

def attention_score(q, keys, w, attention_type):
    """_bahdanau_score: sum_a v[a]*tanh(keys+q) ; normalize=True: v_hat=g*v/|v|, +b inside tanh."""
    v = w["attention/attention_v"].astype(q.dtype)
    if attention_type == "bah_norm":
        g = w["attention/attention_g"].astype(q.dtype)
        b = w["attention/attention_b"].astype(q.dtype)
        nv = g * v / np.sqrt(np.sum(v * v))
        return np.sum(nv * np.tanh(keys + q[:, None, :] + b), axis=2)
    return np.sum(v * np.tanh(keys + q[:, None, :]), axis=2)


def softmax_rows(e):
    m = e.max(axis=1, keepdims=True)
    p = np.exp(e - m)
    return p / p.sum(axis=1, keepdims=True)


def monotonic_attention_parallel(p, prev):
    """monotonic_attention(mode='parallel'): cp=safe_cumprod(1-p, exclusive);
    alpha = p*cp*cumsum(prev/clip(cp,1e-10,1))."""
    tiny = np.finfo(np.float32).tiny    # TF runs in float32: tiny of the tensor dtype
    logs = np.log(np.clip(1.0 - p, tiny, 1.0))
    excl = np.cumsum(logs, axis=1) - logs
    cp = np.exp(excl)
    return p * cp * np.cumsum(prev / np.clip(cp, 1e-10, 1.0), axis=1)


def attention_alignments(q, keys, prev, w, attention_type):
    e = attention_score(q, keys, w, attention_type)
    if attention_type == "bah_mon":
        e = e + w["attention/attention_score_bias"].astype(q.dtype)
        return monotonic_attention_parallel(sigmoid(e), prev)
    return softmax_rows(e)


def initial_alignments(B, T_in, attention_type, dtype):
    a = np.zeros((B, T_in), dtype=dtype)
    if attention_type == "bah_mon":
        a[:, 0] = 1.0                      # TF-sem BahdanauMonotonicAttention.initial_alignments: one_hot(0)
    return a


# ----------------------------------------------------------------------------------------
# the whole forward: models/tacotron.py:21-251 at inference (is_training False)
# ----------------------------------------------------------------------------------------

def forward(w: Dict[str, np.ndarray], hp: OracleHParams, inputs, input_lengths, speaker_id=None,
            num_speakers: int = 1, n_steps: Optional[int] = None, manual_alignments=None,
            dtype=np.float64, taps: Optional[dict] = None, honor_stop: bool = True,
            teacher_frames=None, training: bool = False, bn_updates: Optional[dict] = None):
    """Returns dict(mel [B,n*r,M], linear [B,n*r,F], alignments [B,T_in,n], stop_step).

    `n_steps` = max_iters (tacotron.py:210).  `manual_alignments` [B,T_dec,T_in] switches on the
    manual override of rnn_wrappers.py:313-317.  `teacher_frames` [B,n,M] (optional) replaces the
    fed-back frame at step t>=1 with teacher_frames[:,t-1] -- the TacoTrainingHelper input rule
    (helpers.py:44,66).

    `training=True` is the is_training graph of tacotron.py:26: batch-normalisation with batch statistics
    (moving-average updates returned in `bn_updates`), nothing else changes -- modules.py:24 calls
    tf.layers.dropout WITHOUT training=True, and tf.layers.dropout defaults to training=False, so the prenet
    dropout is the identity in the reference even while training (drop_rate is computed and unused).  A
    training caller passes teacher_frames = mel_targets[:, r-1::r] and n_steps = T_out / r (helpers.py:44-48)
    and honor_stop=False (TacoTrainingHelper finishes on step count only, :62)."""
    inputs = np.asarray(inputs)
    B, T_in = inputs.shape
    r, M = hp.reduction_factor, hp.num_mels
    n = hp.max_iters if n_steps is None else n_steps
    lengths = np.asarray(input_lengths).astype(np.int64)
    f = lambda a: np.asarray(a).astype(dtype)

    x = f(w["embedding"])[inputs]                                          # tacotron.py:34-39
    spk_embed = before_highway = enc_init = att_init = None
    dec_inits = None
    if num_speakers > 1:                                                   # tacotron.py:41-94
        sid = np.zeros(B, np.int64) if speaker_id is None else np.asarray(speaker_id).astype(np.int64)
        if hp.speaker_embedding_size != 1:
            spk_embed = f(w["speaker_embedding"])[sid]
        if hp.model_type == "deepvoice":
            names = ["before_highway", "encoder_rnn_init", "attention_rnn_init"] + \
                    ["decoder_rnn_init_%d" % (i + 1) for i in range(hp.dec_layer_num)]
            vecs = []
            for nm in names:
                if hp.speaker_embedding_size == 1:
                    vecs.append(f(w["spk/%s/table" % nm])[sid])            # :52-66
                else:
                    vecs.append(dense(spk_embed, w, "spk/" + nm, softsign))  # :68-79
            before_highway, enc_init, att_init = vecs[0], vecs[1], vecs[2]
            dec_inits = vecs[3:]
            spk_embed = None                                               # :81
        elif hp.model_type != "simple":
            raise Exception(" [!] Unkown multi-speaker model type: {}".format(hp.model_type))  # :88

    pre = prenet(x, w, "prenet", hp.enc_prenet_sizes)                      # tacotron.py:101-103
    enc = cbhg(pre, lengths, w, "encoder_cbhg", hp.enc_bank_size, hp.enc_maxpool_width,
               hp.enc_highway_depth, hp.enc_proj_sizes, before_highway, enc_init, taps,
               training=training, bn_updates=bn_updates)                   # :105-112
    if hp.attention_type not in ("bah", "bah_norm", "bah_mon"):
        raise Exception(" [!] Unkown attention type: {}".format(hp.attention_type))     # :152
    values = enc                                                           # no memory_sequence_length (A.8)
    keys = dense(values, w, "attention/memory_layer", bias=False)

    h_att = np.zeros((B, hp.attention_state_size), dtype) if att_init is None else att_init.copy()
    hs = [np.zeros((B, hp.dec_rnn_size), dtype) if dec_inits is None else dec_inits[i].copy()
          for i in range(hp.dec_layer_num)]
    ctx = np.zeros((B, enc.shape[-1]), dtype)                              # rnn_wrappers.py:206-209
    alpha = initial_alignments(B, T_in, hp.attention_type, dtype)
    frame = np.zeros((B, M), dtype)                                        # helpers.py:70-72
    Y = np.zeros((B, n, M * r), dtype)
    hist = np.zeros((B, T_in, n), dtype)
    finished = np.zeros(B, bool)
    stop_step = n
    steps = []
    for t in range(n):
        z = prenet(np.concatenate([frame, ctx], axis=-1), w, "decoder/prenet", hp.dec_prenet_sizes)  # rnn_wrappers.py:249,367-370
        if spk_embed is not None:
            z = np.concatenate([z, spk_embed], axis=-1)                    # :372-376
        h_att = gru_cell(z, h_att, w, "decoder/attention_gru")             # :251
        q = dense(h_att, w, "attention/query_layer", bias=False)
        a_new = attention_alignments(q, keys, alpha, w, hp.attention_type)  # :308-309
        if manual_alignments is not None:
            a_new = f(manual_alignments)[:, t, :]                          # :313-317
        alpha = a_new
        ctx = np.einsum("bj,bjd->bd", alpha, values)                       # :322-334
        hist[:, :, t] = alpha                                              # :284-285, tacotron.py:238-239
        cat = [h_att, ctx] + ([spk_embed] if spk_embed is not None else [])
        o = dense(np.concatenate(cat, axis=-1), w, "decoder/concat_projection")  # tacotron.py:166-170
        for i in range(hp.dec_layer_num):
            hs[i] = gru_cell(o, hs[i], w, "decoder/gru_%d" % (i + 1))      # :171-172 ResidualWrapper
            o = hs[i] + o
        y = dense(o, w, "decoder/frame_projection")                        # :178-179
        Y[:, t] = y
        if taps is not None:
            steps.append(dict(h_att=h_att.copy(), ctx=ctx.copy(), alpha=alpha.copy(),
                              h=[h.copy() for h in hs], y=y.copy()))
        frame = y[:, -M:]                                                  # helpers.py:31
        if teacher_frames is not None:
            frame = f(teacher_frames)[:, t]                                # helpers.py:44,66
        finished = finished | np.all(y == 0, axis=1)                       # helpers.py:29
        if honor_stop and finished.all():                                  # TF-sem dynamic_decode loop cond
            stop_step = t + 1
            break
    n_eff = stop_step
    mel = Y[:, :n_eff].reshape(B, n_eff * r, M)                            # tacotron.py:213-214
    post = cbhg(mel, None, w, "post_cbhg", hp.post_bank_size, hp.post_maxpool_width,
                hp.post_highway_depth, hp.post_proj_sizes, taps=taps,
                training=training, bn_updates=bn_updates)                  # :219-224
    if spk_embed is not None:                                              # :226-233 ('simple')
        tiled = np.broadcast_to(spk_embed[:, None, :], (B, post.shape[1], spk_embed.shape[1]))
        post = np.concatenate([tiled, post], axis=-1)
    linear = dense(post, w, "linear")                                      # :235
    out = dict(mel=mel, linear=linear, alignments=hist[:, :, :n_eff], stop_step=n_eff)
    if taps is not None:
        taps.update(prenet=pre, encoder=enc, keys=keys, steps=steps, post=post)
    return out


# ----------------------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------------------

def synthetic_inputs(B, T_in, seed, ragged=False, num_symbols=NUM_SYMBOLS):
    """ids uniform in [2,num_symbols), EOS(1) at T_in-1 (fixed-length) or at len_b~U[T_in/2,T_in)
    with PAD(0) after it (ragged).  input_lengths = argmax(ids==1) (synthesizer.py:120)."""
    rs = np.random.RandomState(seed)
    ids = rs.randint(2, num_symbols, size=(B, T_in)).astype(np.int32)
    if ragged:
        for b in range(B):
            L = int(rs.randint(max(T_in // 2, 1), T_in))
            ids[b, L] = 1
            ids[b, L + 1:] = 0
    else:
        ids[:, T_in - 1] = 1
    lengths = np.argmax(ids == 1, axis=1).astype(np.int32)
    return ids, lengths


CONFIGS = {
    # name: (B, T_in, r, max_iters, num_speakers, model_type)   SURVEY section 8 config names
    "C1": (1, 64, 5, 200, 1, "single"),
    "C2": (32, 128, 4, 128, 1, "single"),
    "C3": (32, 128, 4, 128, 4, "deepvoice"),
    "C5": (8, 512, 4, 1000, 1, "single"),
}


def algorithmic_bytes(hp: OracleHParams, B, T_in, n, num_speakers=1):
    """SURVEY 8(d) streaming model: weights of the feed-forward stages once, decoder weights and
    attention keys/values once per decoder step, activations in/out once.  fp32."""
    shapes = weight_shapes(hp, num_speakers)
    cnt = lambda pred: sum(int(np.prod(s)) for k, s in shapes.items() if pred(k))
    W_dec = cnt(lambda k: k.startswith("decoder/") or k.startswith("attention/query") or
                k.startswith("attention/attention_"))
    W_post = cnt(lambda k: k.startswith("post_cbhg/") or k.startswith("linear/"))
    W_enc = cnt(lambda k: True) - W_dec - W_post
    A, D = hp.attention_size, 2 * hp.enc_rnn_size
    M, F, r = hp.num_mels, hp.num_freq, hp.reduction_factor
    LD = hp.dec_layer_num * hp.dec_rnn_size
    per_step = W_dec + B * T_in * (A + D) + B * (2 * (M + D + hp.attention_state_size + LD) + M * r + 3 * T_in)
    total = W_enc + W_post + n * per_step + B * T_in * (1 + D + A) + B * n * r * (M + F)
    return 4 * total, 4 * per_step


# ----------------------------------------------------------------------------------------
# post-processing next to the path: attention-based trimming, synthesizer.py:242-262
# ----------------------------------------------------------------------------------------

def attention_trim_end(alignment, sequence_len, reduction_factor):
    """alignment [T_in, T_dec] of one utterance -> spec_end_idx = r*jdx + 3 (the `attention_trim and end_of_sentence` branch)."""
    end_idx_counter = 0
    attention_argmax = alignment.argmax(0)
    end_idx = min(sequence_len - 1, max(attention_argmax))
    max_counter = min((attention_argmax == end_idx).sum(), 5)
    jdx = 0
    for jdx, attend_idx in enumerate(attention_argmax):
        if len(attention_argmax) > jdx + 1:
            if attend_idx == end_idx:
                end_idx_counter += 1
            if attend_idx == end_idx and attention_argmax[jdx + 1] > end_idx:
                break
            if end_idx_counter >= max_counter:
                break
        else:
            break
    return reduction_factor * jdx + 3


# ----------------------------------------------------------------------------------------
# training-side pieces that do not need a backward pass (tacotron.py:274-336)
# ----------------------------------------------------------------------------------------

def add_loss(mel_out, mel_tgt, lin_out, lin_tgt, loss_coeff, prioritize_loss=False, sample_rate=24000):
    """tacotron.py:274-302.  Returns dict(loss, mel_loss, linear_loss, loss_without_coeff)."""
    mel_l = np.abs(mel_tgt - mel_out)
    l1 = np.abs(lin_tgt - lin_out)
    c = np.asarray(loss_coeff, dtype=mel_l.dtype)[:, None, None]
    F = lin_out.shape[-1]
    if prioritize_loss:
        up = int(5000 / (sample_rate * 0.5) * F)
        lo = int(165 / (sample_rate * 0.5) * F)
        pr = l1[:, :, lo:up]
        loss = np.mean(mel_l * c) + 0.5 * np.mean(l1 * c) + 0.5 * np.mean(pr * c)
        lin_loss = 0.5 * (np.mean(l1) + np.mean(pr))
    else:
        loss = np.mean(mel_l * c) + np.mean(l1 * c)
        lin_loss = np.mean(l1)
    mel_loss = np.mean(mel_l)
    return dict(loss=loss, mel_loss=mel_loss, linear_loss=lin_loss, loss_without_coeff=mel_loss + lin_loss)


def learning_rate(global_step, initial_learning_rate=0.002, decay_learning_rate_mode=0, is_randomly_initialized=True):
    """tacotron.py:313-325 (step = global_step + 1)."""
    step = float(global_step + 1)
    if decay_learning_rate_mode == 0:
        w = 4000.0 if is_randomly_initialized else 40000.0
        return initial_learning_rate * w ** 0.5 * min(step * w ** -1.5, step ** -0.5)
    return initial_learning_rate * 0.95 ** (step / 3000.0)       # TF-sem exponential_decay(1., step, 3000, 0.95)


def adam_clip_step(params, grads, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8, clip_norm=1.0):
    """tacotron.py:327-336: clip_by_global_norm(1.0) then tf.train.AdamOptimizer (A.14; epsilon outside
    the bias-corrected form).  t = number of this update (1-based).  Flat float arrays; returns
    (params, m, v, global_norm)."""
    gn = np.sqrt(np.sum(np.square(grads.astype(np.float64))))
    g = grads * (clip_norm / max(gn, clip_norm))               # TF-sem clip_by_global_norm
    lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    return params - lr_t * m / (np.sqrt(v) + eps), m, v, gn

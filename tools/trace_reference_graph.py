#!/usr/bin/env python
"""The WIRING of the reference's model, read off its own code: `models/tacotron.py` (initialize / add_loss / add_optimizer),
`models/modules.py`, `models/rnn_wrappers.py` and `models/helpers.py` are executed as they stand, in the build container, against a
stand-in for TensorFlow that computes NOTHING: every `tf.*` call is written down (operation, keyword arguments, which earlier results
it consumes, the variable scope it sits in, the static shape the operation's documented shape rule gives) and answered with a symbol.
TensorFlow's own library classes the reference instantiates or subclasses (GRUCell, MultiRNNCell, OutputProjectionWrapper,
ResidualWrapper, the attention mechanisms, BasicDecoder, dynamic_decode, RNNCell, Helper) are dispatchers here: they record their
constructor arguments and forward calls along the documented structure (a MultiRNNCell feeds cell i's output to cell i + 1; BasicDecoder's
step is cell -> helper.sample -> helper.next_inputs), so that the REFERENCE's own `call` / `next_inputs` methods run and are traced for
one decoder step.

What this pins: which operations the reference builds, in which order, with which arguments and on which tensors -- the claims the
oracle (oracle/taco_oracle.py) and the kernels were written from: BatchNorm after the activation, dropout without `training=`, the
attention memory without `memory_sequence_length`, `sequence_length` on the encoder's BiGRU only, `maximum_iterations`, the residual
and concat wrappers, the loss terms, clip-then-Adam.  What it does not pin: the arithmetic inside TensorFlow's operations (GRUCell's gate
formula, SAME padding, the monotonic attention's normaliser) -- that stays the oracle's restatement of the documented semantics.

    python tools/trace_reference_graph.py            # build container only -> tests/golden/graph_trace.json
"""
import collections
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import json
import os
import re
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("TACO_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")

TRACE = []          # the records of the current run
SCOPE = []          # variable / name scope stack


# ------------------------------------------------------------------------------------------------------------------------------
# symbols
# ------------------------------------------------------------------------------------------------------------------------------
class Dim(object):
    def __init__(self, v):
        self.value = v

    def __int__(self):
        return -1 if self.value is None else int(self.value)

    __index__ = __int__

    def __eq__(self, o):
        return self.value == (o.value if isinstance(o, Dim) else o)

    def __ne__(self, o):
        return not self.__eq__(o)

    def __hash__(self):
        return hash(self.value)

    def __repr__(self):
        return "?" if self.value is None else str(self.value)


class Shape(object):
    def __init__(self, dims):
        self.dims = None if dims is None else [d.value if isinstance(d, Dim) else d for d in dims]

    def __getitem__(self, i):
        if self.dims is None:
            return Dim(None)
        r = self.dims[i]
        return Shape(r) if isinstance(i, slice) else Dim(r)

    def __len__(self):
        return 0 if self.dims is None else len(self.dims)

    def __iter__(self):
        return iter([Dim(d) for d in (self.dims or [])])

    def as_list(self):
        return list(self.dims) if self.dims is not None else None

    def __eq__(self, o):
        return self.as_list() == (o.as_list() if isinstance(o, Shape) else o)

    def __repr__(self):
        return str(self.dims)


def _j(v):
    """a JSON-able description of an argument"""
    if isinstance(v, Sym):
        return {"sym": v.id}
    if isinstance(v, Named):
        return v.name
    if isinstance(v, (Shape, Dim)):
        return repr(v)
    if isinstance(v, (list, tuple)):
        return [_j(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _j(x) for k, x in v.items()}
    if isinstance(v, (int, float, str, bool)) or v is None:
        return v
    if isinstance(v, Recorded):
        return {"object": v._rid}
    if callable(v) and getattr(v, "__name__", "") == "<lambda>":
        return "<lambda>"
    return type(v).__name__


def _syms(v, out):
    if isinstance(v, Sym):
        out.append(v.id)
    elif isinstance(v, (list, tuple)):
        for x in v:
            _syms(x, out)
    elif isinstance(v, dict):
        for x in v.values():
            _syms(x, out)
    elif isinstance(v, tuple) and hasattr(v, "_fields"):
        for x in v:
            _syms(x, out)
    return out


def record(op, args=(), kwargs=None, shape=None, extra=None):
    rid = len(TRACE)
    rec = {"id": rid, "op": op, "scope": "/".join(SCOPE), "in": _syms([list(args), kwargs or {}], []),
           "args": [_j(a) for a in args], "kwargs": {k: _j(v) for k, v in (kwargs or {}).items()},
           "shape": None if shape is None else [d.value if isinstance(d, Dim) else d for d in shape]}
    if extra:
        rec.update(extra)
    TRACE.append(rec)
    return rid


class Sym(object):
    """the answer to a recorded operation"""
    def __init__(self, op, args=(), kwargs=None, shape=None, extra=None):
        self._shape = Shape(shape)
        self.id = record(op, args, kwargs, shape, extra)
        self.op = op
        self.name = op

    @property
    def shape(self):
        return self._shape

    def get_shape(self):
        return self._shape

    def _bin(self, op, o, rev=False):
        sh = self._shape.as_list()
        if isinstance(o, Sym) and o._shape.as_list() is not None and (sh is None or len(o._shape) > len(sh)):
            sh = o._shape.as_list()
        return Sym(op, (o, self) if rev else (self, o), shape=sh)

    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __pow__(self, o): return self._bin("pow", o)
    def __rpow__(self, o): return self._bin("pow", o, True)
    def __ge__(self, o): return self._bin("greater_equal", o)
    def __neg__(self): return Sym("neg", (self,), shape=self._shape.as_list())

    def __getitem__(self, idx):
        idx_t = idx if isinstance(idx, tuple) else (idx,)
        desc = []
        for i in idx_t:
            if isinstance(i, slice):
                desc.append("%s:%s:%s" % tuple("" if x is None else (_j(x) if isinstance(x, Sym) else x) for x in (i.start, i.stop, i.step)))
            else:
                desc.append(_j(i))
        sh = self._shape.as_list()
        out = None
        if sh is not None:
            out = []
            for k, d in enumerate(sh):
                i = idx_t[k] if k < len(idx_t) else slice(None)
                if isinstance(i, slice):
                    out.append(d if (i.start is None and i.stop is None and i.step is None) else None)
            # integer / symbol indices drop their dimension
        return Sym("getitem", (self,), {"index": desc}, shape=out)

    def __iter__(self):
        raise TypeError("a symbol is not iterable")

    def __bool__(self):
        raise TypeError("the reference branched on a tensor value: op %s" % self.op)

    def __hash__(self):
        return id(self)

    def __eq__(self, o):
        return self is o


class Named(object):
    """tf.nn.relu, tf.float32, initialisers ...: things that are passed around by name"""
    def __init__(self, name):
        self.name = name
        self.__name__ = name

    def __call__(self, *a, **k):
        if self.name.startswith("tf.nn.") or self.name in ("tf.tanh", "tf.sigmoid"):      # an activation applied directly
            x = a[0]
            return Sym(self.name, a, k, shape=x.shape.as_list() if isinstance(x, Sym) else None)
        return Named("%s(%s)" % (self.name, ", ".join([repr(_j(x)) for x in a] + ["%s=%r" % (kk, _j(v)) for kk, v in sorted(k.items())])))

    def __repr__(self):
        return self.name


class RecMeta(type):
    """construction of a dispatcher class -- or of a class of the REFERENCE derived from one (its AttentionWrapper, DecoderPrenetWrapper,
    ConcatOutputAndAttentionWrapper, TacoTestHelper, TacoTrainingHelper) -- is recorded with its arguments before __init__ runs"""
    def __call__(cls, *a, **k):
        obj = cls.__new__(cls)
        obj._rid = record("new " + cls.__name__, a, k)
        obj.__init__(*a, **k)
        return obj


class Recorded(object, metaclass=RecMeta):
    def _rec_init(self, cls, args, kwargs):
        pass


class scope_cm(object):
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        SCOPE.append(str(self.name))
        return self

    def __exit__(self, *a):
        SCOPE.pop()
        return False


class null_cm(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ------------------------------------------------------------------------------------------------------------------------------
# shape rules (the documented ones; everything else answers with an unknown shape)
# ------------------------------------------------------------------------------------------------------------------------------
def _sh(x):
    return x.shape.as_list() if isinstance(x, Sym) else None


def _last(x, n):
    s = _sh(x)
    return None if s is None else s[:-1] + [n]


def op_dense(inputs, units=None, **kw):
    return Sym("tf.layers.dense", (inputs,), dict(units=units, **kw), shape=_last(inputs, units))


def op_conv1d(inputs, filters=None, kernel_size=None, **kw):
    return Sym("tf.layers.conv1d", (inputs,), dict(filters=filters, kernel_size=kernel_size, **kw), shape=_last(inputs, filters))


def op_same(name):
    def f(inputs, *a, **kw):
        return Sym(name, (inputs,) + a, kw, shape=_sh(inputs))
    return f


def op_concat(values, axis=None, name=None, **kw):
    shapes = [_sh(v) for v in values]
    out = None
    if all(s is not None for s in shapes) and shapes:
        out = list(shapes[0])
        ax = axis if axis >= 0 else len(out) + axis
        tot = 0
        for s in shapes:
            tot = None if (tot is None or s[ax] is None) else tot + s[ax]
        out[ax] = tot
    return Sym("tf.concat", (list(values),), dict(axis=axis, name=name), shape=out)


class ShapeVec(object):
    """tf.shape(x): indexable, every entry a scalar symbol"""
    def __init__(self, x):
        self.x = x
        self.sym = Sym("tf.shape", (x,))

    def __getitem__(self, i):
        return Sym("getitem", (self.sym,), {"index": [i]}, shape=[])


def op_embedding_lookup(table, ids, **kw):
    s = _sh(ids)
    return Sym("tf.nn.embedding_lookup", (table, ids), kw, shape=None if s is None else s + [_sh(table)[-1]])


def op_get_variable(name, shape=None, **kw):
    return Sym("tf.get_variable", (), dict(name=name, shape=shape, **kw), shape=shape)


def op_placeholder(dtype, shape=None, name=None):
    return Sym("tf.placeholder", (), dict(dtype=dtype, shape=shape, name=name), shape=None if shape is None else list(shape))


def op_expand_dims(x, axis, **kw):
    s = _sh(x)
    ax = axis[0] if isinstance(axis, (list, tuple)) else axis
    if s is not None:
        s = list(s)
        s.insert(ax if ax >= 0 else len(s) + 1 + ax, 1)
    return Sym("tf.expand_dims", (x, axis), kw, shape=s)


def op_tile(x, multiples, **kw):
    s = _sh(x)
    if s is None and isinstance(x, list):
        s = []
        y = x
        while isinstance(y, list):
            s.append(len(y))
            y = y[0]
    out = None
    if s is not None and len(s) == len(multiples):
        out = [(d * m if isinstance(m, int) and d is not None else None) for d, m in zip(s, multiples)]
    return Sym("tf.tile", (x, list(multiples)), kw, shape=out)


def op_reshape(x, shape, **kw):
    return Sym("tf.reshape", (x, list(shape)), kw, shape=[d if isinstance(d, int) and d >= 0 else None for d in shape])


def op_transpose(x, perm=None, **kw):
    s = _sh(x)
    return Sym("tf.transpose", (x,), dict(perm=perm, **kw), shape=None if s is None or perm is None else [s[p] for p in perm])


def op_split(x, num, axis=0, **kw):
    s = _sh(x)
    out = None
    if s is not None:
        out = list(s)
        out[axis] = None if out[axis] is None else out[axis] // num
    return [Sym("tf.split[%d]" % i, (x, num, axis), kw, shape=out) for i in range(num)]


def op_cond(pred, true_fn, false_fn, **kw):
    a, b = true_fn(), false_fn()
    return Sym("tf.cond", (pred, a, b), kw, shape=_sh(b))


def op_matmul(a, b, **kw):
    sa, sb = _sh(a), _sh(b)
    out = None
    if sa is not None and sb is not None:
        out = sa[:-1] + [sb[-1]]
    return Sym("tf.matmul", (a, b), kw, shape=out)


def op_squeeze(x, axis=None, **kw):
    s = _sh(x)
    if s is not None and axis is not None:
        ax = axis if isinstance(axis, (list, tuple)) else [axis]
        s = [d for i, d in enumerate(s) if i not in ax]
    return Sym("tf.squeeze", (x, axis), kw, shape=s)


def op_clip_by_global_norm(t_list, clip_norm, **kw):
    s = Sym("tf.clip_by_global_norm", (list(t_list), clip_norm), kw)
    return [Sym("clipped[%d]" % i, (s,)) for i in range(len(t_list))], Sym("global_norm", (s,))


class TensorArray(Recorded):
    def __init__(self, *a, **k):
        self._rec_init("tf.TensorArray", a, k)
        self.writes = []

    def write(self, index, value):
        Sym("TensorArray.write", (index, value), {"array": self._rid})
        self.writes.append(value)
        return self

    def stack(self):
        v = self.writes[-1] if self.writes else None
        s = _sh(v)
        return Sym("TensorArray.stack", (), {"array": self._rid}, shape=None if s is None else [None] + s)


class AdamOptimizer(Recorded):
    def __init__(self, *a, **k):
        self._rec_init("tf.train.AdamOptimizer", a, k)

    def compute_gradients(self, loss, **k):
        g = Sym("optimizer.compute_gradients", (loss,), k)
        return [(Sym("gradient[%d]" % i, (g,)), Sym("variable[%d]" % i, (g,))) for i in range(2)]

    def apply_gradients(self, grads_and_vars, **k):
        return Sym("optimizer.apply_gradients", (list(grads_and_vars),), k)


# ---- library classes as dispatchers -------------------------------------------------------------------------------------------
def snake(name):
    """TensorFlow's rule for the variable scope a Layer / RNNCell opens around its `call` (base_layer's to_snake_case of the class name)"""
    s1 = re.sub(r"(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    return re.sub(r"([a-z])([A-Z])", r"\1_\2", s1).lower()


class RNNCell(Recorded):
    SCOPED = True           # a cell that defines `call` is run by Layer.__call__ inside a scope named after its class

    def __init__(self, name=None, **k):
        self._name = name
        self._base_name = name or type(self).__name__

    @property
    def name(self):
        return self._base_name

    def __call__(self, inputs, state, scope=None):
        if not self.SCOPED:
            return self.call(inputs, state)
        with scope_cm(snake(type(self).__name__)):
            return self.call(inputs, state)

    def zero_state(self, batch_size, dtype):
        n = self.state_size
        return Sym("%s.zero_state" % type(self).__name__, (batch_size,), {"size": _j(n)}, shape=[None, n if isinstance(n, int) else None])


class GRUCell(RNNCell):
    def __init__(self, num_units, **k):
        RNNCell.__init__(self)
        self.num_units = num_units
        self._rec_init("GRUCell", (num_units,), k)

    state_size = property(lambda self: self.num_units)
    output_size = property(lambda self: self.num_units)

    def call(self, inputs, state):
        o = Sym("GRUCell.call", (inputs, state), {"cell": self._rid, "num_units": self.num_units}, shape=[None, self.num_units])
        return o, o


class MultiRNNCell(RNNCell):
    def __init__(self, cells, state_is_tuple=True):
        RNNCell.__init__(self)
        self._cells = list(cells)
        self._rec_init("MultiRNNCell", ([c for c in cells],), dict(state_is_tuple=state_is_tuple))

    state_size = property(lambda self: tuple(c.state_size for c in self._cells))
    output_size = property(lambda self: self._cells[-1].output_size)

    def zero_state(self, batch_size, dtype):
        return tuple(c.zero_state(batch_size, dtype) for c in self._cells)

    def call(self, inputs, state):
        cur, new = inputs, []
        for i, (c, s) in enumerate(zip(self._cells, state)):
            with scope_cm("cell_%d" % i):
                cur, ns = c(cur, s)
            new.append(ns)
        return cur, tuple(new)


class OutputProjectionWrapper(RNNCell):
    def __init__(self, cell, output_size, activation=None, **k):
        RNNCell.__init__(self)
        self._cell, self._n = cell, output_size
        self._rec_init("OutputProjectionWrapper", (cell, output_size), dict(activation=activation, **k))

    state_size = property(lambda self: self._cell.state_size)
    output_size = property(lambda self: self._n)

    def zero_state(self, batch_size, dtype):
        return self._cell.zero_state(batch_size, dtype)

    def call(self, inputs, state):
        o, ns = self._cell(inputs, state)
        return Sym("OutputProjectionWrapper.linear", (o,), {"wrapper": self._rid, "units": self._n}, shape=[None, self._n]), ns


class ResidualWrapper(RNNCell):
    SCOPED = False          # TF 1.x's ResidualWrapper overrides __call__ itself: no scope of its own

    def __init__(self, cell, **k):
        RNNCell.__init__(self)
        self._cell = cell
        self._rec_init("ResidualWrapper", (cell,), k)

    state_size = property(lambda self: self._cell.state_size)
    output_size = property(lambda self: self._cell.output_size)

    def zero_state(self, batch_size, dtype):
        return self._cell.zero_state(batch_size, dtype)

    def call(self, inputs, state):
        o, ns = self._cell(inputs, state)
        return Sym("ResidualWrapper.add", (inputs, o), {"wrapper": self._rid}, shape=_sh(o)), ns


class AttentionMechanism(Recorded):
    pass


class _Attention(AttentionMechanism):
    KIND = "attention"

    def __init__(self, num_units, memory, *a, **k):
        self._rec_init(self.KIND, (num_units, memory) + a, k)
        self.values = memory
        self.num_units = num_units
        self.batch_size = Sym("attention.batch_size", (memory,), shape=[])
        self.alignments_size = _sh(memory)[1] if _sh(memory) else None

    def initial_alignments(self, batch_size, dtype):
        return Sym("attention.initial_alignments", (batch_size,), {"mechanism": self._rid}, shape=[None, self.alignments_size])

    def __call__(self, query, previous_alignments=None, **k):
        with scope_cm(snake(type(self).__name__)):      # the query layer's variables live here
            return Sym("attention.__call__", (query, previous_alignments), dict(mechanism=self._rid, **k), shape=[None, self.alignments_size])


class BahdanauAttention(_Attention):
    KIND = "BahdanauAttention"


class BahdanauMonotonicAttention(_Attention):
    KIND = "BahdanauMonotonicAttention"


class _BaseAttentionMechanism(_Attention):
    KIND = "_BaseAttentionMechanism"


AttentionWrapperStateBase = collections.namedtuple("AttentionWrapperState", ("cell_state", "attention", "time", "alignments", "alignment_history"))


class AttentionWrapperState(AttentionWrapperStateBase):
    def clone(self, **kwargs):
        return self._replace(**kwargs)


class Helper(Recorded):
    pass


class BasicDecoder(Recorded):
    def __init__(self, cell, helper, initial_state, output_layer=None):
        self.cell, self.helper, self.initial_state = cell, helper, initial_state
        self._rec_init("BasicDecoder", (cell, type(helper).__name__), dict(output_layer=output_layer))


def dynamic_decode(decoder, maximum_iterations=None, **k):
    """records the call and traces ONE step along BasicDecoder's documented structure"""
    Sym("tf.contrib.seq2seq.dynamic_decode", (), dict(maximum_iterations=maximum_iterations, decoder=decoder._rid, **k))
    with scope_cm("decoder"):                      # dynamic_decode's own variable scope
        finished0, first_inputs = decoder.helper.initialize()
        time = Sym("time", (), shape=[])
        outputs, state = decoder.cell(first_inputs, decoder.initial_state)
        sample_ids = decoder.helper.sample(time=time, outputs=outputs, state=state)
        finished, next_inputs, next_state = decoder.helper.next_inputs(time=time, outputs=outputs, state=state, sample_ids=sample_ids)
        Sym("decoder_step.next_inputs", (next_inputs, finished))
        # what the loop carries: the symbols of the initial state / first input, and the symbols of this step that replace them in the next one
        flat = lambda st: [x.id if isinstance(x, Sym) else None for x in nest_flatten(st)]
        Sym("decoder_step.carry", (), {"initial_state": flat(decoder.initial_state), "next_state": flat(next_state), "first_inputs": first_inputs.id,
                                       "next_inputs": next_inputs.id, "time": time.id, "outputs": outputs.id,
                                       "initial_finished": finished0.id, "finished": finished.id})
    n = decoder.cell.output_size
    final_outputs = (Sym("dynamic_decode.rnn_output", (outputs,), shape=[None, None, n]), Sym("dynamic_decode.sample_id", (sample_ids,)))
    return final_outputs, next_state, Sym("dynamic_decode.sequence_lengths", ())


def bidirectional_dynamic_rnn(cell_fw, cell_bw, inputs, **k):
    s = _sh(inputs)
    mk = lambda nm, c: Sym("bidirectional_dynamic_rnn." + nm, (inputs,), dict(cell=c._rid, num_units=c.num_units, **k),
                           shape=None if s is None else s[:-1] + [c.num_units])
    of, ob = mk("output_fw", cell_fw), mk("output_bw", cell_bw)
    return (of, ob), (Sym("bidirectional_dynamic_rnn.state_fw", (of,)), Sym("bidirectional_dynamic_rnn.state_bw", (ob,)))


# ---- nest (structure utilities: real) -------------------------------------------------------------------------------------------
def nest_flatten(s):
    if isinstance(s, (list, tuple)):
        out = []
        for x in s:
            out.extend(nest_flatten(x))
        return out
    return [s]


def nest_map_structure(fn, s):
    if isinstance(s, tuple) and hasattr(s, "_fields"):
        return type(s)(*[nest_map_structure(fn, x) for x in s])
    if isinstance(s, (list, tuple)):
        return type(s)(nest_map_structure(fn, x) for x in s)
    return fn(s)


# ------------------------------------------------------------------------------------------------------------------------------
# the stand-in module tree
# ------------------------------------------------------------------------------------------------------------------------------
SPECIAL = {
    "layers.dense": op_dense, "layers.conv1d": op_conv1d, "layers.dropout": op_same("tf.layers.dropout"),
    "layers.batch_normalization": op_same("tf.layers.batch_normalization"), "layers.max_pooling1d": op_same("tf.layers.max_pooling1d"),
    "concat": op_concat, "shape": ShapeVec, "nn.embedding_lookup": op_embedding_lookup, "get_variable": op_get_variable,
    "placeholder": op_placeholder, "expand_dims": op_expand_dims, "tile": op_tile, "reshape": op_reshape, "transpose": op_transpose,
    "split": op_split, "cond": op_cond, "matmul": op_matmul, "squeeze": op_squeeze, "identity": op_same("tf.identity"),
    "abs": op_same("tf.abs"), "clip_by_global_norm": op_clip_by_global_norm, "TensorArray": TensorArray,
    "train.AdamOptimizer": AdamOptimizer, "variable_scope": scope_cm, "name_scope": lambda *a, **k: null_cm(),
    "control_dependencies": lambda *a, **k: null_cm(), "nn.bidirectional_dynamic_rnn": bidirectional_dynamic_rnn,
    "contrib.seq2seq.dynamic_decode": dynamic_decode, "contrib.seq2seq.BasicDecoder": BasicDecoder, "contrib.seq2seq.Helper": Helper,
    "contrib.seq2seq.BahdanauAttention": BahdanauAttention, "contrib.seq2seq.BahdanauMonotonicAttention": BahdanauMonotonicAttention,
    "contrib.rnn.GRUCell": GRUCell, "contrib.rnn.MultiRNNCell": MultiRNNCell, "contrib.rnn.OutputProjectionWrapper": OutputProjectionWrapper,
    "contrib.rnn.ResidualWrapper": ResidualWrapper, "contrib.rnn.RNNCell": RNNCell,
    "contrib.seq2seq.python.ops.attention_wrapper.BahdanauAttention": BahdanauAttention,
    "contrib.seq2seq.python.ops.attention_wrapper._BaseAttentionMechanism": _BaseAttentionMechanism,
    "contrib.seq2seq.python.ops.attention_wrapper.AttentionMechanism": AttentionMechanism,
    "contrib.seq2seq.python.ops.attention_wrapper.AttentionWrapperState": AttentionWrapperState,
    "contrib.seq2seq.python.ops.attention_wrapper.AttentionWrapper": RNNCell,
    "contrib.data.python.util.nest.flatten": nest_flatten, "contrib.data.python.util.nest.map_structure": nest_map_structure,
    "python.ops.rnn_cell_impl._zero_state_tensors":
        lambda size, batch, dtype: Sym("_zero_state_tensors", (batch,), {"size": _j(size)}, shape=[None, size if isinstance(size, int) else None]),
    "TensorShape": lambda dims: Shape(dims),
}
NAMED = ("nn.relu", "nn.sigmoid", "nn.softsign", "nn.tanh", "tanh", "sigmoid", "float32", "int32", "bool", "truncated_normal_initializer",
         "constant_initializer", "GraphKeys.UPDATE_OPS")


class TFNode(types.ModuleType):
    """`tensorflow` and everything below it: attribute access builds the dotted path; a call of an unknown path is recorded generically"""
    def __init__(self, path):
        types.ModuleType.__init__(self, "tensorflow" + ("." + path if path else ""))
        self.__path__ = []
        self._p = path

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        p = (self._p + "." if self._p else "") + name
        if p in SPECIAL:
            return SPECIAL[p]
        if p in NAMED:
            return Named("tf." + p)
        return TFNode(p)

    def __call__(self, *a, **k):
        return Sym("tf." + self._p, a, k)


class TFFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name == "tensorflow" or name.startswith("tensorflow."):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return TFNode(spec.name[len("tensorflow"):].lstrip("."))

    def exec_module(self, module):
        pass


def install():
    for k in [k for k in sys.modules if k == "tensorflow" or k.startswith("tensorflow.")]:
        del sys.modules[k]
    sys.meta_path.insert(0, TFFinder())


# ------------------------------------------------------------------------------------------------------------------------------
# the reference, by path
# ------------------------------------------------------------------------------------------------------------------------------
def load_reference_models():
    install()
    import tensorflow as tf
    class HP(object):
        def __init__(self, **kw):
            self._v = dict(kw)
            self.__dict__.update(kw)

        def values(self):
            return dict(self._v)
    SPECIAL["contrib.training.HParams"] = HP
    for name in ("hparams", "utils", "utils.infolog", "text", "text.symbols", "models", "models.modules", "models.helpers", "models.rnn_wrappers", "models.tacotron"):
        sys.modules.pop(name, None)
    spec = importlib.util.spec_from_file_location("hparams", os.path.join(REF, "hparams.py"))
    hpm = importlib.util.module_from_spec(spec); sys.modules["hparams"] = hpm; spec.loader.exec_module(hpm)
    u = types.ModuleType("utils"); u.__path__ = []; sys.modules["utils"] = u
    il = types.ModuleType("utils.infolog"); il.log = lambda *a, **k: None; sys.modules["utils.infolog"] = il
    # text.symbols: the reference's own table (text/korean.py:11-21), read from its korean.py loaded by path (its `jamo` import line stubbed)
    j = types.ModuleType("jamo"); j.hangul_to_jamo = j.h2j = j.j2h = None; sys.modules.setdefault("jamo", j)
    rt = types.ModuleType("reftext"); rt.__path__ = [os.path.join(REF, "text")]; sys.modules["reftext"] = rt
    K = importlib.import_module("reftext.korean")
    t = types.ModuleType("text"); t.__path__ = []; sys.modules["text"] = t
    ts = types.ModuleType("text.symbols"); ts.symbols = K.ALL_SYMBOLS; ts.PAD, ts.EOS = K.PAD, K.EOS; sys.modules["text.symbols"] = ts
    pkg = types.ModuleType("models"); pkg.__path__ = [os.path.join(REF, "models")]; sys.modules["models"] = pkg      # models/__init__.py is not executed
    T = importlib.import_module("models.tacotron")
    return T, hpm.hparams, tf


def run_config(model_type, attention_type, num_speakers, training, speaker_embedding_size=None, prioritize_loss=False, decay_mode=0,
               is_randomly_initialized=False, rnn_decoder_test_mode=False):
    del TRACE[:]
    del SCOPE[:]
    T, hp, tf = load_reference_models()
    hp.model_type, hp.attention_type = model_type, attention_type
    if speaker_embedding_size is not None:
        hp.speaker_embedding_size = speaker_embedding_size
    hp.prioritize_loss, hp.decay_learning_rate_mode = prioritize_loss, decay_mode
    m = T.Tacotron(hp)
    inputs = tf.placeholder(tf.int32, [None, None], "inputs")
    input_lengths = tf.placeholder(tf.int32, [None], "input_lengths")
    speaker_id = tf.placeholder(tf.int32, [None], "speaker_id") if num_speakers > 1 else None
    kw = {}
    if training:
        kw = dict(mel_targets=tf.placeholder(tf.float32, [None, None, hp.num_mels], "mel_targets"),
                  linear_targets=tf.placeholder(tf.float32, [None, None, hp.num_freq], "linear_targets"),
                  loss_coeff=tf.placeholder(tf.float32, [None], "loss_coeff"))
    if training:
        kw["is_randomly_initialized"] = is_randomly_initialized
        kw["rnn_decoder_test_mode"] = rnn_decoder_test_mode
    m.initialize(inputs, input_lengths, num_speakers, speaker_id, **kw)
    if training:
        m.add_loss()
        m.add_optimizer(tf.placeholder(tf.int32, [], "global_step"))
    outs = {"mel_outputs": m.mel_outputs.id, "linear_outputs": m.linear_outputs.id, "alignments": m.alignments.id}
    if training:
        for k in ("loss", "mel_loss", "linear_loss", "loss_without_coeff", "learning_rate"):
            outs[k] = getattr(m, k).id
    return {"config": dict(model_type=model_type, attention_type=attention_type, num_speakers=num_speakers, training=training,
                           speaker_embedding_size=hp.speaker_embedding_size, prioritize_loss=prioritize_loss, decay_learning_rate_mode=decay_mode,
                           is_randomly_initialized=is_randomly_initialized, rnn_decoder_test_mode=rnn_decoder_test_mode),
            "hparams": {k: v for k, v in hp.values().items() if isinstance(v, (int, float, str, bool, list))},
            "outputs": outs,
            "trace": [dict(r) for r in TRACE]}


def main():
    if not os.path.isdir(REF):
        sys.exit("no reference checkout at %s (this script runs in the build container only)" % REF)
    runs = [run_config("single", "bah_mon", 1, False), run_config("single", "bah_mon", 1, True), run_config("single", "bah", 1, False),
            run_config("single", "bah_norm", 1, False), run_config("deepvoice", "bah_mon", 3, False), run_config("deepvoice", "bah_mon", 3, False, 1),
            run_config("simple", "bah_mon", 3, False),
            run_config("single", "bah_mon", 1, True, prioritize_loss=True, decay_mode=1, is_randomly_initialized=True),
            run_config("single", "bah_mon", 1, True, rnn_decoder_test_mode=True)]
    os.makedirs(GOLD, exist_ok=True)
    with open(os.path.join(GOLD, "graph_trace.json"), "w") as f:
        json.dump({"generated_by": "tools/trace_reference_graph.py: models/tacotron.py, modules.py, rnn_wrappers.py, helpers.py of /root/reference "
                                   "executed against a recording stand-in for TensorFlow (no arithmetic)", "runs": runs}, f, separators=(",", ":"))
    for r in runs:
        print(r["config"], "%d records" % len(r["trace"]))


if __name__ == "__main__":
    main()

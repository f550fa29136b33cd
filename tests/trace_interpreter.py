"""Runs the reference's GRAPH -- tests/golden/graph_trace.json, the record of every tf.* call the reference's own model code makes
(tools/trace_reference_graph.py) -- numerically: the data flow, the argument order of every concat / add / matmul, which tensor
feeds what, the loop-carried state of the decoder all come from the trace; only the arithmetic of the individual operations is
supplied here, by the oracle's primitives (oracle/taco_oracle.py: dense, conv1d_same, batch_norm_infer, maxpool_same_stride1,
gru_cell, bidirectional_gru, attention_alignments).  If oracle.forward() and this interpreter agree, the oracle's WIRING is the
reference's, mechanically -- for every operand order and every state hand-over, not only for the facts test_reference_graph.py lists.

Test infrastructure only."""
import numpy as np

import taco_oracle as O

ACT = {None: None, "tf.nn.relu": O.relu, "tf.nn.sigmoid": O.sigmoid, "tf.nn.softsign": O.softsign}


def _tf_names_of(t, auto, atype):
    """TensorFlow variable-name stems of a record (scope + TF's per-layer names), used to find its tensors in the canonical weight dict"""
    sc, op, kw = t["scope"], t["op"], t["kwargs"]
    if op == "tf.layers.dense":
        name = kw.get("name")
        if name is None:
            n = auto.get(sc, 0)
            auto[sc] = n + 1
            name = "dense" if n == 0 else "dense_%d" % n
        return sc + "/" + name
    return sc


class Interpreter(object):
    def __init__(self, run, weights, canon_of_tf):
        """canon_of_tf: {tensorflow variable name (without the leading `model/`): canonical weight name}"""
        self.run, self.tr, self.w = run, run["trace"], weights
        self.atype = run["config"]["attention_type"]
        self.canon_of_tf = canon_of_tf
        self.v = {}
        self.auto = {}
        self.arrays = {}            # TensorArray id -> list
        self.keys = {}              # attention mechanism record id -> (keys, memory)

    # ---- helpers ----
    def canon(self, tf_stem, leaf):
        return self.canon_of_tf[tf_stem + "/" + leaf]

    def stem(self, tf_stem):
        """canonical layer name (without /kernel) of a TensorFlow layer scope"""
        c = self.canon(tf_stem, "kernel")
        return c[:-len("/kernel")]

    def val(self, a):
        if isinstance(a, dict) and "sym" in a:
            return self.v[a["sym"]]
        if isinstance(a, list):
            return [self.val(x) for x in a]
        return a

    def B(self):
        return self.feed["inputs"].shape[0]

    # ---- one record ----
    def eval(self, t):
        op, kw, args = t["op"], t["kwargs"], t["args"]
        A = lambda i: self.val(args[i])
        w = self.w
        if op == "tf.placeholder":
            return self.feed.get(kw["name"])
        if op == "tf.shape":
            x = A(0)
            return ("shape", None if x is None else x.shape)
        if op == "getitem":
            x = A(0)
            if isinstance(x, tuple) and x and x[0] == "shape":
                return None if x[1] is None else int(x[1][kw["index"][0]])
            if x is None:
                return None
            idx = []
            for d in kw["index"]:
                if isinstance(d, str):
                    a, b, c = (None if p == "" else int(p) for p in d.split(":"))
                    idx.append(slice(a, b, c))
                elif isinstance(d, dict):
                    idx.append(int(self.v[d["sym"]]))
                else:
                    idx.append(int(d))
            return x[tuple(idx)]
        if op == "tf.get_variable":
            name = (t["scope"] + "/" if t["scope"] else "") + kw["name"]
            return np.asarray(w[self.canon_of_tf[name]], np.float64)
        if op == "tf.nn.embedding_lookup":
            return np.asarray(A(0))[np.asarray(A(1))]
        if op == "tf.layers.dense":
            stem = self.stem(_tf_names_of(t, self.auto, self.atype))
            return O.dense(A(0), w, stem, ACT[kw.get("activation")])
        if op == "tf.layers.dropout":
            assert kw.get("training") is None               # never set: the identity
            return A(0)
        if op == "tf.layers.conv1d":
            stem = self.stem(t["scope"] + "/conv1d")
            y = O.conv1d_same(A(0), w[stem + "/kernel"], w[stem + "/bias"])
            act = ACT[kw.get("activation")]
            return act(y) if act is not None else y
        if op == "tf.layers.batch_normalization":
            c = self.canon(t["scope"] + "/batch_normalization", "gamma")
            if kw["training"]:
                return O.batch_norm_train(A(0), w, c[:-len("/gamma")], self.bn_updates)
            return O.batch_norm_infer(A(0), w, c[:-len("/gamma")])
        if op == "tf.layers.max_pooling1d":
            assert kw["strides"] == 1 and kw["padding"] == "same"
            return O.maxpool_same_stride1(A(0), kw["pool_size"])
        if op == "tf.concat":
            xs = [x for x in A(0)]
            return xs[0] if len(xs) == 1 else np.concatenate(xs, axis=kw["axis"])
        if op in ("add", "sub", "mul", "div", "pow", "greater_equal"):
            a, b = A(0), A(1)
            return {"add": lambda: a + b, "sub": lambda: a - b, "mul": lambda: a * b, "div": lambda: a / b, "pow": lambda: a ** b,
                    "greater_equal": lambda: a >= b}[op]()
        if op == "tf.expand_dims":
            ax = args[1][0] if isinstance(args[1], list) else args[1]
            return np.expand_dims(A(0), ax)
        if op == "tf.tile":
            x, mult = A(0), [int(m) for m in self.val(args[1])]
            return np.tile(np.asarray(x, dtype=np.float64 if np.asarray(x).dtype.kind == "f" else None), mult)
        if op.startswith("tf.split["):
            i = int(op[len("tf.split["):-1])
            return np.split(A(0), args[1], axis=args[2])[i]
        if op == "tf.identity":
            return A(0)
        if op == "tf.squeeze":
            return np.squeeze(A(0), axis=tuple(args[1]))
        if op == "tf.matmul":
            return np.matmul(A(0), A(1))
        if op == "tf.cond":
            pred = A(0)
            return A(1) if pred else A(2)
        if op == "tf.zeros":
            return 0
        if op == "tf.reshape":
            shp = [(-1 if s is None else int(s)) for s in self.val(args[1])]
            return np.reshape(A(0), shp)
        if op == "tf.transpose":
            return np.transpose(A(0), kw["perm"])
        if op == "tf.equal":
            return np.equal(A(0), A(1))
        if op == "tf.reduce_all":
            return np.all(A(0), axis=kw.get("axis"))
        if op == "tf.abs":
            return np.abs(A(0))
        if op == "tf.reduce_mean":
            assert not kw                                   # over every axis
            return np.mean(A(0))
        if op == "tf.cast":
            assert kw["dtype"] == "tf.float32"
            return float(A(0))
        if op == "tf.minimum":
            return min(A(0), A(1))
        if op == "tf.train.exponential_decay":              # TF-sem: lr * rate ** (step / decay_steps), staircase=False
            assert not kw
            return A(0) * A(3) ** (A(1) / A(2))
        if op in ("tf.assert_equal", "attention.batch_size", "time", "decoder_step.next_inputs", "decoder_step.carry",
                  "dynamic_decode.sample_id", "dynamic_decode.sequence_lengths", "tf.contrib.seq2seq.dynamic_decode"):
            return None
        if op.startswith("new "):
            if op in ("new BahdanauMonotonicAttention", "new BahdanauAttention"):
                memory = A(1)
                self.keys[t["id"]] = (O.dense(memory, w, "attention/memory_layer", bias=False), memory)
            if op == "new TensorArray":
                self.arrays[t["id"]] = []
            return None
        if op in ("GRUCell.zero_state", "_zero_state_tensors"):
            return np.zeros((self.B(), kw["size"]))
        if op == "attention.initial_alignments":
            return O.initial_alignments(self.B(), self.keys[kw["mechanism"]][1].shape[1], self.atype, np.float64)
        if op == "attention.__call__":
            keys, _ = self.keys[kw["mechanism"]]
            q = A(0) @ w[self.canon(t["scope"] + "/query_layer", "kernel")]
            return O.attention_alignments(q, keys, A(1), w, self.atype)
        if op == "GRUCell.call":
            return O.gru_cell(A(0), A(1), w, self.canon(t["scope"] + "/gates", "kernel")[:-len("/gates/kernel")])
        if op in ("bidirectional_dynamic_rnn.output_fw", "bidirectional_dynamic_rnn.output_bw"):
            x = A(0)
            L = self.val(kw["sequence_length"]) if kw.get("sequence_length") is not None else None
            h0f = self.val(kw["initial_state_fw"]) if kw.get("initial_state_fw") is not None else None
            h0b = self.val(kw["initial_state_bw"]) if kw.get("initial_state_bw") is not None else None
            scope = self.canon(t["scope"] + "/bidirectional_rnn/fw/gru_cell/gates", "kernel")[:-len("/fw/gates/kernel")]
            init = None if h0f is None else np.concatenate([h0f, h0b], axis=1)
            out = O.bidirectional_gru(x, L, w, scope, init)
            n = kw["num_units"]
            return out[..., :n] if op.endswith("fw") else out[..., n:]
        if op in ("bidirectional_dynamic_rnn.state_fw", "bidirectional_dynamic_rnn.state_bw"):
            return None
        if op == "OutputProjectionWrapper.linear":
            return O.dense(A(0), w, self.stem(t["scope"]))
        if op == "ResidualWrapper.add":
            return A(0) + A(1)
        if op == "TensorArray.write":
            self.arrays[kw["array"]].append(A(1))
            return None
        if op == "TensorArray.stack":
            return np.stack(self.arrays[kw["array"]], axis=0)
        raise NotImplementedError(op)

    # ---- the whole graph ----
    def forward(self, inputs, input_lengths, n_steps=None, speaker_id=None, manual_alignments=None, mel_targets=None, linear_targets=None,
                loss_coeff=None, global_step=None):
        """n_steps = maximum_iterations (None: the value the reference passes to dynamic_decode).  The decoder loop is dynamic_decode's
        (TF-sem): run BasicDecoder's step until every row is finished -- the helper's `finished`, OR-ed over the steps -- or the step
        count reaches maximum_iterations."""
        self.feed = {"inputs": np.asarray(inputs), "input_lengths": np.asarray(input_lengths), "speaker_id": None if speaker_id is None else np.asarray(speaker_id),
                     "is_manual_attention": manual_alignments is not None,
                     "manual_alignments": None if manual_alignments is None else np.asarray(manual_alignments, np.float64),
                     "mel_targets": mel_targets, "linear_targets": linear_targets, "loss_coeff": loss_coeff, "global_step": global_step}
        self.bn_updates = {}
        tr = self.tr
        ddr = [t for t in tr if t["op"] == "tf.contrib.seq2seq.dynamic_decode"][0]
        dd = ddr["id"]
        max_iter = ddr["kwargs"]["maximum_iterations"] if n_steps is None else n_steps
        carry = [t for t in tr if t["op"] == "decoder_step.carry"][0]
        ck = carry["kwargs"]
        out_rec = [t for t in tr if t["op"] == "dynamic_decode.rnn_output"][0]["id"]
        step_ids = list(range(dd + 1, out_rec))
        for t in tr[:dd + 1]:
            self.v[t["id"]] = self.eval(t)
        # the loop: BasicDecoder's step; the carried symbols take the previous step's values
        init_ids = [i for i in ck["initial_state"] if i is not None] + [ck["first_inputs"]]
        next_ids = [i for i in ck["next_state"] if i is not None] + [ck["next_inputs"]]
        outs = []
        finished = None
        step = 0
        while step < max_iter:
            prev = dict(self.v)
            for i in step_ids:
                t = tr[i]
                if step > 0 and i in init_ids:
                    self.v[i] = prev[next_ids[init_ids.index(i)]]
                elif i == ck["time"]:
                    self.v[i] = step
                else:
                    self.v[i] = self.eval(t)
            for a, b in zip(init_ids, next_ids):     # carried symbols created BEFORE the loop (zero states, state.time)
                if a <= dd:
                    self.v[a] = self.v[b]
            outs.append(self.v[ck["outputs"]])
            if finished is None:
                finished = np.asarray(self.v[ck["initial_finished"]], bool)
            finished = finished | np.asarray(self.v[ck["finished"]], bool)
            step += 1
            if finished.all():
                break
        self.n_steps = step
        self.v[out_rec] = np.stack(outs, axis=1)
        for t in tr[out_rec + 1:]:
            if t["op"].startswith("new ") and "Optimizer" in t["op"]:
                break
            self.v[t["id"]] = self.eval(t)
        o = self.run["outputs"]
        res = {"mel": self.v[o["mel_outputs"]], "linear": self.v[o["linear_outputs"]], "alignments": self.v[o["alignments"]], "n_steps": step,
               "bn_updates": self.bn_updates}
        for k in ("loss", "mel_loss", "linear_loss", "loss_without_coeff", "learning_rate"):
            if k in o and self.v.get(o[k]) is not None:
                res[k] = self.v[o[k]]
        return res

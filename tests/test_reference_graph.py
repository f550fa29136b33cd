"""The wiring of the reference's model, as its OWN code builds it (tests/golden/graph_trace.json: models/tacotron.py, modules.py,
rnn_wrappers.py, helpers.py executed against a recording stand-in for TensorFlow by tools/trace_reference_graph.py -- every tf.* call
with its arguments, inputs, scope and static shape; no arithmetic), held against what the oracle and the kernels were written from:

  * every layer the reference creates, with its scope, fan-in and fan-out == a tensor of the product's weight list (names through
    weights.TF_SCOPE_MAP's reference-controlled part, shapes exactly);
  * the claims of DESIGN.md section 1 "reference behaviours reproduced on purpose": BatchNorm after the activation, dropout called without
    `training=` (so it is the identity), attention memory without memory_sequence_length, sequence_length on the encoder's BiGRU only,
    maximum_iterations = max_iters, what the decoder step feeds to what, the loss terms, clip-then-Adam.

What stays unpinned is the arithmetic INSIDE TensorFlow's operations (oracle/taco_oracle.py header)."""
import json
import os

import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def runs():
    with open(os.path.join(GOLD, "graph_trace.json")) as f:
        d = json.load(f)
    return {(r["config"]["model_type"], r["config"]["attention_type"], r["config"]["training"], r["config"]["speaker_embedding_size"]) +
            (("priority",) if r["config"].get("prioritize_loss") else ()) + (("test_mode",) if r["config"].get("rnn_decoder_test_mode") else ()): r for r in d["runs"]}


def _ops(run, op, scope_prefix=None):
    return [t for t in run["trace"] if t["op"] == op and (scope_prefix is None or t["scope"].startswith(scope_prefix))]


def _in_dim(run, t, k=0):
    return run["trace"][t["in"][k]]["shape"][-1]


def _layers(run):
    """(reference scope/name, kernel shape) of every dense / conv1d the reference builds; unnamed tf.layers.dense calls get TensorFlow's
    automatic names (dense, dense_1, ... in creation order inside their scope)"""
    auto = {}
    out = []
    for t in run["trace"]:
        if t["op"] == "tf.layers.dense":
            name = t["kwargs"].get("name")
            if name is None:
                n = auto.get(t["scope"], 0)
                auto[t["scope"]] = n + 1
                name = "dense" if n == 0 else "dense_%d" % n
            out.append((t["scope"] + "/" + name, (_in_dim(run, t), t["kwargs"]["units"])))
        elif t["op"] == "tf.layers.conv1d":
            out.append((t["scope"], (t["kwargs"]["kernel_size"], _in_dim(run, t), t["kwargs"]["filters"])))
    return out


def test_every_layer_of_the_reference_is_a_tensor_of_the_product_with_the_same_shape(runs):
    import taco_amd
    run = runs[("single", "bah_mon", False, 16)]
    spec = dict(taco_amd.weights.weight_spec(taco_amd.hparams.copy(), 1))
    to_product = lambda s: s.replace("inference/decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper/concat_output_and_attention_wrapper/attention_wrapper/decoder_prenet_wrapper/decoder_prenet/", "decoder/prenet/").replace("inference/", "")
    got = {to_product(n): shp for n, shp in _layers(run)}
    assert got.pop("dense") == (512, 1025) and spec["linear/kernel"] == (512, 1025)          # the linear head: the one unnamed dense of scope `inference`
    assert len(got) == 2 + (16 + 2 + 8) + (8 + 2 + 1 + 8) + 2
    for name, shp in got.items():
        assert spec[name + "/kernel"] == tuple(shp), name
    # variables the reference names itself
    emb = [t for t in _ops(run, "tf.get_variable") if t["kwargs"]["name"] == "embedding"][0]
    assert tuple(emb["kwargs"]["shape"]) == spec["embedding"] == (80, 256)
    # recurrent cells: TensorFlow's GRUCell(n) on an input of width i owns gates [i + n, 2n] and candidate [i + n, n]
    cells = {t["id"]: t["args"][0] for t in _ops(run, "new GRUCell")}
    calls = _ops(run, "GRUCell.call")
    want = [("decoder/attention_gru", 128, 256), ("decoder/gru_1", 256, 256), ("decoder/gru_2", 256, 256)]
    assert len(calls) == 3
    for (name, i, n), t in zip(want, calls):
        assert _in_dim(run, t) == i and cells[t["kwargs"]["cell"]] == n
        assert spec[name + "/gates/kernel"] == (i + n, 2 * n) and spec[name + "/candidate/kernel"] == (i + n, n)
    for scope, n in (("encoder_cbhg", 128), ("post_cbhg", 256)):
        fw = _ops(run, "bidirectional_dynamic_rnn.output_fw", "inference/" + scope)[0]
        assert fw["kwargs"]["num_units"] == n and _in_dim(run, fw) == n
        assert spec[scope + "/bigru/fw/gates/kernel"] == (2 * n, 2 * n) and spec[scope + "/bigru/bw/candidate/kernel"] == (2 * n, n)
    # the two OutputProjectionWrappers: [h_att | context] (512) -> 256 in front of the decoder GRUs, 256 -> r * num_mels behind them
    proj = _ops(run, "OutputProjectionWrapper.linear")
    assert [(_in_dim(run, t), t["kwargs"]["units"]) for t in proj] == [(512, 256), (256, 320)]
    assert spec["decoder/concat_projection/kernel"] == (512, 256) and spec["decoder/frame_projection/kernel"] == (256, 320)
    # attention: BahdanauMonotonicAttention(num_units = 256, memory = encoder outputs [.., 256]); the query is the attention GRU's output (256)
    att = _ops(run, "new BahdanauMonotonicAttention")[0]
    assert att["args"][0] == 256 and run["trace"][att["in"][0]]["shape"][-1] == 256
    assert spec["attention/memory_layer/kernel"] == (256, 256) and spec["attention/query_layer/kernel"] == (256, 256) and spec["attention/attention_v"] == (256,)


def test_conv_is_activation_then_batchnorm_and_dropout_is_called_without_training(runs):
    for key in (("single", "bah_mon", False, 16), ("single", "bah_mon", True, 16)):
        run = runs[key]
        tr = run["trace"]
        convs = _ops(run, "tf.layers.conv1d")
        assert len(convs) == 16 + 2 + 8 + 2
        for t in convs:
            bn = tr[t["id"] + 1]                                  # modules.py:123-131: the BatchNorm consumes the conv's (activated) output
            assert bn["op"] == "tf.layers.batch_normalization" and bn["in"] == [t["id"]] and bn["kwargs"]["training"] is key[2]
            assert t["kwargs"]["padding"] == "same"
            last_proj = t["scope"].endswith("proj_2")
            assert t["kwargs"].get("activation") == (None if last_proj else "tf.nn.relu")
        drops = _ops(run, "tf.layers.dropout")
        assert len(drops) == 4                                    # encoder prenet x2, decoder prenet x2
        for t in drops:                                           # modules.py:24: rate and name only -- tf.layers.dropout(training=False) is the identity
            assert sorted(k for k, v in t["kwargs"].items() if v is not None) == ["name", "rate"]
            assert t["kwargs"]["rate"] == (0.8 if key[2] else 0.0)
        pools = _ops(run, "tf.layers.max_pooling1d")
        assert [(t["kwargs"]["pool_size"], t["kwargs"]["strides"], t["kwargs"]["padding"]) for t in pools] == [(2, 1, "same")] * 2
        hw = [t for t in _ops(run, "tf.layers.dense") if t["kwargs"].get("name") == "T"]
        assert len(hw) == 8 and all(t["kwargs"]["activation"] == "tf.nn.sigmoid" and t["kwargs"]["bias_initializer"] == "tf.constant_initializer(-1.0)" for t in hw)


def test_cbhg_wiring(runs):
    run = runs[("single", "bah_mon", False, 16)]
    tr = run["trace"]
    for scope, K, rnn in (("inference/encoder_cbhg", 16, 128), ("inference/post_cbhg", 8, 256)):
        bank = _ops(run, "tf.layers.conv1d", scope + "/conv_bank")
        assert [t["kwargs"]["kernel_size"] for t in bank] == list(range(1, K + 1)) and len({tuple(t["in"]) for t in bank}) == 1
        x = bank[0]["in"][0]                                      # the CBHG's input
        cat = _ops(run, "tf.concat", scope + "/conv_bank")[0]
        assert cat["in"] == [t["id"] + 1 for t in bank] and cat["kwargs"]["axis"] == -1      # the BatchNorm outputs, narrowest first
        pool = _ops(run, "tf.layers.max_pooling1d", scope)[0]
        assert pool["in"] == [cat["id"]]
        p1, p2 = _ops(run, "tf.layers.conv1d", scope + "/proj_1")[0], _ops(run, "tf.layers.conv1d", scope + "/proj_2")[0]
        assert p1["in"] == [pool["id"]] and p2["in"] == [p1["id"] + 1]
        res = [t for t in tr if t["op"] == "add" and t["scope"] == scope][0]
        assert res["in"] == [p2["id"] + 1, x]                     # proj_out + inputs (modules.py:62-69)
        nxt = tr[res["id"] + 1]
        if scope.endswith("post_cbhg"):                           # 80 != rnn_size: the unnamed dense (modules.py:71-73)
            assert nxt["op"] == "tf.layers.dense" and nxt["kwargs"]["units"] == rnn and nxt["kwargs"].get("activation") is None and nxt["in"] == [res["id"]]
        else:
            assert nxt["op"] == "tf.layers.dense" and nxt["kwargs"]["name"] == "H"      # straight into highway_1
        fw = _ops(run, "bidirectional_dynamic_rnn.output_fw", scope)[0]
        has_len = fw["kwargs"].get("sequence_length") is not None
        assert has_len == scope.endswith("encoder_cbhg")          # input_lengths for the encoder, None for the post-net (tacotron.py:222)
        out = [t for t in tr if t["op"] == "tf.concat" and t["scope"] == scope][-1]
        assert out["in"] == [fw["id"], fw["id"] + 1] and out["kwargs"]["axis"] == 2


def test_decoder_step_wiring_and_attention_arguments(runs):
    for atype, cls, extra in (("bah_mon", "BahdanauMonotonicAttention", {}), ("bah", "BahdanauAttention", {}), ("bah_norm", "BahdanauAttention", {"normalize": True})):
        run = runs[("single", atype, False, 16)]
        tr = run["trace"]
        att = _ops(run, "new " + cls)[0]
        # tacotron.py:132-146: (num_units, memory) and nothing else but normalize -- no memory_sequence_length, so the scores are not masked
        assert att["args"][0] == 256 and {k: v for k, v in att["kwargs"].items() if v is not None} == extra
        memory = att["in"][0]
        assert tr[memory]["scope"] == "inference/encoder_cbhg" and tr[memory]["op"] == "tf.concat"
        aw = _ops(run, "new AttentionWrapper")[0]
        assert {k: v for k, v in aw["kwargs"].items() if v is not None} == {"alignment_history": True, "output_attention": False}
        dd = _ops(run, "tf.contrib.seq2seq.dynamic_decode")[0]
        assert dd["kwargs"]["maximum_iterations"] == run["hparams"]["max_iters"] == 200
        step = [t for t in tr if t["scope"].startswith("inference/decoder")]
        cat0 = [t for t in step if t["op"] == "tf.concat"][0]     # cell_input_fn: [previous frame | previous context] -> the prenet
        assert [tr[i]["shape"][-1] for i in cat0["in"]] == [80, 256] and cat0["shape"][-1] == 336
        d1 = [t for t in step if t["op"] == "tf.layers.dense"][0]
        assert d1["in"] == [cat0["id"]] and d1["scope"].endswith("decoder_prenet")
        gru = [t for t in step if t["op"] == "GRUCell.call"]
        q = [t for t in step if t["op"] == "attention.__call__"][0]
        ident = [t for t in step if t["op"] == "tf.identity"][0]
        assert ident["in"] == [gru[0]["id"]] and q["in"][0] == ident["id"]          # the query is the attention GRU's NEW output
        cond = [t for t in step if t["op"] == "tf.cond"][0]       # rnn_wrappers.py:313-317: manual_alignments[:, time, :] or the computed ones
        man = tr[cond["in"][1]]
        assert tr[cond["in"][0]]["kwargs"]["name"] == "is_manual_attention" and man["op"] == "getitem" and man["kwargs"]["index"][0] == "::" and cond["in"][2] == q["id"]
        mm = [t for t in step if t["op"] == "tf.matmul"][0]
        assert mm["in"][1] == memory                              # context = alignments . VALUES = the raw encoder outputs
        cat1 = [t for t in step if t["op"] == "tf.concat" and t["shape"] and t["shape"][-1] == 512][0]
        assert cat1["in"][0] == ident["id"]                       # ConcatOutputAndAttentionWrapper: [cell output | context]
        proj = [t for t in step if t["op"] == "OutputProjectionWrapper.linear"]
        res = [t for t in step if t["op"] == "ResidualWrapper.add"]
        assert proj[0]["in"] == [cat1["id"]] and gru[1]["in"][0] == proj[0]["id"] and res[0]["in"] == [proj[0]["id"], gru[1]["id"]]
        assert gru[2]["in"][0] == res[0]["id"] and res[1]["in"] == [res[0]["id"], gru[2]["id"]] and proj[1]["in"] == [res[1]["id"]]
        # TacoTestHelper (helpers.py:9-32): stop when the r frames equal zero exactly; feed back the LAST of the r frames
        eq = [t for t in step if t["op"] == "tf.equal"][0]
        assert eq["in"][0] == proj[1]["id"]
        fb = [t for t in step if t["op"] == "getitem" and t["in"] == [proj[1]["id"]]][0]
        assert fb["kwargs"]["index"] == ["::", "-80::"]
        # mel reshape and alignments (tacotron.py:213-214, 238-239)
        rs = _ops(run, "tf.reshape")[0]
        assert rs["args"][1][1:] == [-1, 80]
        tp = _ops(run, "tf.transpose")[-1]
        assert tp["kwargs"]["perm"] == [1, 2, 0] and run["outputs"]["alignments"] == tp["id"]


def test_training_graph_teacher_forcing_loss_and_optimizer(runs):
    run = runs[("single", "bah_mon", True, 16)]
    tr = run["trace"]
    ph = {t["kwargs"]["name"]: t["id"] for t in _ops(run, "tf.placeholder")}
    sl = [t for t in tr if t["op"] == "getitem" and t["in"] == [ph["mel_targets"]] and t["kwargs"]["index"] == ["::", "3::4", "::"]]
    assert len(sl) == 1                                           # helpers.py:44: targets[:, r-1::r, :]
    step = [t for t in tr if t["scope"].startswith("inference/decoder")]
    nxt = [t for t in step if t["op"] == "getitem" and t["in"][0] == sl[0]["id"]]
    assert len(nxt) == 1 and nxt[0]["kwargs"]["index"][0] == "::" and nxt[0]["kwargs"]["index"][2] == "::"      # next input = targets[:, time, :]
    loss = [t for t in tr if t["scope"] == "loss"]
    mel_abs, lin_abs = [t for t in loss if t["op"] == "tf.abs"]
    assert tr[mel_abs["in"][0]]["in"] == [ph["mel_targets"], run["outputs"]["mel_outputs"]]
    assert tr[lin_abs["in"][0]]["in"] == [ph["linear_targets"], run["outputs"]["linear_outputs"]]
    means = [t for t in loss if t["op"] == "tf.reduce_mean"]
    assert len(means) == 4                                        # mean(mel * c) + mean(lin * c); linear_loss = mean(lin); mel_loss = mean(mel)
    total = [t for t in loss if t["op"] == "add"][0]
    assert total["in"] == [means[0]["id"], means[1]["id"]] and tr[tr[means[0]["in"][0]]["in"][0]]["id"] == mel_abs["id"]
    opt = [t for t in tr if t["scope"] == "optimizer"]
    adam = [t for t in opt if t["op"] == "new AdamOptimizer"][0]
    assert adam["args"][1:] == [run["hparams"]["adam_beta1"], run["hparams"]["adam_beta2"]] == [0.9, 0.999]
    cg = [t for t in opt if t["op"] == "optimizer.compute_gradients"][0]
    assert cg["in"] == [total["id"]]                              # gradients of `loss` (the coefficient-weighted one)
    clip = [t for t in opt if t["op"] == "tf.clip_by_global_norm"][0]
    assert clip["args"][1] == 1.0
    ap = [t for t in opt if t["op"] == "optimizer.apply_gradients"][0]
    assert ap["kwargs"]["global_step"] == {"sym": ph["global_step"]} and set(t["id"] for t in opt if t["op"].startswith("clipped")) <= set(ap["in"])
    assert [t["args"] for t in opt if t["op"] == "tf.get_collection"] == [["tf.GraphKeys.UPDATE_OPS"]]      # tacotron.py:334


def test_multispeaker_wiring(runs):
    import taco_amd
    from taco_amd.weights import TF_SCOPE_MAP
    dv = runs[("deepvoice", "bah_mon", False, 16)]
    tr = dv["trace"]
    spk = [t for t in _ops(dv, "tf.layers.dense") if t["kwargs"].get("activation") == "tf.nn.softsign"]
    assert [t["kwargs"]["units"] for t in spk] == [128, 256, 256, 256, 256] and all(_in_dim(dv, t) == 16 for t in spk)
    names = dict(_layers(dv))
    # TensorFlow names unnamed layers by creation order inside their scope: the five speaker layers take dense .. dense_4, the linear head dense_5
    assert [n for n in names if n.startswith("inference/dense")] == ["inference/dense", "inference/dense_1", "inference/dense_2", "inference/dense_3", "inference/dense_4", "inference/dense_5"]
    assert names["inference/dense_5"] == (512, 1025)
    for i, k in enumerate(["spk/before_highway", "spk/encoder_rnn_init", "spk/attention_rnn_init", "spk/decoder_rnn_init_1", "spk/decoder_rnn_init_2"]):
        assert TF_SCOPE_MAP[k] == ("dense" if i == 0 else "dense_%d" % i)
    spec = dict(taco_amd.weights.weight_spec(taco_amd.hparams.copy(model_type="deepvoice"), 3))
    assert [spec[k + "/kernel"] for k in ("spk/before_highway", "spk/encoder_rnn_init", "spk/attention_rnn_init", "spk/decoder_rnn_init_1", "spk/decoder_rnn_init_2")] == [(16, 128), (16, 256), (16, 256), (16, 256), (16, 256)]
    # before_highway joins the residual sum; the encoder's initial state is split in two; the attention GRU starts from its vector
    res = [t for t in tr if t["op"] == "add" and t["scope"] == "inference/encoder_cbhg"]
    assert len(res) == 2 and tr[res[1]["in"][1]]["op"] == "tf.tile"
    fw = _ops(dv, "bidirectional_dynamic_rnn.output_fw", "inference/encoder_cbhg")[0]
    assert tr[fw["kwargs"]["initial_state_fw"]["sym"]]["op"] == "tf.split[0]"
    aw = _ops(dv, "new AttentionWrapper")[0]
    assert aw["kwargs"]["initial_cell_state"] == {"sym": spk[2]["id"]}
    step = [t for t in tr if t["scope"].startswith("inference/decoder")]
    gru = [t for t in step if t["op"] == "GRUCell.call"]
    assert gru[1]["in"][1] == spk[3]["id"] and gru[2]["in"][1] == spk[4]["id"]      # decoder_init_state[idx + 1] = the speaker's vector (tacotron.py:183-197)
    # speaker_embedding_size == 1: per-speaker tables instead of dense layers (tacotron.py:52-66)
    tb = runs[("deepvoice", "bah_mon", False, 1)]
    tabs = {t["kwargs"]["name"]: tuple(t["kwargs"]["shape"]) for t in _ops(tb, "tf.get_variable")}
    assert tabs["before_highway"] == (3, 128) and tabs["encoder_rnn_init_state"] == (3, 256) and tabs["attention_rnn_init_state"] == (3, 256)
    assert tabs["decoder_rnn_init_states1"] == (3, 256) and tabs["decoder_rnn_init_states2"] == (3, 256)
    # simple: the embedding is concatenated behind the prenet output, behind [cell output | context] and in front of the post-net output
    sm = runs[("simple", "bah_mon", False, 16)]
    ts = sm["trace"]
    sstep = [t for t in ts if t["scope"].startswith("inference/decoder")]
    sc = [t for t in sstep if t["op"] == "tf.concat" and t["kwargs"].get("name") == "speaker_concat"][0]
    assert [ts[i]["shape"][-1] for i in sc["in"]] == [128, 16]
    cat = [t for t in sstep if t["op"] == "tf.concat" and t["shape"] and t["shape"][-1] == 528][0]
    assert [ts[i]["shape"][-1] for i in cat["in"]] == [256, 256, 16]
    lin = [t for t in _ops(sm, "tf.layers.dense") if t["kwargs"]["units"] == 1025][0]
    head = ts[lin["in"][0]]
    assert head["op"] == "tf.concat" and [ts[i]["shape"][-1] for i in head["in"]] == [16, 512]
    sspec = dict(taco_amd.weights.weight_spec(taco_amd.hparams.copy(model_type="simple"), 3))
    assert sspec["linear/kernel"] == (528, 1025) and sspec["decoder/concat_projection/kernel"] == (528, 256) and sspec["decoder/attention_gru/gates/kernel"] == (128 + 16 + 256, 512)


def _tf_variable_names(run):
    """The variable names a TensorFlow 1.x graph of this trace has: the reference decides the scopes, the layer names and the order in which
    the wrappers nest (traced); TensorFlow decides what a layer of each kind calls its variables inside its scope (conv1d/kernel,
    batch_normalization/gamma, gru_cell/gates/kernel, ...: the documented names, spelled out here)."""
    out = set()
    auto = {}
    atype = run["config"]["attention_type"]
    for t in run["trace"]:
        sc, op, kw = t["scope"], t["op"], t["kwargs"]
        if op == "tf.layers.conv1d":
            out |= {sc + "/conv1d/kernel", sc + "/conv1d/bias"}
        elif op == "tf.layers.batch_normalization":
            out |= {sc + "/batch_normalization/" + v for v in ("gamma", "beta", "moving_mean", "moving_variance")}
        elif op == "tf.layers.dense":
            name = kw.get("name")
            if name is None:
                n = auto.get(sc, 0)
                auto[sc] = n + 1
                name = "dense" if n == 0 else "dense_%d" % n
            out |= {sc + "/" + name + "/kernel", sc + "/" + name + "/bias"}
        elif op == "tf.get_variable":
            out.add((sc + "/" if sc else "") + kw["name"])
        elif op == "GRUCell.call":
            out |= {sc + "/" + g + "/" + v for g in ("gates", "candidate") for v in ("kernel", "bias")}
        elif op == "bidirectional_dynamic_rnn.output_fw":
            out |= {sc + "/bidirectional_rnn/" + d + "/gru_cell/" + g + "/" + v for d in ("fw", "bw") for g in ("gates", "candidate") for v in ("kernel", "bias")}
        elif op == "OutputProjectionWrapper.linear":
            out |= {sc + "/kernel", sc + "/bias"}
        elif op == "attention.__call__":
            out |= {sc + "/query_layer/kernel", sc + "/attention_v"}
            if atype == "bah_mon":
                out.add(sc + "/attention_score_bias")
            if atype == "bah_norm":
                out |= {sc + "/attention_g", sc + "/attention_b"}
        elif op in ("new BahdanauMonotonicAttention", "new BahdanauAttention"):
            out.add(sc + "/memory_layer/kernel")
    return {"model/" + n for n in out}


@pytest.mark.parametrize("key,ns", [(("single", "bah_mon", False, 16), 1), (("single", "bah", False, 16), 1), (("single", "bah_norm", False, 16), 1),
                                    (("deepvoice", "bah_mon", False, 16), 3), (("deepvoice", "bah_mon", False, 1), 3), (("simple", "bah_mon", False, 16), 3)])
def test_checkpoint_variable_names_follow_the_traced_scopes(runs, key, ns):
    """f1 (synthesizer.py:66-67, saver.restore): the TensorFlow variable names the importer / exporter use for every tensor
    (tf_checkpoint.tf_names_for, weights.TF_SCOPE_MAP) are exactly the names the reference's graph has -- scopes, layer names and the nesting
    of the RNN wrappers taken from the trace of the reference's own code, TensorFlow's per-layer variable names spelled out above."""
    import taco_amd
    from taco_amd import tf_checkpoint as T
    run = runs[key]
    hp = taco_amd.hparams.copy(model_type=key[0], attention_type=key[1], speaker_embedding_size=key[3])
    spec = taco_amd.weights.weight_spec(hp, ns)
    mine = set(T.tf_names_for(spec, key[1]).values())
    assert mine == _tf_variable_names(run), (sorted(mine - _tf_variable_names(run))[:5], sorted(_tf_variable_names(run) - mine)[:5])


def test_priority_loss_band_and_learning_rate_schedules(runs):
    """tacotron.py:283-296 with prioritize_loss: the band l1[:, :, lower:upper] and the 0.5 weights; :313-325: both schedules -- the slice
    bounds and the constants the reference's code produces are the ones the oracle (and through it taco_loss_f32 / taco_learning_rate) use."""
    import taco_oracle as O
    run = runs[("single", "bah_mon", True, 16, "priority")]
    tr = run["trace"]
    loss = [t for t in tr if t["scope"] == "loss"]
    band = [t for t in loss if t["op"] == "getitem"][0]
    F, sr = run["hparams"]["num_freq"], run["hparams"]["sample_rate"]
    lo, up = int(165 / (sr * 0.5) * F), int(5000 / (sr * 0.5) * F)
    assert band["kwargs"]["index"] == ["::", "::", "%d:%d:" % (lo, up)] and (lo, up) == (14, 427)
    assert tr[band["in"][0]]["op"] == "tf.abs"
    halves = [t for t in loss if t["op"] == "mul" and 0.5 in t["args"]]
    assert len(halves) == 3                                       # 0.5 * mean(l1 c), 0.5 * mean(l1_priority c), 0.5 * (mean(l1) + mean(l1_priority))
    # the oracle computes the same band (its loss is the checker of the device kernel)
    rs = __import__("numpy").random.RandomState(0)
    a, b = rs.rand(2, 3, F), rs.rand(2, 3, F)
    got = O.add_loss(rs.rand(2, 3, 80), rs.rand(2, 3, 80), a, b, [1.0, 1.0], prioritize_loss=True, sample_rate=sr)
    l1 = abs(b - a)
    assert abs(got["linear_loss"] - 0.5 * (l1.mean() + l1[:, :, lo:up].mean())) < 1e-12
    # learning rate: mode 1 = initial * exponential_decay(1., step, 3000, 0.95); mode 0 (default run) = initial * w**0.5 * min(step * w**-1.5, step**-0.5)
    opt = [t for t in tr if t["scope"] == "optimizer"]
    ed = [t for t in opt if t["op"] == "tf.train.exponential_decay"][0]
    assert ed["args"][0] == 1.0 and ed["args"][2:] == [3000, 0.95]
    run0 = runs[("single", "bah_mon", True, 16)]
    opt0 = [t for t in run0["trace"] if t["scope"] == "optimizer"]
    mul0 = [t for t in opt0 if t["op"] == "mul" and t["in"] == [[x for x in opt0 if x["op"] == "tf.cast"][0]["id"]]][0]
    assert abs(mul0["args"][1] - 40000.0 ** -1.5) < 1e-18           # not randomly initialised: warm-up 40000 steps (tacotron.py:316-319)
    assert abs(O.learning_rate(0, 0.002, 0, False) - 0.002 * 40000.0 ** 0.5 * min(1 * 40000.0 ** -1.5, 1.0)) < 1e-15
    assert abs(O.learning_rate(2999, 0.002, 1, True) - 0.002 * 0.95) < 1e-12


def test_rnn_decoder_test_mode_feeds_back_the_last_frame(runs):
    """helpers.py:59-66 with rnn_decoder_test_mode (the test model of train.py:158-166): the next input is the step's own last frame,
    outputs[:, -num_mels:], not the target frame -- while `finished` still counts target steps."""
    run = runs[("single", "bah_mon", True, 16, "test_mode")]
    tr = run["trace"]
    step = [t for t in tr if t["scope"].startswith("inference/decoder")]
    frame = [t for t in step if t["op"] == "OutputProjectionWrapper.linear"][-1]
    nxt = [t for t in step if t["op"] == "decoder_step.next_inputs"][0]
    fed = tr[nxt["in"][0]]
    assert fed["op"] == "getitem" and fed["in"] == [frame["id"]] and fed["kwargs"]["index"] == ["::", "-80::"]
    fin = tr[nxt["in"][1]]
    assert fin["op"] == "greater_equal"

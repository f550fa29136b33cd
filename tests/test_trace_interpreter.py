"""oracle.forward() == the reference's own graph (tests/golden/graph_trace.json) run with the oracle's per-operation arithmetic
(tests/trace_interpreter.py): the oracle's wiring is the reference's, operand order by operand order, for every model type and
attention type -- at the reference's default widths."""
import json
import os

import numpy as np
import pytest

import taco_oracle as O
from trace_interpreter import Interpreter

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def all_runs():
    with open(os.path.join(GOLD, "graph_trace.json")) as f:
        return json.load(f)["runs"]


@pytest.fixture(scope="module")
def runs(all_runs):
    return [r for r in all_runs if not r["config"]["training"]]


def _setup(model_type="single", atype="bah_mon", ses=16, ns=1, n=3, B=2, T_in=7):
    import taco_amd
    from taco_amd import tf_checkpoint as T
    ohp = O.OracleHParams(max_iters=n, model_type=model_type, attention_type=atype, speaker_embedding_size=ses)
    w = O.init_weights(ohp, ns, 77)
    for k in w:                      # non-trivial BatchNorm statistics and biases, so that operand order and placement matter everywhere
        if k.endswith("moving_variance"):
            w[k] = w[k] * 0 + np.random.RandomState(len(k)).uniform(0.5, 1.5, size=w[k].shape)
        elif k.endswith(("moving_mean", "/bias", "beta")):
            w[k] = w[k] + np.random.RandomState(len(k) + 1).normal(0, 0.1, size=w[k].shape)
    ids, L = O.synthetic_inputs(B, T_in, 78, ragged=True)
    hp = taco_amd.hparams.copy(model_type=model_type, attention_type=atype, speaker_embedding_size=ses)
    names = T.tf_names_for(taco_amd.weights.weight_spec(hp, ns), atype)
    canon_of_tf = {v[len("model/"):]: k for k, v in names.items()}
    return ohp, w, ids, L, canon_of_tf


def _close(got, ref, keys=("mel", "linear", "alignments"), tol=1e-9):
    for k in keys:
        assert np.shape(got[k]) == np.shape(ref[k]), k
        assert np.abs(np.asarray(got[k]) - np.asarray(ref[k])).max() < tol, (k, float(np.abs(np.asarray(got[k]) - np.asarray(ref[k])).max()))


@pytest.mark.parametrize("model_type,atype,ses,ns", [("single", "bah_mon", 16, 1), ("single", "bah", 16, 1), ("single", "bah_norm", 16, 1),
                                                     ("deepvoice", "bah_mon", 16, 3), ("deepvoice", "bah_mon", 1, 3), ("simple", "bah_mon", 16, 3)])
def test_the_oracle_computes_the_graph_the_reference_builds(runs, model_type, atype, ses, ns):
    run = [r for r in runs if (r["config"]["model_type"], r["config"]["attention_type"], r["config"]["speaker_embedding_size"]) == (model_type, atype, ses)][0]
    n = 3
    ohp, w, ids, L, canon = _setup(model_type, atype, ses, ns, n)
    B = ids.shape[0]
    spk = (np.arange(B) % ns).astype(np.int32) if ns > 1 else None
    ref = O.forward(w, ohp, ids, L, speaker_id=spk, num_speakers=ns, n_steps=n, honor_stop=False)
    got = Interpreter(run, w, canon).forward(ids, L, n, speaker_id=spk)
    assert got["mel"].shape == (B, n * ohp.reduction_factor, ohp.num_mels)
    _close(got, ref)


def test_manual_alignments_replace_the_computed_ones_at_state_time(runs):
    """rnn_wrappers.py:313-317: tf.cond(is_manual_attention, manual_alignments[:, state.time, :], computed): the context, the history and the
    NEXT step's previous-alignments all take the manual row"""
    run = [r for r in runs if (r["config"]["model_type"], r["config"]["attention_type"]) == ("single", "bah_mon")][0]
    n, B, T_in = 4, 2, 7
    ohp, w, ids, L, canon = _setup(n=n, B=B, T_in=T_in)
    man = np.random.RandomState(5).dirichlet(np.ones(T_in), size=(B, n))
    ref = O.forward(w, ohp, ids, L, n_steps=n, manual_alignments=man, honor_stop=False)
    got = Interpreter(run, w, canon).forward(ids, L, n, manual_alignments=man)
    _close(got, ref)
    assert np.abs(got["alignments"] - man.transpose(0, 2, 1)).max() < 1e-12


def test_the_decoder_stops_when_every_row_has_emitted_an_all_zero_step(runs):
    """helpers.py:29 + TF-sem dynamic_decode: finished = all(outputs == 0) per row, OR-ed over steps; the loop ends when every row is finished
    (or at maximum_iterations = hp.max_iters, tacotron.py:210) and the finishing step IS part of the output"""
    run = [r for r in runs if (r["config"]["model_type"], r["config"]["attention_type"]) == ("single", "bah_mon")][0]
    ohp, w, ids, L, canon = _setup(n=5)
    w = dict(w)
    for k in ("decoder/frame_projection/kernel", "decoder/frame_projection/bias"):
        w[k] = w[k] * 0
    ref = O.forward(w, ohp, ids, L, n_steps=5)
    got = Interpreter(run, w, canon).forward(ids, L, 5)
    assert got["n_steps"] == ref["stop_step"] == 1
    _close(got, ref)
    ddr = [t for t in run["trace"] if t["op"] == "tf.contrib.seq2seq.dynamic_decode"][0]
    assert ddr["kwargs"]["maximum_iterations"] == run["hparams"]["max_iters"] == 200


def _train_case(all_runs, n=3, **cfg):
    run = [r for r in all_runs if r["config"]["training"] and all(r["config"][k] == v for k, v in cfg.items())][0]
    ohp, w, ids, L, canon = _setup(n=n)
    B, r = ids.shape[0], ohp.reduction_factor
    rs = np.random.RandomState(9)
    mel_t = rs.normal(0, 1, size=(B, n * r, ohp.num_mels))
    lin_t = rs.normal(0, 1, size=(B, n * r, ohp.num_freq))
    coeff = np.array([1.0, 0.5])
    return run, ohp, w, ids, L, canon, mel_t, lin_t, coeff


def test_the_training_graph_teacher_forcing_batch_statistics_loss_and_learning_rate(all_runs):
    """tacotron.py:26,197-206,274-336 + helpers.py:36-72: the is_training graph -- BatchNorm on batch statistics (with its UPDATE_OPS), the
    decoder fed mel_targets[:, r-1::r][:, t-1] at step t>=1 and run for exactly T_out/r steps, both L1 losses, the warm-up schedule"""
    run, ohp, w, ids, L, canon, mel_t, lin_t, coeff = _train_case(all_runs, prioritize_loss=False, rnn_decoder_test_mode=False)
    n, r = 3, ohp.reduction_factor
    upd = {}
    ref = O.forward(w, ohp, ids, L, n_steps=n, teacher_frames=mel_t[:, r - 1::r], training=True, bn_updates=upd, honor_stop=False)
    got = Interpreter(run, w, canon).forward(ids, L, None, mel_targets=mel_t, linear_targets=lin_t, loss_coeff=coeff, global_step=1234)
    assert got["n_steps"] == n                  # from the helper's `time + 1 >= num_steps`, not from maximum_iterations
    _close(got, ref)
    loss = O.add_loss(ref["mel"], mel_t, ref["linear"], lin_t, coeff)
    _close(got, loss, ("loss", "mel_loss", "linear_loss", "loss_without_coeff"), 1e-12)
    assert abs(got["learning_rate"] - O.learning_rate(1234, run["hparams"]["initial_learning_rate"], 0, False)) < 1e-15
    assert set(got["bn_updates"]) == set(upd) and len(upd) == 2 * 28
    for k in upd:
        assert np.abs(got["bn_updates"][k] - upd[k]).max() < 1e-12


def test_the_prioritised_loss_and_the_exponential_schedule(all_runs):
    """tacotron.py:286-295 (0.5 * full band + 0.5 * the 165 Hz..5 kHz band) and :323-325"""
    run, ohp, w, ids, L, canon, mel_t, lin_t, coeff = _train_case(all_runs, prioritize_loss=True)
    n, r = 3, ohp.reduction_factor
    ref = O.forward(w, ohp, ids, L, n_steps=n, teacher_frames=mel_t[:, r - 1::r], training=True, honor_stop=False)
    got = Interpreter(run, w, canon).forward(ids, L, None, mel_targets=mel_t, linear_targets=lin_t, loss_coeff=coeff, global_step=4321)
    _close(got, ref)
    loss = O.add_loss(ref["mel"], mel_t, ref["linear"], lin_t, coeff, prioritize_loss=True, sample_rate=run["hparams"]["sample_rate"])
    _close(got, loss, ("loss", "mel_loss", "linear_loss", "loss_without_coeff"), 1e-12)
    assert abs(got["learning_rate"] - O.learning_rate(4321, run["hparams"]["initial_learning_rate"], 1, True)) < 1e-15


def test_rnn_decoder_test_mode_trains_on_its_own_frames(all_runs):
    """tacotron.py:199-203: TacoTestHelper inside the training graph -- free running (stop rule included), BatchNorm still on batch statistics"""
    run, ohp, w, ids, L, canon, mel_t, lin_t, coeff = _train_case(all_runs, rnn_decoder_test_mode=True)
    n = 3
    ref = O.forward(w, ohp, ids, L, n_steps=n, training=True, honor_stop=False)
    got = Interpreter(run, w, canon).forward(ids, L, n, mel_targets=mel_t, linear_targets=lin_t, loss_coeff=coeff, global_step=0)
    _close(got, ref)


@pytest.mark.parametrize("mutation", ["prenet_concat_order", "residual_dropped", "bn_before_activation"])
def test_the_comparison_notices_a_rewired_graph(runs, mutation):
    """the agreement above is not vacuous: one operand swap / one dropped edge in the trace and the interpreter no longer matches the oracle"""
    import copy
    run = copy.deepcopy([r for r in runs if (r["config"]["model_type"], r["config"]["attention_type"]) == ("single", "bah_mon")][0])
    tr = run["trace"]
    if mutation == "prenet_concat_order":        # rnn_wrappers.py:249 feeds [inputs, attention]; try [attention, inputs]
        t = [t for t in tr if t["op"] == "tf.concat" and t["scope"].endswith("attention_wrapper") and len(t["args"][0]) == 2][0]
        t["args"][0].reverse()
    elif mutation == "residual_dropped":
        t = [t for t in tr if t["op"] == "ResidualWrapper.add"][0]
        t["args"][0] = t["args"][1]
    else:
        t = [t for t in tr if t["op"] == "tf.layers.conv1d" and t["kwargs"].get("activation")][0]
        t["kwargs"]["activation"] = None
    n = 3
    ohp, w, ids, L, canon = _setup(n=n)
    ref = O.forward(w, ohp, ids, L, n_steps=n, honor_stop=False)
    got = Interpreter(run, w, canon).forward(ids, L, n)
    assert np.abs(got["linear"] - ref["linear"]).max() > 1e-6

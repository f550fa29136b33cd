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
def runs():
    with open(os.path.join(GOLD, "graph_trace.json")) as f:
        d = json.load(f)
    return [r for r in d["runs"] if not r["config"]["training"]]


@pytest.mark.parametrize("model_type,atype,ses,ns", [("single", "bah_mon", 16, 1), ("single", "bah", 16, 1), ("single", "bah_norm", 16, 1),
                                                     ("deepvoice", "bah_mon", 16, 3), ("deepvoice", "bah_mon", 1, 3), ("simple", "bah_mon", 16, 3)])
def test_the_oracle_computes_the_graph_the_reference_builds(runs, model_type, atype, ses, ns):
    import taco_amd
    from taco_amd import tf_checkpoint as T
    run = [r for r in runs if (r["config"]["model_type"], r["config"]["attention_type"], r["config"]["speaker_embedding_size"]) == (model_type, atype, ses)][0]
    n = 3
    ohp = O.OracleHParams(max_iters=n, model_type=model_type, attention_type=atype, speaker_embedding_size=ses)
    w = O.init_weights(ohp, ns, 77)
    for k in w:                      # non-trivial BatchNorm statistics and biases, so that operand order and placement matter everywhere
        if k.endswith("moving_variance"):
            w[k] = w[k] * 0 + np.random.RandomState(len(k)).uniform(0.5, 1.5, size=w[k].shape)
        elif k.endswith(("moving_mean", "/bias", "beta")):
            w[k] = w[k] + np.random.RandomState(len(k) + 1).normal(0, 0.1, size=w[k].shape)
    B, T_in = 2, 7
    ids, L = O.synthetic_inputs(B, T_in, 78, ragged=True)
    spk = (np.arange(B) % ns).astype(np.int32) if ns > 1 else None
    ref = O.forward(w, ohp, ids, L, speaker_id=spk, num_speakers=ns, n_steps=n, honor_stop=False)
    hp = taco_amd.hparams.copy(model_type=model_type, attention_type=atype, speaker_embedding_size=ses)
    names = T.tf_names_for(taco_amd.weights.weight_spec(hp, ns), atype)
    canon_of_tf = {v[len("model/"):]: k for k, v in names.items()}
    got = Interpreter(run, w, canon_of_tf).forward(ids, L, n, speaker_id=spk)
    r = ohp.reduction_factor
    assert got["mel"].shape == ref["mel"].shape == (B, n * r, ohp.num_mels)
    for k in ("mel", "linear", "alignments"):
        assert np.abs(got[k] - ref[k]).max() < 1e-9, (k, float(np.abs(got[k] - ref[k]).max()))

#!/usr/bin/env python
"""The turnkey half of pinning the oracle against TensorFlow itself.

TF 1.x cannot be installed where this repository is built, and the reference ships no vectors; this script is what a box that
HAS TensorFlow 1.3/1.4 and a checkout of GSByeon/multi-speaker-tacotron-tensorflow runs to produce reference outputs for the
committed fixtures.  Nothing of the reference is copied: it is imported from the path given on the command line.

Three stages (each is a sub-command; `prepare` and `compare` need only this repository and NumPy):

  prepare  <fixture.npz> <workdir>
           Turns a fixture (inputs + canonical weights, e.g. tests/golden/tiny_forward.npz, or --full-width for a freshly
           generated reference-width case) into what the reference loads with its own code:
             <workdir>/params.json          hyper-parameters in the reference's schema (utils/__init__.py:110-126 load_hparams)
             <workdir>/model.ckpt-0.*       a V2 checkpoint with the reference's variable names (tf_checkpoint.export_tf_checkpoint)
             <workdir>/inputs.npz           inputs, input_lengths, speaker_id, num_speakers
           Restoring that checkpoint with tf.train.Saver is at the same time the test of tf_checkpoint's writer and name map
           against real TensorFlow.

  run      --dry-run <workdir>                                                         (needs neither TensorFlow nor the reference)
           Prints (and writes <workdir>/expected_variables.txt) the variable list `run` expects tf.global_variables() to contain for
           the prepared hyper-parameters -- name and shape, under the scope conventions of TF 1.3 / 1.4 (tf.layers `dense_N`
           numbering in creation order, `gru_cell/{gates,candidate}/{kernel,bias}`, `bidirectional_rnn/{fw,bw}`, the wrapper chain of
           tacotron.py:166-181) -- checks that the name map is a bijection in both directions (canonical -> TF -> canonical) and that
           the prepared checkpoint's index holds exactly those names.  So the first real run fails on arithmetic, not on names.

  run      --reference /path/to/multi-speaker-tacotron-tensorflow <workdir>            (needs TensorFlow 1.x)
           Builds the reference graph exactly as synthesizer.py:28-67 does (create_model(hparams).initialize(inputs, input_lengths,
           num_speakers, speaker_id) under variable_scope('model')), restores the checkpoint, runs
           [linear_outputs, mel_outputs, alignments] (synthesizer.py:122-126,166-167) and writes <workdir>/tf1_outputs.npz.
           The variables TensorFlow actually created are compared with the expected list first; if the names differ (another TF
           minor version, a scope the map does not know) the difference is printed and the values are assigned variable by variable
           through the shape-aware matcher (tf_checkpoint.match_tf_names) instead of tf.train.Saver -- the run still completes.

  compare  <fixture.npz> <workdir> [--tol 1e-3]
           Max-abs differences of tf1_outputs.npz against the oracle outputs stored in the fixture and the alignment-argmax check;
           exit status 1 beyond the tolerance.  A green compare is the pin; copy tf1_outputs.npz next to the fixture as
           tests/golden/<name>.tf1.npz and tests/test_oracle.py picks it up."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_fixture(path):
    g = np.load(path, allow_pickle=False)
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    meta = json.loads(str(g["meta"])) if "meta" in g.files else None
    return g, w, meta


def _fixture_hparams(name_or_meta):
    """OracleHParams of a committed fixture (tiny_forward: tests/golden/make_golden.py) or from the fixture's own meta record."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import taco_oracle as O
    if isinstance(name_or_meta, dict):
        return O.OracleHParams(**name_or_meta["hparams"]), int(name_or_meta["num_speakers"])
    from golden.make_golden import fixture_config
    hp, _, _, _, _, ns = fixture_config()
    return hp, ns


def make_full_width_fixture(path, seed=20260927):
    """A reference-width case small enough for a CPU: B=2, T_in=24 (ragged), 8 decoder steps, single speaker, bah_mon."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import taco_oracle as O
    hp = O.OracleHParams(max_iters=8)
    w = O.init_weights(hp, 1, seed)
    ids, L = O.synthetic_inputs(2, 24, seed + 1, ragged=True)
    out = O.forward(w, hp, ids, L)
    meta = {"hparams": hp.to_dict(), "num_speakers": 1}
    np.savez_compressed(path, inputs=ids, input_lengths=L, speaker_id=np.zeros(2, np.int32), mel=out["mel"], linear=out["linear"],
                        alignments=out["alignments"], meta=json.dumps(meta), **{"w:" + k: v for k, v in w.items()})
    return path


def prepare(args):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import taco_amd
    from taco_amd import tf_checkpoint as T
    if args.full_width:
        make_full_width_fixture(args.fixture)
    g, w, meta = _load_fixture(args.fixture)
    ohp, ns = _fixture_hparams(meta if meta is not None else os.path.basename(args.fixture))
    hp = taco_amd.HParams(**ohp.to_dict())
    os.makedirs(args.workdir, exist_ok=True)
    taco_amd.save_hparams(args.workdir, hp)
    spec = taco_amd.weights.weight_spec(hp, ns)
    T.export_tf_checkpoint(os.path.join(args.workdir, "model.ckpt-0"), w, spec, hp.attention_type, 0, num_shards=args.shards)
    with open(os.path.join(args.workdir, "checkpoint"), "w") as f:          # what tf.train.latest_checkpoint reads
        f.write('model_checkpoint_path: "model.ckpt-0"\nall_model_checkpoint_paths: "model.ckpt-0"\n')
    np.savez(os.path.join(args.workdir, "inputs.npz"), inputs=g["inputs"], input_lengths=g["input_lengths"],
             speaker_id=g["speaker_id"] if "speaker_id" in g.files else np.zeros(len(g["inputs"]), np.int32), num_speakers=ns)
    print("prepared %s: params.json, model.ckpt-0 (%d variables, %d shard(s)), inputs.npz" % (args.workdir, len(spec), args.shards))


def _prepared(workdir):
    """(product hparams, num_speakers, canonical spec, {canonical: expected TF name}) of a prepared work directory"""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import taco_amd
    from taco_amd import tf_checkpoint as T
    hp = taco_amd.load_hparams(taco_amd.hparams.copy(), workdir)
    ns = int(np.load(os.path.join(workdir, "inputs.npz"))["num_speakers"])
    spec = taco_amd.weights.weight_spec(hp, ns)
    return hp, ns, spec, T.tf_names_for(spec, hp.attention_type)


def dry_run(args):
    hp, ns, spec, names = _prepared(args.workdir)
    from taco_amd import tf_checkpoint as T
    shapes = dict(spec)
    expected = {names[c] + ":0": shapes[c] for c, _ in spec}
    lines = ["%s %s" % (n, list(shp)) for n, shp in sorted(expected.items())]
    with open(os.path.join(args.workdir, "expected_variables.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    # both directions: canonical -> TF name is injective, and the matcher that reads TF names recovers every canonical name from it
    assert len(set(names.values())) == len(names), "two canonical weights map to one TF variable name"
    back = T.match_tf_names({names[c]: shapes[c] for c, _ in spec}, spec)
    wrong = [(c, back[c], names[c]) for c, _ in spec if back[c] != names[c]]
    assert not wrong, "name map is not its own inverse: %s" % wrong[:4]
    # and the prepared checkpoint's index holds exactly these variables (+ global_step)
    idx = T.read_index(os.path.join(args.workdir, "model.ckpt-0.index"))
    in_ckpt = set(k for k in idx if k) - {"global_step"}
    want = set(names.values())
    assert in_ckpt == want, "checkpoint index differs from the expected variables: only in checkpoint %s, only expected %s" % (
        sorted(in_ckpt - want)[:6], sorted(want - in_ckpt)[:6])
    print("dry run OK: %d variables for model_type=%s attention_type=%s num_speakers=%d; name map is a bijection; checkpoint index matches; "
          "written to %s" % (len(expected), hp.model_type, hp.attention_type, ns, os.path.join(args.workdir, "expected_variables.txt")))


def run(args):
    if args.dry_run:
        return dry_run(args)
    if not args.reference:
        sys.exit("run needs --reference <checkout of the reference> (or --dry-run)")
    ref = os.path.abspath(args.reference)
    if not os.path.isfile(os.path.join(ref, "models", "tacotron.py")):
        sys.exit("%s does not look like a checkout of the reference (models/tacotron.py missing)" % ref)
    _, _, spec, exp_names = _prepared(args.workdir)           # before the reference's modules shadow `hparams` / `utils` / `text`
    from taco_amd import tf_checkpoint as T
    sys.path.insert(0, ref)
    import tensorflow as tf                                  # TensorFlow 1.x (requirements.txt of the reference: 1.3.0)
    if int(tf.__version__.split(".")[0]) != 1:
        sys.exit("TensorFlow 1.x is required (found %s): the reference uses tf.contrib.seq2seq / tf.contrib.rnn" % tf.__version__)
    from hparams import hparams                              # the reference's modules, imported from its own tree
    from models import create_model
    from utils import load_hparams
    d = np.load(os.path.join(args.workdir, "inputs.npz"))
    ns = int(d["num_speakers"])
    load_hparams(hparams, args.workdir)
    inputs = tf.placeholder(tf.int32, [None, None], "inputs")
    input_lengths = tf.placeholder(tf.int32, [None], "input_lengths")
    speaker_id = tf.placeholder_with_default(tf.zeros([tf.shape(inputs)[0]], dtype=tf.int32), [None], "speaker_id")
    with tf.variable_scope("model"):
        model = create_model(hparams)
        model.initialize(inputs, input_lengths, ns, speaker_id)
    cfg = tf.ConfigProto(allow_soft_placement=True, intra_op_parallelism_threads=1, inter_op_parallelism_threads=2)
    with tf.Session(config=cfg) as sess:
        sess.run(tf.global_variables_initializer())
        gvars = {v.name: v for v in tf.global_variables()}
        names = sorted(gvars)
        with open(os.path.join(args.workdir, "tf1_variables.txt"), "w") as f:
            f.write("\n".join("%s %s" % (v.name, v.shape.as_list()) for v in tf.global_variables()) + "\n")
        expected = set(n + ":0" for n in exp_names.values())
        if set(names) == expected:
            tf.train.Saver().restore(sess, os.path.join(args.workdir, "model.ckpt-0"))
            how = "tf.train.Saver"
        else:
            print("variable names differ from the expected list (TensorFlow %s):" % tf.__version__)
            for n in sorted(set(names) - expected):
                print("  only in the graph   :", n, gvars[n].shape.as_list())
            for n in sorted(expected - set(names)):
                print("  only in the expected:", n)
            match = T.match_tf_names({n[:-2]: tuple(v.shape.as_list()) for n, v in gvars.items()}, spec)    # raises with the details if hopeless
            ckpt = T.read_checkpoint(os.path.join(args.workdir, "model.ckpt-0"))
            for canon, tfn in match.items():
                gvars[tfn + ":0"].load(ckpt[exp_names[canon]], sess)
            how = "assigned one by one through tf_checkpoint.match_tf_names (report the listing above: weights.py::TF_SCOPE_MAP needs it)"
        feed = {model.inputs: d["inputs"], model.input_lengths: d["input_lengths"]}
        if ns > 1:
            feed[model.speaker_id] = d["speaker_id"]
        feed.update(model.get_dummy_feed_dict())
        lin, mel, ali = sess.run([model.linear_outputs, model.mel_outputs, model.alignments], feed_dict=feed)
    np.savez_compressed(os.path.join(args.workdir, "tf1_outputs.npz"), linear=lin, mel=mel, alignments=ali,
                        tf_version=str(tf.__version__), n_variables=len(names))
    print("wrote %s (TensorFlow %s, %d variables restored with %s)" % (os.path.join(args.workdir, "tf1_outputs.npz"), tf.__version__, len(names), how))


def compare(args):
    g, _, _ = _load_fixture(args.fixture)
    t = np.load(os.path.join(args.workdir, "tf1_outputs.npz"))
    ok = True
    for k in ("mel", "linear", "alignments"):
        a, b = np.asarray(t[k], np.float64), np.asarray(g[k], np.float64)
        n = min(a.shape[1], b.shape[1]) if k != "alignments" else None     # TF stops at the stop rule as the oracle does; guard anyway
        if k != "alignments" and a.shape != b.shape:
            print("%s: shapes differ TF %s vs oracle %s (comparing the common %d frames)" % (k, a.shape, b.shape, n))
            a, b = a[:, :n], b[:, :n]
        elif a.shape != b.shape:
            m = min(a.shape[2], b.shape[2])
            a, b = a[:, :, :m], b[:, :, :m]
        err = float(np.abs(a - b).max())
        print("%-10s max|TF1 - oracle| = %.3e %s" % (k, err, "" if err < args.tol else "  <-- beyond %g" % args.tol))
        ok = ok and err < args.tol
    a, b = t["alignments"], g["alignments"]
    m = min(a.shape[2], b.shape[2])
    sel = b[:, :, :m].max(1) > 1e-6
    same = (a[:, :, :m].argmax(1) == b[:, :, :m].argmax(1)) | ~sel
    print("alignment argmax identical at %d of %d compared steps" % (int((same & sel).sum()), int(sel.sum())))
    ok = ok and bool(same.all())
    print("PINNED: the oracle reproduces TensorFlow %s on this fixture" % str(t["tf_version"]) if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("prepare"); p.add_argument("fixture"); p.add_argument("workdir")
    p.add_argument("--full-width", action="store_true", help="generate the fixture first (reference widths, B=2, T_in=24, 8 steps)")
    p.add_argument("--shards", type=int, default=1)
    p = sub.add_parser("run"); p.add_argument("--reference"); p.add_argument("--dry-run", action="store_true"); p.add_argument("workdir")
    p = sub.add_parser("compare"); p.add_argument("fixture"); p.add_argument("workdir"); p.add_argument("--tol", type=float, default=1e-3)
    args = ap.parse_args()
    {"prepare": prepare, "run": run, "compare": compare}[args.cmd](args)


if __name__ == "__main__":
    main()

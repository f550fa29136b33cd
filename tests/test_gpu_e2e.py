"""GPU parity, stage level and end to end, against the oracle (float64) and the golden fixture.
Bar (BASELINE.json north_star): mel / linear within 1e-3 max-abs, alignment argmax identical."""
import os

import numpy as np
import pytest

import taco_oracle as O
from util import tiny_hp, build_model, to_product_hp, maxabs, argmax_match

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "tiny_forward.npz")


def _run(m, ids, L, spk=None, **kw):
    import torch
    lin, al = m.run(inputs=ids, input_lengths=L, speaker_id=spk, **kw)
    torch.cuda.synchronize()
    return m.mel_outputs.cpu().numpy(), lin.cpu().numpy(), al.cpu().numpy()


def _check(hip, ref, tol=1e-3):
    mel, lin, al = hip
    assert mel.shape == ref["mel"].shape and lin.shape == ref["linear"].shape and al.shape == ref["alignments"].shape
    assert maxabs(mel, ref["mel"]) < tol, "mel"
    assert maxabs(lin, ref["linear"]) < tol, "linear"
    assert maxabs(al, ref["alignments"]) < tol, "alignments"
    n, bad = argmax_match(al, ref["alignments"])
    assert bad == 0, "alignment argmax differs at %d of %d steps" % (bad, n)


@pytest.mark.parametrize("atype", ["bah_mon", "bah", "bah_norm"])
@pytest.mark.parametrize("ragged", [False, True])
def test_tiny_forward_all_attention_types(atype, ragged):
    ohp = tiny_hp(attention_type=atype)
    w = O.init_weights(ohp, 1, 21)
    ids, L = O.synthetic_inputs(3, 13, 31, ragged=ragged)
    m = build_model(ohp, w)
    _check(_run(m, ids, L), O.forward(w, ohp, ids, L), tol=2e-4)


def test_stage_level_encoder_decoder_postnet():
    import torch
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 22)
    ids, L = O.synthetic_inputs(4, 10, 32, ragged=True)
    taps = {}
    ref = O.forward(w, ohp, ids, L, taps=taps)
    m = build_model(ohp, w)
    enc = m.encoder(ids, L)
    torch.cuda.synchronize()
    assert maxabs(enc.cpu().numpy(), taps["encoder"]) < 1e-4
    # decoder from the ORACLE's encoder output, with per-step state dump
    mel, al, stop, dbg = m.decoder(taps["encoder"], ohp.max_iters, debug=True)
    torch.cuda.synchronize()
    assert int(stop.item()) == ohp.max_iters
    dbg = dbg.cpu().numpy()
    As, D = ohp.attention_state_size, 2 * ohp.enc_rnn_size
    for t, st in enumerate(taps["steps"]):
        assert maxabs(dbg[t, :, :As], st["h_att"]) < 2e-4, "h_att step %d" % t
        assert maxabs(dbg[t, :, As:As + D], st["ctx"]) < 2e-4, "ctx step %d" % t
        for i, h in enumerate(st["h"]):
            o = As + D + i * ohp.dec_rnn_size
            assert maxabs(dbg[t, :, o:o + ohp.dec_rnn_size], h) < 2e-4, "h_%d step %d" % (i + 1, t)
    assert maxabs(mel.cpu().numpy(), ref["mel"]) < 2e-4
    # post-net from the ORACLE's mel
    lin, post = m.postnet(ref["mel"], return_post=True)
    torch.cuda.synchronize()
    assert maxabs(post.cpu().numpy(), taps["post"][..., :post.shape[-1]]) < 2e-4
    assert maxabs(lin.cpu().numpy(), ref["linear"]) < 2e-4


def test_teacher_forced_decoder_steps():
    """Per-step parity with the fed-back frame replaced by given frames (TacoTrainingHelper rule,
    helpers.py:44,66) so recurrent error growth cannot hide a wrong step."""
    import torch
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 23)
    ids, L = O.synthetic_inputs(2, 9, 33)
    rs = np.random.RandomState(0)
    n = 6
    teacher = rs.uniform(0, 1, (2, n, ohp.num_mels))
    taps = {}
    ref = O.forward(w, ohp, ids, L, n_steps=n, teacher_frames=teacher, taps=taps)
    m = build_model(ohp, w)
    mel, al, _, _ = m.decoder(taps["encoder"], n, teacher_frames=teacher)
    torch.cuda.synchronize()
    assert maxabs(mel.cpu().numpy(), ref["mel"]) < 1e-4
    assert maxabs(al.cpu().numpy(), ref["alignments"]) < 1e-5


def test_golden_fixture_deepvoice():
    g = np.load(GOLDEN)
    from golden.make_golden import fixture_config
    ohp, _, _, _, _, ns = fixture_config()
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    m = build_model(ohp, w, num_speakers=ns)
    hip = _run(m, g["inputs"], g["input_lengths"], g["speaker_id"])
    _check(hip, dict(mel=g["mel"], linear=g["linear"], alignments=g["alignments"]), tol=2e-4)


def test_speaker_embedding_size_one_tables():
    ohp = tiny_hp(model_type="deepvoice", speaker_embedding_size=1)
    w = O.init_weights(ohp, 3, 24)
    ids, L = O.synthetic_inputs(3, 8, 34)
    spk = np.array([1, 2, 0], np.int32)
    m = build_model(ohp, w, num_speakers=3)
    _check(_run(m, ids, L, spk), O.forward(w, ohp, ids, L, speaker_id=spk, num_speakers=3), tol=2e-4)


@pytest.mark.parametrize("atype", ["bah_mon", "bah"])
def test_simple_multispeaker_model(atype):
    """model_type 'simple': speaker embedding concatenated at the attention-GRU input, the concat projection
    and (in front) the linear head (rnn_wrappers.py:372-376,405-413; tacotron.py:226-235)."""
    ohp = tiny_hp(model_type="simple", attention_type=atype)
    w = O.init_weights(ohp, 3, 41)
    ids, L = O.synthetic_inputs(4, 9, 42, ragged=True)
    spk = np.array([2, 0, 1, 2], np.int32)
    m = build_model(ohp, w, num_speakers=3)
    _check(_run(m, ids, L, spk), O.forward(w, ohp, ids, L, speaker_id=spk, num_speakers=3), tol=2e-4)
    # a different speaker changes the output
    mel_a = _run(m, ids, L, spk)[0]
    mel_b = _run(m, ids, L, np.array([0, 0, 1, 2], np.int32))[0]
    assert np.abs(mel_a[0] - mel_b[0]).max() > 1e-4 and np.array_equal(mel_a[1:], mel_b[1:])


def test_simple_full_width():
    ohp = O.OracleHParams(max_iters=8, model_type="simple")
    w = O.init_weights(ohp, 4, 43)
    ids, L = O.synthetic_inputs(5, 40, 44, ragged=True)
    spk = np.array([3, 1, 0, 2, 3], np.int32)
    m = build_model(ohp, w, num_speakers=4)
    _check(_run(m, ids, L, spk), O.forward(w, ohp, ids, L, speaker_id=spk, num_speakers=4), tol=1e-3)


def test_batch_permutation_and_row_independence():
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 25)
    ids, L = O.synthetic_inputs(6, 12, 35, ragged=True)
    m = build_model(ohp, w)
    mel, lin, al = _run(m, ids, L)
    perm = np.array([3, 0, 5, 1, 4, 2])
    mel2, lin2, al2 = _run(m, ids[perm], L[perm])
    assert np.array_equal(mel[perm], mel2) and np.array_equal(lin[perm], lin2) and np.array_equal(al[perm], al2)
    # a batch of one row gives the same row (different MFMA row-tile count, same arithmetic)
    mel1, _, _ = _run(m, ids[2:3], L[2:3])
    assert maxabs(mel1[0], mel[2]) < 1e-5


def test_batch_permutation_at_production_size_with_and_without_batch_invariance():
    """the fused point-wise kernel starts a workgroup's K loops at a step derived from its tile index (CH_ROT), so with 8 or
    more row tiles a row's fp32 accumulation order depends on where in the batch it sits.  Reference widths, 32 rows x 64 inputs (32 tiles
    in the encoder, 16 in the post-net): by default a permuted batch gives the permuted outputs to fp32 rounding; with
    taco_model_set_batch_invariant (Tacotron.set_batch_invariant) to the bit."""
    ohp = O.OracleHParams(max_iters=8)
    w = O.init_weights(ohp, 1, 125)
    ids, L = O.synthetic_inputs(32, 64, 135, ragged=True)
    perm = np.random.RandomState(3).permutation(32)
    m = build_model(ohp, w)
    mel, lin, al = _run(m, ids, L, honor_stop=False)
    mel2, lin2, al2 = _run(m, ids[perm], L[perm], honor_stop=False)
    d = max(maxabs(mel[perm], mel2), maxabs(lin[perm], lin2), maxabs(al[perm], al2))
    print("permuted batch, default (rotated K loops): max difference %.2e, bitwise equal: %s" % (d, d == 0.0))
    assert d < 2e-5
    m.set_batch_invariant(True)
    mel, lin, al = _run(m, ids, L, honor_stop=False)
    mel2, lin2, al2 = _run(m, ids[perm], L[perm], honor_stop=False)
    assert np.array_equal(mel[perm], mel2) and np.array_equal(lin[perm], lin2) and np.array_equal(al[perm], al2)
    # ... and a shard of the batch reproduces its rows to the bit (what data-parallel serving relies on)
    mel3, lin3, al3 = _run(m, ids[8:16], L[8:16], honor_stop=False)
    print("an 8-row shard vs its rows inside the 32-row batch: max difference %.2e" % max(maxabs(mel3, mel[8:16]), maxabs(lin3, lin[8:16])))
    assert maxabs(mel3, mel[8:16]) < 2e-5 and maxabs(lin3, lin[8:16]) < 2e-5       # (other rows per decoder group: another instantiation, same arithmetic to rounding)


def test_pad_ids_beyond_length_do_not_matter_for_other_rows():
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 26)
    ids, L = O.synthetic_inputs(3, 14, 36, ragged=True)
    m = build_model(ohp, w)
    mel, _, _ = _run(m, ids, L)
    ids2 = ids.copy()
    ids2[0, L[0] + 1:] = 5          # scribble over row 0's padding
    mel2, _, _ = _run(m, ids2, L)
    assert np.array_equal(mel[1:], mel2[1:])


def test_stop_rule_cuts_outputs_like_dynamic_decode():
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 27)
    w["decoder/frame_projection/kernel"][:] = 0
    w["decoder/frame_projection/bias"][:] = 0
    ids, L = O.synthetic_inputs(2, 6, 37)
    ref = O.forward(w, ohp, ids, L)
    assert ref["stop_step"] == 1
    m = build_model(ohp, w)
    hip = _run(m, ids, L)
    assert m.stop_step == 1
    _check(hip, ref, tol=2e-4)
    # without honouring the stop rule the full max_iters are returned
    mel, lin, al = _run(m, ids, L, honor_stop=False)
    assert mel.shape[1] == ohp.max_iters * ohp.reduction_factor


def test_manual_attention_override():
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 28)
    ids, L = O.synthetic_inputs(2, 9, 38)
    rs = np.random.RandomState(1)
    man = rs.dirichlet(np.ones(9), (2, ohp.max_iters))          # [B, T_dec, T_in]
    ref = O.forward(w, ohp, ids, L, manual_alignments=man)
    m = build_model(ohp, w)
    hip = _run(m, ids, L, manual_alignments=man, is_manual_attention=True)
    _check(hip, ref, tol=2e-4)
    assert maxabs(hip[2], np.transpose(man, (0, 2, 1))) < 1e-7


def test_graph_replay_equals_eager_and_is_repeatable():
    import ctypes as C
    import torch
    import taco_amd
    from util import dev, ptr, stream
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 29)
    ids, L = O.synthetic_inputs(3, 10, 39)
    m = build_model(ohp, w)
    mel_g, lin_g, al_g = _run(m, ids, L)
    mel_g2, _, _ = _run(m, ids, L)
    assert np.array_equal(mel_g, mel_g2)
    B, T_in, n, r = 3, 10, ohp.max_iters, ohp.reduction_factor
    mel = torch.empty((B, n * r, ohp.num_mels), device="cuda")
    lin = torch.empty((B, n * r, ohp.num_freq), device="cuda")
    al = torch.empty((B, T_in, n), device="cuda")
    stop = torch.zeros((1,), dtype=torch.int32, device="cuda")
    nb = int(m._lib.taco_workspace_bytes(m._handle, B, T_in, n))
    ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    idd, Ld = dev(ids), dev(L)
    taco_amd._lib.check(m._lib.taco_forward_infer(m._handle, stream(), ptr(idd), ptr(Ld), ptr(None), B, T_in, n,
                                                  ptr(None), ptr(mel), ptr(lin), ptr(al), ptr(stop), ptr(ws), nb))
    torch.cuda.synchronize()
    assert np.array_equal(mel.cpu().numpy(), mel_g) and np.array_equal(lin.cpu().numpy(), lin_g)
    assert m.plan_for(B, T_in).num_nodes > 10


def test_overlapped_postnet_equals_sequential():
    """taco_forward_infer runs the post-net feed-forward stages chunk by chunk behind the decoder on a second
    stream (time-window GEMMs with conv halos); the result must be bit-identical to the sequential order."""
    ohp = tiny_hp(max_iters=53, reduction_factor=3)          # 4 chunks of 16 steps, ragged last chunk
    w = O.init_weights(ohp, 1, 61)
    ids, L = O.synthetic_inputs(5, 11, 62, ragged=True)
    m = build_model(ohp, w)
    m._lib.taco_debug_set_overlap(m._handle, 0)
    seq = _run(m, ids, L)
    m._plans.clear()
    m._lib.taco_debug_set_overlap(m._handle, 1)
    ovl = _run(m, ids, L)
    for a, b in zip(seq, ovl):
        assert np.array_equal(a, b)
    _check(ovl, O.forward(w, ohp, ids, L), tol=5e-4)


def test_workspace_too_small_is_an_error_not_a_crash():
    import torch
    import taco_amd
    from util import dev, ptr, stream
    ohp = tiny_hp()
    m = build_model(ohp, O.init_weights(ohp, 1, 30))
    ids, L = O.synthetic_inputs(2, 6, 40)
    ws = torch.empty((1024,), dtype=torch.uint8, device="cuda")
    out = torch.empty((1 << 16,), device="cuda")
    idd, Ld = dev(ids), dev(L)
    rc = m._lib.taco_forward_infer(m._handle, stream(), ptr(idd), ptr(Ld), ptr(None), 2, 6, 3, ptr(None),
                                   ptr(out), ptr(out), ptr(out), ptr(None), ptr(ws), 1024)
    assert rc == taco_amd._lib.TACO_ERR_STATE and b"workspace too small" in m._lib.taco_last_error()


def test_synthesizer_surface(tmp_path):
    import taco_amd
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 31)
    taco_amd.save_hparams(str(tmp_path), to_product_hp(ohp))
    taco_amd.weights.save_weights(str(tmp_path / "model.ckpt-100.safetensors"), w)
    taco_amd.weights.save_weights(str(tmp_path / "model.ckpt-20.safetensors"), O.init_weights(ohp, 1, 99))
    ids, L = O.synthetic_inputs(2, 9, 41)
    s = taco_amd.Synthesizer().load(str(tmp_path), num_speakers=1)
    lin, al = s.synthesize(tokens=ids)
    ref = O.forward(w, ohp, ids, L)
    assert maxabs(lin, ref["linear"]) < 2e-4 and maxabs(al, ref["alignments"]) < 2e-4
    lin1, al1 = s.synthesize(tokens=ids, manual_attention_mode=1)      # second pass with the reference's one-hot alignments (synthesizer.py:173-179):
    from taco_amd.synthesizer import manual_alignments_of            # one decoder step per ENCODER position (pinned on the reference: test_reference_vectors.py)
    want = np.transpose(manual_alignments_of(al, 1), [0, 2, 1])
    assert set(np.unique(al1)) <= {0.0, 1.0} and np.all(al1.sum(2) == 1) and np.array_equal(al1, want)
    s.close()


def test_synthesizer_loads_a_tensorflow_checkpoint_of_the_reference(tmp_path):
    """synthesizer.py:66-67 `saver.restore`: a V2 bundle with the reference's variable names is read without TensorFlow."""
    import taco_amd
    from taco_amd import tf_checkpoint as T
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 32)
    hp = to_product_hp(ohp)
    taco_amd.save_hparams(str(tmp_path), hp)
    T.export_tf_checkpoint(str(tmp_path / "model.ckpt-7000"), w, taco_amd.weights.weight_spec(hp, 1), "bah_mon", 7000)
    ids, L = O.synthetic_inputs(2, 9, 42)
    s = taco_amd.Synthesizer().load(str(tmp_path), num_speakers=1)
    lin, al = s.synthesize(tokens=ids)
    ref = O.forward(w, ohp, ids, L)
    assert maxabs(lin, ref["linear"]) < 2e-4 and maxabs(al, ref["alignments"]) < 2e-4
    s.close()


def test_full_size_C2_parity_and_properties():
    """BASELINE.json configs[1]: B=32, T_in=128, T_mel=512 at full widths against the float64 oracle."""
    B, T_in, r, n, ns, mt = O.CONFIGS["C2"]
    ohp = O.OracleHParams(max_iters=n, reduction_factor=r)
    w = O.init_weights(ohp, 1, 1234 + 1)
    ids, L = O.synthetic_inputs(B, T_in, 1234 + 1)
    m = build_model(ohp, w)
    hip = _run(m, ids, L)
    ref = O.forward(w, ohp, ids, L)
    assert hip[0].shape == (32, 512, 80) and hip[1].shape == (32, 512, 1025) and hip[2].shape == (32, 128, 128)
    _check(hip, ref, tol=1e-3)
    # the three arithmetic levels of the bench line against the same oracle run: bf16 x 3 (default, above), every feed-forward layer on the
    # six-product split (operands split three ways: fp32-grade products on the bf16 pipe; taco_debug_set_bf3 65) and every contraction exact fp32
    errs = {"bf16x3": [maxabs(hip[0], ref["mel"]), maxabs(hip[1], ref["linear"])]}
    for name, mode in (("bf16x6", 65), ("exact fp32", 0)):
        m._lib.taco_debug_set_bf3(m._handle, mode, 0)
        m._plans.clear()
        got = _run(m, ids, L)
        _check(got, ref, tol=1e-3)
        errs[name] = [maxabs(got[0], ref["mel"]), maxabs(got[1], ref["linear"])]
    m._lib.taco_debug_set_bf3(m._handle, 1, 0)
    print("C2 max |err| vs the float64 oracle (mel, linear):", {k: ["%.2e" % x for x in v] for k, v in errs.items()})
    # the six-product split is fp32-grade: as close to the oracle as the exact-fp32 engine (both ~1e-6: fp32 rounding of the recurrences), never worse than the default
    assert errs["bf16x6"][0] <= max(2 * errs["exact fp32"][0], 5e-6) and errs["bf16x6"][1] <= max(2 * errs["exact fp32"][1], 5e-6)


def test_the_one_excused_tie_at_C2_is_a_tie_in_exact_fp32_too():
    """`argmax_match` excuses a differing argmax when the oracle's top two values are closer than 4e-6 of the peak (tests/util.py).
    With tools/parity_margins.py's C2 seed that happens at one step.  This test shows the tie is in the problem, not bought by the
    split-bf16 arithmetic: wherever the default build is excused, the build with EVERY contraction in exact fp32
    (taco_debug_set_bf3 off) computes the two contested positions to within the same margin of each other."""
    from util import argmax_detail
    B, T_in, r, n, ns, mt = O.CONFIGS["C2"]
    ohp = O.OracleHParams(max_iters=n, reduction_factor=r)
    w = O.init_weights(ohp, 1, 1234 + 2 + 1)
    ids, L = O.synthetic_inputs(B, T_in, 99 + 1)
    ref = O.forward(w, ohp, ids, L)["alignments"]
    m = build_model(ohp, w)
    al = _run(m, ids, L)[2]
    m._lib.taco_debug_set_bf3(m._handle, 0, 0)
    m._plans.clear()
    al_x = _run(m, ids, L)[2]
    d, dx = argmax_detail(al, ref), argmax_detail(al_x, ref)
    print("default build:", d, " exact fp32:", dx)
    assert d["mismatch"] == 0 and dx["mismatch"] == 0
    peak = ref.max(axis=1)
    contested = (al.argmax(1) != ref.argmax(1)) & (peak > 1e-6)
    for b, t in zip(*np.nonzero(contested)):
        j_ref, j_hip = int(ref[b, :, t].argmax()), int(al[b, :, t].argmax())
        gap_oracle = (ref[b, j_ref, t] - ref[b, j_hip, t]) / peak[b, t]
        gap_exact = abs(float(al_x[b, j_ref, t]) - float(al_x[b, j_hip, t])) / peak[b, t]
        print("row %d step %d: positions %d / %d, oracle gap %.2e, exact-fp32 gap %.2e (relative to the peak %.3e)" % (b, t, j_ref, j_hip, gap_oracle, gap_exact, peak[b, t]))
        assert gap_oracle <= 4e-6 and gap_exact <= 4e-6


def test_full_size_C3_deepvoice_multispeaker():
    B, T_in, r, n, ns, mt = O.CONFIGS["C3"]
    ohp = O.OracleHParams(max_iters=16, reduction_factor=r, model_type=mt)      # 16 steps keep the oracle quick
    w = O.init_weights(ohp, ns, 1234 + 2)
    ids, L = O.synthetic_inputs(B, T_in, 1234 + 2, ragged=True)
    spk = (np.arange(B) % ns).astype(np.int32)
    m = build_model(ohp, w, num_speakers=ns)
    _check(_run(m, ids, L, spk), O.forward(w, ohp, ids, L, speaker_id=spk, num_speakers=ns), tol=1e-3)


def test_C1_single_utterance_r5():
    B, T_in, r, n, ns, mt = O.CONFIGS["C1"]
    ohp = O.OracleHParams(max_iters=40, reduction_factor=r)
    w = O.init_weights(ohp, 1, 1234)
    ids, L = O.synthetic_inputs(B, T_in, 1234)
    m = build_model(ohp, w)
    _check(_run(m, ids, L), O.forward(w, ohp, ids, L), tol=1e-3)


def test_long_form_C5_shapes_and_roundtrip_properties():
    """C5 (B=8, T_in=512, T_mel=4000) at full size is too slow for the oracle; check size-independent
    properties instead: rows independent (a 2-row sub-batch reproduces its rows), alignments of the
    monotonic mechanism are non-negative with per-step mass <= 1, outputs finite."""
    B, T_in, r, n, ns, mt = O.CONFIGS["C5"]
    ohp = O.OracleHParams(max_iters=n, reduction_factor=r)
    w = O.init_weights(ohp, 1, 1234 + 4)
    ids, L = O.synthetic_inputs(B, T_in, 1234 + 4, ragged=True)
    m = build_model(ohp, w)
    mel, lin, al = _run(m, ids, L)
    assert mel.shape == (8, 4000, 80) and lin.shape == (8, 4000, 1025) and al.shape == (8, 512, 1000)
    assert np.isfinite(mel).all() and np.isfinite(lin).all()
    assert al.min() >= 0 and al.sum(1).max() <= 1 + 1e-4
    mel2, lin2, al2 = _run(m, ids[3:5], L[3:5])
    assert maxabs(mel2, mel[3:5]) < 1e-4 and maxabs(al2, al[3:5]) < 1e-5


@pytest.mark.parametrize("mt,ns", [("single", 1), ("deepvoice", 3)])
def test_plan_pool_lanes_are_independent_and_exact(mt, ns):
    """Four forwards in flight on four streams (PlanPool), each over DIFFERENT inputs, twice around the lanes:
    every result must be bit-identical to the same request served alone, and within tolerance of the checker."""
    ohp = tiny_hp(model_type=mt, speaker_embedding_size=4) if ns > 1 else tiny_hp()
    w = O.init_weights(ohp, ns, 91)
    m = build_model(ohp, w, num_speakers=ns)
    B, T_in = 4, 12
    reqs = []
    for i in range(8):
        ids, L = O.synthetic_inputs(B, T_in, 300 + i, ragged=(i % 2 == 1))
        spk = ((np.arange(B) + i) % ns).astype(np.int32) if ns > 1 else None
        reqs.append((ids, L, spk))
    alone = [_run(m, ids, L, spk, honor_stop=False) for ids, L, spk in reqs]
    pool = m.plan_pool(B, T_in, lanes=4)
    assert pool.engine == "launch-per-stage"          # whole-chip persistent kernels do not share the chip: multi-lane pools avoid them
    got = [None] * len(reqs)
    for base in (0, 4):
        lanes = [pool.submit(*reqs[base + k]) for k in range(4)]
        assert sorted(lanes) == [0, 1, 2, 3]
        with pytest.raises(RuntimeError):
            pool.submit(*reqs[0], lane=lanes[0])            # result not collected yet
        for k in reversed(range(4)):                        # collect out of order
            r = pool.result(lanes[k])
            got[base + k] = (r["mel"].cpu().numpy(), r["linear"].cpu().numpy(), r["alignments"].cpu().numpy())
    with pytest.raises(RuntimeError):
        pool.result(0)
    for i, (ids, L, spk) in enumerate(reqs):
        for a, b in zip(alone[i], got[i]):
            assert np.array_equal(a, b), "request %d differs between pool and serial" % i
    ref = O.forward(w, ohp, reqs[5][0], reqs[5][1], speaker_id=reqs[5][2], num_speakers=ns, honor_stop=False)
    _check(got[5], ref)
    m.check_device_errors()
    pool.close()


@pytest.mark.parametrize("mt,ns", [("single", 1), ("deepvoice", 3)])
def test_plan_pool_coalesces_requests_into_one_forward(mt, ns):
    """PlanPool(coalesce=2): two requests of B rows ride through one plan of 2B rows; five requests over two lanes (the last lane
    is flushed half full).  Every request gets its own rows and its own stop step, equal to serving it alone (to rounding: a
    layer may pick another tile shape for the larger row count) and within tolerance of the checker."""
    ohp = tiny_hp(model_type=mt, speaker_embedding_size=4) if ns > 1 else tiny_hp()
    w = O.init_weights(ohp, ns, 93)
    m = build_model(ohp, w, num_speakers=ns)
    B, T_in = 3, 11
    reqs = []
    for i in range(5):
        ids, L = O.synthetic_inputs(B, T_in, 400 + i, ragged=(i % 2 == 0))
        spk = ((np.arange(B) + i) % ns).astype(np.int32) if ns > 1 else None
        reqs.append((ids, L, spk))
    single = m.plan_pool(B, T_in, lanes=1)
    assert single.engine == "persistent" and m.plan_pool(B, T_in, lanes=1, engine="launch").engine == "launch-per-stage"
    alone = []
    for rq in reqs:
        r = single.result(single.submit(*rq))
        alone.append((r["mel"].cpu().numpy(), r["linear"].cpu().numpy(), r["alignments"].cpu().numpy(), r["stop_step"]))
    single.close()
    pool = m.plan_pool(B, T_in, lanes=2, coalesce=2)
    tickets = [pool.submit(*rq) for rq in reqs[:4]]
    assert tickets == [(0, 0), (0, 1), (1, 0), (1, 1)]
    got = {}
    for i in (3, 0, 2, 1):                                   # collect out of order
        got[i] = pool.result(tickets[i])
    t4 = pool.submit(*reqs[4])                               # half-filled lane: result() flushes it
    assert t4 == (0, 0)
    got[4] = pool.result(t4)
    with pytest.raises(RuntimeError):
        pool.result((0, 1))                                  # that slot held no request
    for i in range(5):
        r = got[i]
        for a, b in zip(alone[i][:3], (r["mel"].cpu().numpy(), r["linear"].cpu().numpy(), r["alignments"].cpu().numpy())):
            assert a.shape == b.shape and maxabs(a, b) < 2e-5, "request %d" % i
        assert r["stop_step"] == alone[i][3], (i, r["stop_step"], alone[i][3])
    ref = O.forward(w, ohp, reqs[3][0], reqs[3][1], speaker_id=reqs[3][2], num_speakers=ns, honor_stop=False)
    _check((got[3]["mel"].cpu().numpy(), got[3]["linear"].cpu().numpy(), got[3]["alignments"].cpu().numpy()), ref)
    m.check_device_errors()
    pool.close()


@pytest.mark.parametrize("mt,ns", [("single", 1), ("simple", 3)])
def test_prenet_layer_folded_into_frame_projection_is_the_same_function(mt, ns):
    """Default: layer 1 of the decoder prenet of step t+1 comes out of step t's frame-projection launch (composite weights
    Wf[:, last frame] . W1[frame rows]); with the fold switched off it is its own launch.  Same function up to rounding."""
    ohp = tiny_hp(model_type=mt, speaker_embedding_size=4, max_iters=11) if ns > 1 else tiny_hp(max_iters=11)
    w = O.init_weights(ohp, ns, 77)
    ids, L = O.synthetic_inputs(5, 12, 78, ragged=True)
    spk = (np.arange(5) % ns).astype(np.int32) if ns > 1 else None
    m = build_model(ohp, w, num_speakers=ns)
    ref = O.forward(w, ohp, ids, L, speaker_id=spk, num_speakers=ns, honor_stop=False)
    folded = _run(m, ids, L, spk, honor_stop=False)
    _check(folded, ref)
    m._plans.clear()
    m._lib.taco_debug_set_fuse_prenet(m._handle, 0)
    plain = _run(m, ids, L, spk, honor_stop=False)
    _check(plain, ref)
    for a, b in zip(folded, plain):
        assert maxabs(a, b) < 2e-5
    m._lib.taco_debug_set_fuse_prenet(m._handle, 1)


@pytest.mark.parametrize("mt,ns", [("single", 1), ("simple", 3), ("deepvoice", 2)])
def test_concat_projection_folded_into_first_decoder_gru_is_the_same_function(mt, ns):
    """Default: the concat projection (rnn_wrappers.py:405-415, tacotron.py:166-170) is folded into the gates launch of decoder GRU 1
    (composite weights Wc . Wg_x; the projection output for the residual comes out of the same launch).  With the fold switched
    off it is its own launch.  Same function up to rounding; both against the oracle."""
    ohp = tiny_hp(model_type=mt, speaker_embedding_size=4, max_iters=11) if ns > 1 else tiny_hp(max_iters=11)
    w = O.init_weights(ohp, ns, 79)
    ids, L = O.synthetic_inputs(5, 12, 80, ragged=True)
    spk = (np.arange(5) % ns).astype(np.int32) if ns > 1 else None
    m = build_model(ohp, w, num_speakers=ns)
    ref = O.forward(w, ohp, ids, L, speaker_id=spk, num_speakers=ns, honor_stop=False)
    folded = _run(m, ids, L, spk, honor_stop=False)
    _check(folded, ref)
    m._plans.clear()
    taco_check = __import__("taco_amd")._lib.check
    taco_check(m._lib.taco_debug_set_fuse_concat(m._handle, 0))
    plain = _run(m, ids, L, spk, honor_stop=False)
    _check(plain, ref)
    for a, b in zip(folded, plain):
        assert maxabs(a, b) < 2e-5
    assert max(maxabs(a, b) for a, b in zip(folded, plain)) > 0        # the two paths really are different launches
    taco_check(m._lib.taco_debug_set_fuse_concat(m._handle, 1))


def test_edge_lengths_zero_and_full_and_no_eos():
    """synthesizer.py:120: input_lengths = argmax(ids == EOS): a row without EOS gets length 0 (its BiGRU output is all zero),
    a row whose first token is EOS too; rows at full length sit beside them in the same batch."""
    import taco_amd
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 51)
    rs = np.random.RandomState(52)
    ids = rs.randint(2, 80, size=(5, 10)).astype(np.int32)
    ids[1, 0] = 1                      # EOS first  -> length 0
    ids[2, 9] = 1                      # EOS last   -> length 9
    ids[3, 4] = 1; ids[3, 5:] = 0      # padded
    L = taco_amd.input_lengths_from_tokens(ids)          # row 0 and 4: no EOS at all -> 0
    assert list(L) == [0, 0, 9, 4, 0]
    m = build_model(ohp, w)
    hip = _run(m, ids, L, honor_stop=False)
    ref = O.forward(w, ohp, ids, L, honor_stop=False)
    _check(hip, ref)
    enc = m.encoder(ids, L).cpu().numpy()
    assert not enc[0].any() and not enc[1].any() and enc[2, :9].any() and not enc[2, 9:].any()


@pytest.mark.parametrize("B", [1, 64])
def test_batch_limits(B):
    ohp = tiny_hp(max_iters=3)
    w = O.init_weights(ohp, 1, 53)
    ids, L = O.synthetic_inputs(B, 7, 54, ragged=True)
    m = build_model(ohp, w)
    _check(_run(m, ids, L, honor_stop=False), O.forward(w, ohp, ids, L, honor_stop=False))


@pytest.mark.parametrize("B", [65, 96, 130])
def test_more_than_64_rows_run_as_passes_of_equal_size(B):
    """The reference puts no cap on the batch (synthesizer.py:120-131: whatever `texts` holds; eval.py:86-119 --batch_size): the C ABI
    serves B > 64 as ceil(B / 64) passes over one workspace (taco_forward_infer and taco_plan_create alike).  Against the oracle on
    the whole batch, ragged lengths, and against the same rows served alone (rows never interact at inference)."""
    ohp = tiny_hp(max_iters=3)
    w = O.init_weights(ohp, 1, 55)
    ids, L = O.synthetic_inputs(B, 7, 56, ragged=True)
    m = build_model(ohp, w)
    hip = _run(m, ids, L, honor_stop=False)                 # Tacotron.run replays a captured plan: taco_plan_create
    _check(hip, O.forward(w, ohp, ids, L, honor_stop=False))
    rows = (B + (B + 63) // 64 - 1) // ((B + 63) // 64)     # rows per pass
    alone = _run(m, ids[rows:2 * rows], L[rows:2 * rows], honor_stop=False)
    for a, b in zip(hip, alone):
        assert np.array_equal(a[rows:2 * rows], b)          # the second pass == its rows as a batch of their own, to the bit


def test_more_than_64_rows_on_the_persistent_engine_and_the_eager_entry_point():
    """Reference widths (the persistent decoder and scans), 96 rows = two passes of 48: eager taco_forward_infer == the captured plan ==
    each pass's rows served alone, to the bit; the stop word is the maximum over the passes."""
    import torch
    import taco_amd
    from util import dev, ptr, stream
    ohp = O.OracleHParams(max_iters=3)
    w = O.init_weights(ohp, 1, 57)
    B, T_in, n, r = 96, 16, 3, ohp.reduction_factor
    ids, L = O.synthetic_inputs(B, T_in, 58, ragged=True)
    m = build_model(ohp, w)
    assert "96 rows = 2 passes of 48" in m.engine_plan(B, T_in), m.engine_plan(B, T_in)
    mel_g, lin_g, al_g = _run(m, ids, L, honor_stop=False)
    assert m.stop_step == n
    for lo in (0, 48):
        mel_a, lin_a, al_a = _run(m, ids[lo:lo + 48], L[lo:lo + 48], honor_stop=False)
        assert np.array_equal(mel_g[lo:lo + 48], mel_a) and np.array_equal(lin_g[lo:lo + 48], lin_a) and np.array_equal(al_g[lo:lo + 48], al_a)
    ref = O.forward(w, ohp, ids[44:52], L[44:52], honor_stop=False)      # eight rows across the seam between the passes
    assert maxabs(mel_g[44:52], ref["mel"]) < 1e-3 and maxabs(lin_g[44:52], ref["linear"]) < 1e-3 and maxabs(al_g[44:52], ref["alignments"]) < 1e-3
    mel = torch.empty((B, n * r, ohp.num_mels), device="cuda")
    lin = torch.empty((B, n * r, ohp.num_freq), device="cuda")
    al = torch.empty((B, T_in, n), device="cuda")
    stop = torch.zeros((1,), dtype=torch.int32, device="cuda")
    nb = int(m._lib.taco_workspace_bytes(m._handle, B, T_in, n))
    assert nb < 1.2 * int(m._lib.taco_workspace_bytes(m._handle, 48, T_in, n)) + 4096          # the workspace of ONE pass
    ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    idd, Ld = dev(ids), dev(L)
    taco_amd._lib.check(m._lib.taco_forward_infer(m._handle, stream(), ptr(idd), ptr(Ld), ptr(None), B, T_in, n,
                                                  ptr(None), ptr(mel), ptr(lin), ptr(al), ptr(stop), ptr(ws), nb))
    torch.cuda.synchronize()
    assert np.array_equal(mel.cpu().numpy(), mel_g) and np.array_equal(lin.cpu().numpy(), lin_g) and np.array_equal(al.cpu().numpy(), al_g)
    assert int(stop.item()) == n
    # every row finished at step 1 in both passes (helpers.py:29): the combined stop step is 1
    w2 = {k: v.copy() for k, v in w.items()}
    w2["decoder/frame_projection/kernel"][:] = 0
    w2["decoder/frame_projection/bias"][:] = 0
    m2 = build_model(ohp, w2)
    _run(m2, ids, L)
    assert m2.stop_step == 1


def test_bad_shapes_are_errors():
    ohp = tiny_hp(max_iters=2)
    m = build_model(ohp, O.init_weights(ohp, 1, 55))
    ids, L = O.synthetic_inputs(3, 5, 56)
    with pytest.raises(Exception):
        m.run(inputs=ids[0], input_lengths=L[:1])          # rank-1 inputs


def test_single_decoder_step_and_longest_supported_input():
    """n_steps = 1 (initial alignments only) and T_in = 2048, the attention kernel's LDS limit; 2049 is refused."""
    import taco_amd
    ohp = tiny_hp(max_iters=1)
    w = O.init_weights(ohp, 1, 57)
    m = build_model(ohp, w)
    ids, L = O.synthetic_inputs(2, 9, 58)
    _check(_run(m, ids, L, honor_stop=False), O.forward(w, ohp, ids, L, honor_stop=False))
    ohp2 = tiny_hp(max_iters=2)
    w2 = O.init_weights(ohp2, 1, 59)
    m2 = build_model(ohp2, w2)
    ids, L = O.synthetic_inputs(2, 2048, 60, ragged=True)
    _check(_run(m2, ids, L, honor_stop=False), O.forward(w2, ohp2, ids, L, honor_stop=False))
    ids, L = O.synthetic_inputs(1, 2049, 61)
    with pytest.raises(taco_amd._lib.TacoError):
        m2.run(inputs=ids, input_lengths=L)


@pytest.mark.parametrize("atype", ["bah_mon", "bah", "bah_norm"])
@pytest.mark.parametrize("slices", [4, 3])
def test_split_attention_for_small_batches_is_the_same_function(atype, slices):
    """k_att_scores + k_att_context (rows spread over several workgroups; the default for B <= 16, T_in >= 256) vs the
    one-workgroup-per-row kernel and vs the oracle, incl. manual alignments and ragged slice boundaries (T_in = 37)."""
    ohp = tiny_hp(attention_type=atype, max_iters=9)
    w = O.init_weights(ohp, 1, 71)
    ids, L = O.synthetic_inputs(3, 37, 72, ragged=True)
    ref = O.forward(w, ohp, ids, L, honor_stop=False)
    m = build_model(ohp, w)
    m._lib.taco_debug_set_att_split(m._handle, 0)
    one = _run(m, ids, L, honor_stop=False)
    m._plans.clear()
    m._lib.taco_debug_set_att_split(m._handle, slices)
    split = _run(m, ids, L, honor_stop=False)
    _check(one, ref); _check(split, ref)
    for a, b in zip(one, split):
        assert maxabs(a, b) < 2e-5
    man = np.random.RandomState(5).rand(3, 9, 37).astype(np.float32)
    man /= man.sum(-1, keepdims=True)
    m._plans.clear()
    got = _run(m, ids, L, manual_alignments=man, is_manual_attention=True, honor_stop=False)
    _check(got, O.forward(w, ohp, ids, L, manual_alignments=man, honor_stop=False), tol=2e-4)
    m._lib.taco_debug_set_att_split(m._handle, -1)


@pytest.mark.parametrize("model_type,ns,B,T_in,n", [("single", 1, 3, 37, 5), ("deepvoice", 3, 5, 70, 3), ("single", 1, 2, 130, 2)])
def test_pointwise_chain_kernel_matches_the_oracle_and_the_per_layer_path(model_type, ns, B, T_in, n):
    """csrc/taco_chain.h: [dense ->] highway x 4 -> BiGRU input projection of a CBHG as ONE launch (modules.py:72-96).  Reference
    widths (the kernel exists for widths 128 / 256), row counts that are not multiples of the 64-row tile, T below and above the
    tile height (the time-reversed store of the backward direction crosses batch rows inside a tile), ragged lengths incl. 0,
    the deepvoice residual in front of the encoder's highways: both CBHG stages against the float64 oracle, and the fused tail
    against one launch per layer."""
    import torch
    ohp = O.OracleHParams(max_iters=n, model_type=model_type)
    w = O.init_weights(ohp, ns, 911)
    ids, L = O.synthetic_inputs(B, T_in, 912, ragged=True)
    L = L.copy(); L[-1] = 0 if B > 2 else L[-1]
    spk = (np.arange(B) % ns).astype(np.int32) if ns > 1 else None
    taps = {}
    ref = O.forward(w, ohp, ids, L, speaker_id=spk, num_speakers=ns, taps=taps)
    m = build_model(ohp, w, num_speakers=ns)
    got = {}
    for flag, name in ((1, "chain"), (5, "per-layer")):           # taco_debug_set_bf3 bit 2: one launch per point-wise layer
        m._lib.taco_debug_set_bf3(m._handle, flag, 0)
        enc = m.encoder(ids, L, spk)
        lin, post = m.postnet(ref["mel"], return_post=True, speaker_id=spk)
        torch.cuda.synchronize()
        got[name] = (enc.cpu().numpy(), post.cpu().numpy(), lin.cpu().numpy())
    m._lib.taco_debug_set_bf3(m._handle, 1, 0)
    for name, (enc, post, lin) in got.items():
        assert maxabs(enc, taps["encoder"]) < 1e-4, name
        assert maxabs(post, taps["post"][..., :post.shape[-1]]) < 2e-4 and maxabs(lin, ref["linear"]) < 2e-4, name
    for a, b in zip(got["chain"], got["per-layer"]):
        assert maxabs(a, b) < 2e-5


@pytest.mark.parametrize("B,T_in,T_mel,oracle", [(3, 37, 20, True), (2, 130, 280, True), (5, 257, 132, True), (1, 64, 400, True), (9, 128, 129, True),
                                                 (16, 128, 256, False), (32, 128, 512, False), (43, 100, 384, False), (64, 128, 512, False)])
def test_fused_cbhg_front_matches_the_oracle_and_the_two_launch_path(B, T_in, T_mel, oracle):
    """csrc/taco_front.h: conv bank -> max-pool -> proj_1 of a CBHG as ONE launch (modules.py:35-59; the bank tensor never reaches
    memory, the contraction of proj_1 is split over parts of the bank's channels and summed in a fixed order).  Reference widths
    (the kernel exists for the encoder's 16 x 128 over 128 channels and the post-net's 8 x 256 over 80), frame counts below, at and
    above the 128-frame tile incl. one frame more than a tile, batch sizes that give 1 ... 16 parts, ragged lengths.  Both CBHG
    stages and the linear head behind them (csrc/taco_head.h: row sweep with a vector-ALU tail column, row counts that are not multiples
    of its 64-row tile) against the float64 oracle's own stage functions (the large shapes, where NumPy would take minutes, only against the
    two-launch path: C2 itself is held to the oracle by test_full_size_C2_parity_and_properties), the fused front against bank and
    proj_1 as two launches, and twice the same call bit for bit (no atomics)."""
    import torch
    ohp = O.OracleHParams(max_iters=max(2, T_mel // 4))
    w = O.init_weights(ohp, 1, 1234)
    ids, L = O.synthetic_inputs(B, T_in, 77, ragged=True)
    rs = np.random.RandomState(78)
    mel = rs.uniform(-0.2, 1.2, (B, T_mel, ohp.num_mels))
    f64 = lambda a: np.asarray(a, np.float64)
    if oracle:
        pre = O.prenet(f64(w["embedding"])[ids], w, "prenet", ohp.enc_prenet_sizes)
        enc_ref = O.cbhg(pre, L.astype(np.int64), w, "encoder_cbhg", ohp.enc_bank_size, ohp.enc_maxpool_width, ohp.enc_highway_depth, ohp.enc_proj_sizes)
        post_ref = O.cbhg(f64(mel), None, w, "post_cbhg", ohp.post_bank_size, ohp.post_maxpool_width, ohp.post_highway_depth, ohp.post_proj_sizes)
        lin_ref = O.dense(post_ref, w, "linear")
    m = build_model(ohp, w)
    got = {}
    # taco_debug_set_bf3 bit 3: front off; bit 4: fused front, but proj_1's epilogue and proj_2 as launches of their own instead of the
    # point-wise chain's fused entry (csrc/taco_chain.h)
    # bit 5: the linear head on k_gemm_bf3's 64 x 256 tiles instead of the row sweep (csrc/taco_head.h; runs from 256 rows on)
    for flag, name in ((1, "fused"), (17, "fused front, separate proj_2"), (9, "two launches"), (33, "linear head on GEMM tiles"), (1, "fused again")):
        m._lib.taco_debug_set_bf3(m._handle, flag, 0)
        enc = m.encoder(ids, L)
        lin, post = m.postnet(mel, return_post=True)
        torch.cuda.synchronize()
        got[name] = (enc.cpu().numpy(), post.cpu().numpy(), lin.cpu().numpy())
    m._lib.taco_debug_set_bf3(m._handle, 1, 0)
    m.check_device_errors()
    for name, (enc, post, lin) in got.items():
        if not oracle:
            break
        e = (maxabs(enc, enc_ref), maxabs(post, post_ref), maxabs(lin, lin_ref))
        print("%s: encoder %.2e  post %.2e  linear %.2e" % ((name,) + e))
        assert e[0] < 1e-4 and e[1] < 2e-4 and e[2] < 2e-4, (name, e)
    for other in ("two launches", "fused front, separate proj_2", "linear head on GEMM tiles"):
        d = [maxabs(a, b) for a, b in zip(got["fused"], got[other])]
        print("fused vs %s: encoder %.2e  post %.2e  linear %.2e" % ((other,) + tuple(d)))
        assert max(d) < 3e-5, (other, d)
    for a, b in zip(got["fused"], got["fused again"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("atype,bias,B,T_in,n", [("bah_mon", 1.5, 32, 128, 128), ("bah", None, 32, 128, 128), ("bah_norm", None, 8, 128, 128),
                                                 ("bah_mon", 6.0, 2, 512, 1000), ("bah", None, 2, 512, 1000)])
def test_alignment_argmax_is_compared_on_every_step_of_a_full_horizon(atype, bias, B, T_in, n):
    """`north_star`: alignments bit-identical in argmax.  With Glorot weights and attention_score_bias 0 the 'parallel' monotonic
    recurrence loses its mass after ~85 of 128 steps (p ~ 0.5: the exclusive cumprod of (1 - p) falls under the 1e-10 clip of
    monotonic_attention ~33 positions past the start, TF-sem A.10), and argmax_match masks every step whose oracle peak is <= 1e-6 --
    a third of the steps of the C2 test.  Here the criterion bites on (nearly) EVERY step of a full horizon: at the C2 size and on
    1000-step rows (C5's horizon), with the monotonic mechanism under a score bias that keeps the attended position's mass alive
    (p near 1: the mass advances a fraction of a position per step) and with the two softmax mechanisms, whose peak is >= 1 / T_in
    by construction and whose argmax wanders over the whole input.  Asserts masked-by-floor < 5 % and zero mismatches."""
    from util import argmax_detail
    ohp = O.OracleHParams(max_iters=n, attention_type=atype)
    w = O.init_weights(ohp, 1, 4242)
    if bias is not None:
        w["attention/attention_score_bias"] = np.array(bias, np.float32)
    ids, L = O.synthetic_inputs(B, T_in, 4243, ragged=True)
    ref = O.forward(w, ohp, ids, L, honor_stop=False)
    m = build_model(ohp, w)
    hip = _run(m, ids, L, honor_stop=False)
    d = argmax_detail(hip[2], ref["alignments"])
    moved = int((np.diff(ref["alignments"].argmax(1), axis=1) != 0).sum())
    print("%s, bias %s, B=%d, T_in=%d, %d steps: masked by the floor %.1f %% of %d steps; the oracle's argmax changes position %d times"
          % (atype, bias, B, T_in, n, 100.0 * d["masked_by_floor"] / d["steps"], d["steps"], moved))
    assert d["masked_by_floor"] < 0.05 * d["steps"], d
    _check(hip, ref, tol=1e-3)


# ---- full size x full horizon against oracle outputs committed as fixtures (tests/golden/make_full_size_golden.py) ----
def _golden(name):
    p = os.path.join(os.path.dirname(__file__), "golden", name)
    if not os.path.exists(p):
        pytest.skip("%s not generated (python tests/golden/make_full_size_golden.py)" % name)
    return np.load(p)


def test_C3_all_32_rows_all_128_steps_against_the_committed_oracle_outputs():
    """BASELINE.json configs[2] at full size AND full horizon: 32 rows x 128 decoder steps, deepvoice, 4 speakers (round 5 ran 16 steps, or 8 of
    the rows).  The oracle's float64 outputs are a fixture (40 s of CPU when made); weights and inputs are re-created from its seed."""
    import make_full_size_golden as G
    g = _golden("full_C3.npz")
    hp, ns, seed, ids, L, spk = G.c3_case()
    assert int(g["seed"]) == seed and np.array_equal(g["inputs"], ids) and np.array_equal(g["input_lengths"], L) and np.array_equal(g["speaker_id"], spk)
    m = build_model(hp, O.init_weights(hp, ns, seed), num_speakers=ns)
    assert "k_decoder_xcd<4>" in m.engine_plan(32, 128)
    mel, lin, al = _run(m, ids, L, spk, honor_stop=False)
    st = int(g["linear_stride"])
    errs = {"mel": maxabs(mel, g["mel"]), "linear": maxabs(lin[:, ::st], g["linear"]), "alignments": maxabs(al, g["alignments"])}
    print("C3 full size x full horizon vs the oracle fixture:", errs)
    assert max(errs.values()) < 1e-3, errs
    n, bad = argmax_match(al, g["alignments"].astype(np.float64))
    assert bad == 0 and n > 0.9 * 32 * 128, (n, bad)           # (95 % of the steps have a peak above the floor of tests/util.py)


def test_C5_all_8_rows_all_1000_steps_against_the_committed_oracle_outputs():
    """BASELINE.json configs[4] at full size and full horizon: 8 rows x T_in 512 x 1000 decoder steps (round 5 compared 2 of the 8 rows).  The fixture
    holds every 4th decoder step's frames, the linear output on a frame stride, the alignment arg-max / peak / runner-up of EVERY step and every
    8th step's alignments."""
    import make_full_size_golden as G
    g = _golden("full_C5.npz")
    hp, seed, ids, L = G.c5_case()
    assert int(g["seed"]) == seed and np.array_equal(g["inputs"], ids) and np.array_equal(g["input_lengths"], L)
    m = build_model(hp, O.init_weights(hp, 1, seed))
    mel, lin, al = _run(m, ids, L, honor_stop=False)
    B, n, r, M = 8, hp.max_iters, hp.reduction_factor, hp.num_mels
    ss, ls, as_ = int(g["step_stride"]), int(g["linear_stride"]), int(g["align_stride"])
    errs = {"mel": maxabs(mel.reshape(B, n, r * M)[:, ::ss], g["mel_steps"]), "linear": maxabs(lin[:, ::ls], g["linear"]),
            "alignments": maxabs(al[:, :, ::as_], g["alignments"])}
    print("C5 full size x full horizon vs the oracle fixture:", errs)
    assert max(errs.values()) < 1e-3, errs
    # arg-max of every step whose oracle peak is above the floor and is not tied with the runner-up at fp32 resolution
    peak, second = g["align_peak"], g["align_second"]
    sel = (peak > 1e-20) & ((peak - second) > 2e-5 * peak)           # (tests/util.py: the floor and the tie criterion of argmax_match)
    got = al.argmax(axis=1)
    print("C5 arg-max: %d of %d steps compared (the monotonic mass of random weights leaks past the last encoder step)" % (int(sel.sum()), sel.size))
    assert sel.sum() > 1200 and np.array_equal(got[sel], g["align_argmax"][sel].astype(got.dtype))

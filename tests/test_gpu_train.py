"""Training path through the C ABI vs the float64 checkers: the NumPy oracle's training-mode
forward and tests/torch_formulation.py's reverse-mode autograd of the same graph."""
import numpy as np
import pytest

import taco_oracle as O
import torch_formulation as TF
from util import tiny_hp, to_product_hp, maxabs

pytestmark = pytest.mark.gpu


def _setup(atype="bah_mon", B=3, T_in=9, T_out=12, seed=5, ragged=True, **kw):
    hp = tiny_hp(attention_type=atype, **kw)
    w = O.init_weights(hp, 1, seed)
    ids, L = O.synthetic_inputs(B, T_in, seed + 6, ragged=ragged)
    rs = np.random.RandomState(seed + 1)
    mt = rs.rand(B, T_out, hp.num_mels)
    lt = rs.rand(B, T_out, hp.num_freq)
    co = rs.uniform(0.5, 1.5, size=B)
    return hp, w, ids, L, mt, lt, co


def _trainer(hp, w):
    import taco_amd
    return taco_amd.Trainer(to_product_hp(hp), w)


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("atype", ["bah_mon", "bah", "bah_norm"])
def test_training_forward_matches_oracle(atype):
    import torch
    hp, w, ids, L, mt, lt, co = _setup(atype)
    r = hp.reduction_factor
    upd = {}
    ref = O.forward(w, hp, ids, L, n_steps=mt.shape[1] // r, honor_stop=False, teacher_frames=mt[:, r - 1::r], training=True, bn_updates=upd)
    tr = _trainer(hp, w)
    losses = tr.forward_backward(ids, L, mt, lt, co, backward=False, keep_outputs=True)
    torch.cuda.synchronize()
    assert maxabs(tr.mel_outputs.cpu().numpy(), ref["mel"]) < 1e-4
    assert maxabs(tr.linear_outputs.cpu().numpy(), ref["linear"]) < 1e-4
    assert maxabs(tr.alignments.cpu().numpy(), ref["alignments"]) < 1e-4
    want = O.add_loss(ref["mel"], mt, ref["linear"], lt, co)
    got = losses.cpu().numpy()
    for i, k in enumerate(("loss", "mel_loss", "linear_loss", "loss_without_coeff")):
        assert abs(got[i] - want[k]) < 1e-5 * max(1.0, abs(want[k])), k
    # a forward-only pass (loss fetch, test model) leaves the BatchNorm moving averages alone: the reference runs UPDATE_OPS only
    # as a dependency of `optimize` (tacotron.py:334)
    now = tr.get_weights()
    for k in w:
        assert np.array_equal(now[k], np.asarray(w[k], np.float32)), k
    # a warm-up forward+backward with frozen statistics (what Trainer.capture does before recording the step) does not either
    tr.forward_backward(ids, L, mt, lt, co, backward=True, freeze_moving_averages=True)
    torch.cuda.synchronize()
    now = tr.get_weights()
    for k in w:
        assert np.array_equal(now[k], np.asarray(w[k], np.float32)), k
    # forward + backward: the moving averages are updated in the flat parameter buffer, once
    tr.forward_backward(ids, L, mt, lt, co, backward=True)
    torch.cuda.synchronize()
    now = tr.get_weights()
    assert len(upd) == 2 * (hp.enc_bank_size + hp.post_bank_size + len(hp.enc_proj_sizes) + len(hp.post_proj_sizes))
    for k, v in upd.items():
        assert maxabs(now[k], v) < 1e-5, k
    for k in w:
        if k not in upd:
            assert np.array_equal(now[k], np.asarray(w[k], np.float32)), k


@pytest.mark.parametrize("atype,ragged", [("bah_mon", True), ("bah", False), ("bah_norm", True)])
def test_gradients_match_autograd(atype, ragged):
    import torch
    hp, w, ids, L, mt, lt, co = _setup(atype, ragged=ragged)
    loss, g, _ = TF.train_grads(w, hp, ids, L, mt, lt, co)
    tr = _trainer(hp, w)
    losses = tr.forward_backward(ids, L, mt, lt, co)
    torch.cuda.synchronize()
    assert abs(float(losses[0]) - loss) < 1e-5
    got = tr.grad_dict()
    gn = np.sqrt(sum(float((v ** 2).sum()) for v in g.values()))
    bad = []
    for k, v in g.items():
        err = float(np.abs(got[k] - v).max())
        scale = max(float(np.abs(v).max()), 1e-3 * gn)
        if err > 2e-3 * scale:
            bad.append((k, err, float(np.abs(v).max())))
    assert not bad, bad[:8]
    for k in w:
        if k.endswith(("/moving_mean", "/moving_variance")):
            assert not got[k].any(), k
    tr.close()


def _grad_report(got, g):
    gn = np.sqrt(sum(float((v ** 2).sum()) for v in g.values()))
    worst = []
    for k, v in g.items():
        err = float(np.abs(got[k] - v).max())
        worst.append((err / max(float(np.abs(v).max()), 1e-3 * gn), k, err))
    worst.sort(reverse=True)
    return worst, gn


def test_gradients_long_input_clipped_monotonic_and_priority_loss():
    """T_in = 72: the exclusive cumprod of (1-p) falls below the 1e-10 clip of monotonic_attention('parallel'), so the
    clip's zero-gradient branch is exercised; prioritize_loss adds the 165 Hz..5 kHz band term (tacotron.py:283-296)."""
    import torch
    hp, w, ids, L, mt, lt, co = _setup("bah_mon", B=5, T_in=72, T_out=24, seed=9, prioritize_loss=True, max_iters=9)
    assert hp.prioritize_loss
    loss, g, out = TF.train_grads(w, hp, ids, L, mt, lt, co, prioritize_loss=True, sample_rate=24000)
    tr = _trainer(hp, w)
    losses = tr.forward_backward(ids, L, mt, lt, co, keep_outputs=True)
    torch.cuda.synchronize()
    assert abs(float(losses[0]) - loss) < 1e-5
    assert maxabs(tr.alignments.cpu().numpy(), out["alignments"]) < 1e-4
    worst, gn = _grad_report(tr.grad_dict(), g)
    assert worst[0][0] < 2e-3, worst[:5]


def test_gradients_mid_size_batch_of_17():
    """Wider model (reference widths / 4), odd batch, several 64-row tiles in every weight-gradient GEMM."""
    import torch
    hp = O.OracleHParams.scaled(4, num_mels=20, num_freq=65, enc_bank_size=6, post_bank_size=5, max_iters=10, reduction_factor=4)
    w = O.init_weights(hp, 1, 3)
    B, T_in, T_out = 17, 21, 36
    ids, L = O.synthetic_inputs(B, T_in, 4, ragged=True)
    rs = np.random.RandomState(8)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    loss, g, _ = TF.train_grads(w, hp, ids, L, mt, lt)
    tr = _trainer(hp, w)
    losses = tr.forward_backward(ids, L, mt, lt)
    torch.cuda.synchronize()
    assert abs(float(losses[0]) - loss) < 1e-5
    worst, gn = _grad_report(tr.grad_dict(), g)
    assert worst[0][0] < 2e-3, worst[:5]


def test_train_step_matches_clip_and_adam_of_the_checker_and_loss_goes_down():
    """train.py:217-219: one step = fwd + bwd + clip_by_global_norm(1.0) + Adam with the schedule's learning rate; the
    updated parameters must equal the checker's update applied to the checker's gradients.  Then 30 more steps on the
    same batch must reduce the loss."""
    import torch
    hp, w, ids, L, mt, lt, co = _setup("bah_mon", seed=12)
    loss0, g, _ = TF.train_grads(w, hp, ids, L, mt, lt, co)
    tr = _trainer(hp, w)
    names = [k for k, _ in tr.spec]
    flat = lambda d: np.concatenate([np.asarray(d[k], np.float64).reshape(-1) for k in names])
    gflat = flat({k: (g[k] if k in g else np.zeros_like(np.asarray(w[k], np.float64))) for k in names})
    lr = O.learning_rate(0, 0.002, 0, True)
    assert abs(tr.learning_rate - lr) < 1e-12 + 1e-6 * lr
    want, _, _, gn = O.adam_clip_step(flat(w), gflat, np.zeros_like(gflat), np.zeros_like(gflat), 1, lr)
    step, lwc = tr.train_step(ids, L, mt, lt, co)
    torch.cuda.synchronize()
    assert step == 1 and abs(float(tr.adam.gnorm) - gn) < 1e-4 * gn
    now = tr.get_weights()
    upd = {}
    O.forward(w, hp, ids, L, n_steps=mt.shape[1] // hp.reduction_factor, honor_stop=False,
              teacher_frames=mt[:, hp.reduction_factor - 1::hp.reduction_factor], training=True, bn_updates=upd)
    for k in names:
        ref = upd[k] if k in upd else want[tr.offsets[k][0]:tr.offsets[k][0] + tr.offsets[k][1]].reshape(np.shape(w[k]))
        # Adam's first update is lr * g/(|g| + eps): elements with |g| ~ 1e-8 are ill-conditioned -> compare loosely, in units of lr
        assert maxabs(now[k], ref) < 0.02 * lr + 1e-6, k
    first = float(lwc)
    for _ in range(30):
        step, lwc = tr.train_step(ids, L, mt, lt, co)
    torch.cuda.synchronize()
    assert step == 31 and float(lwc) < first - 0.005      # warm-up schedule: lr is only 5e-7 * step here
    assert np.isfinite(tr.params.cpu().numpy()).all()
    tr.close()


def test_training_rejects_unsupported_configurations():
    import taco_amd
    hp, w, ids, L, mt, lt, co = _setup()
    tr = _trainer(hp, w)
    with pytest.raises(taco_amd._lib.TacoError):
        tr.forward_backward(ids, L, mt[:, :11], lt[:, :11])          # T_out not a multiple of r
    with pytest.raises(taco_amd._lib.TacoError):
        big = np.zeros((3, hp.reduction_factor * (hp.max_iters + 1), hp.num_mels))
        tr.forward_backward(ids, L, big, np.zeros((3, big.shape[1], hp.num_freq)))   # T_out / r > max_iters


def test_tacotron_training_surface_mirrors_train_py():
    """train.py:145-166,215-219 through the model object: initialize(..., targets) + add_loss + add_optimizer, a test model
    sharing the variables with rnn_decoder_test_mode, steps via train_step, then the trained weights drive the inference model."""
    import torch
    import taco_amd
    hp, w, ids, L, mt, lt, co = _setup("bah_mon", seed=21)
    php = to_product_hp(hp)
    model = taco_amd.create_model(php)
    model.load_weights(w)
    model.initialize(ids, L, 1, None, mel_targets=mt, linear_targets=lt, loss_coeff=co, is_randomly_initialized=True)
    model.add_loss()
    model.add_optimizer(0)
    ref = O.forward(w, hp, ids, L, n_steps=mt.shape[1] // hp.reduction_factor, honor_stop=False,
                    teacher_frames=mt[:, hp.reduction_factor - 1::hp.reduction_factor], training=True)
    want = O.add_loss(ref["mel"], mt, ref["linear"], lt, co)
    assert abs(float(model.loss) - want["loss"]) < 1e-5 and abs(float(model.loss_without_coeff) - want["loss_without_coeff"]) < 1e-5
    assert maxabs(model.linear_outputs.cpu().numpy(), ref["linear"]) < 1e-4
    assert abs(model.learning_rate - O.learning_rate(0)) < 1e-9
    # the test model: same variables, decoder fed its own outputs (helpers.py:63-64); forward/loss only
    test_model = taco_amd.create_model(php).share_variables_with(model)
    test_model.initialize(ids, L, 1, None, mel_targets=mt, linear_targets=lt, loss_coeff=co, rnn_decoder_test_mode=True)
    test_model.add_loss()
    fb = O.forward(w, hp, ids, L, n_steps=mt.shape[1] // hp.reduction_factor, honor_stop=False, training=True)
    assert maxabs(test_model.mel_outputs.cpu().numpy(), fb["mel"]) < 1e-4
    assert abs(float(test_model.loss) - O.add_loss(fb["mel"], mt, fb["linear"], lt, co)["loss"]) < 1e-5
    with pytest.raises(taco_amd._lib.TacoError):
        test_model.train_step()
    losses = []
    for _ in range(5):
        step, lwc = model.train_step(ids, L, mt, lt, co)
        losses.append(float(lwc))
    assert step == 5 and losses[-1] < losses[0]
    # trained weights -> inference model
    tw = model.trained_weights()
    inf = taco_amd.create_model(php)
    inf.load_weights(tw)
    inf.initialize(None, None, 1, None)
    lin, al = inf.run(inputs=ids, input_lengths=L, honor_stop=False)
    torch.cuda.synchronize()
    chk = O.forward(tw, hp, ids, L, honor_stop=False)
    assert maxabs(lin.cpu().numpy(), chk["linear"]) < 1e-3


@pytest.mark.parametrize("B,T_in,T_out", [(4, 16, 32), (4, 37, 44), (3, 21, 100)])
def test_gradients_at_full_reference_widths(B, T_in, T_out):
    """The reference's real layer widths (hparams.py:33-69: 256-wide embedding/attention/decoder, 16x128 and 8x256 conv banks,
    1025 linear bins) on a short batch, so every kernel runs with its production tile shapes.  The lengths that are not multiples
    of 16 put a 16-row block of k_wgrad_bf3 across two batch rows: with a negative tap shift its unmasked fast path used to read
    the previous batch row's tail where SAME padding has zeros (2-6 % error in the conv taps' gradients; found by the round-3
    advisor, invisible at T in {16, 32, 128, 512})."""
    import torch
    hp = O.OracleHParams(max_iters=max(16, T_out // 4))
    w = O.init_weights(hp, 1, 41)
    ids, L = O.synthetic_inputs(B, T_in, 42, ragged=True)
    rs = np.random.RandomState(43)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    co = rs.uniform(0.5, 1.5, size=B)
    loss, g, out = TF.train_grads(w, hp, ids, L, mt, lt, co)
    tr = _trainer(hp, w)
    losses = tr.forward_backward(ids, L, mt, lt, co, keep_outputs=True)
    torch.cuda.synchronize()
    assert abs(float(losses[0]) - loss) < 2e-5
    assert maxabs(tr.mel_outputs.cpu().numpy(), out["mel"]) < 1e-4 and maxabs(tr.linear_outputs.cpu().numpy(), out["linear"]) < 1e-4
    worst, gn = _grad_report(tr.grad_dict(), g)
    # (B T_out 2048 max-pool and ReLU decisions: on the longest case one of them can fall on a near-tie that fp32 and the float64
    # checker resolve differently -- a finite step of ~1e-5 of the gradient norm in a few small tensors, in BOTH weight-gradient
    # kernels alike; the boundary bug this parametrisation exists for shows in the split-vs-exact comparison below, at 2-6 %)
    # ... and the L1 losses (tacotron.py:274-302) are not differentiable where an output meets its target: an output that the fp32
    # forward puts on the other side of its target than the float64 checker does flips d|y - t|/dy = sign(y - t) for that element -- a
    # finite step of 2 / N in the output gradient whatever the forward's accuracy.  Counted here; the tight bound holds when none occurs.
    flips = int((np.sign(tr.linear_outputs.cpu().numpy() - lt) != np.sign(out["linear"] - lt)).sum() +
                (np.sign(tr.mel_outputs.cpu().numpy() - mt) != np.sign(out["mel"] - mt)).sum())
    tol = 3e-3 if (B * T_out <= 200 and flips == 0) else 5e-2
    print("outputs on the other side of their target than the float64 checker's: %d of %d" % (flips, lt.size + mt.size))
    assert worst[0][0] < tol, worst[:5]
    # the relaxed bound above must not hide what this parametrisation exists for.  The conv taps' gradients -- every
    # [k, cin, cout] kernel of both CBHGs, the tensors the batch-row-boundary bug moved by 2-6 % -- are sums over all B * T frames: one
    # flipped L1 sign or one near-tie moves a few ELEMENTS of them (the max-norm measure above sees that: up to 1e-2 of a small tensor's
    # scale, measured), not the tensor: in the Frobenius norm they keep the TIGHT bound whatever `flips` is, for both weight-gradient
    # kernels (the exact-fp32 one below).
    def conv_taps(got_):
        rep = []
        for k, v in g.items():
            if ("/conv_bank/" in k or "/proj_" in k) and k.endswith("/kernel"):
                rep.append((float(np.linalg.norm(got_[k] - v)) / max(float(np.linalg.norm(v)), 1e-3 * gn), k))
        return sorted(rep, reverse=True)
    ct = conv_taps(tr.grad_dict())
    assert len(ct) == 16 + 8 + 4
    print("conv-tap gradients vs float64 autograd (Frobenius, relative), worst three:", ct[:3])
    assert ct[0][0] < 3e-3, ct[:5]
    # the weight gradients above came from the split-bf16 matrix-core kernel (k_wgrad_bf3, the default); the exact-fp32 MFMA kernel
    # (k_wgrad) must give the same gradients to the split's ~1e-5
    got = tr.grad_dict()
    tr.set_exact_wgrad(True)
    try:
        tr.forward_backward(ids, L, mt, lt, co)
        torch.cuda.synchronize()
        exact = tr.grad_dict()
    finally:
        tr.set_exact_wgrad(False)
    worst_x, _ = _grad_report(exact, g)
    d = max(maxabs(got[k], exact[k]) / max(float(np.abs(exact[k]).max()), 1e-3 * gn) for k in exact)
    print("weight gradients: split-bf16 vs float64 autograd %.2e, exact fp32 vs autograd %.2e, split vs exact %.2e (relative, per tensor, worst)" % (worst[0][0], worst_x[0][0], d))
    assert worst_x[0][0] < tol and d < 1e-3
    assert conv_taps(exact)[0][0] < 3e-3, conv_taps(exact)[:5]
    tr.close()


def test_train_step_through_the_collective_path():
    """The data-parallel step with a real RCCL process group (one rank: the all-reduce and the division by the world size
    must leave the single-process result untouched)."""
    import os
    import torch
    import torch.distributed as dist
    import taco_amd
    hp, w, ids, L, mt, lt, co = _setup("bah_mon", seed=33)
    a = _trainer(hp, w)
    a.train_step(ids, L, mt, lt, co)
    torch.cuda.synchronize()
    want = a.params.clone()
    a.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        b = _trainer(hp, w)
        step, _ = b.train_step(ids, L, mt, lt, co)
        torch.cuda.synchronize()
        # gradients are accumulated with fp32 atomics (order not fixed): allow a few Adam-step quanta (lr = 5e-7 here)
        assert step == 1 and float((b.params - want).abs().max()) < 2e-6
        b.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_sync_bn_two_ranks_equal_one_process_on_the_global_batch(tmp_path):
    """with SyncBN a data-parallel step over W shards IS the reference's single-device step over the whole batch
    (BatchNorm statistics, moving averages and every gradient).  Two processes (gloo transport, both on this GPU) take half of a
    batch of 4 each; one process takes all 4 with plain BatchNorm.  Also shows the unsynchronised DP step is a different function."""
    import os
    import subprocess
    import sys
    import torch
    hp, w, ids, L, mt, lt, co = _setup("bah_mon", B=4, seed=41)
    one = _trainer(hp, w)
    l1 = one.forward_backward(ids, L, mt, lt, co).cpu().numpy().copy()
    torch.cuda.synchronize()
    g1, p1 = one.grads.cpu().numpy().copy(), one.params.cpu().numpy().copy()
    # the same two shards without synchronisation: a different (per-shard statistics) function
    ga = []
    for r in range(2):
        t = _trainer(hp, w)
        t.forward_backward(ids[2 * r:2 * r + 2], L[2 * r:2 * r + 2], mt[2 * r:2 * r + 2], lt[2 * r:2 * r + 2], co[2 * r:2 * r + 2])
        ga.append(t.grads.cpu().numpy().copy()); t.close()
    one.close()
    np.savez(os.path.join(tmp_path, "case.npz"), ids=ids, L=L, mt=mt, lt=lt, co=co, seed=41)
    port = str(29600 + os.getpid() % 300)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_syncbn_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path)], env=env) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=600) == 0
    res = np.load(os.path.join(tmp_path, "result.npz"))
    scale = np.abs(g1).max()
    assert np.abs(res["losses"] - l1).max() < 2e-5, (res["losses"], l1)
    worst = float(np.abs(res["grads"] - g1).max() / scale)
    assert worst < 2e-5, worst                                         # summation order (shards, atomics) only
    assert float(np.abs(res["params"] - p1).max()) < 2e-5               # the moving statistics were updated with global-batch values
    # one exchange per BatchNorm layer forward (2 CBHGs x {bank, proj_1, proj_2}) and per layer / per bank backward
    assert int(res["exchanges"]) == 12, int(res["exchanges"])
    unsync = float(np.abs(0.5 * (ga[0] + ga[1]) - g1).max() / scale)
    print("syncbn: worst %.2e  unsync %.2e  loss diff %.2e" % (worst, unsync, float(np.abs(res["losses"] - l1).max())))
    assert unsync > 20 * worst, (unsync, worst)


@pytest.mark.parametrize("ses,atype", [(4, "bah_mon"), (1, "bah")])
def test_deepvoice_multispeaker_training_gradients(ses, atype):
    """model_type 'deepvoice' (tacotron.py:52-94): speaker embedding -> five softsign dense layers (or five per-speaker tables when
    speaker_embedding_size == 1) -> encoder residual offset + initial states of the encoder BiGRU, the attention GRU and the decoder
    GRUs; all of it trained.  Forward, loss and every gradient vs float64 autograd."""
    import torch
    import taco_amd
    ns = 3
    hp = tiny_hp(model_type="deepvoice", speaker_embedding_size=ses, attention_type=atype)
    w = O.init_weights(hp, ns, 61)
    B, T_in, T_out = 5, 10, 12
    ids, L = O.synthetic_inputs(B, T_in, 62, ragged=True)
    rs = np.random.RandomState(63)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    co = rs.uniform(0.5, 1.5, size=B)
    spk = np.array([2, 0, 1, 2, 2], np.int32)
    loss, g, out = TF.train_grads(w, hp, ids, L, mt, lt, co, speaker_id=spk, num_speakers=ns)
    tr = taco_amd.Trainer(to_product_hp(hp), w, num_speakers=ns)
    losses = tr.forward_backward(ids, L, mt, lt, co, keep_outputs=True, speaker_id=spk)
    torch.cuda.synchronize()
    assert abs(float(losses[0]) - loss) < 1e-5
    assert maxabs(tr.linear_outputs.cpu().numpy(), out["linear"]) < 1e-4 and maxabs(tr.alignments.cpu().numpy(), out["alignments"]) < 1e-4
    worst, gn = _grad_report(tr.grad_dict(), g)
    assert worst[0][0] < 2e-3, worst[:6]
    spk_names = [k for k in g if k.startswith("spk/") or k == "speaker_embedding"]
    assert spk_names and all(np.abs(g[k]).max() > 0 for k in spk_names)
    step, lwc = tr.train_step(ids, L, mt, lt, co, speaker_id=spk)
    assert step == 1 and np.isfinite(float(lwc))
    tr.close()


@pytest.mark.parametrize("ses,atype", [(4, "bah_mon"), (8, "bah_norm")])
def test_simple_multispeaker_training_gradients(ses, atype):
    """model_type 'simple' (tacotron.py:47-50, rnn_wrappers.py:172-195,253-263, tacotron.py:226-235): the speaker embedding row is
    concatenated to the attention-GRU input, to the concat-projection input and (tiled over time) to the linear-head input.
    Forward, loss and every gradient -- the widened kernels and the embedding table included -- vs float64 autograd."""
    import torch
    import taco_amd
    ns = 3
    hp = tiny_hp(model_type="simple", speaker_embedding_size=ses, attention_type=atype)
    w = O.init_weights(hp, ns, 71)
    B, T_in, T_out = 5, 10, 12
    ids, L = O.synthetic_inputs(B, T_in, 72, ragged=True)
    rs = np.random.RandomState(73)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    co = rs.uniform(0.5, 1.5, size=B)
    spk = np.array([2, 0, 1, 2, 2], np.int32)
    loss, g, out = TF.train_grads(w, hp, ids, L, mt, lt, co, speaker_id=spk, num_speakers=ns)
    tr = taco_amd.Trainer(to_product_hp(hp), w, num_speakers=ns)
    losses = tr.forward_backward(ids, L, mt, lt, co, keep_outputs=True, speaker_id=spk)
    torch.cuda.synchronize()
    assert abs(float(losses[0]) - loss) < 1e-5
    assert maxabs(tr.linear_outputs.cpu().numpy(), out["linear"]) < 1e-4 and maxabs(tr.alignments.cpu().numpy(), out["alignments"]) < 1e-4
    got = tr.grad_dict()
    worst, gn = _grad_report(got, g)
    assert worst[0][0] < 2e-3, worst[:6]
    assert np.abs(g["speaker_embedding"]).max() > 0 and np.abs(got["speaker_embedding"][1]).max() > 0
    assert got["linear/kernel"].shape[0] == ses + 2 * hp.post_rnn_size and np.abs(got["linear/kernel"][:ses]).max() > 0
    step, lwc = tr.train_step(ids, L, mt, lt, co, speaker_id=spk)
    assert step == 1 and np.isfinite(float(lwc))
    tr.close()


@pytest.mark.parametrize("model_type,atype,B,r", [("single", "bah_mon", 9, 4), ("simple", "bah", 5, 5), ("deepvoice", "bah_norm", 3, 4), ("single", "bah_mon", 37, 4)])
def test_rnn_decoder_test_mode_on_the_persistent_kernel(model_type, atype, B, r):
    """rnn_decoder_test_mode (helpers.py:63-64; the test model of train.py:158-166, run every test_interval): the decoder is fed the LAST
    of the r frames it just emitted.  At the reference widths that is a mode of the TAPE instantiation of k_decoder_xcd (the frame is
    exchanged between the group's members, then the prenet layer reads it with the raw kernel rows of the teacher-form pack): outputs and
    loss against the float64 oracle's fed-back forward, rows per group 2 / 1 / 1 / 8, and against the launch-per-stage loop."""
    import torch
    import taco_amd
    ns = 1 if model_type == "single" else 3
    n = 7
    hp = O.OracleHParams(max_iters=n, model_type=model_type, attention_type=atype, reduction_factor=r)
    w = O.init_weights(hp, ns, 91)
    T_in, T_out = 19, n * r
    ids, L = O.synthetic_inputs(B, T_in, 92, ragged=True)
    rs = np.random.RandomState(93)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    co = rs.uniform(0.5, 1.5, size=B)
    spk = (np.arange(B) % ns).astype(np.int32) if ns > 1 else None
    fb = O.forward(w, hp, ids, L, speaker_id=spk, num_speakers=ns, n_steps=n, honor_stop=False, training=True)
    tr = taco_amd.Trainer(to_product_hp(hp), w, num_speakers=ns)
    got = {}
    for mode in (1, 0):
        tr.set_decoder_engine(mode)
        losses = tr.forward_backward(ids, L, mt, lt, co, backward=False, keep_outputs=True, rnn_decoder_test_mode=True, speaker_id=spk)
        torch.cuda.synchronize()
        tr.check_device_errors()
        info = tr.decoder_engine_info()
        assert mode == 0 or info["protocol"] in (1, 2), (mode, info)      # (the word keeps the last persistent launch's answer when the launch-per-stage loop runs)
        got[mode] = (tr.mel_outputs.cpu().numpy(), tr.linear_outputs.cpu().numpy(), tr.alignments.cpu().numpy(), float(losses[0]))
        assert maxabs(got[mode][0], fb["mel"]) < 1e-4 and maxabs(got[mode][1], fb["linear"]) < 2e-4 and maxabs(got[mode][2], fb["alignments"]) < 1e-4, mode
        assert abs(got[mode][3] - O.add_loss(fb["mel"], mt, fb["linear"], lt, co)["loss"]) < 2e-5
    assert maxabs(got[1][0], got[0][0]) < 2e-5 and maxabs(got[1][2], got[0][2]) < 2e-5
    # and the teacher-forced step right after it on the same trainer (the teacher buffer of the kernel's LDS was the fed-back frame's)
    ref = O.forward(w, hp, ids, L, speaker_id=spk, num_speakers=ns, n_steps=n, honor_stop=False, teacher_frames=mt[:, r - 1::r], training=True)
    tr.set_decoder_engine(1)
    tr.forward_backward(ids, L, mt, lt, co, backward=False, keep_outputs=True, speaker_id=spk)
    torch.cuda.synchronize()
    assert maxabs(tr.mel_outputs.cpu().numpy(), ref["mel"]) < 1e-4


@pytest.mark.parametrize("model_type,atype,B", [("single", "bah_mon", 9), ("deepvoice", "bah", 3), ("simple", "bah_norm", 5), ("single", "bah_mon", 33),
                                                ("deepvoice", "bah_mon", 20), ("single", "bah_norm", 18)])
def test_training_forward_on_the_persistent_kernels(model_type, atype, B):
    """At the reference widths the teacher-forced decoder loop and the post-net scan of the training forward are the persistent
    kernels of inference with tape outputs (k_decoder_xcd<RG, true>: teacher frames in, gates / states / scores / alignments out;
    k_bigru_duo<RG, true>: gate tape) -- rows per group 1 / 2 / 8, deepvoice initial states, the 'simple' speaker term.  Forward and loss
    against float64 autograd; the gradients (all of them are computed from that tape) as a whole against autograd and tensor by
    tensor against the launch-per-stage engine."""
    import torch
    import taco_amd
    ns = 1 if model_type == "single" else 3
    hp = O.OracleHParams(max_iters=8, model_type=model_type, attention_type=atype)
    w = O.init_weights(hp, ns, 81)
    T_in, T_out = 14, 8 * hp.reduction_factor
    ids, L = O.synthetic_inputs(B, T_in, 82, ragged=True)
    rs = np.random.RandomState(83)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    co = rs.uniform(0.5, 1.5, size=B)
    spk = (np.arange(B) % ns).astype(np.int32) if ns > 1 else None
    loss, g, out = TF.train_grads(w, hp, ids, L, mt, lt, co, speaker_id=spk, num_speakers=ns)
    tr = taco_amd.Trainer(to_product_hp(hp), w, num_speakers=ns)
    losses = tr.forward_backward(ids, L, mt, lt, co, keep_outputs=True, speaker_id=spk)
    torch.cuda.synchronize()
    tr.check_device_errors()
    info = tr.decoder_engine_info()
    assert info["has_pack"] and info["protocol"] in (1, 2), info
    assert info["bptt_protocol"] in (1, 2), info                   # every model type back-propagates through the one persistent launch
    assert abs(float(losses[0]) - loss) < 2e-5
    assert maxabs(tr.mel_outputs.cpu().numpy(), out["mel"]) < 1e-4 and maxabs(tr.linear_outputs.cpu().numpy(), out["linear"]) < 1e-4
    assert maxabs(tr.alignments.cpu().numpy(), out["alignments"]) < 1e-4
    got = tr.grad_dict()
    worst, gn = _grad_report(got, g)
    # Against float64 autograd: the gradient as a whole (2e-3 of its norm).  Tensor by tensor the comparison is not meaningful at
    # these widths on a few dozen rows: a ReLU / max-pool decision that float64 and float32 take differently on a near-tie moves
    # individual small tensors by percents of their scale -- identically on BOTH engines (tools/scratch/ab_train_engines.py:
    # e.g. encoder_cbhg/highway_2/H/kernel 3.2e-2 on the launch-per-stage and on the persistent engine alike) -- so the
    # tensor-by-tensor yardstick here is the launch-per-stage engine, whose own gradients are pinned tensor by tensor against
    # autograd by the tests above at shapes without such ties.
    err2 = np.sqrt(sum(float(((got[k] - g[k]) ** 2).sum()) for k in g))
    print("gradient vs float64 autograd: |diff| / |g| = %.2e; worst tensors %s" % (err2 / gn, [(round(x[0], 4), x[1]) for x in worst[:3]]))
    assert err2 < 2e-3 * gn, (err2, gn)
    tr.set_bptt_engine(False)                     # persistent forward, the decoder's BPTT as the chain of per-stage launches
    tr.forward_backward(ids, L, mt, lt, co, keep_outputs=True, speaker_id=spk)
    torch.cuda.synchronize()
    assert tr.decoder_engine_info()["bptt_protocol"] == 0
    mid = tr.grad_dict()
    worst_b = sorted(((maxabs(got[k], mid[k]), k) for k in mid), reverse=True)
    print("persistent BPTT (k_decoder_bwd_xcd) vs per-stage BPTT on the same tape, worst tensors (absolute):", worst_b[:3], " |g| =", gn)
    assert worst_b[0][0] < 2e-5 * gn, worst_b[:4]
    tr.set_bptt_engine(True)
    tr.set_decoder_engine(0)                      # the same step on the launch-per-stage engine
    tr.forward_backward(ids, L, mt, lt, co, keep_outputs=True, speaker_id=spk)
    torch.cuda.synchronize()
    assert tr.decoder_engine_info()["protocol"] in (1, 2)          # (the info words still show the earlier persistent launch)
    ref = tr.grad_dict()
    worst_e = sorted(((maxabs(got[k], ref[k]), k) for k in ref), reverse=True)
    print("persistent vs launch-per-stage engine, worst tensors (absolute):", worst_e[:3], " |g| =", gn)
    assert worst_e[0][0] < 2e-5 * gn, worst_e[:4]
    tr.set_decoder_engine(1)
    step, lwc = tr.train_step(ids, L, mt, lt, co, speaker_id=spk)      # packs regenerated on the device (k_dx_fold + index-map gather)
    losses2 = tr.forward_backward(ids, L, mt, lt, co, backward=False, speaker_id=spk)
    torch.cuda.synchronize()
    tr.check_device_errors()
    assert step == 1 and np.isfinite(float(lwc)) and float(losses2[3]) < float(lwc) + 1e-3
    tr.close()


def test_captured_step_survives_an_eager_step_of_a_larger_shape():
    """Trainer.capture bakes the workspace address into the graph; an eager step with a shape that needs a larger workspace must
    not move or free that buffer, and the replay that follows it must still be the step (the trainer records the
    step afresh on that transition): capture at a small shape, step eagerly at a larger one, replay the small shape again and
    compare every parameter with a trainer that never captured anything.  The capture itself (warm-up + recording) must not
    advance the BatchNorm moving averages."""
    import torch
    hp, w, ids, L, mt, lt, co = _setup("bah_mon", seed=31)
    rs = np.random.RandomState(32)
    B2, T2, To2 = ids.shape[0] + 3, ids.shape[1] + 9, mt.shape[1] + 2 * hp.reduction_factor
    ids2, L2 = O.synthetic_inputs(B2, T2, 33, ragged=True)
    mt2, lt2 = rs.rand(B2, To2, hp.num_mels), rs.rand(B2, To2, hp.num_freq)
    a, b = _trainer(hp, w), _trainer(hp, w)
    a.capture(ids, L, mt, lt, co)
    torch.cuda.synchronize()
    w0 = a.get_weights()
    for k in w:
        assert np.array_equal(w0[k], np.asarray(w[k], np.float32)), "capture changed %s" % k
    for tr in (a, b):
        tr.train_step(ids, L, mt, lt, co)             # a: graph replay, b: eager
        tr.train_step(ids2, L2, mt2, lt2)              # a: eager beside the graph, with a larger workspace
        tr.train_step(ids, L, mt, lt, co)             # a: replay into the graph's own (untouched) workspace
    torch.cuda.synchronize()
    assert a._graph is not None and a._ws_eager is not None and a._ws_eager.numel() > a._ws.numel()
    wa, wb = a.get_weights(), b.get_weights()
    for k in wa:
        # (Adam's update is lr * m / sqrt(v): fp32-atomic summation order in the weight gradients moves near-zero-gradient entries by
        # a visible fraction of a step; a replay into freed memory is off by orders of magnitude or not finite)
        assert np.isfinite(wa[k]).all() and maxabs(wa[k], wb[k]) < 1e-3 * max(1.0, float(np.abs(wb[k]).max())), k


def test_C4_shard_shape_forward_and_properties():
    """BASELINE.json configs[3], one shard of the data-parallel step: B=32, T_in=128, T_out=512 at the reference widths
    (train.py:145-166,217-219; hparams.py batch_size 32).  The teacher-forced training forward against the float64 oracle at full
    size; then the size-independent properties of the step: finite gradients, two runs agree to the tolerance of fp32 atomics,
    five steps on the batch lower the loss."""
    import torch
    B, T_in, T_out = 32, 128, 512
    hp = O.OracleHParams(max_iters=T_out // 4)
    w = O.init_weights(hp, 1, 1234 + 3)
    ids, L = O.synthetic_inputs(B, T_in, 1234 + 3, ragged=True)
    rs = np.random.RandomState(1234 + 3)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)       # the normalised range of audio/__init__.py:161-162
    r = hp.reduction_factor
    ref = O.forward(w, hp, ids, L, n_steps=T_out // r, honor_stop=False, teacher_frames=mt[:, r - 1::r], training=True, dtype=np.float64)
    tr = _trainer(hp, w)
    losses = tr.forward_backward(ids, L, mt, lt, backward=False, keep_outputs=True)
    torch.cuda.synchronize()
    errs = {k: maxabs(getattr(tr, k + "_outputs" if k != "alignments" else k).cpu().numpy(), ref[k]) for k in ("mel", "linear", "alignments")}
    assert max(errs.values()) < 1e-3, errs
    want = O.add_loss(ref["mel"], mt, ref["linear"], lt, np.ones(B))
    got = losses.cpu().numpy()
    for i, k in enumerate(("loss", "mel_loss", "linear_loss", "loss_without_coeff")):
        assert abs(got[i] - want[k]) < 1e-4 * max(1.0, abs(want[k])), (k, got[i], want[k])
    tr.set_deterministic(False)                  # first the fp32-atomics mode (the default until round 3)
    tr.forward_backward(ids, L, mt, lt)
    torch.cuda.synchronize()
    g1 = tr.grads.detach().clone()
    tr.forward_backward(ids, L, mt, lt, freeze_moving_averages=True)
    torch.cuda.synchronize()
    g2 = tr.grads
    assert bool(torch.isfinite(g1).all())
    scale = float(g1.abs().max())
    rerun = float((g1 - g2).abs().max())
    # the weight gradients are sums of 16 K rows accumulated with fp32 atomics in whatever order the workgroups arrive: 24 reruns on one
    # box spread between 0.5e-4 and 1.4e-4 of the largest gradient; a stale operand or a missed update would be of the order of the scale
    assert rerun < 1e-3 * scale, ("two runs of the same step differ beyond fp32-atomic summation order", rerun, scale)
    # ... and with ordered two-stage sums instead of atomics (Trainer.set_deterministic) the step is reproducible to the bit, like the
    # reference's single-device step; its gradients are the same up to that summation order
    tr.set_deterministic(True)
    tr.forward_backward(ids, L, mt, lt, freeze_moving_averages=True)
    torch.cuda.synchronize()
    d1 = tr.grads.detach().clone()
    tr.forward_backward(ids, L, mt, lt, freeze_moving_averages=True)
    torch.cuda.synchronize()
    rerun_det = float((d1 - tr.grads).abs().max())
    print("rerun difference: atomics %.2e of the gradient scale, deterministic %.2e" % (rerun / scale, rerun_det / scale))
    assert rerun_det == 0.0, rerun_det
    assert float((d1 - g1).abs().max()) < 1e-3 * scale
    trace = []
    for _ in range(5):
        _, lwc = tr.train_step(ids, L, mt, lt)
        trace.append(float(lwc))
    torch.cuda.synchronize()
    assert trace[-1] < trace[0] and np.isfinite(trace[-1]), trace


def test_C4_horizon_gradients_on_a_two_row_slice():
    """The C4 horizon (T_in=128, T_out=512: 128 teacher-forced decoder steps, 512-frame post-net) at the reference widths on two
    rows: every gradient tensor against float64 reverse-mode autograd of the independent torch formulation."""
    import torch
    hp = O.OracleHParams(max_iters=128)
    w = O.init_weights(hp, 1, 1234 + 13)
    B, T_in, T_out = 2, 128, 512
    ids, L = O.synthetic_inputs(B, T_in, 1234 + 13, ragged=True)
    rs = np.random.RandomState(1234 + 13)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    loss, g, _ = TF.train_grads(w, hp, ids, L, mt, lt)
    tr = _trainer(hp, w)
    losses = tr.forward_backward(ids, L, mt, lt)
    torch.cuda.synchronize()
    assert abs(float(losses[0]) - loss) < 1e-4
    worst, gn = _grad_report(tr.grad_dict(), g)
    assert worst[0][0] < 2e-3, worst[:5]


@pytest.mark.parametrize("B,T", [(5, 37), (32, 48), (40, 21), (1, 19), (12, 33), (17, 20)])
def test_whole_chip_bigru_scans_forward_tape_and_backward(B, T):
    """k_bigru_oct<UPW, true> (9 to 32 rows: B = 12, 17, 32) / k_bigru_duo<RG, true> (gate tape) + k_bigru_duo_bwd (BPTT through TF's GRUCell, A.6/A.7) against the kernels they replace in
    training (k_bigru_res + k_bigru_rows_bwd) and against float64 autograd of the recurrence itself, with what the post-net never
    has but the kernels support: ragged lengths (0 and T included) and initial states, rows per group 1 / 2 / 4 / 8."""
    import ctypes as C
    import torch
    import taco_amd
    from util import dev, ptr, stream
    hp = O.OracleHParams(max_iters=4)
    w = O.init_weights(hp, 1, 91)
    tr = _trainer(hp, w)
    H = hp.post_rnn_size
    rs = np.random.RandomState(92 + B)
    xproj = (rs.randn(B, T, 6 * H) * 0.4).astype(np.float32)
    lens = rs.randint(0, T + 1, size=B).astype(np.int32); lens[0] = T
    if B > 1:
        lens[1] = 0
    h0 = (rs.randn(B, 2 * H) * 0.5).astype(np.float32)
    dout = rs.randn(B, T, 2 * H).astype(np.float32)
    xd, ld, hd, dd = dev(xproj), dev(lens), dev(h0), dev(dout)
    scratch = torch.empty((2 << 20,), dtype=torch.uint8, device="cuda")
    res = {}
    for eng in (1, 0):
        o = {k: torch.full(s, float("nan"), device="cuda") for k, s in (("out", (B, T, 2 * H)), ("gsave", (B, T, 6 * H)), ("dg", (B, T, 6 * H)),
                                                                        ("rh", (B, T, 2 * H)), ("dh0", (B, 2 * H)))}
        taco_amd._lib.check(tr._lib.taco_train_debug_bigru(tr._h, stream(), ptr(xd), ptr(ld), ptr(hd), ptr(dd), B, T, eng, ptr(o["out"]), ptr(o["gsave"]),
                                                           ptr(o["dg"]), ptr(o["rh"]), ptr(o["dh0"]), ptr(scratch), scratch.numel()))
        torch.cuda.synchronize()
        tr.check_device_errors()
        res[eng] = {k: v.cpu().numpy() for k, v in o.items()}
    for k in ("out", "gsave", "dg", "rh", "dh0"):
        assert np.isfinite(res[1][k]).all(), k
        assert maxabs(res[1][k], res[0][k]) < 2e-5 * max(1.0, float(np.abs(res[0][k]).max())), k
    # float64 autograd of the recurrence (gates r|u, candidate; h' = u h + (1-u) c; A.7 masking and reverse_sequence time mapping)
    tw = {k: torch.tensor(np.asarray(v, np.float64)) for k, v in w.items() if k.startswith("post_cbhg/bigru/")}
    xp = torch.tensor(xproj.astype(np.float64), requires_grad=True)
    h0t = torch.tensor(h0.astype(np.float64), requires_grad=True)
    outs = torch.zeros(B, T, 2 * H, dtype=torch.float64)
    pieces = []
    for d, name in enumerate(("fw", "bw")):
        Wg = tw["post_cbhg/bigru/%s/gates/kernel" % name][H:]; Wc = tw["post_cbhg/bigru/%s/candidate/kernel" % name][H:]
        for b in range(B):
            h = h0t[b, d * H:(d + 1) * H]
            L = int(lens[b])
            for s in range(L):
                t = (L - 1 - s) if d else s
                xg = xp[b, s, d * 3 * H:d * 3 * H + 2 * H] if d else xp[b, t, :2 * H]       # the backward columns are stored time-reversed: row s
                xc = xp[b, s, d * 3 * H + 2 * H:(d + 1) * 3 * H] if d else xp[b, t, 2 * H:3 * H]
                g = torch.sigmoid(xg + h @ Wg)
                r, u = g[:H], g[H:]
                c = torch.tanh(xc + (r * h) @ Wc)
                h = u * h + (1 - u) * c
                pieces.append((b, t, d, h))
    loss = sum((hh * torch.tensor(dout[b, t, d * H:(d + 1) * H].astype(np.float64))).sum() for b, t, d, hh in pieces)
    loss.backward()
    for b, t, d, hh in pieces:
        outs[b, t, d * H:(d + 1) * H] = hh.detach()
    assert maxabs(res[1]["out"], outs.numpy()) < 2e-5
    assert maxabs(res[1]["dh0"], h0t.grad.numpy()) < 2e-4 * max(1.0, float(h0t.grad.abs().max()))
    # dg is the gradient of the pre-activations = of the hoisted projection; the kernel writes it at TRUE time, the projection's
    # backward columns live at scan time: map them back for the comparison
    want = np.zeros((B, T, 6 * H))
    gx = xp.grad.numpy()
    for b in range(B):
        L = int(lens[b])
        want[b, :L, :3 * H] = gx[b, :L, :3 * H]
        want[b, :L, 3 * H:] = gx[b, :L, 3 * H:][::-1]
    assert maxabs(res[1]["dg"], want) < 2e-4 * max(1.0, float(np.abs(want).max()))
    tr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("atype,B,T_in,n", [("bah_mon", 20, 150, 5), ("bah", 7, 300, 4), ("bah_mon", 64, 70, 3), ("bah_mon", 2, 600, 3)])
def test_persistent_bptt_on_long_inputs_equals_the_per_stage_chain(atype, B, T_in, n):
    """k_decoder_bwd_xcd on inputs whose positions no longer fit two per lane (the chunked normaliser-backward path), on all 64 rows, and on an
    input too long for its LDS (T_in = 600 at one row per group still fits; the usable check falls back by itself otherwise): every
    gradient tensor against the per-stage BPTT reading the same tape."""
    import torch
    import taco_amd
    hp = O.OracleHParams(max_iters=8, attention_type=atype)
    w = O.init_weights(hp, 1, 91)
    T_out = n * hp.reduction_factor
    ids, L = O.synthetic_inputs(B, T_in, 92, ragged=True)
    rs = np.random.RandomState(93)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    tr = taco_amd.Trainer(to_product_hp(hp), w)
    tr.forward_backward(ids, L, mt, lt, None)
    info = tr.decoder_engine_info()
    tr.check_device_errors()
    assert info["protocol"] in (1, 2) and info["bptt_protocol"] in (1, 2), info
    got = tr.grad_dict()
    tr.set_bptt_engine(False)
    tr.forward_backward(ids, L, mt, lt, None)
    torch.cuda.synchronize()
    assert tr.decoder_engine_info()["bptt_protocol"] == 0
    ref = tr.grad_dict()
    gn = np.sqrt(sum(float((ref[k] ** 2).sum()) for k in ref))
    worst = sorted(((maxabs(got[k], ref[k]), k) for k in ref), reverse=True)
    print("persistent vs per-stage BPTT, worst tensors (absolute):", worst[:3], " |g| =", gn)
    assert worst[0][0] < 2e-5 * gn, worst[:4]
    tr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("atype,B", [("bah_mon", 9), ("bah", 20)])
def test_split_bf16_training_gemms_track_the_exact_engine(atype, B):
    """The GEMM engines of the step.  Default (mode 3): forward GEMMs exact, data gradients on the split-bf16 kernels -- held here to 5e-5 of the
    gradient norm against the all-exact step, with a bit-identical loss.  Opt-in speed mode (Trainer.set_exact_gemm(False)): the feed-forward GEMMs and their data gradients on the split-bf16
    kernels of inference, weight planes re-split on the device after every optimizer step.  Against the default exact-fp32 engine on
    the same inputs: loss to 1e-5, the gradient as a whole to 3e-3 of its norm (measured 1.2e-3 at 9 rows: the ~1e-5 relative product
    error is amplified by the BatchNorm backward's cancellations, and a ReLU / max-pool near-tie may resolve differently, which moves
    single tensors by more), and the planes follow a parameter update."""
    import torch
    import taco_amd
    hp = O.OracleHParams(max_iters=8, attention_type=atype)
    w = O.init_weights(hp, 1, 101)
    T_in, T_out = 14, 8 * hp.reduction_factor
    ids, L = O.synthetic_inputs(B, T_in, 102, ragged=True)
    rs = np.random.RandomState(103)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    tr = taco_amd.Trainer(to_product_hp(hp), w)
    tr.set_exact_gemm(True)                                # every GEMM on the exact-fp32 MFMA: the yardstick
    le = tr.forward_backward(ids, L, mt, lt, None).cpu().numpy().copy()
    ref = tr.grad_dict()
    tr.set_exact_gemm(3)                                   # the engine of rounds 2-3: forward exact, data gradients split-bf16
    ld = tr.forward_backward(ids, L, mt, lt, None).cpu().numpy().copy()
    dflt = tr.grad_dict()
    gn0 = np.sqrt(sum(float((ref[k] ** 2).sum()) for k in ref))
    e0 = np.sqrt(sum(float(((dflt[k] - ref[k]) ** 2).sum()) for k in ref))
    print("default engine (forward exact, data gradients split-bf16) vs all-exact: loss %.1e, gradient |diff| / |g| = %.2e" % (abs(ld[0] - le[0]), e0 / gn0))
    assert ld[0] == le[0] and e0 < 5e-5 * gn0              # the forward is the same arithmetic; measured 4e-6 .. 6e-6
    tr.set_exact_gemm(4)                                   # forward on the six-product split (operands split three ways: fp32-grade), data gradients split-bf16
    l6 = tr.forward_backward(ids, L, mt, lt, None, keep_outputs=True).cpu().numpy().copy()
    six = tr.grad_dict()
    mel6 = tr.mel_outputs.cpu().numpy().copy()
    e6 = np.sqrt(sum(float(((six[k] - ref[k]) ** 2).sum()) for k in ref))
    print("six-product forward vs all-exact: loss %.1e, gradient |diff| / |g| = %.2e" % (abs(l6[0] - le[0]), e6 / gn0))
    # fp32-grade products (the loss is the same to the last printed digit), but not the same BITS as the fp32-input MFMA: an output or
    # pre-activation that sits within an ulp of a kink (L1 target, ReLU zero, max-pool tie) may fall on its other side -- a finite step
    # of ~1e-4 of the gradient norm per occurrence, as between any two fp32 evaluation orders; measured 2.3e-4 at 9 rows (one flip)
    assert abs(l6[0] - le[0]) < 2e-6 and e6 < 1e-3 * gn0
    tr.set_exact_gemm(False)
    lf = tr.forward_backward(ids, L, mt, lt, None).cpu().numpy().copy()
    got = tr.grad_dict()
    gn = np.sqrt(sum(float((ref[k] ** 2).sum()) for k in ref))
    e2 = np.sqrt(sum(float(((got[k] - ref[k]) ** 2).sum()) for k in ref))
    print("split-bf16 vs exact GEMMs: loss %.3e, gradient |diff| / |g| = %.2e" % (abs(lf[0] - le[0]), e2 / gn))
    assert abs(lf[0] - le[0]) < 1e-5 and e2 < 3e-3 * gn
    step, _ = tr.train_step(ids, L, mt, lt, None)          # the planes are regenerated from the updated parameters
    l1 = tr.forward_backward(ids, L, mt, lt, None, backward=False).cpu().numpy().copy()
    tr.set_exact_gemm(3)
    l2 = tr.forward_backward(ids, L, mt, lt, None, backward=False).cpu().numpy().copy()
    assert step == 1 and abs(l1[0] - l2[0]) < 1e-5 and l1[0] < lf[0]
    tr.close()


@pytest.mark.gpu
def test_a_device_fault_poisons_the_step_and_adam_skips_the_update():
    """ADVICE r02 (medium), training side: a persistent kernel that gave up must not feed garbage into the optimizer.  The sticky device
    error word is raised by hand (taco_debug_raise_device_error): the next step's losses and first gradient element are NaN, the
    update is skipped (parameters and moments bit-identical), check_device_errors() raises and clears, and the step after that trains."""
    import torch
    import taco_amd
    hp = O.OracleHParams(max_iters=8)
    w = O.init_weights(hp, 1, 111)
    B, T_in, T_out = 4, 12, 8 * hp.reduction_factor
    ids, L = O.synthetic_inputs(B, T_in, 112, ragged=True)
    rs = np.random.RandomState(113)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    tr = taco_amd.Trainer(to_product_hp(hp), w)
    tr.train_step(ids, L, mt, lt, None)
    torch.cuda.synchronize()
    before = tr.params.clone(); m0 = tr.adam.m.clone()
    tr.raise_device_error_for_test()
    step, lwc = tr.train_step(ids, L, mt, lt, None)
    torch.cuda.synchronize()
    assert not np.isfinite(float(lwc)) and not np.isfinite(float(tr.grads[0]))
    assert torch.equal(tr.adam.m, m0)
    now = tr.params
    for name, shape in tr.spec:        # every trained parameter is untouched (the BatchNorm moving statistics are written by the forward pass itself)
        if "moving_" in name:
            continue
        o, c = tr.offsets[name]
        assert torch.equal(now[o:o + c], before[o:o + c]), name
    with pytest.raises(taco_amd._lib.TacoError):
        tr.check_device_errors()
    step, lwc = tr.train_step(ids, L, mt, lt, None)       # the word was acknowledged: this one is a normal step again
    torch.cuda.synchronize()
    o, c = tr.offsets["linear/kernel"]
    assert np.isfinite(float(lwc)) and not torch.equal(tr.params[o:o + c], before[o:o + c])
    tr.check_device_errors()
    tr.close()


def test_a_restored_checkpoint_continues_the_run_to_the_bit(tmp_path):
    """train.py:175 (tf.train.Saver), :189-193 (saver.restore of the most recent checkpoint), :242-244 (saver.save every checkpoint_interval):
    parameters, BatchNorm moving statistics, Adam's m / v and the step counters are persisted -- a run that is stopped after three steps,
    restored into a NEW trainer and stepped twice more ends with the same bits as the run that never stopped (deterministic reductions are
    the default).  The weights file of a checkpoint is a weight pack the inference model loads; `--initialize_path` (train.py:194-203)
    restores everything and resets global_step only."""
    import torch
    import taco_amd
    hp, w, ids, L, mt, lt, co = _setup("bah_mon", seed=31)
    a = _trainer(hp, w)
    for _ in range(3):
        a.train_step(ids, L, mt, lt, co)
    ck = a.save_checkpoint(str(tmp_path))
    assert ck.endswith("model.ckpt-3.safetensors") and taco_amd.train_ops.list_train_checkpoints(str(tmp_path)) == [(3, ck)]
    for _ in range(2):
        step_a, loss_a = a.train_step(ids, L, mt, lt, co)
    torch.cuda.synchronize()
    b = _trainer(hp, O.init_weights(hp, 1, 99))              # different initial values: everything must come from the checkpoint
    assert b.restore_checkpoint(str(tmp_path)) == 3 and b.global_step == 3 and b.adam.adam_t == 3
    for _ in range(2):
        step_b, loss_b = b.train_step(ids, L, mt, lt, co)
    torch.cuda.synchronize()
    assert step_a == step_b == 5 and float(loss_a) == float(loss_b)
    assert torch.equal(a.params, b.params) and torch.equal(a.adam.m, b.adam.m) and torch.equal(a.adam.v, b.adam.v)
    moving = [k for k, _ in a.spec if k.endswith(("moving_mean", "moving_variance"))]
    w0, wa = w, a.get_weights()
    assert moving and any(maxabs(w0[k], wa[k]) > 1e-6 for k in moving)          # the moving statistics did move, and were part of the state
    # --initialize_path: the schedule restarts, Adam's bias-correction powers do not
    c = _trainer(hp, O.init_weights(hp, 1, 98))
    assert c.restore_checkpoint(ck, reset_global_step=True) == 0 and c.adam.adam_t == 3 and c.global_step == 0
    assert abs(c.learning_rate - O.learning_rate(0, 0.002, 0, True)) < 1e-12
    # the weights file alone serves inference (synthesizer.py:66-67 restores the same variables)
    m = taco_amd.create_model(to_product_hp(hp))
    m.load_weights(taco_amd.weights.load_weights(ck))
    m.initialize(None, None, 1, None)
    lin, al = m.run(inputs=ids, input_lengths=L)
    assert np.isfinite(lin.cpu().numpy()).all()
    # a weight pack without optimizer state is refused, so is a state written for other hyper-parameters
    taco_amd.weights.save_weights(str(tmp_path / "model.ckpt-9.safetensors"), w)
    with pytest.raises(taco_amd._lib.TacoError):
        b.restore_checkpoint(str(tmp_path / "model.ckpt-9.safetensors"))
    hp2, w2 = tiny_hp(attention_type="bah_mon", enc_bank_size=3), None
    d = _trainer(hp2, O.init_weights(hp2, 1, 5))
    with pytest.raises(taco_amd._lib.TacoError):
        d.restore_checkpoint(ck)
    for t in (a, b, c, d):
        t.close()


@pytest.mark.parametrize("persist,bptt,B,planes", [(1, 1, 12, 1), (1, 1, 32, 1), (10, 1, 12, 1), (11, 1, 12, 1), (1, 1, 40, 1), (1, 1, 4, 1), (1, 0, 12, 1),
                                                   (1, 1, 12, 2), (1, 1, 37, 2)])
def test_a_poisoned_workspace_does_not_reach_the_gradients(persist, bptt, B, planes):
    """the zero fills of the post-net scan's gate tape (forward) and of its gate-gradient / r*h buffers (backward) are skipped when
    the scan has no lengths, on the contract that every scan kernel variant writes every element (k_bigru_oct<.., true> / k_bigru_duo<.., true> /
    k_bigru_res, k_bigru_oct_bwd / k_bigru_duo_bwd / k_bigru_rows_bwd).  Enforced here: the whole step workspace is filled with NaN bit
    patterns between two identical steps -- per kernel choice (taco_debug_set_persistent 1 / 10 / 11, rows that select the one-row clusters
    or the 32-CU groups, the per-stage BPTT engine) -- and the second step's losses and gradients must be finite and equal to the first's to
    the bit (ordered reductions are the default).  planes = 2 (round 6): every eligible weight gradient from pre-split planes -- the plane
    scratch (rows padded to 64, columns to 32) is part of the poisoned workspace, and the conv banks' BatchNorm output is no longer stored."""
    import ctypes as C
    import torch
    hp = O.OracleHParams(max_iters=4)
    w = O.init_weights(hp, 1, 141)
    T_in, T_out = 10, 16
    ids, L = O.synthetic_inputs(B, T_in, 142, ragged=True)
    rs = np.random.RandomState(143)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    tr = _trainer(hp, w)
    mh = C.c_void_p(tr._lib.taco_train_model(tr._h))
    assert tr._lib.taco_debug_set_persistent(mh, persist) == 0
    tr.set_bptt_engine(bool(bptt))
    if planes != 1:
        tr.set_wgrad_planes(planes)
    l1 = tr.forward_backward(ids, L, mt, lt, freeze_moving_averages=True).clone()
    assert planes != 2 or tr.planes_problems() >= 40
    torch.cuda.synchronize()
    g1 = tr.grads.clone()
    assert bool(torch.isfinite(g1).all()) and bool(torch.isfinite(l1).all())
    tr._ws.fill_(0xFF)                                        # every fp32 word of the workspace a NaN
    tr.grads.fill_(float("nan"))
    l2 = tr.forward_backward(ids, L, mt, lt, freeze_moving_averages=True)
    torch.cuda.synchronize()
    tr.check_device_errors()
    assert bool(torch.isfinite(tr.grads).all()) and bool(torch.isfinite(l2).all())
    assert torch.equal(tr.grads, g1) and torch.equal(l2, l1)
    tr.close()


def test_C4_shard_gradients_at_32_rows_against_the_committed_autograd_fixture():
    """BASELINE.json configs[3], one data-parallel shard at its real size: B = 32, T_in = 128, T_out = 512 at the reference widths -- every
    gradient tensor's norm and a fixed sample of 2000 of its elements against float64 reverse-mode autograd of the independent torch
    formulation (tests/golden/full_C4_grads.npz, made by tests/golden/make_full_size_golden.py; round 5 compared a 2-row slice)."""
    import os
    import torch
    import make_full_size_golden as G
    p = os.path.join(os.path.dirname(__file__), "golden", "full_C4_grads.npz")
    if not os.path.exists(p):
        pytest.skip("full_C4_grads.npz not generated (python tests/golden/make_full_size_golden.py C4)")
    g = np.load(p)
    hp, seed, ids, L, mt, lt = G.c4_case()
    assert int(g["seed"]) == seed
    tr = _trainer(hp, O.init_weights(hp, 1, seed))
    losses = tr.forward_backward(ids, L, mt, lt)
    torch.cuda.synchronize()
    tr.check_device_errors()
    assert abs(float(losses[0]) - float(g["loss"])) < 1e-4
    got = tr.grad_dict()
    gnorm = float(np.sqrt(sum(float(g["norm:" + k]) ** 2 for k in g["names"])))
    worst = []
    for k in (str(x) for x in g["names"]):
        v = np.asarray(got[k], np.float64).reshape(-1)
        idx = G.grad_sample_index(k, v.size)
        ref, nrm = g["sample:" + k], float(g["norm:" + k])
        scale = max(float(g["max:" + k]), 1e-3 * gnorm)
        worst.append((float(np.abs(v[idx] - ref).max()) / scale, abs(float(np.sqrt((v * v).sum())) - nrm) / max(nrm, 1e-3 * gnorm), k))
    worst.sort(reverse=True)
    print("C4 shard, B = 32: worst sampled element error / tensor scale, norm error:", worst[:4])
    assert worst[0][0] < 2e-3 and max(w[1] for w in worst) < 2e-3, worst[:5]
    tr.close()


@pytest.mark.parametrize("wide,B,T_in,T_out,atype", [(True, 4, 16, 32, "bah_mon"), (True, 3, 21, 100, "bah"), (True, 5, 37, 44, "bah_mon"),
                                                     (False, 3, 9, 12, "bah_mon"), (False, 5, 13, 21, "bah_norm")])
def test_weight_gradients_from_pre_split_planes_equal_the_in_kernel_split(wide, B, T_in, T_out, atype):
    """csrc/taco_wgrad_planes.h (round 6): the operands of a weight gradient converted ONCE into bf16 planes (fragment-major, the conv
    taps' shifts and batch-row masks applied to copies of the narrower operand; a whole conv bank as one product launch) give the
    gradients of k_wgrad_bf3, which converts inside the product loop -- same three-way split, same six products, fp32 sums in another
    slice order.  Mode 2 sends EVERY eligible problem through the planes (mode 1, the default, only the large ones: at these test sizes
    none), so every shape class is covered: odd widths (1025 bins, 80 mels, the scaled model's 8 / 36), row counts that are not
    multiples of 64, taps across batch-row boundaries at lengths that are not multiples of 16, +-1-step shifts of the recurrent kernels.
    Both engines are also held to the float64 autograd."""
    import torch
    if wide:
        hp = O.OracleHParams(max_iters=max(16, T_out // 4), attention_type=atype)
        w = O.init_weights(hp, 1, 41)
        ids, L = O.synthetic_inputs(B, T_in, 42, ragged=True)
        rs = np.random.RandomState(43)
        mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
        co = rs.uniform(0.5, 1.5, size=B)
    else:
        hp, w, ids, L, mt, lt, co = _setup(atype, B=B, T_in=T_in, T_out=T_out)
    tr = _trainer(hp, w)
    got, count = {}, {}
    for mode in (0, 2, 1):
        tr.set_wgrad_planes(mode)
        tr.forward_backward(ids, L, mt, lt, co)
        torch.cuda.synchronize()
        got[mode] = tr.grad_dict()
        count[mode] = tr.planes_problems()
    print("weight gradients from pre-split planes: mode 0 -> %d, mode 2 -> %d, mode 1 -> %d problems" % (count[0], count[2], count[1]))
    assert count[0] == 0 and count[1] == 0 and count[2] >= 40       # (both conv banks, proj_1 / proj_2, highways, GRU kernels, head, decoder)
    gn = max(float(np.abs(v).max()) for v in got[0].values())
    worst = sorted(((maxabs(got[2][k], got[0][k]) / max(float(np.abs(got[0][k]).max()), 1e-3 * gn), k) for k in got[0]), reverse=True)
    print("planes (every eligible problem) vs in-kernel split, relative per tensor, worst three:", worst[:3])
    assert worst[0][0] < 2e-5, worst[:5]
    moved = [k for k in got[0] if not np.array_equal(got[2][k], got[0][k])]      # (one slice on both paths = the same order of the same products)
    print("%d of %d gradient tensors took another summation order" % (len(moved), len(got[0])))
    for k in got[0]:       # the default mode at these sizes: nothing is large enough, bit-identical to mode 0
        assert np.array_equal(got[1][k], got[0][k]), k
    loss, g, _ = TF.train_grads(w, hp, ids, L, mt, lt, co)
    rep, _ = _grad_report(got[2], g)
    assert rep[0][0] < 5e-2, rep[:5]       # (loose: the L1 sign flips of test_gradients_at_full_reference_widths move both engines alike; the tight check is the one above)
    tr.close()

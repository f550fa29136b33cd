"""A SECOND, independently written formulation of the reference's inference forward
(models/tacotron.py:21-251) on torch CPU float64 functional ops -- test infrastructure only.

It deliberately differs in structure from oracle/taco_oracle.py: channels-first F.conv1d with
explicit asymmetric F.pad, F.max_pool1d on a right-padded tensor, batch-norm via F.batch_norm,
GRU gates via separate x/h matmuls on split kernels, the monotonic normaliser through its
sequential recurrence q_j = (1-p_{j-1}) q_{j-1} + prev_j (not the cumprod/cumsum closed form),
backward GRU by index arithmetic instead of reverse_sequence.  Agreement of the two catches
transcription errors in either; it cannot validate the TF-semantics assumptions themselves."""
import torch
import torch.nn.functional as F

DT = torch.float64


def _t(w, name):
    v = w[name]
    return v if torch.is_tensor(v) else torch.as_tensor(v, dtype=DT)


def dense(x, w, name, bias=True):
    y = x @ _t(w, name + "/kernel")
    return y + _t(w, name + "/bias") if bias else y


def conv_bn(x, w, name, act, training=False):
    """x [B,T,C] -> conv1d SAME (modules.py:123-131) -> act -> BN (moving statistics, or batch statistics when training)."""
    k = _t(w, name + "/kernel")                       # [k, in, out]
    kw = k.shape[0]
    pl = (kw - 1) // 2
    xc = F.pad(x.permute(0, 2, 1), (pl, kw - 1 - pl))
    y = F.conv1d(xc, k.permute(2, 1, 0), _t(w, name + "/bias"))
    if act:
        y = F.relu(y)
    if training:      # batch mean / biased variance over (B,T); running statistics are not touched here
        y = F.batch_norm(y, None, None, _t(w, name + "/gamma"), _t(w, name + "/beta"), training=True, eps=1e-3)
    else:
        y = F.batch_norm(y, _t(w, name + "/moving_mean"), _t(w, name + "/moving_variance"), _t(w, name + "/gamma"),
                         _t(w, name + "/beta"), training=False, eps=1e-3)
    return y.permute(0, 2, 1)


def gru_step(x, h, w, name):
    n = h.shape[-1]
    i = x.shape[-1]
    gk, ck = _t(w, name + "/gates/kernel"), _t(w, name + "/candidate/kernel")
    g = torch.sigmoid(x @ gk[:i] + h @ gk[i:] + _t(w, name + "/gates/bias"))
    r, u = g[:, :n], g[:, n:]
    c = torch.tanh(x @ ck[:i] + (r * h) @ ck[i:] + _t(w, name + "/candidate/bias"))
    return u * h + (1 - u) * c


def bigru(x, lengths, w, scope, init=None):
    B, T, _ = x.shape
    n = w[scope + "/fw/candidate/bias"].shape[0]
    L = torch.full((B,), T, dtype=torch.long) if lengths is None else torch.as_tensor(lengths, dtype=torch.long)
    dirs = []
    for d, nm in enumerate(("fw", "bw")):
        rows = []
        for b in range(B):                             # row by row: no masking logic at all
            h = torch.zeros(1, n, dtype=DT) if init is None else init[b:b + 1, d * n:(d + 1) * n].clone()
            order = range(int(L[b])) if d == 0 else range(int(L[b]) - 1, -1, -1)
            seq = [torch.zeros(n, dtype=DT)] * T       # positions past the length stay zero
            for t in order:
                h = gru_step(x[b:b + 1, t], h, w, scope + "/" + nm)
                seq[t] = h[0]
            rows.append(torch.stack(seq, 0))
        dirs.append(torch.stack(rows, 0))
    return torch.cat(dirs, -1)


def cbhg(x, lengths, w, scope, K, depth, nproj, before=None, init=None, training=False):
    bank = torch.cat([conv_bn(x, w, "%s/conv_bank/conv1d_%d" % (scope, k), True, training) for k in range(1, K + 1)], -1)
    mp = F.max_pool1d(F.pad(bank.permute(0, 2, 1), (0, 1), value=float("-inf")), 2, 1).permute(0, 2, 1)
    p = mp
    for i in range(nproj):
        p = conv_bn(p, w, "%s/proj_%d" % (scope, i + 1), i < nproj - 1, training)
    h = p + x
    if before is not None:
        h = h + before[:, None, :]
    if (scope + "/dense/kernel") in w:
        h = dense(h, w, scope + "/dense")
    for i in range(depth):
        nm = "%s/highway_%d" % (scope, i + 1)
        Hh, Tt = F.relu(dense(h, w, nm + "/H")), torch.sigmoid(dense(h, w, nm + "/T"))
        h = Hh * Tt + h * (1 - Tt)
    return bigru(h, lengths, w, scope + "/bigru", init)


def monotonic_sequential(p, prev):
    qs = [prev[:, 0]]
    for j in range(1, p.shape[1]):
        qs.append((1 - p[:, j - 1]) * qs[-1] + prev[:, j])
    return p * torch.stack(qs, 1)


def monotonic_closed_form(p, prev):
    """tf.contrib.seq2seq.monotonic_attention(mode='parallel') op for op -- what tf.gradients differentiates:
    p * safe_cumprod(1-p, exclusive) * cumsum(prev / clip(safe_cumprod, 1e-10, 1)).  The clips stop gradients
    where they are active, which the sequential recurrence above would not reproduce."""
    tiny = torch.finfo(p.dtype).tiny
    lg = torch.log(torch.clamp(1 - p, tiny, 1.0))
    cp = torch.exp(torch.cumsum(lg, 1) - lg)                 # exclusive cumsum
    return p * cp * torch.cumsum(prev / torch.clamp(cp, 1e-10, 1.0), 1)


def forward(w, hp, ids, lengths, speaker_id=None, num_speakers=1, n_steps=None, manual=None, training=False,
            teacher_frames=None, as_numpy=True):
    ids = torch.as_tensor(ids, dtype=torch.long)
    B, T_in = ids.shape
    r, M = hp.reduction_factor, hp.num_mels
    n = hp.max_iters if n_steps is None else n_steps
    x = _t(w, "embedding")[ids]
    spk = before = enc_init = att_init = dec_inits = None
    if num_speakers > 1:
        sid = torch.as_tensor(speaker_id, dtype=torch.long)
        if hp.speaker_embedding_size != 1:
            spk = _t(w, "speaker_embedding")[sid]
        if hp.model_type == "deepvoice":
            names = ["before_highway", "encoder_rnn_init", "attention_rnn_init"] + \
                    ["decoder_rnn_init_%d" % (i + 1) for i in range(hp.dec_layer_num)]
            vs = [(_t(w, "spk/%s/table" % nm)[sid] if hp.speaker_embedding_size == 1 else F.softsign(dense(spk, w, "spk/" + nm)))
                  for nm in names]
            before, enc_init, att_init, dec_inits = vs[0], vs[1], vs[2], vs[3:]
            spk = None
    for i in range(len(hp.enc_prenet_sizes)):
        x = F.relu(dense(x, w, "prenet/dense_%d" % (i + 1)))
    enc = cbhg(x, lengths, w, "encoder_cbhg", hp.enc_bank_size, hp.enc_highway_depth, len(hp.enc_proj_sizes), before, enc_init,
               training)
    keys = dense(enc, w, "attention/memory_layer", bias=False)
    v = _t(w, "attention/attention_v")
    h_att = torch.zeros(B, hp.attention_state_size, dtype=DT) if att_init is None else att_init.clone()
    hs = [torch.zeros(B, hp.dec_rnn_size, dtype=DT) if dec_inits is None else dec_inits[i].clone() for i in range(hp.dec_layer_num)]
    ctx = torch.zeros(B, enc.shape[-1], dtype=DT)
    alpha = torch.zeros(B, T_in, dtype=DT)
    if hp.attention_type == "bah_mon":
        alpha[:, 0] = 1
    frame = torch.zeros(B, M, dtype=DT)
    ys, als = [], []
    for t in range(n):
        z = torch.cat([frame, ctx], -1)
        for i in range(len(hp.dec_prenet_sizes)):
            z = F.relu(dense(z, w, "decoder/prenet/dense_%d" % (i + 1)))
        if spk is not None:
            z = torch.cat([z, spk], -1)
        h_att = gru_step(z, h_att, w, "decoder/attention_gru")
        qv = dense(h_att, w, "attention/query_layer", bias=False)
        if hp.attention_type == "bah_norm":
            nv = _t(w, "attention/attention_g") * v / v.norm()
            e = (nv * torch.tanh(keys + qv[:, None] + _t(w, "attention/attention_b"))).sum(-1)
        else:
            e = (v * torch.tanh(keys + qv[:, None])).sum(-1)
        if hp.attention_type == "bah_mon":
            mono = monotonic_closed_form if training else monotonic_sequential
            alpha = mono(torch.sigmoid(e + _t(w, "attention/attention_score_bias")), alpha)
        else:
            alpha = F.softmax(e, dim=1)
        if manual is not None:
            alpha = torch.as_tensor(manual, dtype=DT)[:, t]
        ctx = torch.bmm(alpha[:, None], enc)[:, 0]
        als.append(alpha)
        o = dense(torch.cat([h_att, ctx] + ([spk] if spk is not None else []), -1), w, "decoder/concat_projection")
        for i in range(hp.dec_layer_num):
            hs[i] = gru_step(o, hs[i], w, "decoder/gru_%d" % (i + 1))
            o = o + hs[i]
        y = dense(o, w, "decoder/frame_projection")
        ys.append(y)
        frame = y[:, -M:]
        if teacher_frames is not None:
            frame = torch.as_tensor(teacher_frames, dtype=DT)[:, t]
    mel = torch.stack(ys, 1).reshape(B, n * r, M)
    post = cbhg(mel, None, w, "post_cbhg", hp.post_bank_size, hp.post_highway_depth, len(hp.post_proj_sizes),
                training=training)
    if spk is not None:
        post = torch.cat([spk[:, None].expand(B, post.shape[1], spk.shape[1]), post], -1)
    linear = dense(post, w, "linear")
    if not as_numpy:
        return dict(mel=mel, linear=linear, alignments=torch.stack(als, 2))
    return dict(mel=mel.detach().numpy(), linear=linear.detach().numpy(), alignments=torch.stack(als, 2).detach().numpy())


def add_loss(mel_out, mel_tgt, lin_out, lin_tgt, coeff, prioritize_loss=False, sample_rate=24000):
    """tacotron.py:274-302 on torch tensors; returns the scalar `loss` the optimizer minimises."""
    ml = (mel_tgt - mel_out).abs()
    l1 = (lin_tgt - lin_out).abs()
    c = coeff[:, None, None]
    if prioritize_loss:
        Fq = lin_out.shape[-1]
        up, lo = int(5000 / (sample_rate * 0.5) * Fq), int(165 / (sample_rate * 0.5) * Fq)
        return (ml * c).mean() + 0.5 * (l1 * c).mean() + 0.5 * (l1[:, :, lo:up] * c).mean()
    return (ml * c).mean() + (l1 * c).mean()


def train_grads(w, hp, ids, lengths, mel_targets, linear_targets, loss_coeff=None, speaker_id=None, num_speakers=1,
                prioritize_loss=False, sample_rate=24000):
    """The training graph of tacotron.py:26,199-202,274-302 + tf.gradients: teacher-forced forward with batch-stat
    BN, L1 losses, reverse-mode autograd.  Returns (loss, dict name -> d loss / d weight as float64 arrays,
    forward outputs).  Moving statistics get no gradient (they are not trainable variables)."""
    r = hp.reduction_factor
    mt = torch.as_tensor(mel_targets, dtype=DT)
    lt = torch.as_tensor(linear_targets, dtype=DT)
    B, T_out, _ = mt.shape
    assert T_out % r == 0
    co = torch.ones(B, dtype=DT) if loss_coeff is None else torch.as_tensor(loss_coeff, dtype=DT)
    wt = {}
    for k, v in w.items():
        t = torch.tensor(v, dtype=DT)
        if not k.endswith(("/moving_mean", "/moving_variance")):
            t.requires_grad_(True)
        wt[k] = t
    out = forward(wt, hp, ids, lengths, speaker_id, num_speakers, n_steps=T_out // r, training=True,
                  teacher_frames=mt[:, r - 1::r], as_numpy=False)
    loss = add_loss(out["mel"], mt, out["linear"], lt, co, prioritize_loss, sample_rate)
    loss.backward()
    grads = {k: (t.grad.numpy() if t.grad is not None else None) for k, t in wt.items() if t.requires_grad}
    return float(loss.detach()), grads, {k: v.detach().numpy() for k, v in out.items()}

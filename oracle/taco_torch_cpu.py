"""oracle/taco_torch_cpu.py -- TEST / BENCH INFRASTRUCTURE, never imported by the product.

The CPU baseline SURVEY section 8(d) names: an fp32 PyTorch-CPU, eager, op-for-op restatement of the TF1 inference graph of
models/tacotron.py:21-251 (one torch op where the reference has one TensorFlow op: conv1d, dense, max_pooling1d, the GRUCell's two
matmuls, the attention score / normaliser, ...), so that its timing has the same shape as the reference's CPU path: a Python-side loop
over decoder steps (tf.while_loop there) around small dense ops.  It is NOT TensorFlow and its arithmetic is pinned only through the
NumPy oracle it restates (oracle/taco_oracle.py; tests/test_oracle.py::test_torch_cpu_restatement_equals_the_numpy_oracle holds the
two to each other); "parity unpinned" applies to it exactly as to the oracle.

Only bench.py's cpu_baseline leg and tests/ use it.  Citations: see the function of the same name in taco_oracle.py."""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


class TorchCpuTacotron:
    """Weights converted once (as a TF session holds its variables); forward() is the timed part."""

    def __init__(self, w, hp, num_speakers=1, dtype=torch.float32):
        self.hp, self.ns, self.dt = hp, num_speakers, dtype
        self.w = {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))).to(dtype) for k, v in w.items()}
        # tf.layers.conv1d kernels [k, in, out] -> torch [out, in, k]
        self.ck = {k: v.permute(2, 1, 0).contiguous() for k, v in self.w.items() if v.dim() == 3}

    # ---- modules.py ----
    def dense(self, x, name, act=None, bias=True):                                  # A.1
        y = x @ self.w[name + "/kernel"]
        if bias:
            y = y + self.w[name + "/bias"]
        return act(y) if act is not None else y

    def conv1d_bn(self, x, name, act):                                              # modules.py:123-131, A.2: conv -> activation -> BN
        k = self.ck[name + "/kernel"].shape[2]
        pl = (k - 1) // 2
        y = F.conv1d(F.pad(x.transpose(1, 2), (pl, k - 1 - pl)), self.ck[name + "/kernel"], self.w[name + "/bias"]).transpose(1, 2)
        if act is not None:
            y = act(y)
        w = self.w
        return w[name + "/gamma"] * (y - w[name + "/moving_mean"]) / torch.sqrt(w[name + "/moving_variance"] + BN_EPS) + w[name + "/beta"]

    @staticmethod
    def maxpool(x, width):                                                          # A.3: pad right with -inf
        if width == 1:
            return x
        pl = (width - 1) // 2
        xp = F.pad(x.transpose(1, 2), (pl, width - 1 - pl), value=float("-inf"))
        return F.max_pool1d(xp, width, 1).transpose(1, 2)

    def prenet(self, x, scope, sizes):                                              # modules.py:18-25
        for i in range(len(sizes)):
            x = self.dense(x, "%s/dense_%d" % (scope, i + 1), torch.relu)
        return x

    def highway(self, x, name):                                                     # modules.py:105-120
        H = self.dense(x, name + "/H", torch.relu)
        T = self.dense(x, name + "/T", torch.sigmoid)
        return H * T + x * (1.0 - T)

    def gru_cell(self, x, h, name):                                                 # A.6
        n = h.shape[-1]
        g = torch.sigmoid(torch.cat([x, h], -1) @ self.w[name + "/gates/kernel"] + self.w[name + "/gates/bias"])
        r, u = g[..., :n], g[..., n:]
        c = torch.tanh(torch.cat([x, r * h], -1) @ self.w[name + "/candidate/kernel"] + self.w[name + "/candidate/bias"])
        return u * h + (1.0 - u) * c

    def dynamic_gru(self, x, lengths, name, h0):                                    # A.7
        B, T, _ = x.shape
        n = self.w[name + "/candidate/bias"].shape[0]
        h = torch.zeros(B, n, dtype=self.dt) if h0 is None else h0.clone()
        out = []
        for t in range(T):
            hn = self.gru_cell(x[:, t], h, name)
            if lengths is None:
                h = hn
                out.append(hn)
            else:
                act = (t < lengths)[:, None]
                h = torch.where(act, hn, h)
                out.append(torch.where(act, hn, torch.zeros_like(hn)))
        return torch.stack(out, 1)

    @staticmethod
    def reverse_sequence(x, lengths):
        if lengths is None:
            return x.flip(1)
        y = x.clone()
        for b in range(x.shape[0]):
            L = int(lengths[b])
            y[b, :L] = x[b, :L].flip(0)
        return y

    def bigru(self, x, lengths, scope, init):                                       # modules.py:82-96
        h0f = h0b = None
        if init is not None:
            n = init.shape[1] // 2
            h0f, h0b = init[:, :n], init[:, n:]
        fw = self.dynamic_gru(x, lengths, scope + "/fw", h0f)
        bw = self.reverse_sequence(self.dynamic_gru(self.reverse_sequence(x, lengths), lengths, scope + "/bw", h0b), lengths)
        return torch.cat([fw, bw], -1)

    def cbhg(self, x, lengths, scope, K, mpw, depth, projs, before_highway=None, rnn_init=None):   # modules.py:27-96
        bank = torch.cat([self.conv1d_bn(x, "%s/conv_bank/conv1d_%d" % (scope, k), torch.relu) for k in range(1, K + 1)], -1)
        p = self.maxpool(bank, mpw)
        for i in range(len(projs)):
            p = self.conv1d_bn(p, "%s/proj_%d" % (scope, i + 1), None if i == len(projs) - 1 else torch.relu)
        hi = p + x
        if before_highway is not None:
            hi = hi + before_highway[:, None, :]
        if (scope + "/dense/kernel") in self.w:
            hi = self.dense(hi, scope + "/dense")
        for i in range(depth):
            hi = self.highway(hi, "%s/highway_%d" % (scope, i + 1))
        return self.bigru(hi, lengths, scope + "/bigru", rnn_init)

    # ---- attention (A.8-A.10) ----
    def alignments(self, q, keys, prev):
        hp, w = self.hp, self.w
        v = w["attention/attention_v"]
        if hp.attention_type == "bah_norm":
            e = torch.sum(w["attention/attention_g"] * v / torch.sqrt(torch.sum(v * v)) * torch.tanh(keys + q[:, None, :] + w["attention/attention_b"]), 2)
        else:
            e = torch.sum(v * torch.tanh(keys + q[:, None, :]), 2)
        if hp.attention_type == "bah_mon":
            p = torch.sigmoid(e + w["attention/attention_score_bias"])
            tiny = float(np.finfo(np.float32).tiny)
            logs = torch.log(torch.clamp(1.0 - p, tiny, 1.0))
            cp = torch.exp(torch.cumsum(logs, 1) - logs)
            return p * cp * torch.cumsum(prev / torch.clamp(cp, 1e-10, 1.0), 1)
        return torch.softmax(e, 1)

    # ---- models/tacotron.py:21-251, inference ----
    @torch.no_grad()
    def forward(self, inputs, input_lengths, speaker_id=None, n_steps=None):
        hp, w, dt = self.hp, self.w, self.dt
        ids = torch.from_numpy(np.asarray(inputs).astype(np.int64))
        lengths = torch.from_numpy(np.asarray(input_lengths).astype(np.int64))
        B, T_in = ids.shape
        r, M = hp.reduction_factor, hp.num_mels
        n = hp.max_iters if n_steps is None else n_steps
        x = w["embedding"][ids]
        spk = before_highway = enc_init = att_init = dec_inits = None
        if self.ns > 1:
            sid = torch.zeros(B, dtype=torch.int64) if speaker_id is None else torch.from_numpy(np.asarray(speaker_id).astype(np.int64))
            if hp.speaker_embedding_size != 1:
                spk = w["speaker_embedding"][sid]
            if hp.model_type == "deepvoice":
                names = ["before_highway", "encoder_rnn_init", "attention_rnn_init"] + ["decoder_rnn_init_%d" % (i + 1) for i in range(hp.dec_layer_num)]
                softsign = lambda t: t / (1.0 + t.abs())
                vecs = [w["spk/%s/table" % nm][sid] if hp.speaker_embedding_size == 1 else self.dense(spk, "spk/" + nm, softsign) for nm in names]
                before_highway, enc_init, att_init, dec_inits, spk = vecs[0], vecs[1], vecs[2], vecs[3:], None
        pre = self.prenet(x, "prenet", hp.enc_prenet_sizes)
        enc = self.cbhg(pre, lengths, "encoder_cbhg", hp.enc_bank_size, hp.enc_maxpool_width, hp.enc_highway_depth, hp.enc_proj_sizes, before_highway, enc_init)
        values = enc
        keys = self.dense(values, "attention/memory_layer", bias=False)
        h_att = torch.zeros(B, hp.attention_state_size, dtype=dt) if att_init is None else att_init.clone()
        hs = [torch.zeros(B, hp.dec_rnn_size, dtype=dt) if dec_inits is None else dec_inits[i].clone() for i in range(hp.dec_layer_num)]
        ctx = torch.zeros(B, enc.shape[-1], dtype=dt)
        alpha = torch.zeros(B, T_in, dtype=dt)
        if hp.attention_type == "bah_mon":
            alpha[:, 0] = 1.0
        frame = torch.zeros(B, M, dtype=dt)
        Y, hist = [], []
        for t in range(n):
            z = self.prenet(torch.cat([frame, ctx], -1), "decoder/prenet", hp.dec_prenet_sizes)
            if spk is not None:
                z = torch.cat([z, spk], -1)
            h_att = self.gru_cell(z, h_att, "decoder/attention_gru")
            alpha = self.alignments(self.dense(h_att, "attention/query_layer", bias=False), keys, alpha)
            ctx = torch.einsum("bj,bjd->bd", alpha, values)
            hist.append(alpha)
            o = self.dense(torch.cat([h_att, ctx] + ([spk] if spk is not None else []), -1), "decoder/concat_projection")
            for i in range(hp.dec_layer_num):
                hs[i] = self.gru_cell(o, hs[i], "decoder/gru_%d" % (i + 1))
                o = hs[i] + o
            y = self.dense(o, "decoder/frame_projection")
            Y.append(y)
            frame = y[:, -M:]
        mel = torch.stack(Y, 1).reshape(B, n * r, M)
        post = self.cbhg(mel, None, "post_cbhg", hp.post_bank_size, hp.post_maxpool_width, hp.post_highway_depth, hp.post_proj_sizes)
        if spk is not None:
            post = torch.cat([spk[:, None, :].expand(B, post.shape[1], spk.shape[1]), post], -1)
        linear = self.dense(post, "linear")
        return dict(mel=mel.numpy(), linear=linear.numpy(), alignments=torch.stack(hist, 2).numpy())

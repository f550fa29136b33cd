"""CPU numerics experiment for DESIGN section 6 item (6): would a GRU recurrence whose recurrent products run on the bf16 matrix cores as
SPLIT products (state split into three bf16 planes by its producer, weights held as two or three bf16 planes, the significant cross
products accumulated in fp32) stay as close to the float64 oracle as today's exact-fp32 FMAs do -- over the 512 dependent steps of the
post-net scan at C2 and the 4000 of C5?  Emulated in NumPy (round-to-nearest-even bf16 planes, float32 accumulation); no GPU involved.
The x-projection (the hoisted GEMM) is given in float32 to every variant, as in the product.

    python tools/sim_split_recurrence.py [--T 512] [--B 4] [--H 256] [--seed 0]
    python tools/sim_split_recurrence.py --decoder [--B 4] [--T_in 48] [--steps 96] [--atype bah_mon]

--decoder: the whole oracle forward at the reference widths in float32, with every matrix product INSIDE the decoder loop (prenet,
attention GRU, query, concat projection, the two GRUs, frame projection) on the chosen engine; reports mel / alignment error against
the float64 oracle and how many (row, step) attention argmaxes differ (ties below 1e-6 of the peak not counted).

Prints max |h_t - h_t(float64)| over all steps and the value at the last step for:
    fp32        plain float32 recurrence (what k_bigru_duo computes, up to summation order)
    s3x3_6      state 3 planes x weights 3 planes, six products (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi)
    s3x2_5      state 3 planes x weights 2 planes, five products (hi*hi, hi*lo_w, mid*hi, mid*lo_w, lo*hi)
    s2x2_3      state 2 planes x weights 2 planes, three products (hi*hi, hi*lo, lo*hi)
    bf16        single bf16 plane each (for scale)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))


def bf16(x):
    """float32 -> nearest-even bfloat16, returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def planes(x, n):
    x = np.asarray(x, np.float32)
    out, rest = [], x
    for _ in range(n):
        p = bf16(rest)
        out.append(p)
        rest = (rest - p).astype(np.float32)
    return out


def mm32(a, b):
    return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)


class Split(object):
    def __init__(self, na, nw, pairs):
        self.na, self.nw, self.pairs = na, nw, pairs

    def prep(self, W):
        return planes(W, self.nw)

    def mm(self, a, Wp):
        ap = planes(a, self.na)
        acc = np.zeros((a.shape[0], Wp[0].shape[1]), np.float32)
        for i, j in sorted(self.pairs, key=lambda p: -(p[0] + p[1])):      # small terms first
            acc = (acc + mm32(ap[i], Wp[j])).astype(np.float32)
        return acc


class Plain(object):
    def __init__(self, dt):
        self.dt = dt

    def prep(self, W):
        return W.astype(self.dt)

    def mm(self, a, Wp):
        return a.astype(self.dt) @ Wp


VARIANTS = {
    "fp32": Plain(np.float32),
    "s3x3_6": Split(3, 3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]),
    "s3x2_5": Split(3, 2, [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)]),
    "s2x2_3": Split(2, 2, [(0, 0), (0, 1), (1, 0)]),
    "bf16": Split(1, 1, [(0, 0)]),
}


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def scan(xg, xc, Wg, Wc, eng, dt):
    """xg [T,B,2H], xc [T,B,H]: x-part of gates / candidate incl. biases; Wg [H,2H], Wc [H,H] recurrent halves"""
    T, B, H2 = xg.shape
    H = H2 // 2
    Wgp, Wcp = eng.prep(Wg), eng.prep(Wc)
    h = np.zeros((B, H), dt)
    out = np.zeros((T, B, H), np.float64)
    for t in range(T):
        g = sigmoid((xg[t].astype(dt) + eng.mm(h, Wgp)).astype(dt))
        r, u = g[:, :H], g[:, H:]
        c = np.tanh((xc[t].astype(dt) + eng.mm((r * h).astype(dt), Wcp)).astype(dt))
        h = (u * h + (1 - u) * c).astype(dt)
        out[t] = h
    return out


def decoder_experiment(a):
    import taco_oracle as O
    hp = O.OracleHParams(max_iters=a.steps, attention_type=a.atype)
    w = O.init_weights(hp, 1, 1234 + a.seed)
    ids, L = O.synthetic_inputs(a.B, a.T_in, 77 + a.seed, ragged=True)
    ref = O.forward(w, hp, ids, L, n_steps=a.steps, honor_stop=False)
    plain_dense, plain_gru = O.dense, O.gru_cell
    print("decoder loop, %s, B=%d T_in=%d steps=%d; |mel| max %.3f" % (a.atype, a.B, a.T_in, a.steps, np.abs(ref["mel"]).max()))
    for name, eng in VARIANTS.items():
        cache = {}

        def W(key, arr):
            if key not in cache:
                cache[key] = eng.prep(np.asarray(arr, np.float32))
            return cache[key]

        def in_loop(nm):
            return nm.startswith("decoder/") or nm == "attention/query_layer"

        def dense(x, w_, nm, act=None, bias=True):
            if not in_loop(nm) or x.ndim != 2:
                return plain_dense(x, w_, nm, act, bias)
            y = eng.mm(x, W(nm, w_[nm + "/kernel"])).astype(np.float32)
            if bias:
                y = y + w_[nm + "/bias"].astype(np.float32)
            return act(y) if act is not None else y

        def gru_cell(x, h, w_, nm):
            n = h.shape[-1]
            g = O.sigmoid(eng.mm(np.concatenate([x, h], -1), W(nm + "/g", w_[nm + "/gates/kernel"])).astype(np.float32) + w_[nm + "/gates/bias"].astype(np.float32))
            r, u = g[..., :n], g[..., n:]
            c = np.tanh(eng.mm(np.concatenate([x, r * h], -1), W(nm + "/c", w_[nm + "/candidate/kernel"])).astype(np.float32) + w_[nm + "/candidate/bias"].astype(np.float32))
            return (u * h + (1.0 - u) * c).astype(np.float32)

        O.dense, O.gru_cell = dense, (lambda x, h, w_, nm: gru_cell(x, h, w_, nm) if nm.startswith("decoder/") else plain_gru(x, h, w_, nm))
        try:
            got = O.forward(w, hp, ids, L, n_steps=a.steps, honor_stop=False, dtype=np.float32)
        finally:
            O.dense, O.gru_cell = plain_dense, plain_gru
        ra, ga = ref["alignments"], got["alignments"].astype(np.float64)
        am_r, am_g = ra.argmax(1), ga.argmax(1)
        diff = 0
        for b, t in zip(*np.nonzero(am_r != am_g)):
            peak = ra[b, :, t].max()
            if peak - ra[b, am_g[b, t], t] > 1e-6 * peak:
                diff += 1
        print("  %-7s mel max err %.3e   alignment max err %.3e   argmax differs at %d of %d (row, step)"
              % (name, np.abs(got["mel"] - ref["mel"]).max(), np.abs(ga - ra).max(), diff, am_r.size))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--H", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--decoder", action="store_true")
    ap.add_argument("--T_in", type=int, default=48)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--atype", default="bah_mon")
    a = ap.parse_args()
    if a.decoder:
        return decoder_experiment(a)
    rs = np.random.RandomState(a.seed)
    H, I = a.H, a.H
    lim_g, lim_c = np.sqrt(6.0 / (I + H + 2 * H)), np.sqrt(6.0 / (I + H + H))       # Glorot-uniform, TF's GRUCell kernels [I+H, 2H] / [I+H, H]
    Wg_full = rs.uniform(-lim_g, lim_g, size=(I + H, 2 * H))
    Wc_full = rs.uniform(-lim_c, lim_c, size=(I + H, H))
    x = rs.normal(0, 1.0, size=(a.T, a.B, I))                                           # highway outputs are O(1)
    xg = (x @ Wg_full[:I] + 1.0).astype(np.float32)                                     # gate bias 1.0 (TF-sem)
    xc = (x @ Wc_full[:I]).astype(np.float32)
    Wg, Wc = Wg_full[I:], Wc_full[I:]
    ref = scan(xg.astype(np.float64), xc.astype(np.float64), Wg, Wc, Plain(np.float64), np.float64)
    print("GRU recurrence H=%d, B=%d, T=%d, |h| max %.3f, rms %.3f" % (H, a.B, a.T, np.abs(ref).max(), np.sqrt((ref ** 2).mean())))
    for name, eng in VARIANTS.items():
        got = scan(xg, xc, Wg.astype(np.float32), Wc.astype(np.float32), eng, np.float32)
        err = np.abs(got - ref)
        print("  %-7s max over all steps %.3e   last step %.3e   rms %.3e" % (name, err.max(), err[-1].max(), np.sqrt((err ** 2).mean())))


if __name__ == "__main__":
    main()

"""Full-size oracle outputs for the BASELINE.json configurations whose GPU tests only ran slices:

  full_C3.npz        C3 = 32 rows x 128 decoder steps, deepvoice, 4 speakers (the GPU test ran 16 steps, or 8 of the 32 rows)
  full_C5.npz        C5 = 8 rows x T_in 512 x 1000 decoder steps (the GPU test compared 2 of the 8 rows)
  full_C4_grads.npz  C4 shard = the train step's gradients at B = 32, T_in = 128, T_out = 512 (the GPU test compared a 2-row slice)

Everything comes from the float64 oracle (oracle/taco_oracle.py; gradients: reverse-mode autograd of tests/torch_formulation.py), seeded
exactly like the slice tests, so the fixtures are oracle-made: 'parity unpinned' applies to them as to every oracle comparison.  The
files hold float32 subsets sized for the repository -- mel in full (C3) or every 4th decoder step (C5), the linear output on a frame
stride, alignments in full (C3) or their arg-max / peak per step plus every 8th step (C5), gradients as per-tensor norms plus a fixed random
sample of elements -- together with the inputs.  Weights are re-created by the tests from the seeds recorded here.
Minutes of CPU per file:  python tests/golden/make_full_size_golden.py [C3] [C5] [C4]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import taco_oracle as O  # noqa: E402

C3_LINEAR_STRIDE, C5_LINEAR_STRIDE, C5_STEP_STRIDE, C5_ALIGN_STRIDE, GRAD_SAMPLES = 8, 64, 4, 8, 2000


def c3_case():
    B, T_in, r, n, ns, mt = O.CONFIGS["C3"]
    hp = O.OracleHParams(max_iters=n, reduction_factor=r, model_type=mt)
    seed = 1234 + 2
    ids, L = O.synthetic_inputs(B, T_in, seed, ragged=True)
    spk = (np.arange(B) % ns).astype(np.int32)
    return hp, ns, seed, ids, L, spk


def c5_case():
    B, T_in, r, n, ns, mt = O.CONFIGS["C5"]
    hp = O.OracleHParams(max_iters=n, reduction_factor=r)
    seed = 1234 + 4
    ids, L = O.synthetic_inputs(B, T_in, seed, ragged=True)
    return hp, seed, ids, L


def c4_case():
    B, T_in, T_out = 32, 128, 512
    hp = O.OracleHParams(max_iters=T_out // 4)
    seed = 1234 + 3
    ids, L = O.synthetic_inputs(B, T_in, seed, ragged=True)
    rs = np.random.RandomState(seed)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    return hp, seed, ids, L, mt, lt


def grad_sample_index(name, size):
    """The fixed sample of a gradient tensor's elements (all of them when the tensor is small)."""
    if size <= GRAD_SAMPLES:
        return np.arange(size)
    import zlib
    return np.sort(np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF).choice(size, GRAD_SAMPLES, replace=False))


def make_c3():
    hp, ns, seed, ids, L, spk = c3_case()
    w = O.init_weights(hp, ns, seed)
    t = time.time()
    out = O.forward(w, hp, ids, L, speaker_id=spk, num_speakers=ns, honor_stop=False)
    np.savez_compressed(os.path.join(HERE, "full_C3.npz"), seed=seed, inputs=ids, input_lengths=L, speaker_id=spk,
                        mel=out["mel"].astype(np.float32), alignments=out["alignments"].astype(np.float32),
                        linear_stride=C3_LINEAR_STRIDE, linear=out["linear"][:, ::C3_LINEAR_STRIDE].astype(np.float32),
                        stop_step=int(out["stop_step"]))
    print("full_C3.npz: %.0f s" % (time.time() - t))


def make_c5():
    hp, seed, ids, L = c5_case()
    w = O.init_weights(hp, 1, seed)
    t = time.time()
    out = O.forward(w, hp, ids, L, honor_stop=False)
    B, n, r, M = ids.shape[0], hp.max_iters, hp.reduction_factor, hp.num_mels
    mel_steps = out["mel"].reshape(B, n, r * M)[:, ::C5_STEP_STRIDE]
    al = out["alignments"]                                                       # [B, T_in, n]
    np.savez_compressed(os.path.join(HERE, "full_C5.npz"), seed=seed, inputs=ids, input_lengths=L,
                        step_stride=C5_STEP_STRIDE, mel_steps=mel_steps.astype(np.float32),
                        align_argmax=al.argmax(axis=1).astype(np.int16), align_peak=al.max(axis=1).astype(np.float64),
                        align_second=np.sort(al, axis=1)[:, -2, :].astype(np.float64),
                        align_stride=C5_ALIGN_STRIDE, alignments=al[:, :, ::C5_ALIGN_STRIDE].astype(np.float32),
                        linear_stride=C5_LINEAR_STRIDE, linear=out["linear"][:, ::C5_LINEAR_STRIDE].astype(np.float32),
                        stop_step=int(out["stop_step"]))
    print("full_C5.npz: %.0f s" % (time.time() - t))


def make_c4():
    import torch_formulation as TF
    hp, seed, ids, L, mt, lt = c4_case()
    w = O.init_weights(hp, 1, seed)
    t = time.time()
    loss, g, _ = TF.train_grads(w, hp, ids, L, mt, lt)
    names = sorted(g)
    save = {"seed": seed, "loss": float(loss), "names": np.array(names)}
    for k in names:
        v = np.asarray(g[k], np.float64).reshape(-1)
        save["norm:" + k] = float(np.sqrt((v * v).sum()))
        save["max:" + k] = float(np.abs(v).max()) if v.size else 0.0
        save["sample:" + k] = v[grad_sample_index(k, v.size)]
    np.savez_compressed(os.path.join(HERE, "full_C4_grads.npz"), **save)
    print("full_C4_grads.npz: %.0f s, loss %.6f, %d tensors" % (time.time() - t, loss, len(names)))


if __name__ == "__main__":
    which = sys.argv[1:] or ["C3", "C5", "C4"]
    for k in which:
        {"C3": make_c3, "C5": make_c5, "C4": make_c4}[k]()

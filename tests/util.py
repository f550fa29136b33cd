"""Shared helpers for the parity tests: tiny configs, oracle <-> product hparams, C-ABI op calls."""
import ctypes as C

import numpy as np

import taco_oracle as O


def tiny_hp(**kw):
    """Shrunken widths (the reference's SCALE_FACTOR idea, hparams.py:3-6): everything /8."""
    base = dict(num_mels=8, num_freq=36, enc_bank_size=5, post_bank_size=4, max_iters=7, reduction_factor=3)
    base.update(kw)
    return O.OracleHParams.scaled(8, **base)


def to_product_hp(ohp):
    import taco_amd
    return taco_amd.HParams(**ohp.to_dict())


def build_model(ohp, weights, num_speakers=1):
    import taco_amd
    m = taco_amd.create_model(to_product_hp(ohp))
    m.load_weights(weights)
    m.initialize(None, None, num_speakers, None)
    return m


def dev(x, dtype=None):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0


def argmax_match(a_hip, a_ref, floor=1e-6, tie=4e-6):
    """alignment argmax over the encoder axis must be identical wherever the reference's peak is
    above `floor` (below it the monotonic mass has leaked past the last encoder step and fp32/fp64
    underflow differently); returns (n_checked, n_mismatch).  A step whose float64 peak and the value the HIP path picked differ
    by less than `tie` of the peak is a tie at fp32 resolution (2^-23 per operation, a few operations deep) and no mismatch:
    at C2 with tools/parity_margins.py's seed one of 2784 steps has its top two positions 1.3e-6 apart, and the exact-fp32 path
    computes them EQUAL."""
    a_hip, a_ref = np.asarray(a_hip), np.asarray(a_ref)
    peak = a_ref.max(axis=1)
    sel = peak > floor
    picked = np.take_along_axis(a_ref, a_hip.argmax(axis=1)[:, None, :], axis=1)[:, 0, :]     # the oracle's value where HIP peaks
    mism = (a_hip.argmax(axis=1) != a_ref.argmax(axis=1)) & sel & ((peak - picked) > tie * peak)
    return int(sel.sum()), int(mism.sum())

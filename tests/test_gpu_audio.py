"""Spectrogram -> waveform on the GPU vs oracle/audio_oracle.py (float64 restatement of
audio/__init__.py + librosa's stft/istft)."""
import numpy as np
import pytest

import audio_oracle as A

pytestmark = pytest.mark.gpu


class _HP(object):
    def __init__(self, a):
        self.__dict__.update(a.__dict__)


def _cmp(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("iters", [0, 1, 4])
def test_small_stft_parameters_match_oracle(iters):
    import torch, taco_amd
    ahp = A.AudioHParams(num_freq=65, sample_rate=1600, frame_length_ms=50, frame_shift_ms=12.5, griffin_lim_iters=3)   # n_fft 128, hop 20, win 80
    rs = np.random.RandomState(iters)
    B, T = 3, 37
    spec = rs.rand(B, T, 65) * 1.2 - 0.1                      # exercises the clip to [0,1]
    u = rs.rand(B, T, 65)
    gl = taco_amd.GriffinLim(_HP(ahp))
    wav = gl.inv_spectrogram(spec, init_uniform=u, iters=iters).cpu().numpy()
    assert wav.shape == (B, 20 * (T - 1))
    for b in range(B):
        ref = A.inv_spectrogram(spec[b].T, ahp, u[b].T, iters=iters)
        assert _cmp(wav[b], ref) < 2e-4, (b, _cmp(wav[b], ref))
    gl.close()


def test_reference_stft_parameters_one_iteration():
    """n_fft 2048, hop 300, win 1200 (24 kHz, 50 ms / 12.5 ms), 1025 bins."""
    import torch, taco_amd
    ahp = A.AudioHParams()
    rs = np.random.RandomState(5)
    B, T = 2, 24
    spec = rs.rand(B, T, 1025)
    u = rs.rand(B, T, 1025)
    gl = taco_amd.GriffinLim(_HP(ahp))
    wav = gl.inv_spectrogram(spec, init_uniform=u, iters=1).cpu().numpy()
    assert wav.shape == (B, 300 * (T - 1))
    for b in range(B):
        ref = A.inv_spectrogram(spec[b].T, ahp, u[b].T, iters=1)
        assert _cmp(wav[b], ref) < 5e-4
    gl.close()


def test_full_iterations_reach_the_same_spectral_consistency():
    """60 iterations amplify rounding differences sample by sample; what must agree is the quality of the result:
    the magnitude of the STFT of the produced waveform vs the requested magnitudes (spectral convergence)."""
    import torch, taco_amd
    ahp = A.AudioHParams(num_freq=129, sample_rate=3200, frame_length_ms=50, frame_shift_ms=12.5, griffin_lim_iters=60)  # n_fft 256, hop 40, win 160
    rs = np.random.RandomState(9)
    T = 60
    t = np.arange(40 * (T - 1)) / 3200.0
    y0 = np.sin(2 * np.pi * 220 * t) + 0.5 * np.sin(2 * np.pi * 555 * t * (1 + 0.2 * t))
    mag = np.abs(A.stft(y0, ahp))                                            # [F, T] a consistent spectrogram
    # encode it the way the model's output would be: normalised dB of mag^(1/power) ...
    S = mag ** (1 / ahp.power)
    spec = np.clip((20 * np.log10(np.maximum(1e-5, S)) - ahp.ref_level_db - ahp.min_level_db) / -ahp.min_level_db, 0, 1)
    target = A.db_to_amp(A.denormalize(spec, ahp) + ahp.ref_level_db) ** ahp.power
    u = rs.rand(129, T)
    gl = taco_amd.GriffinLim(_HP(ahp))
    wav = gl.inv_spectrogram(spec.T[None], init_uniform=u.T[None]).cpu().numpy()[0]
    ref = A.inv_spectrogram(spec, ahp, u)
    def sc(w):                                                               # undo the final inverse pre-emphasis first
        y = np.append(w[0], w[1:] - ahp.preemphasis * w[:-1])
        return np.linalg.norm(np.abs(A.stft(y, ahp)) - target) / np.linalg.norm(target)
    assert sc(ref) < 0.25
    assert abs(sc(wav) - sc(ref)) < 0.03, (sc(wav), sc(ref))
    gl.close()


def test_hash_initial_phases_and_errors():
    import torch, taco_amd
    ahp = A.AudioHParams(num_freq=65, sample_rate=1600)
    gl = taco_amd.GriffinLim(_HP(ahp))
    spec = np.random.RandomState(1).rand(2, 30, 65)
    a = gl.inv_spectrogram(spec, seed=7, iters=2).cpu().numpy()
    b = gl.inv_spectrogram(spec, seed=7, iters=2).cpu().numpy()
    c = gl.inv_spectrogram(spec, seed=8, iters=2).cpu().numpy()
    assert np.array_equal(a, b) and not np.array_equal(a, c) and np.isfinite(a).all()
    with pytest.raises(taco_amd._lib.TacoError):
        gl.inv_spectrogram(spec[:, :3], iters=1)               # too short for reflect padding
    with pytest.raises(Exception):
        gl.inv_spectrogram(spec[:, :, :10])
    gl.close()


def test_synthesizer_vocode_surface(tmp_path):
    import taco_amd
    import taco_oracle as O
    from util import tiny_hp, to_product_hp
    ohp = tiny_hp(num_freq=65, max_iters=20)
    hp = to_product_hp(ohp)
    hp.add_hparam("sample_rate", 1600); hp.add_hparam("griffin_lim_iters", 3)
    w = O.init_weights(ohp, 1, 31)
    taco_amd.save_hparams(str(tmp_path), hp)
    taco_amd.weights.save_weights(str(tmp_path / "model.ckpt-1.safetensors"), w)
    ids, L = O.synthetic_inputs(2, 9, 41)
    s = taco_amd.Synthesizer().load(str(tmp_path), num_speakers=1)
    lin, al = s.synthesize(tokens=ids, vocode=True)
    assert len(s.wavs) == 2 and all(w.ndim == 1 and np.isfinite(w).all() and len(w) > 0 for w in s.wavs)
    assert len(s.wavs[0]) == 20 * max(int(s.spec_end_idx[0]) - 1, 1)
    s.close()

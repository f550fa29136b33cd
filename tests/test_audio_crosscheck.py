"""The librosa calls of audio/__init__.py:87-96 (stft / istft with n_fft, hop_length, win_length and librosa's defaults) cannot run here; the
oracle restates them (oracle/audio_oracle.py).  Two independent implementations of the same documented convention are in this image and
are held against the restatement here: torch.stft / torch.istft (whose centre padding, window padding and sum-square normalisation are
documented to follow librosa's) and scipy.signal.stft.  Not a pin on librosa itself -- a pin on the convention by two other readers of it."""
import numpy as np
import pytest
import torch

import audio_oracle as A


def _signal(n, seed):
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 24000.0
    return 0.3 * np.sin(2 * np.pi * 220.0 * t) + 0.1 * rs.normal(size=n)


@pytest.mark.parametrize("num_freq,n", [(1025, 24000), (1025, 7777), (257, 5000)])
def test_stft_equals_torch_and_scipy(num_freq, n):
    hp = A.AudioHParams(num_freq=num_freq, frame_length_ms=50 if num_freq == 1025 else 20, frame_shift_ms=12.5 if num_freq == 1025 else 5)
    n_fft, hop, win = hp.stft_parameters()
    y = _signal(n, 1)
    S = A.stft(y, hp)
    St = torch.stft(torch.from_numpy(y), n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win, periodic=True, dtype=torch.float64),
                    center=True, pad_mode="reflect", return_complex=True).numpy()
    assert S.shape == St.shape == (num_freq, 1 + n // hop)
    assert np.abs(S - St).max() < 1e-9 * max(1.0, np.abs(S).max())
    import scipy.signal as ss
    w = A.padded_window(n_fft, win)
    _, _, Ss = ss.stft(y, window=w, nperseg=n_fft, noverlap=n_fft - hop, boundary="even", padded=False, return_onesided=True)
    Ss = Ss * w.sum()                                        # scipy scales by the window's sum
    k = min(S.shape[1], Ss.shape[1])
    assert k >= S.shape[1] - 1
    assert np.abs(S[:, :k] - Ss[:, :k]).max() < 1e-9 * max(1.0, np.abs(S).max())


@pytest.mark.parametrize("seed", [0, 1])
def test_istft_equals_torch_on_inconsistent_spectrograms(seed):
    """Griffin-Lim inverts spectrograms that are NOT the STFT of any signal: the overlap-add and the sum-square division must agree there too"""
    hp = A.AudioHParams()
    n_fft, hop, win = hp.stft_parameters()
    rs = np.random.RandomState(seed)
    T = 40
    S = rs.normal(size=(hp.num_freq, T)) + 1j * rs.normal(size=(hp.num_freq, T))
    y = A.istft(S, hp)
    S_t = S.copy()
    S_t[0].imag = 0                                          # irfft ignores the imaginary part of DC / Nyquist; torch (C2R) documents the same
    S_t[-1].imag = 0
    yt = torch.istft(torch.from_numpy(S_t), n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win, periodic=True, dtype=torch.float64),
                     center=True).numpy()
    assert y.shape == yt.shape == (hop * (T - 1),)
    assert np.abs(y - yt).max() < 1e-10 * max(1.0, np.abs(y).max())
    y0 = A.istft(S_t, hp)
    assert np.abs(y - y0).max() < 1e-12


def test_round_trip_and_one_griffin_lim_iteration_against_torch():
    hp = A.AudioHParams()
    n_fft, hop, win = hp.stft_parameters()
    y = _signal(hop * 30, 5)
    assert np.abs(A.istft(A.stft(y, hp), hp) - y).max() < 1e-10
    win_t = torch.hann_window(win, periodic=True, dtype=torch.float64)
    mag = np.abs(A.stft(y, hp))
    u = np.random.RandomState(2).rand(*mag.shape)
    got = A.griffin_lim(mag, hp, u, iters=2)
    ang = torch.from_numpy(np.exp(2j * np.pi * u))
    M = torch.from_numpy(mag).to(torch.complex128)
    yt = torch.istft(M * ang, n_fft, hop_length=hop, win_length=win, window=win_t, center=True)
    for _ in range(2):
        St = torch.stft(yt, n_fft, hop_length=hop, win_length=win, window=win_t, center=True, pad_mode="reflect", return_complex=True)
        yt = torch.istft(M * torch.exp(1j * torch.angle(St)), n_fft, hop_length=hop, win_length=win, window=win_t, center=True)
    assert np.abs(got - yt.numpy()).max() < 1e-8 * max(1.0, np.abs(got).max())

"""CPU restatement of the reference's spectrogram -> waveform step (audio/__init__.py:54-56,76-96,118-122,149-165), test
infrastructure only (same rules as taco_oracle.py: nothing in the product imports it).

The reference delegates the transforms to librosa (`librosa.stft` / `librosa.istft`, not installed here).  They are restated
from librosa's documented algorithm (0.5/0.6 era, the versions contemporary with the reference):
  stft : periodic Hann window of win_length, zero-padded (centred) to n_fft; signal reflect-padded by n_fft/2 (center=True);
         frame t = y_pad[t*hop : t*hop + n_fft] * window -> rfft
  istft: irfft of every frame * the same window, overlap-added at t*hop, divided by the window sum-square where it exceeds
         tiny, then n_fft/2 trimmed from both ends.
PARITY: the librosa-free pieces (stft_parameters, denormalize, db_to_amp, the power law, inv_preemphasis) are pinned bit for bit on the
reference's own functions (tools/make_reference_vectors.py -> tests/golden/audio_vectors.npz, tests/test_reference_vectors.py).  The
STFT / ISTFT are UNPINNED on librosa itself (it cannot run here); they are held against two other implementations of the same documented
convention -- torch.stft / torch.istft and scipy.signal.stft (tests/test_audio_crosscheck.py) -- and by round trip / Parseval checks."""
import numpy as np


class AudioHParams:
    def __init__(self, num_freq=1025, sample_rate=24000, frame_length_ms=50, frame_shift_ms=12.5, preemphasis=0.97,
                 min_level_db=-100, ref_level_db=20, power=1.5, griffin_lim_iters=60):
        self.num_freq, self.sample_rate = num_freq, sample_rate
        self.frame_length_ms, self.frame_shift_ms = frame_length_ms, frame_shift_ms
        self.preemphasis, self.min_level_db, self.ref_level_db = preemphasis, min_level_db, ref_level_db
        self.power, self.griffin_lim_iters = power, griffin_lim_iters

    def stft_parameters(self):                      # audio/__init__.py:118-122
        n_fft = (self.num_freq - 1) * 2
        return n_fft, int(self.frame_shift_ms / 1000 * self.sample_rate), int(self.frame_length_ms / 1000 * self.sample_rate)


def padded_window(n_fft, win_length):
    n = np.arange(win_length)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / win_length)           # scipy get_window('hann', win_length, fftbins=True)
    lpad = (n_fft - win_length) // 2
    out = np.zeros(n_fft)
    out[lpad:lpad + win_length] = w
    return out


def stft(y, hp):
    n_fft, hop, win = hp.stft_parameters()
    w = padded_window(n_fft, win)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    out = np.empty((1 + n_fft // 2, n_frames), np.complex128)
    for t in range(n_frames):
        out[:, t] = np.fft.rfft(yp[t * hop:t * hop + n_fft] * w)
    return out


def window_sumsquare(n_frames, hp):
    n_fft, hop, win = hp.stft_parameters()
    w2 = padded_window(n_fft, win) ** 2
    x = np.zeros(n_fft + hop * (n_frames - 1))
    for t in range(n_frames):
        x[t * hop:t * hop + n_fft] += w2
    return x


def istft(S, hp):
    n_fft, hop, win = hp.stft_parameters()
    w = padded_window(n_fft, win)
    n_frames = S.shape[1]
    y = np.zeros(n_fft + hop * (n_frames - 1))
    for t in range(n_frames):
        y[t * hop:t * hop + n_fft] += w * np.fft.irfft(S[:, t], n_fft)
    wss = window_sumsquare(n_frames, hp)
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:-(n_fft // 2)]


def denormalize(S, hp):
    return np.clip(S, 0, 1) * -hp.min_level_db + hp.min_level_db


def db_to_amp(x):
    return np.power(10.0, x * 0.05)


def inv_preemphasis(x, hp):                          # scipy.signal.lfilter([1], [1, -k], x)
    y = np.empty_like(x)
    acc = 0.0
    for i, v in enumerate(x):
        acc = v + hp.preemphasis * acc
        y[i] = acc
    return y


def griffin_lim(S, hp, init_uniform, iters=None):
    """audio/__init__.py:76-85.  S [F, T] magnitudes; init_uniform [F, T] in [0,1) stands for np.random.rand(*S.shape)."""
    angles = np.exp(2j * np.pi * init_uniform)
    Sc = np.abs(S).astype(np.complex128)
    y = istft(Sc * angles, hp)
    for _ in range(hp.griffin_lim_iters if iters is None else iters):
        angles = np.exp(1j * np.angle(stft(y, hp)))
        y = istft(Sc * angles, hp)
    return y


def inv_spectrogram(spec_FT, hp, init_uniform, iters=None):
    """audio/__init__.py:54-56: spec_FT [num_freq, T] (the synthesizer passes linear_outputs.T, synthesizer.py:264)."""
    S = db_to_amp(denormalize(spec_FT, hp) + hp.ref_level_db)
    return inv_preemphasis(griffin_lim(S ** hp.power, hp, init_uniform, iters), hp)

"""Spectrogram -> waveform on the GPU: `inv_spectrogram` of the reference's audio/__init__.py:54-56 (denormalise, dB -> amplitude,
^power, Griffin-Lim with librosa-semantics STFT/ISTFT, inverse pre-emphasis), the step synthesizer.py:264 runs on the CPU for
every utterance.  All arithmetic is in libtaco_hip (taco_gl_*); PyTorch holds the buffers."""
import ctypes as C

import numpy as np
import torch

from . import _lib


class GriffinLim(object):
    def __init__(self, hparams, device="cuda:0"):
        g = lambda k, d: getattr(hparams, k, d)
        self.hp = _lib.TacoAudioHParams(
            num_freq=int(g("num_freq", 1025)), sample_rate=int(g("sample_rate", 24000)), griffin_lim_iters=int(g("griffin_lim_iters", 60)),
            frame_length_ms=float(g("frame_length_ms", 50)), frame_shift_ms=float(g("frame_shift_ms", 12.5)),
            preemphasis=float(g("preemphasis", 0.97)), min_level_db=float(g("min_level_db", -100)), ref_level_db=float(g("ref_level_db", 20)),
            power=float(g("power", 1.5)))
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TacoError(_lib.TACO_ERR_ARG, "GriffinLim runs on a GPU (got %s); there is no CPU fallback" % device)
        self._lib = _lib.load_library()
        self._h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self._lib.taco_gl_create(C.byref(self.hp), idx, C.byref(self._h)))
        self._ws = None

    def num_samples(self, T):
        return int(self._lib.taco_gl_num_samples(self._h, T))

    def inv_spectrogram(self, linear, init_uniform=None, seed=0, iters=None):
        """linear [B, T, num_freq] (model layout; numpy or tensor) -> waveforms [B, hop*(T-1)] (device tensor).
        init_uniform [B, T, num_freq] in [0,1) replaces the reference's np.random.rand initial phases (default: hash of seed)."""
        dev = self.device
        x = (linear if torch.is_tensor(linear) else torch.as_tensor(np.asarray(linear))).to(dev, torch.float32).contiguous()
        u = None if init_uniform is None else (init_uniform if torch.is_tensor(init_uniform) else torch.as_tensor(np.asarray(init_uniform))).to(dev, torch.float32).contiguous()
        B, T, F = x.shape
        if F != self.hp.num_freq:
            raise Exception("last dimension must be num_freq = %d, got %d" % (self.hp.num_freq, F))
        nb = int(self._lib.taco_gl_workspace_bytes(self._h, B, T))
        if self._ws is None or self._ws.numel() < nb:
            self._ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        wav = torch.empty((B, self.num_samples(T)), dtype=torch.float32, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        with torch.cuda.device(dev):
            _lib.check(self._lib.taco_gl_inv_spectrogram(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream), p(x), p(u),
                                                         C.c_ulonglong(int(seed)), B, T, -1 if iters is None else int(iters), p(wav),
                                                         p(self._ws), self._ws.numel()))
        return wav

    def close(self):
        if getattr(self, "_h", None):
            self._lib.taco_gl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

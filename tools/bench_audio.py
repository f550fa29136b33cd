#!/usr/bin/env python
"""Spectrogram -> waveform (Griffin-Lim, 60 iterations) at the C2 output shape: B=32 utterances x 512 frames x 1025 bins.
Prints one JSON line: audio seconds produced per second, with the NumPy oracle (FFT-based, one utterance) timed beside it."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, taco_amd
import audio_oracle as A
B, T, F = 32, 512, 1025
hp = taco_amd.hparams
gl = taco_amd.GriffinLim(hp)
rs = np.random.RandomState(0)
spec = torch.from_numpy(rs.rand(B, T, F).astype(np.float32)).cuda()
for _ in range(2):
    wav = gl.inv_spectrogram(spec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n):
    wav = gl.inv_spectrogram(spec)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
L = wav.shape[1]; audio_s = B * L / hp.sample_rate
gflop = 2.0 * 2 * (B * (T + 7)) * 1200 * 2050 * 61 / 1e9      # two windowed-DFT products per iteration (+1 synthesis)
ahp = A.AudioHParams()
t0 = time.perf_counter(); A.inv_spectrogram(spec[0].cpu().numpy().astype(np.float64).T, ahp, rs.rand(F, T)); cpu_s = time.perf_counter() - t0
print(json.dumps({"metric": "audio seconds synthesised per second (Griffin-Lim, 60 iterations)", "value": audio_s / (ms / 1e3), "unit": "x realtime",
                  "ms_per_batch": ms, "batch": "B=%d x T=%d frames x %d bins -> %d samples each (%.1f s of audio at %d Hz)" % (B, T, F, L, L / hp.sample_rate, hp.sample_rate),
                  "dft_gemm_TFLOPs_equiv": gflop / ms, "finite": bool(torch.isfinite(wav).all()),
                  "cpu_baseline": {"kind": "port", "sample": "oracle/audio_oracle.py (NumPy FFT, float64), 1 utterance", "seconds_per_utterance": cpu_s,
                                   "value": (L / hp.sample_rate) / cpu_s, "unit": "x realtime", "cores": 1}}))

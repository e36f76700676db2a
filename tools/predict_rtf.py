#!/usr/bin/env python3
"""Real-time factor of long-file inference (utils/predict_long.py:30-79): predict.predict_long over a long synthetic signal -- device-side
framing (a strided view of the padded signal), the HIP forward per batch of windows, the cropped concatenation -- timed end to end INCLUDING the host->device copy of
the signal and the device->host copy of the prediction.  seconds of 44.1 kHz audio processed per second of wall time.
    python tools/predict_rtf.py [minutes of audio, default 10]      (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from signaltrain_amd import nn_proc, predict
nn_proc._QUIET = True
MIN = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
SR = 44100
rng = np.random.default_rng(0)
n = int(MIN * 60 * SR)
sig = (0.3 * np.sin(2 * np.pi * 220.0 * np.arange(n) / SR) * (0.5 + 0.5 * np.sin(2 * np.pi * 0.5 * np.arange(n) / SR)) + 0.01 * rng.standard_normal(n)).astype(np.float32)
for scale in (1, 8):
    for dt in ("f32", "bf16_all"):
        torch.manual_seed(0)
        m = nn_proc.st_model(scale_factor=scale, shrink_factor=4, num_knobs=4).to("cuda:0")
        m.set_compute_dtype(dt)
        for bs in ((200, 1024) if scale == 1 else (64, 200)):       # 200 = the reference's default (predict_long.py:30)
            kn = np.array([0.1, -0.2, 0.3, 0.0], np.float32)
            y = predict.predict_long(sig[: 20 * m.in_chunk_size], kn, m, m.in_chunk_size, m.out_chunk_size, sr=SR, batch_size=bs)      # warm-up (workspace, first launches)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            y = predict.predict_long(sig, kn, m, m.in_chunk_size, m.out_chunk_size, sr=SR, batch_size=bs)
            torch.cuda.synchronize(); dtm = time.perf_counter() - t0
            assert np.isfinite(y).all() and len(y) == n - (m.in_chunk_size - m.out_chunk_size)
            nwin = (n - m.in_chunk_size) // m.out_chunk_size + 1
            print(f"window {m.in_chunk_size:6d} -> {m.out_chunk_size:5d}  {dt:9s} batch {bs:5d}: {MIN:.0f} min of audio ({nwin} windows) in {dtm * 1e3:8.1f} ms"
                  f" = {MIN * 60 / dtm:9.0f} x real time  ({nwin / dtm / 1e3:7.1f} k windows/s)")

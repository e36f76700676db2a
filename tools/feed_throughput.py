#!/usr/bin/env python3
"""GPU: rate of the device-side comp_4c data feed (signals: audio_device.py, effect: st_compressor_4c), windows per second."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from signaltrain_amd import audio, datasets
for L, ysz, B in ((8192, 2048, 256), (8192, 2048, 2048), (65536, 16256, 64)):
    ds = datasets.SynthAudioDataSet(L, audio.Compressor_4c(), y_size=ysz)
    for _ in range(3): ds.batch_device(B)
    torch.cuda.synchronize(); t0 = time.time(); n = 20
    for _ in range(n): ds.batch_device(B)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print(f"L={L} B={B}: {dt*1e3:.2f} ms/batch = {B/dt:.0f} windows/s generated on the device")

#!/usr/bin/env python3
"""GPU: rate of the device-side comp_4c data feed (csrc/st_feed.h: st_synth_comp4c = generator kernel + lane-per-window compressor stage),
windows per second, and the two kernels' own times (library event profiling)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from signaltrain_amd import audio, datasets, _lib
lib = _lib.load()
for L, ysz, B in ((8192, 2048, 256), (8192, 2048, 2048), (65536, 16256, 64), (65536, 16256, 2048)):
    ds = datasets.SynthAudioDataSet(L, audio.Compressor_4c(), y_size=ysz)
    for _ in range(3): ds.batch_device(B)
    torch.cuda.synchronize(); t0 = time.time(); n = 20
    for _ in range(n): ds.batch_device(B)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    lib.st_profile_enable(1)
    for _ in range(5): ds.batch_device(B)
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 14); lib.st_profile_report(buf, len(buf)); lib.st_profile_enable(0)
    ks = "  ".join(f"{l.split()[0]} {float(l.split()[1]) / int(l.split()[2]) * 1e3:.0f} us" for l in buf.value.decode().strip().splitlines())
    print(f"L={L} B={B}: {dt*1e3:.3f} ms/batch = {B/dt:.0f} windows/s generated on the device   [{ks}]")

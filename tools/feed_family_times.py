#!/usr/bin/env python3
"""Generator time per signal family (st_synth_comp4c with a fixed chooser), B windows of L samples: where the feed's GPU time goes.
    python tools/feed_family_times.py [L] [B]      (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from signaltrain_amd import datasets, audio, _lib
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
lib = _lib.load()
ds = datasets.SynthAudioDataSet(L, audio.Compressor_4c(), datapoints=B, y_size=L // 4)
for ch in (-1, 0, 1, 2, 4, 6, 7):
    for _ in range(2): ds.batch_device(B, "cuda:0", chooser=ch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ds.batch_device(B, "cuda:0", chooser=ch)
    e1.record(); torch.cuda.synchronize()
    print(f"L={L} B={B} chooser {ch:3d}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per call (all launches of the call, back to back: the lane-per-window smoothing stage is a constant ~290 us at L = 8192 / ~2.3 ms at 65536 of it)")

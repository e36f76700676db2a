#!/usr/bin/env python3
"""End-to-end training-loop throughput of train.train() at the 65536-sample window (BASELINE configs[4] geometry: scale_factor 8, B = 64, f16_all) with the on-the-fly device feed.
    python tools/train_throughput_s8.py [dtype] [device|recycle] [steps per epoch, default 600]      (GPU box; recycle = index gathers from a device-resident set: the loop
    without the generator; the reference's default epoch is 200 000 windows = 3125 steps of 64, over which the per-epoch work -- validation, status line -- amortises)"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from signaltrain_amd import train, audio, nn_proc
nn_proc._QUIET = True
os.chdir(tempfile.mkdtemp())
DT = sys.argv[1] if len(sys.argv) > 1 else "f16_all"
FEED = sys.argv[2] if len(sys.argv) > 2 else "device"
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 600
torch.manual_seed(0); np.random.seed(0)
t0 = time.time()
train.train(effect=audio.Compressor_4c(), epochs=3, n_data_points=64 * STEPS, batch_size=64, device=torch.device("cuda:0"), scale_factor=8,
            num_workers=2, device_feed=(True if FEED == "device" else "recycle"), compute_dtype=DT)
print(f"==> scale 8, dtype={DT}, feed={FEED}: total wall {time.time() - t0:.1f} s for {3 * 64 * STEPS} training windows ({STEPS} steps per epoch) (see the loop's own windows/s line above)")

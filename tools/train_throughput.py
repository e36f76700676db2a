#!/usr/bin/env python3
"""End-to-end training-loop throughput of train.train() on one GPU: the device feed (every minibatch generated on the GPU),
the device-resident recycled dataset (datasets.DeviceRecycledDataSet) and the reference-style CPU DataLoader feed.  Prints windows/s of the whole loop (data feed + step + the driver's bookkeeping)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from signaltrain_amd import train, audio, nn_proc
nn_proc._QUIET = True
os.chdir(tempfile.mkdtemp())
B = 256
DT = sys.argv[1] if len(sys.argv) > 1 else "f32"
for feed, npts, workers in (("device", 256 * 1500, 2), ("recycle", 256 * 1500, 2)):
    torch.manual_seed(0); np.random.seed(0)
    t0 = time.time()
    train.train(effect=audio.Compressor_4c(), epochs=3, n_data_points=npts, batch_size=B, device=torch.device("cuda:0"),
                num_workers=workers, device_feed={"device": True, "recycle": "recycle"}[feed], compute_dtype=DT)
    print(f"==> dtype={DT} feed={feed}: total wall {time.time() - t0:.1f} s for {3 * npts} training windows (see the loop's own windows/s line above)")
    for f in ("modelcheckpoint.tar", "vl_avg_out.dat", "val_err_mae.dat"):
        if os.path.exists(f): os.remove(f)

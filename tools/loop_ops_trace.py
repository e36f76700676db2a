#!/usr/bin/env python3
"""Which torch operators does train.train() issue per step?  Counts every aten op dispatched during a short run by the innermost frame of this
package that caused it (the step itself is ONE C call: whatever shows up here with a count near the number of steps is host-side overhead).
    python tools/loop_ops_trace.py [dtype] [scale_factor] [batch] [steps per epoch]      (GPU box)"""
import os, sys, tempfile, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
from signaltrain_amd import train, audio, nn_proc
nn_proc._QUIET = True
os.chdir(tempfile.mkdtemp())
DT = sys.argv[1] if len(sys.argv) > 1 else "bf16_all"
SF = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 200
counts = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        where = "?"
        for fr in reversed(traceback.extract_stack()[:-1]):
            if "signaltrain_amd" in fr.filename:
                where = f"{os.path.basename(fr.filename)}:{fr.lineno}"; break
        counts[(str(func), where)] += 1
        return func(*args, **(kwargs or {}))


torch.manual_seed(0); np.random.seed(0)
with Log():
    train.train(effect=audio.Compressor_4c(), epochs=1, n_data_points=B * STEPS, batch_size=B, device=torch.device("cuda:0"), scale_factor=SF,
                num_workers=2, device_feed=True, compute_dtype=DT)
print(f"\n==> {STEPS} training steps (+ {STEPS // 4} validation batches); aten ops by call site:")
for (op, where), n in counts.most_common(40):
    print(f"{n:7d}  {op:45s} {where}")

#!/usr/bin/env python3
"""Per-stage cycle breakdown of ae_bwd_kernel for one wave (s_memtime ticks at 100 MHz; diagnostics only)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tests import gpu_checks as G
from signaltrain_amd import _lib
from signaltrain_amd.engine import StepEngine
lib = _lib.load()
B = 256
geo, X, Y, KN, P = G.make_case(8, 3)
import numpy as np
X = np.tile(X, (32, 1)); Y = np.tile(Y, (32, 1)); KN = np.tile(KN, (32, 1))
d = G.dims_of(geo, B, 4)
eng = StepEngine(d, G.DEV); eng.load_state_dict(P)
x, kn, y = G.t(X), G.t(KN), G.t(Y)
for _ in range(3): eng.loss_backward(x, kn, y)
torch.cuda.synchronize()
buf = (C.c_uint64 * 32)()
lib.st_debug_read_stage_cycles(buf)
lib.st_set_debug(256)
N = 5
for _ in range(N): eng.loss_backward(x, kn, y)
torch.cuda.synchronize()
lib.st_set_debug(0)
lib.st_debug_read_stage_cycles(buf)
names = ["loads issue", "fwd L1", "fwd L2", "fwd L3-5", "fwd L6-7", "fwd L8", "fwd L9", "d-out", "bwd 9", "bwd 8", "bwd 7", "bwd 6-4", "bwd 3", "bwd 2", "bwd 1", "store dv", "latch", "takeover"]
tot = sum(buf[:18])
groups = 17 * N     # wave 0 of block 0 processes ceil(8448/512) groups per launch
print("s_memtime ticks (constant 100 MHz clock => 1 tick = 10 ns = ~24 shader cycles); per group:")
for i, nme in enumerate(names):
    print(f"  {nme:12s} {buf[i]/groups:9.1f} ticks  {100.0*buf[i]/max(tot,1):5.1f} %")
print(f"  total        {tot/groups:9.1f} ticks/group = {tot/groups*10:.0f} ns/group")

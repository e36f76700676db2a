#!/usr/bin/env python3
"""Diagnostics (GPU): the two-kernel autoencoder backward (st_ae_split.h, st_set_tuning(8001)) under the 16-bit '_all' precisions against the
single kernel (8000): fused-step parity against the rounding oracle for both, and the largest gradient difference between the two forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import gpu_checks as G
from signaltrain_amd import _lib
lib = _lib.load()
for half in ("bf16", "f16"):
    grads = {}
    for tune in (8000, 8001):
        _lib.check(lib.st_set_tuning(tune), "tune")
        tol = (G.mixed_mode.FUSED_TOL if half == "bf16" else G.mixed_mode.FUSED_TOL_F16)[2]
        with G.mixed_mode(2, half=half, tol_scale=tol):
            res = G.run_fused(B=5, seed=3, K=4, steps=2)
            geo, X, Y, KN, P = G.make_case(5, 3, K=4)
            eng = G.new_engine(G.dims_of(geo, 5, 4)); eng.load_state_dict(P)
            eng.loss_backward(G.t(X), G.t(KN), G.t(Y))
            grads[tune] = eng.grads.clone()
        worst = max(res, key=lambda r: r["err"] / max(r["tol"] * r["scale"], 1e-30))
        print(half, "tune", tune, "failed:", sum(not r["ok"] for r in res), "of", len(res), " worst", worst["name"], f"{worst['err'] / max(worst['tol'] * worst['scale'], 1e-30):.3f} of tolerance")
    a, b = grads[8000], grads[8001]
    print(half, "single vs split: max |diff| / max |grad| =", float((a - b).abs().max() / a.abs().max()), " nan:", bool(torch.isnan(b).any()))
_lib.check(lib.st_set_tuning(8002), "tune")

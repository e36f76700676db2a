#!/usr/bin/env python3
"""Diagnostics: a fused-step parity case under the 16-bit operand pipeline (st_gemm16.h) and under gemm_half_kernel (st_set_tuning(9600)):
prints err / (tol * scale) of the worst tensors for both, and whether the two paths agree bit for bit on the gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import gpu_checks as G
from signaltrain_amd import _lib

def run(tune, **kw):
    _lib.check(_lib.load().st_set_tuning(tune), "tune")
    with G.mixed_mode(2, half="f16", tol_scale=G.mixed_mode.FUSED_TOL_F16[2]):
        res = G.run_fused(**kw)
    return res

kw = dict(B=2, seed=5, K=4, steps=2, scale=8, scheme="legacy")
for tune in (9601, 9600):
    res = run(tune, **kw)
    worst = sorted(res, key=lambda r: -(r["err"] / max(r["tol"] * r["scale"], 1e-30)))[:8]
    print("tune", tune, "failed:", sum(not r["ok"] for r in res), "of", len(res))
    for r in worst:
        print(f"   {r['name']:55s} err/tol*scale = {r['err'] / max(r['tol'] * r['scale'], 1e-30):.3f}  err {r['err']:.3e} scale {r['scale']:.3e}")
_lib.load().st_set_tuning(9601)

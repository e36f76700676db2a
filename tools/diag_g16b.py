#!/usr/bin/env python3
"""Diagnostics: the 16-bit operand pipeline (st_gemm16.h) against gemm_half_kernel on fp32 operands, tensor by tensor (same inputs, same rounding
points: differences can only come from the fp32 accumulation order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import gpu_checks as G
from signaltrain_amd import _lib
from signaltrain_amd.engine import StepEngine

def run(tune, dtype, scale, scheme, B):
    _lib.check(_lib.load().st_set_tuning(tune), "tune")
    geo, X, Y, KN, P = G.make_case(B, 5, K=4, scale=scale, scheme=scheme)
    d = G.dims_of(geo, B, 4)
    e = StepEngine(d, G.DEV, compute_dtype=dtype); e.load_state_dict(P)
    outs = e.loss_backward(G.t(X), G.t(KN), G.t(Y), want_outputs=True)
    torch.cuda.synchronize()
    return [o.clone() for o in outs], e.layout.views(e.grads.clone()), e.scalars.clone()

for dtype, scale, scheme, B in (("f16_all", 8, "legacy", 2), ("bf16_all", 1, "lean", 3), ("f16", 1, "lean", 3)):
    o1, g1, s1 = run(9601, dtype, scale, scheme, B)
    o0, g0, s0 = run(9600, dtype, scale, scheme, B)
    print(f"== {dtype} scale {scale} {scheme} B={B}: scalars new {s1[:5].tolist()} old {s0[:5].tolist()}")
    for name, a, b in (("y_hat", o1[0], o0[0]), ("mag", o1[1], o0[1]), ("mag_hat", o1[2], o0[2])):
        print(f"   {name:50s} max|new-old| / max|old| = {float((a - b).abs().max() / b.abs().max()):.3e}")
    for k in g1:
        den = float(g0[k].abs().max())
        if den > 0:
            r = float((g1[k] - g0[k]).abs().max()) / den
            if r > 0: print(f"   grad {k:45s} {r:.3e}")
_lib.load().st_set_tuning(9601)

#!/usr/bin/env python3
"""Error of the fp32 MFMA path and of the three-plane bfloat16 split (compute_dtype "f32x3") against the float64 oracle:
forward outputs and gradients of one batch through the fused entry points.  Run on the GPU box (the oracle is the checker)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tests import gpu_checks as G
from oracle import st_oracle as O
from signaltrain_amd.engine import StepEngine
B, K = 6, 4
geo, X, Y, KN, P = G.make_case(B, 21, K=K)
P64 = {k: v.astype(np.float64) for k, v in P.items()}
X64, KN64, Y64 = X.astype(np.float64), KN.astype(np.float64), Y.astype(np.float64)
loss, Gr, c = O.model_loss_bwd(X64, KN64, Y64, P64, geo)
y_ref, mag_ref, _ = O.model_fwd(X64, KN64, P64, geo)
rel = lambda a, r: float(np.abs(a - r).max() / max(np.abs(r).max(), 1e-300))
for mode in ("f32", "f32x3", "bf16"):
    eng = StepEngine(G.dims_of(geo, B, K), G.DEV, compute_dtype=mode); eng.load_state_dict(P)
    y_hat, mag, mag_hat = eng.forward(G.t(X), G.t(KN))
    eng.loss_backward(G.t(X), G.t(KN), G.t(Y)); torch.cuda.synchronize()
    g = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in eng.layout.views(eng.grads).items()}
    eg = {k: rel(g[k], np.asarray(Gr[k], np.float64).reshape(g[k].shape)) for k in g}
    stft = [k for k in eg if "dft" in k]
    print(f"{mode:6s} max rel. error vs float64:  y_hat {rel(y_hat.cpu().numpy().astype(np.float64), y_ref):.2e}   |STFT| {rel(mag.cpu().numpy().astype(np.float64), mag_ref):.2e}"
          f"   STFT-basis gradients {max(eg[k] for k in stft):.2e}   autoencoder gradients {max(eg[k] for k in eg if k not in stft):.2e}")

#!/usr/bin/env python3
"""Round-3 golden vector from the *imported reference* (build container only); complements capture_golden.py / capture_golden_r2.py.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_r3.py
Writes (small, committed) and asserts oracle/st_oracle.py against every value:

  tests/golden/g8b_scale8_backward.npz   BASELINE configs[4] geometry (65536-sample window, lean scheme: T = 174, OT = 46, y = 16256), B = 1:
                                          loss, every autoencoder gradient, and the four STFT gradients fingerprinted (L1 norm, sampled rows,
                                          random projections) from the reference's autograd -- the wide-autoencoder path's backward was pinned
                                          by the reference at scale 1 only (G4, G4b); the oracle's backward is the same code at another T.
Parameters and inputs are those of G8 (tools/capture_golden.py): the reference's seed-218 initialisation of the autoencoders (stored in
g8_scale8.npz), regenerated + perturbed STFT bases, one comp_4c window from numpy Generator 8 -- with the target from the same draw.
"""
import os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _ref_import import import_reference                       # noqa: E402
from oracle import st_oracle as O                                # noqa: E402
from tests.golden_util import perturb_stft, projections, ae_keys, SAMPLE_ROWS  # noqa: E402

R = import_reference()
nn_proc, loss_functions = R.nn_proc, R.loss_functions
torch.set_num_threads(8)


def report(name, a, b, tol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.max(np.abs(a - b)) if a.size else 0.0
    scale = max(np.max(np.abs(b)), 1e-30) if b.size else 1.0
    print(f"  {name:40s} max|d|={err:.3e} rel={err/scale:.3e}")
    assert err <= tol * max(scale, 1e-30) + 1e-30, f"oracle mismatch on {name}: {err} vs scale {scale}"


g8 = np.load(os.path.join(OUT, "g8_scale8.npz"))
geo = O.geometry(8, 4)
P = O.init_params(geo, 4)
for k in ae_keys():
    P[k] = g8["ae_" + k]
perturb_stft(P, seed=9)
model = nn_proc.st_model(scale_factor=8, shrink_factor=4, num_knobs=4)
with torch.no_grad():
    model.load_state_dict({k: torch.from_numpy(P[k]) for k in P})
model.train()
rng = np.random.default_rng(8)
X, Y, KN = O.synth_comp4c_batch(1, geo["L"], geo["y"], rng)
assert np.array_equal(X, g8["x"]) and np.array_equal(KN, g8["knobs"]), "inputs of G8 not reproduced"
Y = (Y * np.float32(1.3)).astype(np.float32)                     # a gain error, so that the gradients are not noise-level
F = geo["F"]
print("G8b scale-8 backward")
y_hat, mag, mag_hat = model.forward(torch.from_numpy(X), torch.from_numpy(KN))
sbf = torch.exp((7. / F) * torch.arange(0., F)).expand_as(mag_hat).float()
loss = loss_functions.calc_loss(y_hat.float(), torch.from_numpy(Y).float(), mag_hat.float(), scale_by_freq=sbf)
model.zero_grad(); loss.backward()
gref = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}
ol, og, _ = O.model_loss_bwd(X.astype(np.float64), KN.astype(np.float64), Y.astype(np.float64), P, geo)
report("loss", ol, loss.item(), 3e-5)
for k in gref:
    report("grad " + k.replace("mpaec.", "")[:34], og[k], gref[k], 3e-4)
PROJ = projections(seed=13)
out = dict(y=Y, loss=np.float64(loss.item()))
for k in ae_keys():
    out["g_" + k] = gref[k]
for k in O.STFT_KEYS:
    g = gref[k][:, 0, :]
    out["l1_" + k] = np.float64(np.abs(g.astype(np.float64)).sum())
    out["rows_" + k] = g[SAMPLE_ROWS]
    out["proj_" + k] = PROJ @ g.astype(np.float64)
np.savez_compressed(os.path.join(OUT, "g8b_scale8_backward.npz"), **out)
print(f"g8b_scale8_backward.npz {os.path.getsize(os.path.join(OUT, 'g8b_scale8_backward.npz')) / 1024:.1f} KiB")
print("golden capture (round 3) OK")

#!/usr/bin/env python3
"""Round-6 golden vectors from the *imported reference* (build container only); complements capture_golden{,_r2,_r3,_r4,_r5}.py.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_r6.py
Writes (small, committed):

  tests/golden/g14_edges.npz   three edge cases of the model surface through the REFERENCE's forward, loss (called as train.py:115-121 calls it) and fp32 autograd:
        c0  a model WITHOUT knobs (num_knobs = 0: nn_proc.py:92-93 concatenates an empty [B, 0] tensor, fnn_addknobs is Linear(16, 16)),
        c1  ONE knob,
        c2  DIGITAL SILENCE: window 0 all zeros (input and target), window 1 half a window of zeros then signal, window 2 plain (K = 4).  At an exactly
            silent bin re = im = 0: the reference's autograd takes d |.| = 0 (torch.norm's sub-gradient) and d atan2(im, re + 1e-7) / d im = 1e7 (nn_proc.py:309-310),
            and a silent FRAME multiplies that by a frame of zeros in the analysis weight gradient.
        Per case: loss, y_hat, the maxima of mag / mag_hat, all 36 autoencoder gradients (full), the four STFT gradients fingerprinted (L1 norm, maximum, sampled
        rows, random projections) and -- as in G13 -- the measured distance of the reference's fp32 analysis-basis gradients from the float64 oracle
        (`ref_vs_f64`: the honest bound for those two tensors).  Inputs and parameters are NOT stored: tests regenerate them with tests.gpu_checks.make_case
        (numpy, portable generators) and apply the same edits.
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
from tests.golden_util import projections, SAMPLE_ROWS, g14_case  # noqa: E402

R = import_reference()
nn_proc, loss_functions = R.nn_proc, R.loss_functions
torch.set_num_threads(8)
PROJ = projections(seed=23)
out = {}
for ci in range(3):
    geo, X, Y, KN, P, K = g14_case(ci)
    model = nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=K)
    with torch.no_grad():
        model.load_state_dict({k: torch.from_numpy(P[k]) for k in P})
    model.train()
    F = geo["F"]
    y_hat, mag, mag_hat = model.forward(torch.from_numpy(X), torch.from_numpy(KN))
    sbf = torch.exp((7. / F) * torch.arange(0., F)).expand_as(mag_hat).float()
    loss = loss_functions.calc_loss(y_hat.float(), torch.from_numpy(Y).float(), mag_hat.float(), scale_by_freq=sbf)
    model.zero_grad(); loss.backward()
    gref = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}
    assert all(np.isfinite(v).all() for v in gref.values()), "the reference's own gradients are not finite"
    l64, g64, c64 = O.model_loss_bwd(X.astype(np.float64), KN.astype(np.float64), Y.astype(np.float64), {k: v.astype(np.float64) for k, v in P.items()}, geo)
    pre = f"c{ci}_"
    out[pre + "loss"] = np.float64(loss.item())
    out[pre + "y_hat"] = y_hat.detach().numpy().astype(np.float32)
    out[pre + "mag_max"] = np.float64(mag.detach().abs().max().item()); out[pre + "mag_hat_max"] = np.float64(mag_hat.detach().abs().max().item())
    e_y = float(np.abs(out[pre + "y_hat"] - c64["out"]).max() / np.abs(c64["out"]).max())
    print(f"G14 case {ci} (K = {K}): loss ref {loss.item():.6e} oracle64 {l64:.6e} (rel {abs(loss.item() - l64) / abs(l64):.1e}); y_hat {e_y:.1e}; "
          f"exactly-zero STFT bins: {int((c64['mag'] == 0).sum())} of {c64['mag'].size}")
    assert abs(loss.item() - l64) <= 3e-5 * abs(l64) and e_y <= 1e-5
    for k in gref:
        r = gref[k].astype(np.float64); sc = max(np.abs(g64[k]).max(), 1e-30)
        e64 = float(np.abs(r.reshape(g64[k].shape) - g64[k]).max() / sc)
        if k in O.STFT_KEYS:
            g = gref[k][:, 0, :]
            out[pre + "l1_" + k] = np.float64(np.abs(g.astype(np.float64)).sum()); out[pre + "max_" + k] = np.float64(np.abs(g).max())
            out[pre + "rows_" + k] = g[SAMPLE_ROWS]; out[pre + "proj_" + k] = PROJ @ g.astype(np.float64)
            out[pre + "ref_vs_f64_" + k] = np.float64(e64)
            print(f"  {k.replace('mpaec.', ''):52s} reference fp32 vs oracle f64 {e64:.2e}")
            if "analysis" not in k:
                assert e64 < 2e-5, (k, e64)
        else:
            assert e64 < 1e-4, (k, e64)                      # autoencoder gradients: fp32 level (north_star: 1e-4)
            out[pre + "g_" + k] = gref[k].astype(np.float32)
np.savez_compressed(os.path.join(OUT, "g14_edges.npz"), **out)
print(f"g14_edges.npz {os.path.getsize(os.path.join(OUT, 'g14_edges.npz')) / 1024:.1f} KiB")
print("golden capture (round 6) OK")

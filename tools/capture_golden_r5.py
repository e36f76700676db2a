#!/usr/bin/env python3
"""Round-5 golden vector from the *imported reference* (build container only); complements capture_golden{,_r2,_r3,_r4}.py.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_r5.py
Writes (small, committed):

  tests/golden/g13_near_silent_backward.npz   two windows-sets of the randomized sweep whose analysis-basis gradients missed the suite's 2e-4 on the device
        (profiles/r04_fuzz_parity.txt: fp32, B = 4 / seed 768 / K = 12 and B = 2 / seed 229 / K = 16; comp_4c windows with near-silent bins) through the REFERENCE's
        own fp32 autograd: loss, the 36 autoencoder gradients' maxima, and the four STFT gradients fingerprinted (L1 norm, sampled rows, random projections, the
        tensor maximum).  What it pins (VERDICT round 4, next #1 iii):
          * the oracle's float32 run reproduces the reference's fp32 gradients as closely as two fp32 evaluations can agree on that tensor, and
          * the reference's fp32 analysis-basis gradient ITSELF sits 2e-4 ... 1e-3 of the tensor maximum away from the float64 evaluation of the same formulas --
            the distance the device showed.  d atan2(im, re + 1e-7) = (-im, re) / (re^2 + im^2) (nn_proc.py:309-310 under autograd) amplifies the ~1e-7 |X|max
            rounding of re / im by 1 / mag at near-silent bins; the float64 oracle is the better VALUE, not the reference's arithmetic.
        The fixture therefore stores, per case, `ref_vs_f64` (measured here) -- tests/test_oracle_golden.py asserts the oracle's float32 run lands within 3 x of it
        and that the float64 oracle agrees with the reference to the same bound.
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
from tests.golden_util import projections, SAMPLE_ROWS           # noqa: E402
from tests import gpu_checks as G                                # noqa: E402  (make_case only: numpy)

R = import_reference()
nn_proc, loss_functions = R.nn_proc, R.loss_functions
torch.set_num_threads(8)
CASES = [dict(B=4, seed=768, K=12), dict(B=2, seed=229, K=16)]
PROJ = projections(seed=17)
out = {}
for ci, kw in enumerate(CASES):
    geo, X, Y, KN, P = G.make_case(kw["B"], kw["seed"], K=kw["K"])
    model = nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=kw["K"])
    with torch.no_grad():
        model.load_state_dict({k: torch.from_numpy(P[k]) for k in P})
    model.train()
    F = geo["F"]
    y_hat, mag, mag_hat = model.forward(torch.from_numpy(X), torch.from_numpy(KN))
    sbf = torch.exp((7. / F) * torch.arange(0., F)).expand_as(mag_hat).float()
    loss = loss_functions.calc_loss(y_hat.float(), torch.from_numpy(Y).float(), mag_hat.float(), scale_by_freq=sbf)
    model.zero_grad(); loss.backward()
    gref = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}
    l64, g64, c64 = O.model_loss_bwd(X.astype(np.float64), KN.astype(np.float64), Y.astype(np.float64), {k: v.astype(np.float64) for k, v in P.items()}, geo)
    l32, g32, _ = O.model_loss_bwd(X, KN, Y, P, geo)
    print(f"G13 case {ci} {kw}: loss ref {loss.item():.6e} oracle64 {l64:.6e}; smallest non-zero |STFT| / max = {np.min(c64['mag'][c64['mag'] > 0]) / c64['mag'].max():.1e}")
    pre = f"c{ci}_"
    worst_an = 0.0
    out[pre + "cfg"] = np.array([kw["B"], kw["seed"], kw["K"]]); out[pre + "loss"] = np.float64(loss.item())
    for k in gref:
        r = gref[k].astype(np.float64); sc = max(np.abs(g64[k]).max(), 1e-30)
        e64, e32 = np.abs(r - g64[k]).max() / sc, np.abs(r - g32[k].astype(np.float64)).max() / sc
        if k in O.STFT_KEYS:
            print(f"  {k.replace('mpaec.', ''):52s} reference fp32 vs oracle f64 {e64:.2e}   vs oracle f32 {e32:.2e}   (oracle f32 vs f64 {np.abs(g32[k] - g64[k]).max() / sc:.2e})")
            g = gref[k][:, 0, :]
            out[pre + "l1_" + k] = np.float64(np.abs(g.astype(np.float64)).sum()); out[pre + "max_" + k] = np.float64(np.abs(g).max())
            out[pre + "rows_" + k] = g[SAMPLE_ROWS]; out[pre + "proj_" + k] = PROJ @ g.astype(np.float64)
            out[pre + "ref_vs_f64_" + k] = np.float64(e64)
            # the conditioning claim itself: fp32 against fp32 is no closer than fp32 against float64 on the analysis bases, and the synthesis bases are at fp32 rounding level
            out[pre + "ref_vs_f32_" + k] = np.float64(e32)
            if "analysis" in k:
                worst_an = max(worst_an, e64)
            else:
                assert e64 < 2e-5, (k, e64)
        else:
            assert e64 < 1e-4, (k, e64)                      # autoencoder gradients: the oracle reproduces the reference at the fp32 level (north_star: 1e-4)
            out[pre + "max_" + k] = np.float64(np.abs(gref[k]).max())
    # the conditioning claim itself: the reference's OWN fp32 autograd is further than the suite's 2e-4 from the float64 value on an analysis basis (the synthesis
    # bases and the autoencoders sit at fp32 rounding level, asserted above)
    assert worst_an > 2e-4, worst_an
np.savez_compressed(os.path.join(OUT, "g13_near_silent_backward.npz"), **out)
print(f"g13_near_silent_backward.npz {os.path.getsize(os.path.join(OUT, 'g13_near_silent_backward.npz')) / 1024:.1f} KiB")
print("golden capture (round 5) OK")

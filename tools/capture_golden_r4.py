#!/usr/bin/env python3
"""Round-4 golden vector from the *imported reference* (build container only); complements capture_golden.py / _r2 / _r3.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_r4.py
Writes (small, committed) and asserts oracle/host_audio.py against every value, BIT FOR BIT:

  tests/golden/g11_host_signals.npz   the reference's synthetic test signals (signaltrain/audio.py:296-334 synth_input_sample) for the compressor's
                                       chooser set {0,1,2,4,6,7} (datasets.py:317) at fixed seeds of numpy's global generator, 2048-sample windows,
                                       and one SynthAudioDataSet.gen_single_chunk item (datasets.py:312-334: signal, Beta knobs, compressor target, augment)
                                       per seed.  oracle/host_audio.py is the only checker of the device feed's signal families (csrc/st_feed.h); this pins
                                       it to the reference draw for draw (chooser 7 drew amp_n after pluck()).
  tests/golden/g12_knob_grad.npz      d loss / d knobs by the reference's autograd (knobs.requires_grad_(): nn_proc.py:92-93 repeats the knob settings
                                       over the rows of a window and concatenates them in front of fnn_addknobs of BOTH autoencoders) for the G3 / G4
                                       inputs and weights, loss = calc_loss with the frequency weighting of train.py; plus a second upstream gradient
                                       (a plain sum of squares of all three model outputs).  Pins oracle.st_oracle's d_knobs (model_loss_bwd cache).
The numba-jitted compressor loop of the reference runs through the jit stub of tools/_ref_import.py (plain Python, same arithmetic).
"""
import os, sys, types
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _ref_import import import_reference, REF                    # noqa: E402
from oracle import host_audio as H                               # noqa: E402

R = import_reference()
ref_audio = R.audio
# the reference's datasets.py uses a package-relative import (from . import audio): give it a package to live in
pkg = types.ModuleType("signaltrain"); pkg.__path__ = [REF]; pkg.audio = ref_audio
sys.modules["signaltrain"] = pkg; sys.modules["signaltrain.audio"] = ref_audio
import importlib                                                 # noqa: E402
ref_datasets = importlib.import_module("signaltrain.datasets")

N = 2048
SR = 44100
t = np.arange(N, dtype=np.float32) / SR
CHOOSERS = (0, 1, 2, 4, 6, 7)
SEEDS = (11, 12)
out = {"n": N, "sr": SR, "choosers": np.array(CHOOSERS), "seeds": np.array(SEEDS)}
print("G11 host signal generators")
for c in CHOOSERS:
    for s in SEEDS:
        np.random.seed(s); ref = ref_audio.synth_input_sample(t, c)
        np.random.seed(s); mine = H.synth_input_sample(t, c)
        d = float(np.max(np.abs(ref - mine)))
        print(f"  chooser {c} seed {s}: max|d| = {d:.3e}")
        assert np.array_equal(ref, mine), f"host_audio.synth_input_sample differs from the reference for chooser {c}, seed {s}"
        out[f"sig_c{c}_s{s}"] = ref.astype(np.float64)

# whole items: chooser drawn inside, knobs, compressor target, polarity augmentation
from signaltrain_amd import audio as A                            # noqa: E402  (the mirror's Effect: its host go() is the gcc-built compressor)
ref_eff = ref_audio.Compressor_4c()
my_eff = A.Compressor_4c()
assert np.array_equal(np.asarray(ref_eff.knob_ranges, np.float64), np.asarray(my_eff.knob_ranges, np.float64))
ds = ref_datasets.SynthAudioDataSet(N, ref_eff, sr=SR, datapoints=4, y_size=N // 2, augment=True)
for s in (21, 22, 23):
    np.random.seed(s); rx, ry, rk = ds.gen_single_chunk()
    np.random.seed(s); mx, my, mk = H.gen_single_chunk(t, my_eff, N // 2, augment=True)
    assert np.array_equal(np.asarray(rk, np.float64), np.asarray(mk, np.float64)), "knob draws differ"
    assert np.array_equal(np.asarray(rx, np.float64), np.asarray(mx, np.float64)), "input signal differs"
    dy = float(np.max(np.abs(np.asarray(ry, np.float64) - np.asarray(my, np.float64))))
    print(f"  item seed {s}: knobs, x bit-identical; compressor target max|d| = {dy:.3e}")
    assert dy <= 2e-6, "compressor target differs"                # float32 recursion: C helper vs the reference's python loop, libm pow/log10
    out[f"item_x_s{s}"] = np.asarray(rx, np.float64); out[f"item_y_s{s}"] = np.asarray(ry, np.float64); out[f"item_k_s{s}"] = np.asarray(rk, np.float64)
out["item_seeds"] = np.array((21, 22, 23))
np.savez_compressed(os.path.join(OUT, "g11_host_signals.npz"), **out)

# ---- G12: gradient w.r.t. the knob settings -------------------------------------------------------------------------------------------
print("G12 knob gradient")
import torch                                                     # noqa: E402
from oracle import st_oracle as O                                # noqa: E402
from tests.test_oracle_golden import golden_params               # noqa: E402
geo = O.geometry(1, 4)
P = golden_params(OUT, geo)
g3 = np.load(os.path.join(OUT, "g3_forward.npz"))
X, KN, Y = g3["x"], g3["knobs"], g3["y"]
torch.manual_seed(218); np.random.seed(218)
model = R.nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=4)
with torch.no_grad():
    model.load_state_dict({k: torch.from_numpy(P[k]) for k in P})
model.train()
xt, yt = torch.from_numpy(X), torch.from_numpy(Y)
kt = torch.from_numpy(KN).clone().requires_grad_(True)
y_hat, mag, mag_hat = model.forward(xt, kt)
F_ = geo["F"]
sbf = torch.exp((7. / F_) * torch.arange(0., F_)).expand_as(mag_hat).float()
loss = R.loss_functions.calc_loss(y_hat.float(), yt.float(), mag_hat.float(), scale_by_freq=sbf)
loss.backward()
gk_ref = kt.grad.detach().numpy().astype(np.float64)
_, _, c64 = O.model_loss_bwd(X.astype(np.float64), KN.astype(np.float64), Y.astype(np.float64), P, geo)
_, _, c32 = O.model_loss_bwd(X, KN, Y, P, geo)
for nm, c in (("f64 oracle", c64), ("f32 oracle", c32)):
    err = float(np.max(np.abs(c["d_knobs"] - gk_ref))); sc = float(np.max(np.abs(gk_ref)))
    print(f"  d knobs ({nm}): max|d| = {err:.3e}  scale {sc:.3e}  rel {err / sc:.2e}")
    assert err <= (2e-4 if nm[1] == "3" else 2e-5) * sc, "oracle knob gradient differs from the reference's autograd"
np.savez_compressed(os.path.join(OUT, "g12_knob_grad.npz"), d_knobs=gk_ref, loss=np.float64(loss.item()))
print("golden capture r4 OK ->", os.path.join(OUT, "g11_host_signals.npz"), os.path.getsize(os.path.join(OUT, "g11_host_signals.npz")), "bytes;",
      os.path.join(OUT, "g12_knob_grad.npz"), os.path.getsize(os.path.join(OUT, "g12_knob_grad.npz")), "bytes")

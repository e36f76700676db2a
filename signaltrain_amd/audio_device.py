"""Device-side synthetic test signals for the comp_4c task -- SURVEY.md 8(f)-1, the rest of the GPU data feed.

signaltrain/audio.py:85-196 and :296-334 (`synth_input_sample` and the generators it calls for the compressor's chooser set
{0, 1, 2, 4, 6, 7}, datasets.py:317) restated as BATCHED device computations: every window of a minibatch draws its own
chooser, tone count, amplitudes, frequencies, onsets ... from a device `torch.Generator` (no host RNG, no host loop), the
waveforms are evaluated elementwise over [B, L] and normalised per window exactly as `normish` / `pinknoise` do.  The 1/f noise
keeps the reference's construction -- an inverse real FFT of a REAL spectrum `(2 u - 1) / sqrt(k + 1)` (so it is an even
sequence, like the reference's) -- through `torch.fft.irfft` (rocFFT): the feed is not the hot path (SURVEY.md 8, rows
"next"), the one sequential, expensive part of it -- the compressor -- is the hand-written HIP kernel `st_compressor_4c`.

Parity is distributional (the draws come from a different generator than numpy's): tests compare amplitude ranges, onset
statistics and spectra with the host generators of signaltrain_amd/audio.py; the effect itself is pinned by golden G9.
"""
import math

import torch

COMPRESSOR_CHOOSERS = (0, 1, 2, 4, 6, 7)        # datasets.py:317


def _u(gen, *shape, device):
    return torch.rand(*shape, generator=gen, device=device, dtype=torch.float32)


def _sign(gen, *shape, device):
    return torch.where(_u(gen, *shape, device=device) < 0.5, -1.0, 1.0)


def _normish(y, gen, lo=0.6, hi=0.9):
    """audio.py:78-81: y / max|y| * U(lo, hi), per window."""
    B = y.shape[0]
    peak = y.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    return y / peak * ((hi - lo) * _u(gen, B, 1, device=y.device) + lo)


def pinknoise(B, N, gen, device):
    """audio.py:85-94: irfft of a real spectrum (2u-1)/sqrt(k+1), k = 0..N/2, normalised to unit peak."""
    nf = N // 2 + 1
    spec = (2.0 * _u(gen, B, nf, device=device) - 1.0) / torch.sqrt(torch.arange(nf, device=device, dtype=torch.float32) + 1.0)
    y = torch.fft.irfft(spec.to(torch.complex64), n=N, dim=1)
    return y / y.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)


def randsine(t, B, gen, amp_range=(0.2, 0.9), freq_range=(5.0, 150.0)):
    """audio.py:96-104: one or two cosines with random amplitude / frequency / onset, then normish."""
    dev, tl = t.device, float(t[-1])
    n_tones = torch.randint(1, 3, (B, 1), generator=gen, device=dev)
    y = torch.zeros(B, t.shape[0], device=dev)
    for i in range(2):
        amp = amp_range[0] + (amp_range[1] - amp_range[0]) * _u(gen, B, 1, device=dev)
        freq = freq_range[0] + (freq_range[1] - freq_range[0]) * _u(gen, B, 1, device=dev)
        t0 = _u(gen, B, 1, device=dev) * tl
        y = y + torch.where(n_tones > i, amp, torch.zeros_like(amp)) * torch.cos(freq * (t[None, :] - t0))
    return _normish(y, gen)


def box(t, B, gen):
    """audio.py:106-124 with delta = 0: three plateaus; sample i_up - 1 keeps the end height (the reference's slice bounds)."""
    dev, n = t.device, t.shape[0]
    h0, h1, h2 = 0.15 * _u(gen, B, 1, device=dev), 0.35 * _u(gen, B, 1, device=dev) + 0.6, 0.2 * _u(gen, B, 1, device=dev) + 0.1
    i_up = (0.3 * _u(gen, B, 1, device=dev) * n).floor()
    i_dn = torch.minimum(i_up + ((0.3 + 0.35 * _u(gen, B, 1, device=dev)) * n).floor(), torch.full_like(i_up, n - 1))
    i = torch.arange(n, device=dev, dtype=torch.float32)[None, :]
    x = h2.expand(B, n).clone()
    x = torch.where(i < i_up - 1, h0, x)
    x = torch.where((i >= i_up) & (i < i_dn), h1, x)
    return x


def expdecay(t, B, gen):
    """audio.py:126-136: plateau, then an exponential decay from a random onset."""
    dev, tl = t.device, float(t[-1])
    t0 = 0.35 * _u(gen, B, 1, device=dev) * tl
    hi, lo = 0.35 * _u(gen, B, 1, device=dev) + 0.6, 0.1 * _u(gen, B, 1, device=dev) + 0.1
    decay = 12.0 * _u(gen, B, 1, device=dev)
    tt = t[None, :]
    return torch.where(tt < t0, lo, torch.exp(-decay * (tt - t0)) * hi)


def pluck(t, B, gen, freq_range=(50.0, 6400.0)):
    """audio.py:138-148: one to three sines with random sign / phase under an exponential decay, then normish."""
    dev, tl = t.device, float(t[-1])
    n_tones = torch.randint(1, 4, (B, 1), generator=gen, device=dev)
    y = torch.zeros(B, t.shape[0], device=dev)
    for i in range(3):
        amp0 = (0.45 * _u(gen, B, 1, device=dev) + 0.5) * _sign(gen, B, 1, device=dev)
        t0 = (2.0 * _u(gen, B, 1, device=dev) - 1.0) * 0.3 * tl
        freq = freq_range[0] + (freq_range[1] - freq_range[0]) * _u(gen, B, 1, device=dev)
        y = y + torch.where(n_tones > i, amp0, torch.zeros_like(amp0)) * torch.sin(freq * (t[None, :] - t0))
    return _normish(y * expdecay(t, B, gen), gen)


def synth_input_batch(B, chunk_size, sr, gen, device, choosers=COMPRESSOR_CHOOSERS, chooser=None):
    """B windows of audio.synth_input_sample (audio.py:296-334): x [B, chunk_size] float32 on `device`, and the chooser drawn
    for each window.  chooser: force one signal family for every window (tests)."""
    t = torch.arange(chunk_size, device=device, dtype=torch.float32) / float(sr)
    n = chunk_size
    if chooser is None:
        pick = torch.tensor(choosers, device=device)[torch.randint(0, len(choosers), (B,), generator=gen, device=device)]
    else:
        pick = torch.full((B,), int(chooser), device=device)
    c = pick[:, None]
    sine, plk, bx = randsine(t, B, gen), pluck(t, B, gen), box(t, B, gen)
    pink = pinknoise(B, n, gen, device)
    white = 2.0 * _u(gen, B, n, device=device) - 1.0
    y = torch.where(c == 0, sine, torch.zeros_like(sine))
    y = torch.where(c == 1, sine + 0.2 * _u(gen, B, 1, device=device) * pink + 0.2 * _u(gen, B, 1, device=device) * white, y)
    y = torch.where(c == 2, plk, y)
    y = torch.where(c == 4, bx, y)
    y = torch.where(c == 6, bx * white, y)
    y = torch.where(c == 7, plk + (0.3 * _u(gen, B, 1, device=device) + 0.1) * pink, y)
    known = torch.zeros_like(c, dtype=torch.bool)
    for k in (0, 1, 2, 4, 6, 7):
        known |= c == k
    if not bool(known.all()):
        raise NotImplementedError("audio_device.synth_input_batch: only the compressor's chooser set {0,1,2,4,6,7} is built")
    # audio.py:333: random polarity + 1e-8 of uniform noise (keeps log10 away from exact zeros downstream)
    return y * _sign(gen, B, 1, device=device) + _u(gen, B, n, device=device) * 1e-8, pick


def random_ends(B, K, gen, device):
    """audio.py:20-21: Beta(0.8, 0.8) -- emphasises the ends of the knob range -- as a ratio of Gamma draws."""
    conc = torch.full((B, K), 0.8, device=device)
    # torch has no generator-aware Beta: Gamma(a) by Marsaglia-Tsang on a + 1 with the u^(1/a) boost, from this generator's draws
    def gamma(a):
        d = a + 1.0 - 1.0 / 3.0
        cc = 1.0 / torch.sqrt(9.0 * d)
        out = torch.zeros_like(a); todo = torch.ones_like(a, dtype=torch.bool)
        for _ in range(16):
            z = torch.randn(a.shape, generator=gen, device=device)
            v = (1.0 + cc * z) ** 3
            u = _u(gen, *a.shape, device=device)
            ok = (v > 0) & (torch.log(u.clamp_min(1e-30)) < 0.5 * z * z + d - d * v + d * torch.log(v.clamp_min(1e-30)))
            take = todo & ok
            out = torch.where(take, d * v, out); todo = todo & ~take
            if not bool(todo.any()):
                break
        return out * _u(gen, *a.shape, device=device).clamp_min(1e-30) ** (1.0 / a)
    x, y = gamma(conc), gamma(conc)
    return x / (x + y)

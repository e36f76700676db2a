"""Host-side restatement of the reference's synthetic test signals -- TEST / CHECKER INFRASTRUCTURE ONLY.

signaltrain/audio.py:23-49 (sliding_window), :78-196 (normish, pinknoise, randsine, box, expdecay, pluck) and :296-334 (synth_input_sample
for the compressor's chooser set {0,1,2,4,6,7}, datasets.py:317) with numpy's global generator, draw for draw like the reference, plus the
item / batch assembly of SynthAudioDataSet.gen_single_chunk (datasets.py:312-334).  PINNED: tools/capture_golden_r4.py runs the imported
reference's synth_input_sample / gen_single_chunk at fixed numpy seeds, asserts this module reproduces them bit for bit and writes
tests/golden/g11_host_signals.npz (re-checked without the reference by tests/test_oracle_golden.py).  The product generates its training data on the GPU
(signaltrain_amd/csrc/st_feed.h, signaltrain_amd/audio_device.py); these functions are what the distributional tests of that feed compare against
and what the long-file inference test windows its signal with.  Only tests/ import this module."""
import numpy as np


def random_ends(size=1):
    """audio.py:20-21: Beta(0.8,0.8), emphasises the range ends."""
    return np.random.beta(0.8, 0.8, size=size)


def sliding_window(x, size, overlap=0):
    """audio.py:23-49: stack a 1-D array into overlapping windows (zero padded to fit)."""
    step = size - overlap
    remainder = (x.shape[-1] - size) % step
    if remainder != 0:
        x = np.pad(x, (0, step - remainder), mode='constant')
    nwin = (x.shape[-1] - size) // step + 1
    shape = x.shape[:-1] + (nwin, size)
    strides = x.strides[:-1] + (step * x.strides[-1], x.strides[-1])
    return np.lib.stride_tricks.as_strided(x, shape=shape, strides=strides, writeable=False)


def normish(y, amp_range=None, randfunc=np.random.rand):
    lo, hi = (0.6, 0.9) if amp_range is None else amp_range
    return y / np.max(np.abs(y)) * ((hi - lo) * randfunc() + lo)


def pinknoise(N):
    nf = N // 2 + 1
    noise = 2 * np.random.random(nf) - 1
    y = np.fft.irfft(noise / np.sqrt(np.arange(nf) + 1.)).real
    return y / np.max(np.abs(y))


def randsine(t, randfunc=np.random.rand, amp_range=(0.2, 0.9), freq_range=(5, 150), n_tones=None, t0_fac=None):
    y = np.zeros(t.shape[0])
    n_tones = np.random.randint(1, 3) if n_tones is None else n_tones
    for _ in range(n_tones):
        amp = amp_range[0] + (amp_range[1] - amp_range[0]) * randfunc()
        freq = freq_range[0] + (freq_range[1] - freq_range[0]) * randfunc()
        t0 = randfunc() * t[-1] if t0_fac is None else t0_fac * t[-1]
        y += amp * np.cos(freq * (t - t0))
    return normish(y, randfunc=randfunc)


def box(t, randfunc=np.random.rand, t0_fac=None):
    h0, h1, h2 = 0.15 * randfunc(), 0.35 * randfunc() + 0.6, 0.2 * randfunc() + 0.1
    n = len(t)
    i_up = int(0.3 * randfunc() * n) if t0_fac is None else int(t0_fac * n)
    i_dn = min(i_up + int((0.3 + 0.35 * randfunc()) * n), n - 1)
    x = h2 * np.ones(n).astype(t.dtype, copy=False)
    x[0:max(i_up - 1, 0)] = h0
    x[i_up:i_dn] = h1
    return x


def expdecay(t, randfunc=np.random.rand, t0_fac=None):
    t0 = 0.35 * randfunc() * t[-1] if t0_fac is None else t0_fac * t[-1]
    hi, lo = 0.35 * randfunc() + 0.6, 0.1 * randfunc() + 0.1
    x = np.exp(-12 * randfunc() * (t - t0)) * hi
    x[np.where(t < t0)] = lo
    return x


def pluck(t, randfunc=np.random.rand, freq_range=(50, 6400), n_tones=None, t0_fac=None):
    y = np.zeros(t.shape[0])
    n_tones = np.random.randint(1, 4) if n_tones is None else n_tones
    for _ in range(n_tones):
        amp0 = (0.45 * randfunc() + 0.5) * np.random.choice([-1, 1])
        t0 = (2. * randfunc() - 1) * 0.3 * t[-1] if t0_fac is None else t0_fac * t[-1]
        freq = freq_range[0] + (freq_range[1] - freq_range[0]) * randfunc()
        y += amp0 * np.sin(freq * (t - t0))
    return normish(y * expdecay(t, t0_fac=t0_fac), randfunc=randfunc)


def synth_input_sample(t, chooser=None, randfunc=np.random.rand, t0_fac=None):
    """audio.py:296-334 for the compressor chooser set {0,1,2,4,6,7} (datasets.py:317)."""
    if chooser is None:
        chooser = np.random.choice([0, 1, 2, 4, 6, 7])
    n = t.shape[0]
    if chooser == 0:
        y = randsine(t, t0_fac=t0_fac)
    elif chooser == 1:
        y = randsine(t, t0_fac=t0_fac) + 0.2 * np.random.rand() * pinknoise(n) + 0.2 * np.random.rand() * (2 * np.random.rand(n) - 1)
    elif chooser == 2:
        y = pluck(t, t0_fac=t0_fac)
    elif chooser == 4:
        y = box(t, t0_fac=t0_fac)
    elif chooser == 6:
        y = box(t, t0_fac=t0_fac) * (2 * np.random.rand(n) - 1)
    elif chooser == 7:
        amp_n = 0.3 * randfunc() + 0.1            # drawn BEFORE pluck() (audio.py:317-318)
        y = pluck(t, t0_fac=t0_fac) + amp_n * pinknoise(n)
    else:
        raise NotImplementedError(f"signaltrain_amd.audio: test signal {chooser} is not built (compressor set is 0,1,2,4,6,7)")
    return y * np.random.choice([-1, 1]) + np.random.rand(n) * 1e-8


def gen_single_chunk(t, effect, y_size, augment=True, chooser=None, knobs=None):
    """datasets.py:312-334: one (x, y, knobs) item; `effect` is a signaltrain_amd.audio Effect (its host go())."""
    if chooser is None:
        chooser = np.random.choice([0, 1, 2, 4, 6, 7])            # datasets.py:317
    x = synth_input_sample(t, chooser)
    if knobs is None:
        knobs = random_ends(len(effect.knob_ranges)) - 0.5
    y, x = effect.go(x, knobs)
    y = y[-y_size:]
    if augment and np.random.choice([True, False]):               # do_augment, datasets.py:27-29
        x, y = -x, -y
    return x, y, knobs


def batch(B, chunk_size, effect, y_size, sr=44100, augment=True):
    """B stacked items as float32 arrays."""
    t = np.arange(chunk_size, dtype=np.float32) / sr
    xs, ys, ks = zip(*(gen_single_chunk(t, effect, y_size, augment) for _ in range(B)))
    return np.stack(xs).astype(np.float32), np.stack(ys).astype(np.float32), np.stack(ks).astype(np.float32)

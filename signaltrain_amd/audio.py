"""Effects and audio files -- the parts of signaltrain/audio.py the training driver needs: the Effect classes (:449-537, knob names / ranges,
normalised <-> world knob coordinates), the comp_4c target effect (compressor_4controls :380-426: on the GPU through st_compressor_4c /
st_synth_comp4c, on the host through the gcc-built helper for file datasets), wav reading / writing (:207-262) and file-defined effects
(:624-670).  The synthetic test signals themselves are generated on the GPU (csrc/st_feed.h, audio_device.py); their numpy restatement lives
on the checker side (oracle/host_audio.py)."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_AUD = None


def _audio_lib():
    global _AUD
    if _AUD is None:
        path = os.path.join(_HERE, "libst_audio.so")
        if os.path.isfile(path):
            lib = C.CDLL(path)
            lib.st_compressor_4controls.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_double] * 5
            lib.st_compressor_4controls.restype = None
            _AUD = lib
        else:
            _AUD = False
    return _AUD


def compressor_4controls(x, thresh=-24.0, ratio=2.0, attackTime=0.01, releaseTime=0.01, sr=44100.0):
    """audio.py:380-426.  Uses the gcc-built helper when present, else a numpy/python loop."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    lib = _audio_lib()
    if lib:
        y = np.empty_like(x)
        lib.st_compressor_4controls(x.ctypes.data, y.ctypes.data, x.size, float(thresh), float(ratio),
                                    float(attackTime), float(releaseTime), float(sr))
        return y
    N = len(x)
    alphaA = np.exp(-np.log(9) / (sr * attackTime)); alphaR = np.exp(-np.log(9) / (sr * releaseTime))
    x_dB = np.maximum(20 * np.log10(np.abs(x) + 1e-8), -96).astype(np.float32)
    gc = np.zeros(N, dtype=np.float32)
    i = x_dB > thresh
    gc[i] = thresh + (x_dB[i] - thresh) / ratio - x_dB[i]
    lin = np.zeros(N, dtype=np.float32)
    prev = 0.0
    g = gc.tolist()
    out = [0.0] * N
    for n in range(1, N):
        prev = float(np.float32((1 - alphaA) * g[n] + alphaA * prev if g[n] < prev else (1 - alphaR) * g[n] + alphaR * prev))
        out[n] = prev
    lin = np.power(10.0, np.asarray(out, dtype=np.float32) / 20)
    return (lin * x).astype(np.float32)


class Effect:
    """audio.py:449-480."""

    def __init__(self, sr=44100.0, dtype=np.float32):
        self.name = 'Generic Effect'; self.knob_names = ['knob']
        self.knob_ranges = np.array([[0, 1]], dtype=dtype); self.sr = sr; self.is_inverse = False

    def knobs_wc(self, knobs_nn):
        return (self.knob_ranges[:, 0] + (knobs_nn + 0.5) * (self.knob_ranges[:, 1] - self.knob_ranges[:, 0])).tolist()

    def info(self):
        print(f'Effect: {self.name}.  Knobs:')
        for n, r in zip(self.knob_names, self.knob_ranges):
            print(f'                            {n}: {r[0]} to {r[1]}')

    def go_wc(self, x, knobs_wc):
        raise Exception("This effect's go_wc() is undefined")

    def go(self, x, knobs_nn, **kwargs):
        return self.go_wc(x, self.knobs_wc(knobs_nn), **kwargs)


class Compressor_4c(Effect):
    """audio.py:493-500."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.name = 'Compressor_4c'
        self.knob_names = ['threshold', 'ratio', 'attackTime', 'releaseTime']
        self.knob_ranges = np.array([[-30, 0], [1, 5], [1e-3, 4e-2], [1e-3, 4e-2]])

    def go_wc(self, x, knobs_w):
        return compressor_4controls(x, thresh=knobs_w[0], ratio=knobs_w[1], attackTime=knobs_w[2],
                                    releaseTime=knobs_w[3], sr=self.sr), x

    def go_device(self, x, knobs_nn, y_size=None):
        """Batched effect on the GPU (st_compressor_4c): x [B,L] and knobs_nn [B,4] in [-.5,.5] as device tensors ->
        y [B,y_size] (the last y_size samples, datasets.py:327-330).  No CPU fallback."""
        import ctypes as C
        import torch
        from . import _lib
        if x.device.type != "cuda":
            raise RuntimeError("Compressor_4c.go_device needs ROCm device tensors")
        x = x.to(torch.float32).contiguous(); B, L = x.shape
        y_size = L if y_size is None else int(y_size)
        lo = torch.as_tensor(self.knob_ranges[:, 0], dtype=torch.float32, device=x.device)
        hi = torch.as_tensor(self.knob_ranges[:, 1], dtype=torch.float32, device=x.device)
        kw = (lo + (knobs_nn.to(torch.float32) + 0.5) * (hi - lo)).contiguous()          # Effect.knobs_wc, audio.py:455
        y = torch.empty(B, y_size, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):             # the library launches on the current device
            _lib.check(_lib.load().st_compressor_4c(_lib.ptr(x), _lib.ptr(kw), float(self.sr), B, L, y_size, _lib.ptr(y),
                                                    C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "st_compressor_4c")
        return y


class Compressor_4c_Large(Compressor_4c):
    """audio.py:503-510."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.name = 'Compressor_4c_Large'
        self.knob_ranges = np.array([[-50, 0], [1.5, 10], [1e-3, 1], [1e-3, 1]])


# ------------------------------------------------------------------------------------------------ wav files + file-defined effects
def mu_compand(y, mu=32):
    """audio.py:339-340: mu-law companding (run_train.py --compand, datasets.py:218-220, utils/predict_long.py:38-40)."""
    return np.sign(y) * np.log(1 + mu * np.abs(y)) / np.log(1 + mu)


def mu_decompand(y, mu=32):
    """audio.py:343-344."""
    return np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)


def read_audio_file(filename, sr=44100, mono=True, norm=False, dtype=np.float32, **_ignored):
    """audio.py:207-255: a wav file as float in [-1, 1] (int16 / 32767), first channel if `mono`.  A file at another sample rate is resampled to
    `sr` with a polyphase filter (scipy.signal.resample_poly; the reference calls librosa.resample there -- another low-pass design, so such
    files agree with the reference's to the filters' pass-band ripple, not bit for bit)."""
    from scipy.io import wavfile
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        read_sr, signal = wavfile.read(filename)
    if mono and signal.ndim > 1:
        signal = signal[:, 0]
    if signal.dtype == np.int16:
        signal = np.array(signal / 32767.0, dtype=dtype)
    signal = signal.astype(dtype, copy=False)
    if read_sr != int(sr):
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(read_sr))
        print(f"read_audio_file: {filename}: resampling {read_sr} Hz -> {int(sr)} Hz")
        signal = resample_poly(signal.astype(np.float64), int(sr) // g, int(read_sr) // g, axis=0).astype(dtype)
    if norm:
        m = np.max(np.abs(signal))
        signal = signal / m if m > 0 else signal
    return signal, sr


def write_audio_file(filename, data, sr=44100):
    """audio.py:258-262."""
    from scipy.io import wavfile
    wavfile.write(filename, sr, data)


class FileEffect(Effect):
    """audio.py:624-670: an effect that exists only as recordings -- <path>/Train, <path>/Val with input_* / target_* pairs and
    <path>/effect_info.ini ([effect] name, knob_names, knob_ranges [, inverse]); e.g. the LA2A of BASELINE configs[3]."""

    def __init__(self, path, sr=44100):
        super().__init__(sr=sr)
        import ast
        import configparser
        import glob
        if path is None or not glob.glob(path + "/Train/target*") or not glob.glob(path + "/Val/target*") or not glob.glob(path + "/effect_info.ini"):
            raise FileNotFoundError(f"FileEffect: no Train/ + Val/ target files or effect_info.ini under {path}")
        cfg = configparser.ConfigParser(); cfg.read(path + "/effect_info.ini")
        self.name = cfg["effect"]["name"].strip("'\"") + "(files)"
        self.knob_names = list(ast.literal_eval(cfg.get("effect", "knob_names")))       # literal_eval instead of the reference's eval
        self.knob_ranges = np.array(ast.literal_eval(cfg.get("effect", "knob_ranges")), dtype=np.float64)
        if cfg["effect"].get("inverse"):
            self.is_inverse = True; self.name = "De-" + self.name

    def go_wc(self, x, knobs_w):
        return None                      # there is no plugin to call: the targets are recordings

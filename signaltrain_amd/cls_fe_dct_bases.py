"""Cosine-modulated (DCT-IV style) learned-basis front/back end -- mirror of signaltrain/cls_fe_dct_bases.py
(core_modulation :57-97, Analysis :100-136, Synthesis :139-179).  SURVEY.md row a15: not used by st_model, kept as a
drop-in alternative front end.  Same parameter names/shapes as the reference (`conv_analysis.weight [C,1,KW]`,
`conv_analysis.bias [C]`, `conv_synthesis.weight [C,1,KW]`); forward AND autograd run in libsignaltrain_hip.so on the
framed fp32-MFMA GEMM family (st_fe_* entry points).  ROCm device only -- no CPU fallback."""
import ctypes as C
import numpy as np
import torch
import torch.nn as nn

from . import _lib


def core_modulation(freq_subbands, window_size):
    """cls_fe_dct_bases.py:57-97 ('scott' method): cosine window x cos(pi/C (k+.5)(n+.5+C/2)) sqrt(2/C), float32 [C, KW]."""
    w = np.sin(np.pi / window_size * (np.arange(window_size) + 0.5))           # scipy.signal.cosine(window_size)
    kvec = np.arange(0, freq_subbands) + 0.5
    nvec = np.arange(0, window_size) + 0.5 + freq_subbands / 2
    cos_an = w * np.cos(np.pi / freq_subbands * kvec[np.newaxis].T * nvec) * np.sqrt(2. / freq_subbands)
    return cos_an.astype(np.float32, copy=False)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Conv(nn.Module):
    """Parameter holder with the attribute names of nn.Conv1d / nn.ConvTranspose1d (.weight [C,1,KW], .bias [C] or None)."""

    def __init__(self, w, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.from_numpy(np.ascontiguousarray(w[:, None, :])))
        if bias:
            # nn.Conv1d default bias init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)), fan_in = 1 * KW
            b = 1.0 / np.sqrt(w.shape[1])
            self.bias = nn.Parameter(torch.empty(w.shape[0]).uniform_(-b, b))
        else:
            self.register_parameter("bias", None)


class _AnalysisFn(torch.autograd.Function):
    @staticmethod
    @_lib.on_arg_device
    def forward(ctx, wave, W, bias, hop, pad):
        lib = _lib.load()
        x = wave.contiguous().float(); B, L = x.shape
        Cn, _, KW = W.shape
        T = lib.st_fe_frames(L, KW, hop, pad)
        out = torch.empty(B, T, Cn, device=x.device)
        Wc = W.detach().contiguous(); bc = None if bias is None else bias.detach().contiguous()
        _lib.check(lib.st_fe_analysis_fwd(_lib.ptr(x), B, L, _lib.ptr(Wc), _lib.ptr(bc), Cn, KW, hop, pad, _lib.ptr(out), _stream()),
                   "st_fe_analysis_fwd")
        ctx.save_for_backward(x, Wc); ctx.geom = (hop, pad, bias is not None)
        return out

    @staticmethod
    @_lib.on_arg_device
    def backward(ctx, g_out):
        lib = _lib.load()
        x, W = ctx.saved_tensors; hop, pad, has_bias = ctx.geom
        B, L = x.shape; Cn, _, KW = W.shape
        g = g_out.contiguous().float()
        ws = torch.empty(lib.st_fe_ws_floats(B, L, Cn, KW, hop, pad), device=x.device)
        gW = torch.empty_like(W); gb = torch.empty(Cn, device=x.device) if has_bias else None
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _lib.check(lib.st_fe_analysis_bwd(_lib.ptr(x), B, L, _lib.ptr(W), Cn, KW, hop, pad, _lib.ptr(g), _lib.ptr(ws), _lib.ptr(gW),
                                          _lib.ptr(gb), _lib.ptr(gx), _stream()), "st_fe_analysis_bwd")
        return gx, gW, gb, None, None


class _SynthesisFn(torch.autograd.Function):
    @staticmethod
    @_lib.on_arg_device
    def forward(ctx, x_ft, W, hop, crop):
        lib = _lib.load()
        xf = x_ft.contiguous().float(); B, T, Cn = xf.shape
        KW = W.shape[2]
        n = (T - 1) * hop + KW - 2 * crop
        Wc = W.detach().contiguous()
        ws = torch.empty(B * T * KW, device=xf.device)
        out = torch.empty(B, 1, n, device=xf.device)
        _lib.check(lib.st_fe_synthesis_fwd(_lib.ptr(xf), B, T, _lib.ptr(Wc), Cn, KW, hop, crop, _lib.ptr(ws), _lib.ptr(out), _stream()),
                   "st_fe_synthesis_fwd")
        ctx.save_for_backward(xf, Wc); ctx.geom = (hop, crop)
        return out

    @staticmethod
    @_lib.on_arg_device
    def backward(ctx, g_wave):
        lib = _lib.load()
        xf, W = ctx.saved_tensors; hop, crop = ctx.geom
        B, T, Cn = xf.shape; KW = W.shape[2]
        n = (T - 1) * hop + KW - 2 * crop
        g = g_wave.contiguous().float().reshape(B, n)
        ws = torch.empty(lib.st_fe_ws_floats(B, n + 2 * crop, Cn, KW, hop, crop), device=xf.device)
        gW = torch.empty_like(W)
        gx = torch.empty_like(xf) if ctx.needs_input_grad[0] else None
        _lib.check(lib.st_fe_synthesis_bwd(_lib.ptr(xf), B, T, _lib.ptr(W), Cn, KW, hop, crop, _lib.ptr(g), _lib.ptr(ws), _lib.ptr(gW),
                                           _lib.ptr(gx), _stream()), "st_fe_synthesis_bwd")
        return gx, gW, None, None


class Analysis(nn.Module):
    """cls_fe_dct_bases.py:100-136.  forward accepts the reference's numpy waveform [B, L] or a device tensor."""

    def __init__(self, ft_size=1024, w_size=2048, hop_size=1024, shrink=False):
        super().__init__()
        self.batch_size = None; self.time_domain_samples = None
        self.sz, self.wsz, self.hop = ft_size, w_size, hop_size
        self.conv_analysis = _Conv(core_modulation(self.sz, self.wsz), bias=True)      # Conv1d(1, sz, wsz, padding=sz, stride=hop, bias=True)

    def initialize(self):
        with torch.no_grad():
            self.conv_analysis.weight.copy_(torch.from_numpy(core_modulation(self.sz, self.wsz)[:, None, :]))

    def forward(self, wave_form):
        dev = self.conv_analysis.weight.device
        if dev.type != "cuda":
            raise RuntimeError("signaltrain_amd front end runs on a ROCm device only (no CPU fallback); call .cuda() first")
        if isinstance(wave_form, np.ndarray):                          # cls_fe_dct_bases.py:130
            wave_form = torch.from_numpy(wave_form).to(dev).requires_grad_(True)
        return _AnalysisFn.apply(wave_form, self.conv_analysis.weight, self.conv_analysis.bias, self.hop, self.sz)


class Synthesis(nn.Module):
    """cls_fe_dct_bases.py:139-179: ConvTranspose1d(sz, 1, wsz, stride=hop, bias=False), crop sz samples from each end."""

    def __init__(self, ft_size=1024, w_size=2048, hop_size=1024):
        super().__init__()
        self.batch_size = None; self.time_domain_samples = None
        self.sz, self.wsz, self.hop = ft_size, w_size, hop_size
        self.half_N = int(self.sz / 2 + 1)
        self.conv_synthesis = _Conv(core_modulation(self.sz, self.wsz), bias=False)
        self.h_tanh = torch.nn.Hardtanh(); self.tanh = torch.nn.Tanh()

    def initialize(self):
        with torch.no_grad():
            self.conv_synthesis.weight.copy_(torch.from_numpy(core_modulation(self.sz, self.wsz)[:, None, :]))

    def forward(self, x_ft):
        if x_ft.device.type != "cuda":
            raise RuntimeError("signaltrain_amd front end runs on a ROCm device only (no CPU fallback)")
        return _SynthesisFn.apply(x_ft, self.conv_synthesis.weight, self.hop, self.sz)

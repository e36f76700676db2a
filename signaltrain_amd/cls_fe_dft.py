"""Learned-basis DFT front/back end -- mirror of signaltrain/cls_fe_dft.py (Analysis :12-58, Synthesis :61-163).

The modules hold the same parameters (names, shapes [N,1,N], init) as the reference's Conv1d /
ConvTranspose1d; forward/backward run in libsignaltrain_hip.so (framed fp32-MFMA GEMMs)."""
import ctypes as C
import numpy as np
import torch
import torch.nn as nn

from . import _lib


def hamming(N):
    """scipy.signal.hamming(N) (symmetric) used at cls_fe_dft.py:38,148."""
    n = np.arange(N, dtype=np.float64)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * n / (N - 1))


def _dft_bases(N, window):
    f = np.fft.fft(np.eye(N), norm='ortho')                       # cls_fe_dft.py:37,88
    return ((np.real(f) * window).astype(np.float32)[:, None, :], (np.imag(f) * window).astype(np.float32)[:, None, :])


class _Basis(nn.Module):
    """Parameter holder with the attribute name `.weight` ([N,1,N]) of Conv1d/ConvTranspose1d."""

    def __init__(self, w):
        super().__init__()
        self.weight = nn.Parameter(torch.from_numpy(w.copy()))


def _dims(B, L, N, H, T, OT, K=0):
    d = _lib.st_dims(); d.B, d.L, d.N, d.H, d.T, d.OT, d.F, d.K = B, L, N, H, T, OT, N // 2 + 1, K
    d.y = (OT - 1) * H - N
    return d


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _AnalysisFn(torch.autograd.Function):
    @staticmethod
    @_lib.on_arg_device
    def forward(ctx, wave, Wr, Wi, N, H):
        lib = _lib.load()
        B, L = wave.shape
        T = (L + 2 * N - N) // H + 1                                # Conv1d(k=N, stride=H, padding=N)
        d = _dims(B, L, N, H, T, 1)
        d.y = 4; d.OT = 1                                           # unused by this op
        re = torch.empty(B, T, N // 2 + 1, device=wave.device); im = torch.empty_like(re)
        x = wave.contiguous().float(); Wr_ = Wr.contiguous(); Wi_ = Wi.contiguous()
        _lib.check(lib.st_analysis_fwd(C.byref(_relax(d)), _lib.ptr(x), _lib.ptr(Wr_), _lib.ptr(Wi_), 1.0, _lib.ptr(re),
                                       _lib.ptr(im), None, None, _stream()), "st_analysis_fwd")
        ctx.save_for_backward(x, Wr_, Wi_); ctx.geom = (B, L, N, H, T)
        return re, im

    @staticmethod
    @_lib.on_arg_device
    def backward(ctx, g_re, g_im):
        lib = _lib.load()
        (x, Wr, Wi) = ctx.saved_tensors
        B, L, N, H, T = ctx.geom
        gx = None
        if ctx.needs_input_grad[0]:
            # d/d(wave) of the Conv1d pair (cls_fe_dft.py:55-56) = conv-transpose of the output gradients with the same kernels, cropped by
            # the padding: the generic front-end entry does it for C channels -- the F used rows of both bases stacked (padded to C % 16 == 0)
            F = N // 2 + 1; Cp = -(-2 * F // 16) * 16
            Wc = torch.zeros(Cp, N, device=x.device); Wc[:F] = Wr.reshape(N, N)[:F]; Wc[F:2 * F] = Wi.reshape(N, N)[:F]
            gc = torch.zeros(B * T, Cp, device=x.device); gc[:, :F] = g_re.reshape(B * T, F); gc[:, F:2 * F] = g_im.reshape(B * T, F)
            wsf = torch.empty(lib.st_fe_ws_floats(B, L, Cp, N, H, N), device=x.device)
            gWc = torch.empty(Cp, N, device=x.device); gx = torch.empty(B, L, device=x.device)
            _lib.check(lib.st_fe_analysis_bwd(_lib.ptr(x), B, L, _lib.ptr(Wc), Cp, N, H, N, _lib.ptr(gc), _lib.ptr(wsf), _lib.ptr(gWc), None,
                                              _lib.ptr(gx), _stream()), "st_fe_analysis_bwd")
        d = _relax(_dims(B, L, N, H, T, 1))
        F = N // 2 + 1; KP = lib.st_kp(F)
        dG = torch.zeros(B * T, KP, device=x.device)
        dG[:, :F] = g_re.reshape(B * T, F); dG[:, KP // 2:KP // 2 + F] = g_im.reshape(B * T, F)
        ws = torch.empty(lib.st_wgrad_ws_floats(C.byref(d)), device=x.device)
        gWr = torch.zeros(N, 1, N, device=x.device); gWi = torch.zeros(N, 1, N, device=x.device)
        npart = torch.empty(lib.st_norm_partials(C.byref(d)), device=x.device)
        _lib.check(lib.st_analysis_wgrad(C.byref(d), _lib.ptr(dG), _lib.ptr(x), 1.0, _lib.ptr(ws), _lib.ptr(gWr), _lib.ptr(gWi),
                                         _lib.ptr(npart), _stream()), "st_analysis_wgrad")
        return gx, gWr, gWi, None, None


def _relax(d):
    """Standalone front/back-end calls only use part of st_dims; fill the rest consistently."""
    if d.OT < 3:
        d.OT = max(3, -(-d.N // d.H) + 1)
    d.y = (d.OT - 1) * d.H - d.N
    while d.y <= 0 or d.y % 4:
        d.OT += 1; d.y = (d.OT - 1) * d.H - d.N
    if d.T < d.OT:
        d.T = d.OT
    return d


class Analysis(nn.Module):
    """cls_fe_dft.py:12-58: forward(wave[B,L]) -> (re, im) [B,T,N/2+1]."""

    def __init__(self, ft_size=1024, hop_size=384):
        super().__init__()
        self.sz, self.hop, self.half_N = ft_size, hop_size, int(ft_size / 2. + 1)
        wr, wi = _dft_bases(ft_size, hamming(ft_size))             # cls_fe_dft.py:36-48
        self.conv_analysis_real = _Basis(wr)
        self.conv_analysis_imag = _Basis(wi)

    def forward(self, wave_form):
        return _AnalysisFn.apply(wave_form, self.conv_analysis_real.weight, self.conv_analysis_imag.weight, self.sz, self.hop)


class _SynthesisFn(torch.autograd.Function):
    @staticmethod
    @_lib.on_arg_device
    def forward(ctx, real, imag, Sr, Si, N, H):
        lib = _lib.load()
        B, OT, F = real.shape
        d = _dims(B, 4 * ((OT - 1) * H - N), N, H, OT, OT)
        KP = lib.st_kp(F)
        AA = torch.zeros(B * OT, KP, device=real.device)
        AA[:, :F] = real.reshape(B * OT, F); AA[:, KP // 2:KP // 2 + F] = imag.reshape(B * OT, F)
        Sfold = torch.empty(KP, N, device=real.device); frs = torch.zeros(lib.st_synth_frame_slabs(C.byref(d)), B * OT, N, device=real.device)
        wave = torch.empty(B, d.y, device=real.device)
        Sr_, Si_ = Sr.contiguous(), Si.contiguous()
        _lib.check(lib.st_synth_fold(C.byref(d), _lib.ptr(Sr_), _lib.ptr(Si_), _lib.ptr(Sfold), _stream()), "fold")
        _lib.check(lib.st_synthesis_frames(C.byref(d), _lib.ptr(AA), _lib.ptr(Sfold), _lib.ptr(frs), _stream()), "frames")
        _lib.check(lib.st_ola_loss(C.byref(d), _lib.ptr(frs), None, None, _lib.ptr(wave), None, None, _stream()), "ola")
        ctx.save_for_backward(AA, Sfold); ctx.geom = (B, OT, F, N, H)
        return wave

    @staticmethod
    @_lib.on_arg_device
    def backward(ctx, g_wave):
        lib = _lib.load()
        AA, Sfold = ctx.saved_tensors
        B, OT, F, N, H = ctx.geom
        d = _dims(B, 4 * ((OT - 1) * H - N), N, H, OT, OT)
        KP = lib.st_kp(F)
        dsyn = g_wave.contiguous().float()
        dAA = torch.zeros(lib.st_synth_slabs(C.byref(d)), B * OT, KP, device=AA.device)
        _lib.check(lib.st_synthesis_dgrad(C.byref(d), _lib.ptr(dsyn), _lib.ptr(Sfold), _lib.ptr(dAA), _stream()), "dgrad")
        ws = torch.empty(lib.st_wgrad_ws_floats(C.byref(d)), device=AA.device)
        gSr = torch.zeros(N, 1, N, device=AA.device); gSi = torch.zeros(N, 1, N, device=AA.device)
        npart = torch.empty(lib.st_norm_partials(C.byref(d)), device=AA.device)
        _lib.check(lib.st_synthesis_wgrad(C.byref(d), _lib.ptr(AA), _lib.ptr(dsyn), _lib.ptr(ws), _lib.ptr(gSr), _lib.ptr(gSi),
                                          _lib.ptr(npart), _stream()), "wgrad")
        dAA = dAA.sum(0)
        g_re = dAA[:, :F].reshape(B, OT, F); g_im = dAA[:, KP // 2:KP // 2 + F].reshape(B, OT, F)
        return g_re, g_im, gSr, gSi, None, None


class Synthesis(nn.Module):
    """cls_fe_dft.py:61-163: forward(real, imag [B,OT,N/2+1]) -> wave [B,(OT-1)*H-N]."""

    def __init__(self, ft_size=1024, hop_size=384):
        super().__init__()
        self.sz, self.hop, self.half_N = ft_size, hop_size, int(ft_size / 2 + 1)
        sr, si = _dft_bases(ft_size, Synthesis.GLA(ft_size, hop_size, ft_size))   # cls_fe_dft.py:87-100
        self.conv_synthesis_real = _Basis(sr)
        self.conv_synthesis_imag = _Basis(si)

    def forward(self, real, imag):
        return _SynthesisFn.apply(real.contiguous().float(), imag.contiguous().float(),
                                  self.conv_synthesis_real.weight, self.conv_synthesis_imag.weight, self.sz, self.hop)

    @staticmethod
    def flip(x, dim):
        return x.flip(dim)

    @staticmethod
    def GLA(wsz, hop, N=4096):
        """LSEE-MSTFT synthesis window (Griffin & Lim 1984), cls_fe_dft.py:133-163."""
        synw = hamming(wsz)
        prod = synw ** 2.
        env = np.zeros(wsz)
        redundancy = wsz // hop
        for k in range(-redundancy, redundancy + 1):
            ind = hop * k + np.arange(1, wsz + 1)
            valid = (ind > 0) & (ind <= wsz)
            env[ind[valid] - 1] += prod[np.arange(wsz)[valid]]
        return synw / env

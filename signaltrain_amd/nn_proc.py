"""SignalTrain model -- mirror of signaltrain/nn_proc.py (AsymAutoEncoder :28-126, AsymMPAEC :264-340,
st_model :344-393) with the same constructor arguments, attributes, parameter names/shapes and
state_dict layout; all arithmetic runs in libsignaltrain_hip.so.

`st_model.forward` is differentiable through torch.autograd (custom Function -> HIP backward kernels), so
the reference's training loop (loss.backward(); model.clip_grad_norm_(); optimizer.step()) works
unchanged.  The fast path is `StepEngine.train_step` (one fused call per step), used by train.train().
"""
import ctypes as C
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .cls_fe_dft import Analysis, Synthesis
from .engine import StepEngine, param_names

_QUIET = False


def _say(*a):
    if not _QUIET:
        print(*a)


class _Lin(nn.Module):
    """Parameter holder with nn.Linear's attribute names (weight [out,in], bias [out])."""

    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i)); self.bias = nn.Parameter(torch.zeros(o))
        self.in_features, self.out_features = i, o


class AsymAutoEncoder(nn.Module):
    """nn_proc.py:28-126: 9 Linear layers on the time-frame axis, knobs concatenated mid-way, ELU."""

    def __init__(self, T=25, R=64, K=3, OT=None, use_bias=True, use_dropout=False):
        super().__init__()
        if not use_bias or use_dropout:
            raise NotImplementedError("signaltrain_amd: use_bias=False / use_dropout=True are not built (unused by the reference)")
        self._T, self._R, self._K = T, R, K
        self._OT = T if OT is None else OT
        self.use_bias, self.use_dropout = use_bias, use_dropout
        _say("AsymAutoEncoder __init__: T, R, K, OT = ", T, R, K, OT)
        rf = 2
        self.fnn_enc = _Lin(T, R); self.fnn_enc2 = _Lin(R, R // rf); self.fnn_enc3 = _Lin(R // rf, R // rf ** 2)
        self.fnn_enc4 = _Lin(R // rf ** 2, R // rf ** 2)
        self.fnn_addknobs = _Lin(R // rf ** 2 + K, R // rf ** 2)
        self.fnn_dec4 = _Lin(R // rf ** 2, R // rf ** 2); self.fnn_dec3 = _Lin(R // rf ** 2, R // rf)
        self.fnn_dec2 = _Lin(R // rf, R); self.fnn_dec = _Lin(R, self._OT)
        self.layer_list = [self.fnn_enc, self.fnn_enc2, self.fnn_enc3, self.fnn_enc4, self.fnn_addknobs,
                           self.fnn_dec4, self.fnn_dec3, self.fnn_dec2, self.fnn_dec]
        self.initialize()

    def initialize(self):                                            # nn_proc.py:71-75
        for x in self.layer_list:
            torch.nn.init.xavier_normal_(x.weight)
            x.bias.data.zero_()

    def acts_device(self, x_input, knobs, skip_connections):
        """The ten return_acts tensors of nn_proc.py:77-126 ([B, F, width] each) from the library's diagnostic kernel (st_ae_acts): plain
        fp32 FMAs per (window, bin) row -- the training kernels keep these activations in registers."""
        lib = _lib.load()
        x = x_input.contiguous().float(); kn = knobs.contiguous().float() if self._K else None      # K = 0: an empty tensor has no address; the C ABI takes NULL then
        B, T, F = x.shape
        d = _lib.st_dims(); d.B, d.N, d.F, d.T, d.OT, d.K, d.H = B, 2 * (F - 1), F, T, self._OT, self._K, 384
        d.y = (d.OT - 1) * d.H - d.N; d.L = max(4 * d.y, 4)
        offs, total = _lib.param_offsets(d)
        packed = torch.zeros(offs[22] - offs[4], device=x.device)
        k = 0
        for l in self.layer_list:
            for t in (l.weight, l.bias):
                o = offs[4 + k] - offs[4]; packed[o:o + t.numel()] = t.detach().reshape(-1); k += 1
        flat = torch.empty(int(lib.st_ae_acts_floats(C.byref(d))), device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.st_ae_acts(C.byref(d), _lib.ptr(x), _lib.ptr(kn), _lib.ptr(packed), 1 if skip_connections == 'sf' else 0, _lib.ptr(flat),
                                      C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "st_ae_acts")
        acts, o = [], 0
        for w in (64, 32, 16, 16, 16 + self._K, 16, 16, 32, 64, self._OT):
            acts.append(flat[o:o + B * F * w].view(B, F, w)); o += B * F * w
        return acts

    def forward(self, x_input, knobs, skip_connections='res', return_acts=False):
        """Standalone use of one autoencoder ('sf' and '' modes, the two the model uses)."""
        if skip_connections not in ('sf', ''):
            raise NotImplementedError("signaltrain_amd: only skip_connections 'sf' and '' are built (nn_proc.py:315-316)")
        lib = _lib.load()
        B, T, F = x_input.shape
        d = _lib.st_dims(); d.B, d.N, d.F, d.T, d.OT, d.K, d.H = B, 2 * (F - 1), F, T, self._OT, self._K, 384
        d.y = (d.OT - 1) * d.H - d.N; d.L = max(4 * d.y, 4)
        offs, total = _lib.param_offsets(d)
        pg = offs[22] - offs[4]
        packed = torch.zeros(pg, device=x_input.device)
        k = 0
        for l in self.layer_list:
            for t in (l.weight, l.bias):
                o = offs[4 + k] - offs[4]; packed[o:o + t.numel()] = t.detach().reshape(-1); k += 1
        x = x_input.contiguous().float(); kn = knobs.contiguous().float() if self._K else None      # K = 0: an empty tensor has no address; the C ABI takes NULL then
        KP = lib.st_kp(F)
        mh = torch.empty(B, self._OT, F, device=x.device); ph = torch.empty_like(mh)
        AA = torch.empty(B * self._OT, KP, device=x.device)
        nws = int(lib.st_ae_fwd_ws_floats(C.byref(d)))                 # 0 unless the geometry is wide (T > 32 or OT > 16)
        ws = torch.empty(nws, device=x.device) if nws else None
        with torch.cuda.device(x.device):                             # the library launches on the current device
            st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            _lib.check(lib.st_ae_fwd(C.byref(d), _lib.ptr(x), _lib.ptr(x), _lib.ptr(kn), _lib.ptr(packed), _lib.ptr(packed),
                                     _lib.ptr(mh), _lib.ptr(ph), _lib.ptr(AA), None, _lib.ptr(ws), st), "st_ae_fwd")
        out = mh if skip_connections == 'sf' else ph - x[:, T - self._OT:, :]
        return out, (self.acts_device(x_input, knobs, skip_connections) if return_acts else [])


class _STModelFn(torch.autograd.Function):
    """forward/backward of the whole model through the fused HIP entry points.

    The saved-for-backward state (re, im, mag, phs, AA, ...) lives in the engine's ONE workspace, not in ctx: a second
    forward (a validation batch, another micro-batch) before `backward()` overwrites it.  The forward stamps the engine's
    generation counter into ctx; a stale stamp makes the backward RECOMPUTE the forward from the saved inputs (correct
    gradients, one extra forward) instead of silently differentiating the wrong activations."""

    @staticmethod
    def forward(ctx, model, x, knobs, *params):
        eng = model._engine
        y_hat, mag, mag_hat = eng.forward(x, knobs, save_for_backward=True)
        ctx.model, ctx.generation, ctx.engine = model, eng.generation, eng
        ctx.save_for_backward(x, knobs)
        return y_hat, mag, mag_hat

    @staticmethod
    def backward(ctx, g_y, g_mag, g_mh):
        x, knobs = ctx.saved_tensors
        eng = ctx.model._engine
        if g_y is None:
            g_y = torch.zeros(x.shape[0], eng.dims.y, device=x.device)
        gk = None
        if ctx.needs_input_grad[2]:
            # knobs that require grad (nn_proc.py:92-93 under autograd): the exact per-window route of st_model_knob_grad -- slow (one forward + backward
            # per window: the reference's training never asks for it), and it leaves the LAST window's state in the workspace
            gk = eng.knob_grad(x, knobs, g_y, g_mh, g_mag)
        if gk is not None or eng is not ctx.engine or eng.generation != ctx.generation:
            eng.forward(x, knobs, save_for_backward=True)          # the workspace was reused since this graph's forward: rebuild its state
        eng.backward(x, knobs, g_y, g_mh, g_mag)
        gx = eng.input_grad(x, g_y) if ctx.needs_input_grad[1] else None     # something trainable upstream of the model
        # one flat clone, then per-parameter views of it: autograd accumulates into .grad, so the engine's gradient buffer
        # (overwritten by the next backward) must not be handed out itself
        flat = eng.grads.clone()
        grads = tuple(eng.layout.views(flat).values())
        return (None, gx, gk) + grads


class AsymMPAEC(nn.Module):
    """nn_proc.py:264-340: Analysis -> (mag, phase) autoencoders -> Synthesis, knob conditioned."""

    def __init__(self, expected_time_frames, ft_size=1024, hop_size=384, decomposition_rank=64, n_knobs=4, output_tf=None):
        super().__init__()
        _say("AsymMPAEC: expected_time_frames, ft_size, hop_size, decomposition_rank, n_knobs, output_tf = ",
             expected_time_frames, ft_size, hop_size, decomposition_rank, n_knobs, output_tf)
        if decomposition_rank != 64:
            raise NotImplementedError("signaltrain_amd: decomposition_rank is fixed at 64 (the reference never changes it)")
        self.output_tf = expected_time_frames if output_tf is None else output_tf
        self.expected_time_frames, self.ft_size, self.hop_size, self.n_knobs = expected_time_frames, ft_size, hop_size, n_knobs
        self.dft_analysis = Analysis(ft_size=ft_size, hop_size=hop_size)
        self.dft_synthesis = Synthesis(ft_size=ft_size, hop_size=hop_size)
        self.aenc = AsymAutoEncoder(T=expected_time_frames, R=decomposition_rank, K=n_knobs, OT=self.output_tf)
        self.phs_aenc = AsymAutoEncoder(T=expected_time_frames, R=decomposition_rank, K=n_knobs, OT=self.output_tf)
        self._engine = None

    def reinitialize(self):
        self.aenc.initialize(); self.phs_aenc.initialize()

    def clip_grad_norm_(self):                                       # nn_proc.py:299-302
        torch.nn.utils.clip_grad_norm_(list(self.dft_analysis.parameters()) + list(self.dft_synthesis.parameters()),
                                       max_norm=1., norm_type=1)

    # ------------------------------------------------------------------ engine plumbing
    def _ordered_params(self):
        sd = dict(self.named_parameters(prefix="mpaec"))
        return [sd[k] for k in param_names()]

    def set_compute_dtype(self, dtype):
        """'f32' (default), 'bf16' / 'f16' (16-bit operands, fp32 accumulation in the STFT GEMMs) or 'bf16_all' / 'f16_all' (also in
        the nine Linear layers of both autoencoders) -- StepEngine.compute_dtype."""
        if dtype not in _lib.PREC:
            raise ValueError(f"compute dtype must be one of {tuple(_lib.PREC)}")
        self.compute_dtype = dtype
        if self._engine is not None:
            self._engine.set_arithmetic(dtype)

    def _ensure_engine(self, x):
        """Flatten the 40 parameters into the engine's buffer (parameters become views of it)."""
        B = x.shape[0]
        eng = self._engine
        ps = self._ordered_params()
        xdev = x.device if x.device.index is not None else torch.device("cuda", torch.cuda.current_device())
        if eng is not None and eng.device == xdev and B <= eng.max_batch and \
                all(p.data_ptr() == v.data_ptr() for p, v in zip(ps, eng.named.values())):
            return eng
        d = _lib.st_dims()
        d.B, d.L, d.N, d.H, d.T, d.OT, d.F, d.K = B, x.shape[1], self.ft_size, self.hop_size, self.expected_time_frames, \
            self.output_tf, self.ft_size // 2 + 1, self.n_knobs
        d.y = (d.OT - 1) * d.H - d.N
        new = StepEngine(d, x.device, max_batch=max(B, eng.max_batch if eng is not None else 0),
                         compute_dtype=getattr(self, "compute_dtype", "f32"))
        with torch.no_grad():
            for p, v in zip(ps, new.named.values()):
                v.copy_(p.detach().to(x.device).reshape(v.shape))
                p.data = v
        self._engine = new
        return new

    def forward(self, x_cuda, knobs_cuda, return_acts=False):
        if x_cuda.device.type != "cuda":
            raise RuntimeError("signaltrain_amd.st_model runs on a ROCm device only (no CPU fallback); move inputs to cuda")
        x = x_cuda.contiguous().float(); kn = knobs_cuda.contiguous().float()
        self._ensure_engine(x)
        y_hat, mag, mag_hat = _STModelFn.apply(self, x, kn, *self._ordered_params())
        if not return_acts:
            return y_hat, mag, mag_hat
        return y_hat, mag, mag_hat, self._acts(x, kn, y_hat, mag, mag_hat)

    def _acts(self, x, kn, y_hat, mag, mag_hat):
        """The 30-entry activation list of nn_proc.py:311-338 (diagnostics for utils/viz.py), from the HIP path's own buffers: re, im, phs,
        phs_hat, an_real, an_imag are views of the forward state the fused forward left in the engine's workspace (st_workspace_offsets),
        the 2 x 10 layer activations come from the library's diagnostic kernel (st_ae_acts)."""
        eng = self._engine
        with torch.no_grad():
            d = eng._dims(x.shape[0])
            offs = (C.c_int64 * 8)()
            _lib.check(eng.lib.st_workspace_offsets(C.byref(d), offs), "st_workspace_offsets")
            wsf = eng.ws.view(torch.float32)
            KP = int(eng.lib.st_kp(d.F))
            view = lambda i, *shape: wsf[offs[i]:offs[i] + int(np.prod(shape))].view(*shape).clone()
            re, im, phs = view(0, d.B, d.T, d.F), view(1, d.B, d.T, d.F), view(3, d.B, d.T, d.F)
            phs_hat = view(5, d.B, d.OT, d.F)
            AA = wsf[offs[6]:offs[6] + d.B * d.OT * KP].view(d.B, d.OT, KP)
            an_real, an_imag = AA[:, :, :d.F].clone(), AA[:, :, KP // 2:KP // 2 + d.F].clone()
            acts = [re, im, mag, phs]
            acts += self.aenc.acts_device(mag, kn, 'sf')
            acts += self.phs_aenc.acts_device(phs, kn, '')
            x_fwdsyn = y_hat / 2 - x[:, -y_hat.shape[1]:] / 2         # nn_proc.py:332,340 inverted: y_hat = 2 (x_fwdsyn + x_tail / 2)
            acts += [mag_hat, phs_hat, an_real, an_imag, x_fwdsyn, y_hat / 2]
        return acts


class st_model(nn.Module):
    """nn_proc.py:344-393: geometry wrapper around AsymMPAEC."""

    def __init__(self, scale_factor=1, shrink_factor=4, num_knobs=3, sr=44100, scale_scheme='lean'):
        super().__init__()
        chunk_size = int(8192 * scale_factor)                       # nn_proc.py:357
        out_chunk_size = int(chunk_size / shrink_factor)            # nn_proc.py:358
        self.scale_factor, self.shrink_factor = scale_factor, shrink_factor
        self.in_chunk_size, self.out_chunk_size = chunk_size, out_chunk_size
        self.num_knobs = num_knobs
        _say("Input chunk size =", chunk_size); _say("Intended Output chunk size =", out_chunk_size); _say("Sample rate =", sr)
        ft_size, hop_size = 1024, 384
        if scale_scheme != 'lean':                                  # nn_proc.py:374-376
            ft_size, hop_size = int(ft_size * scale_factor), int(hop_size * scale_factor)
        expected_time_frames = int(np.ceil(chunk_size / float(hop_size)) + np.ceil(ft_size / float(hop_size)))
        output_time_frames = int(np.ceil(out_chunk_size / float(hop_size)) + np.ceil(ft_size / float(hop_size)))
        y_size = (output_time_frames - 1) * hop_size - ft_size
        if y_size != out_chunk_size:
            _say(f"Warning: y_size ({y_size}) should equal out_chunk_size ({out_chunk_size})")
            _say(f"    Setting out_chunk_size = y_size = {y_size}")
        self.out_chunk_size = y_size
        self.mpaec = AsymMPAEC(expected_time_frames, ft_size=ft_size, hop_size=hop_size, n_knobs=num_knobs,
                               output_tf=output_time_frames)

    def clip_grad_norm_(self):
        self.mpaec.clip_grad_norm_()

    def set_compute_dtype(self, dtype):
        """Arithmetic of the accelerated path: 'f32' (default) | 'f32x3' | 'bf16' | 'bf16_all' | 'f16' | 'f16_all' (see AsymMPAEC.set_compute_dtype)."""
        self.mpaec.set_compute_dtype(dtype)

    def forward(self, x_cuda, knobs_cuda, return_acts=False):
        return self.mpaec.forward(x_cuda, knobs_cuda, return_acts=return_acts)

    def engine(self, example_x):
        """The fused StepEngine sharing this model's parameters (fast training path)."""
        return self.mpaec._ensure_engine(example_x)

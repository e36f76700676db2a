"""Fused train-step engine over libsignaltrain_hip.so.

Owns (as torch tensors -- PyTorch is the allocator, nothing more) the flat fp32 parameter /
gradient / Adam-moment buffers laid out by `st_param_offsets` and the workspace the C library
carves.  One engine = one GPU = one process; data parallelism (dp.py) all-reduces `grads`.

Reference mapping: one `train_step` call == the body of train_loop's minibatch iteration,
signaltrain/train.py:112-151 (forward, calc_loss, backward, clip_grad_norm_, Adam.step).
"""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib

AE_LAYERS = ("fnn_enc", "fnn_enc2", "fnn_enc3", "fnn_enc4", "fnn_addknobs",
             "fnn_dec4", "fnn_dec3", "fnn_dec2", "fnn_dec")          # nn_proc.py:64-65
STFT_KEYS = ("mpaec.dft_analysis.conv_analysis_real.weight",
             "mpaec.dft_analysis.conv_analysis_imag.weight",
             "mpaec.dft_synthesis.conv_synthesis_real.weight",
             "mpaec.dft_synthesis.conv_synthesis_imag.weight")
COMPUTE_DTYPES = tuple(_lib.PREC)       # "f32", "bf16", "bf16_all", "f16", "f16_all", "f32x3"


def param_names():
    """state_dict() key order of the reference st_model."""
    keys = list(STFT_KEYS)
    for ae in ("aenc", "phs_aenc"):
        for n in AE_LAYERS:
            keys += [f"mpaec.{ae}.{n}.weight", f"mpaec.{ae}.{n}.bias"]
    return keys


def param_shapes(d):
    R = 64
    ae = [(R, d.T), (R // 2, R), (R // 4, R // 2), (R // 4, R // 4), (R // 4, R // 4 + d.K),
          (R // 4, R // 4), (R // 2, R // 4), (R, R // 2), (d.OT, R)]           # nn_proc.py:46-61
    shapes = [(d.N, 1, d.N)] * 4
    for _ in range(2):
        for (o, i) in ae:
            shapes += [(o, i), (o,)]
    return shapes


class ParamLayout:
    """Names, shapes and float offsets of the 40 tensors inside the flat buffers."""

    def __init__(self, d):
        self.names = param_names()
        self.shapes = param_shapes(d)
        self.offsets, self.total = _lib.param_offsets(d)
        self.n_stft = self.offsets[4]

    def views(self, flat):
        out = OrderedDict()
        for n, s, o in zip(self.names, self.shapes, self.offsets):
            out[n] = flat[o:o + int(np.prod(s))].view(*s)
        return out


class _OptimizerView:
    """What misc.save_checkpoint needs from an optimizer: state_dict() in torch.optim.Adam's layout (train.py:228)."""

    def __init__(self, engine):
        self.engine = engine

    def state_dict(self):
        return self.engine.optimizer_state_dict()


def workspace_bytes_upto(lib, dims, max_batch):
    """Bytes of ONE workspace that serves every batch of 1 .. max_batch windows at every arithmetic level and clip scope: st_workspace_bytes_max (the exact
    st_workspace_bytes is a function of all three and is monotonic in none of them: INTEGRATION.md "Sizing the workspace")."""
    n = int(lib.st_workspace_bytes_max(C.byref(dims.with_batch(int(max_batch)))))
    if n <= 0:
        _lib.check(-1, "st_workspace_bytes_max")
    return n


class StepEngine:
    """Holds device state for one model replica and drives the HIP step."""

    def __init__(self, dims, device="cuda:0", max_batch=None, compute_dtype="f32", loss_scale=None, clip_all=None):
        """compute_dtype: "f32" (default, the parity path); "f32x3" = fp32-grade STFT GEMMs on the bf16 matrix pipe (operands as three
        bfloat16 planes, six partial products, fp32 accumulation; same tolerance as "f32"); "bf16" / "f16" = 16-bit operands with fp32
        accumulation in the STFT GEMMs (BASELINE configs[2], [3] / [4]); "bf16_all" / "f16_all" = also in the autoencoder layers.  Parameters, gradients,
        loss and Adam are fp32 in every mode.  loss_scale: static loss scale of the step (default: 1 except 2**12 for the f16
        modes -- Apex's amp.scale_loss, train.py:134-135); clip_all: L1 clip over all parameters instead of the STFT tensors
        (train.py:136, what the reference does with Apex on; default: only in the f16 modes)."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("signaltrain_amd.StepEngine needs a ROCm device (there is no CPU fallback)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._arith_checked = set()
        self.set_arithmetic(compute_dtype, loss_scale, clip_all)
        self.dims = dims
        self.max_batch = int(max_batch or dims.B)
        self.layout = ParamLayout(dims)
        n = self.layout.total
        z = lambda: torch.zeros(n, dtype=torch.float32, device=self.device)
        self.params, self.grads, self.m, self.v = z(), z(), z(), z()
        dmax = dims.with_batch(self.max_batch)
        # set_arithmetic() may change the mode of a live engine and the workspace depends on it (fp32 autoencoder layers keep their activations: 294 MB at B = 256;
        # the 16-bit modes carry operand copies): size it for the LARGEST mode, whatever arithmetic level `dims` happens to carry
        # ... and for every batch up to max_batch: the size is NOT monotonic in the batch (the split-K slab counts of the weight-gradient / synthesis GEMMs are picked per
        # batch: at the default geometry 585 windows need 85.6 MB more than 586, 178 windows 70 MB more than 179 at shrink 1) and an engine sized for its largest batch
        # serves smaller ones (a last partial batch, predict_long's remainder, validation): st_workspace_bytes_max, tens of ms of host arithmetic once per engine.
        nbytes = max(workspace_bytes_upto(self.lib, dims, self.max_batch), int(self.lib.st_workspace_bytes(C.byref(dmax))))
        self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.scalars = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.stage = None            # packed live analysis gradient rows (data parallel, allocated on first use)
        self.named = self.layout.views(self.params)
        self.named_grads = self.layout.views(self.grads)
        self.step_count = 0
        self.generation = 0          # bumped by every call that overwrites the saved-for-backward state in the workspace
        self.dp = None               # st_dp* of the library-owned RCCL communicator (dp.DataParallel attaches it)
        self._pending = None

    def set_arithmetic(self, compute_dtype, loss_scale=None, clip_all=None):
        if compute_dtype not in _lib.PREC:
            raise ValueError(f"compute_dtype must be one of {COMPUTE_DTYPES}")
        self.compute_dtype = compute_dtype
        half = compute_dtype.startswith("f16")
        self.loss_scale = float(loss_scale) if loss_scale is not None else (4096.0 if half else 1.0)
        self.clip_all = bool(half if clip_all is None else clip_all)

    # ---------------------------------------------------------------- parameters / optimizer state
    def load_state_dict(self, sd):
        for k, v in self.named.items():
            src = sd[k]
            src = torch.as_tensor(np.asarray(src)) if not torch.is_tensor(src) else src
            v.copy_(src.to(device=self.device, dtype=torch.float32).reshape(v.shape))

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.named.items())

    def optimizer_view(self):
        return _OptimizerView(self)

    def optimizer_state_dict(self, lr=None):
        """Adam state in torch.optim.Adam.state_dict() layout (40 per-parameter entries keyed by index, views of the flat
        moment buffers copied to the host) -- what the reference writes into its checkpoints (misc.py:28)."""
        from . import misc
        m, v = self.layout.views(self.m.detach().cpu()), self.layout.views(self.v.detach().cpu())
        return misc.adam_state_dict(self.step_count, self.lr if lr is None else lr,
                                    [t.clone() for t in m.values()], [t.clone() for t in v.values()])

    def load_optimizer_state_dict(self, osd):
        """Restore exp_avg / exp_avg_sq / step from a torch.optim.Adam state_dict (the reference's 'optimizer' checkpoint entry;
        the reference itself never reads it back, train.py:229).  Returns the restored learning rate, or None if `osd` is not
        in that layout (nothing is changed then)."""
        from . import misc
        flat = misc.flatten_optimizer_state(osd, self.layout.shapes)
        if flat is None:
            return None
        for buf, key in ((self.m, "exp_avg"), (self.v, "exp_avg_sq")):
            for view, src in zip(self.layout.views(buf).values(), flat[key]):
                view.copy_(torch.from_numpy(src).to(self.device).reshape(view.shape))
        self.step_count = int(flat["step"])
        self.lr = float(flat["lr"])
        return self.lr

    lr = 0.0        # last learning rate handed to train_step (recorded for optimizer_state_dict)

    # ---------------------------------------------------------------- plumbing
    def _dims(self, B):
        """The geometry for a batch of B windows + this engine's arithmetic: precision, loss scale and clip scope travel in
        st_dims with every call (there is no process-wide switch)."""
        if B > self.max_batch:
            raise RuntimeError(f"batch {B} exceeds the engine's workspace (max_batch={self.max_batch})")
        d = self.dims.with_batch(B)
        d.prec, d.loss_scale, d.clip_all = _lib.PREC[self.compute_dtype], self.loss_scale, int(self.clip_all)
        if (B, self.compute_dtype) not in self._arith_checked:
            # no silent arithmetic switch: where this geometry / batch cannot take the requested level (st_effective_prec, e.g. an odd batch
            # on the wide autoencoder path) say so once per (batch, dtype) and keep it readable in effective_dtype(B)
            self._arith_checked.add((B, self.compute_dtype))
            eff = int(self.lib.st_effective_prec(C.byref(d)))
            if eff != d.prec:
                import warnings
                name = {v: k for k, v in _lib.PREC.items()}.get(eff, str(eff))
                warnings.warn(f"signaltrain_amd: compute_dtype={self.compute_dtype!r} with batch {B} at this geometry (T={d.T}, OT={d.OT}) runs the "
                              f"autoencoder layers in fp32 (effective arithmetic {name!r}): the wide autoencoder path needs an even batch for 16-bit "
                              f"Linear layers", RuntimeWarning, stacklevel=3)
        return d

    def effective_dtype(self, B=None):
        """The arithmetic a batch of B windows actually runs in (st_effective_prec): compute_dtype, except where the geometry cannot take it."""
        d = self.dims.with_batch(self.max_batch if B is None else B)
        d.prec, d.loss_scale, d.clip_all = _lib.PREC[self.compute_dtype], self.loss_scale, int(self.clip_all)
        eff = int(self.lib.st_effective_prec(C.byref(d)))
        return {v: k for k, v in _lib.PREC.items()}.get(eff, str(eff))

    def _stream(self):
        """The current stream OF THIS ENGINE'S DEVICE (not of whatever device happens to be current)."""
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _call(self, name, *args):
        """One C-ABI call with this engine's device current: the library launches on the device that is current at call time
        (it never calls hipSetDevice itself), so an engine on cuda:1 must not depend on the caller's current device."""
        with torch.cuda.device(self.device):
            _lib.check(getattr(self.lib, name)(*args), name)

    def _prep(self, x, knobs, y=None):
        f = lambda t: None if t is None else t.to(device=self.device, dtype=torch.float32).contiguous()
        x, knobs, y = f(x), f(knobs), f(y)
        d = self._dims(x.shape[0])
        assert x.shape == (d.B, d.L) and knobs.shape == (d.B, d.K), (x.shape, knobs.shape, d.as_dict())
        if y is not None:
            assert y.shape == (d.B, d.y), (y.shape, d.y)
        if d.K == 0:
            knobs = None      # a model WITHOUT knobs (nn_proc.py:92-93 concatenates an empty [B, 0] tensor): an empty tensor has no data pointer; the C ABI takes NULL for K == 0
        return d, x, knobs, y

    # ---------------------------------------------------------------- forward / backward / step
    def forward(self, x, knobs, save_for_backward=False):
        """st_model.forward (nn_proc.py:392): returns (y_hat[B,y], mag[B,T,F], mag_hat[B,OT,F])."""
        d, x, knobs, _ = self._prep(x, knobs)
        y_hat = torch.empty(d.B, d.y, dtype=torch.float32, device=self.device)
        mag = torch.empty(d.B, d.T, d.F, dtype=torch.float32, device=self.device)
        mag_hat = torch.empty(d.B, d.OT, d.F, dtype=torch.float32, device=self.device)
        self.generation += 1
        self._call("st_model_fwd", C.byref(d), _lib.ptr(self.params), _lib.ptr(x), _lib.ptr(knobs),
                   _lib.ptr(y_hat), _lib.ptr(mag), _lib.ptr(mag_hat), _lib.ptr(self.ws), 1 if save_for_backward else 0, self._stream())
        return y_hat, mag, mag_hat

    def knob_grad(self, x, knobs, g_y_hat, g_mag_hat=None, g_mag=None):
        """d / d knobs [B, K] for the upstream gradients of backward() (st_model_knob_grad; nn_proc.py:92-93 under autograd).  The exact, slow route:
        one forward + backward per window.  Overwrites the workspace's saved-for-backward state (call forward(save_for_backward=True) again before
        backward()); the parameter gradients of these passes go to a scratch buffer, self.grads is left alone."""
        d, x, knobs, _ = self._prep(x, knobs)
        if d.K == 0:
            return torch.empty(d.B, 0, dtype=torch.float32, device=self.device)
        d.loss_scale = 0.0
        f = lambda t: None if t is None else t.to(device=self.device, dtype=torch.float32).contiguous()
        g_y_hat, g_mag_hat, g_mag = f(g_y_hat), f(g_mag_hat), f(g_mag)
        if getattr(self, "_knob_scratch", None) is None:
            self._knob_scratch = torch.zeros_like(self.grads)
        out = torch.empty(d.B, d.K, dtype=torch.float32, device=self.device)
        self.generation += 1
        self._call("st_model_knob_grad", C.byref(d), _lib.ptr(self.params), _lib.ptr(self._knob_scratch), _lib.ptr(x), _lib.ptr(knobs),
                   _lib.ptr(g_y_hat), _lib.ptr(g_mag_hat), _lib.ptr(g_mag), _lib.ptr(self.ws), _lib.ptr(out), self._stream())
        return out

    def backward(self, x, knobs, g_y_hat, g_mag_hat=None, g_mag=None):
        """Autograd backward for arbitrary upstream gradients (after forward(save_for_backward=True))."""
        d, x, knobs, _ = self._prep(x, knobs)
        d.loss_scale = 0.0                 # the upstream gradient is whatever autograd hands over: no loss scale of ours in it
        f = lambda t: None if t is None else t.to(device=self.device, dtype=torch.float32).contiguous()
        g_y_hat, g_mag_hat, g_mag = f(g_y_hat), f(g_mag_hat), f(g_mag)
        self._call("st_model_bwd", C.byref(d), _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(x),
                   _lib.ptr(knobs), _lib.ptr(g_y_hat), _lib.ptr(g_mag_hat), _lib.ptr(g_mag), _lib.ptr(self.ws), self._stream())
        return self.grads

    def input_grad(self, x, g_y_hat):
        """d loss / d x of the whole model (right after backward() on the same batch): half the conv-transpose of the analysis output
        gradient (nn_proc.py:307 feeds x/2) plus the skip connection's share on the last y samples (nn_proc.py:340)."""
        d = self._dims(x.shape[0])
        scratch = torch.empty(self.lib.st_model_input_grad_ws_floats(C.byref(d)), dtype=torch.float32, device=self.device)
        gx = torch.empty(d.B, d.L, dtype=torch.float32, device=self.device)
        self._call("st_model_input_grad", C.byref(d), _lib.ptr(self.params), _lib.ptr(self.ws), _lib.ptr(scratch), _lib.ptr(gx), self._stream())
        gx.mul_(0.5)
        if g_y_hat is not None:
            gx[:, d.L - d.y:] += g_y_hat.to(device=self.device, dtype=torch.float32)
        return gx

    def loss_backward(self, x, knobs, y, want_outputs=False):
        """forward + calc_loss (loss_functions.py:26-36 with scale_by_freq) + backward; fills self.grads (times the loss
        scale, if one is set).  self.scalars[0..4] = loss, mean log-cosh, L1 term, L1 norm of the (unscaled) STFT grads, clip coefficient."""
        d, x, knobs, y = self._prep(x, knobs, y)
        self._pending = (d, x, knobs, y)   # the dims clip_adam() carves the workspace with are those of the LAST backward
        outs = (None, None, None)
        if want_outputs:
            outs = (torch.empty(d.B, d.y, dtype=torch.float32, device=self.device),
                    torch.empty(d.B, d.T, d.F, dtype=torch.float32, device=self.device),
                    torch.empty(d.B, d.OT, d.F, dtype=torch.float32, device=self.device))
        self.generation += 1
        self._call("st_loss_backward", C.byref(d), _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(x),
                   _lib.ptr(knobs), _lib.ptr(y), _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(outs[2]), _lib.ptr(self.ws),
                   _lib.ptr(self.scalars), self._stream())
        return outs

    def _stage_buf(self):
        if self.stage is None:
            self.stage = torch.zeros(2 * self.dims.F * self.dims.N, dtype=torch.float32, device=self.device)
        return self.stage

    def loss_backward_p1(self, x, knobs, y):
        """Forward + backward up to (excluding) the analysis weight gradient; see dp.DataParallel."""
        d, x, knobs, y = self._prep(x, knobs, y)
        self._pending = (d, x, knobs, y)
        self.generation += 1
        self._call("st_loss_backward_p1", C.byref(d), _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(x),
                   _lib.ptr(knobs), _lib.ptr(y), _lib.ptr(self.ws), self._stream())

    def loss_backward_p2(self):
        """Analysis weight gradient; the 2F live rows also land packed in self.stage (see grad_buckets)."""
        d, x = self._pending[:2]
        self._call("st_loss_backward_p2_staged", C.byref(d), _lib.ptr(self.grads), _lib.ptr(self._stage_buf()), _lib.ptr(x),
                   _lib.ptr(self.ws), _lib.ptr(self.scalars), self._stream())

    def finish_buckets(self):
        """After the all-reduce of grad_buckets(): copy the reduced analysis rows from the packed staging buffer back into grads."""
        self._call("st_unstage_analysis", C.byref(self._pending[0]), _lib.ptr(self.grads), _lib.ptr(self.stage), self._stream())

    N_STAGES = 4

    def loss_backward_stage(self, stage, x=None, knobs=None, y=None):
        """Stage `stage` (0..3, in order) of forward + loss + backward; stage 0 takes the minibatch.  After stage s the
        range stage_bucket(s) of self.grads is final (st_loss_backward_stage in include/signaltrain_hip.h)."""
        if stage == 0:
            self._pending = self._prep(x, knobs, y)
            self.generation += 1
        d, x, knobs, y = self._pending
        self._call("st_loss_backward_stage", C.byref(d), _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(x),
                   _lib.ptr(knobs), _lib.ptr(y), _lib.ptr(self.ws), _lib.ptr(self.scalars), int(stage), self._stream())

    def stage_bucket(self, stage):
        """Gradient range that is final after loss_backward_stage(stage): synthesis bases (4.2 MB at N=1024), both
        autoencoders (67 KB), live rows [0,F) of the real analysis basis, live rows of the imaginary one (2.1 MB each)."""
        o, d = self.layout.offsets, self.dims
        live = d.F * d.N
        return (self.grads[o[2]:o[4]], self.grads[o[4]:], self.grads[o[0]:o[0] + live], self.grads[o[1]:o[1] + live])[stage]

    def grad_buckets(self):
        """Two-phase form (loss_backward_p1 / _p2): the buffers to all-reduce, in the order they become final --
        [synthesis + autoencoders] = grads[offs[2]:] after phase 1, then the packed staging copy [2F][N] of the live analysis
        rows written by phase 2 (4.2 MB; the contiguous range of grads holding them would span 2 MB of structurally-zero rows).
        Call finish_buckets() after the second all-reduce."""
        return [self.grads[self.layout.offsets[2]:], self._stage_buf()]

    def train_step(self, x, knobs, y, lr, betas=(0.9, 0.999), eps=1e-8):
        """One optimisation step (train.py:112-151).  `lr` is the value sitting in param_groups at step
        time, i.e. lr_sched[max(i-1,0)] in the reference loop (train.py:150).  No host sync."""
        d, x, knobs, y = self._prep(x, knobs, y)
        self._pending = None               # a later clip_adam() must not carve the workspace with an older backward's dims
        self.step_count += 1; self.generation += 1; self.lr = float(lr)
        self._call("st_train_step", C.byref(d), _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.m),
                   _lib.ptr(self.v), _lib.ptr(x), _lib.ptr(knobs), _lib.ptr(y), _lib.ptr(self.ws),
                   _lib.ptr(self.scalars), float(lr), float(betas[0]), float(betas[1]), float(eps),
                   int(self.step_count), self._stream())
        return self.scalars

    # ---------------------------------------------------------------- the step as one HIP graph
    def graph_capture(self, batch, lr_table, betas=(0.9, 0.999), eps=1e-8):
        """Capture st_train_step for `batch` windows into a HIP graph (st_graph_create).  The step counter and the learning rate
        live on the device (scalars[6], scalars[7]; `lr_table` = the 1-cycle table, uploaded once): graph_step() then needs no
        per-step arguments.  The minibatch is read from the fixed buffers self.gx / self.gk / self.gy."""
        d = self._dims(int(batch))
        dev = self.device
        self.gx = torch.zeros(d.B, d.L, dtype=torch.float32, device=dev)
        self.gk = torch.zeros(d.B, d.K, dtype=torch.float32, device=dev) if d.K else None      # K = 0: NULL (see _prep)
        self.gy = torch.zeros(d.B, d.y, dtype=torch.float32, device=dev)
        self.g_lr = torch.as_tensor(np.asarray(lr_table, dtype=np.float32), device=dev).contiguous()
        self.scalars[6] = float(self.step_count)
        self._graph_dims = d
        handle = C.c_void_p()
        side = torch.cuda.Stream(device=dev)             # capture needs a non-default stream
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.device(dev):
            _lib.check(self.lib.st_graph_create(C.byref(d), _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.m), _lib.ptr(self.v),
                                                _lib.ptr(self.gx), _lib.ptr(self.gk), _lib.ptr(self.gy), _lib.ptr(self.ws), _lib.ptr(self.scalars),
                                                _lib.ptr(self.g_lr), int(self.g_lr.numel()), float(betas[0]), float(betas[1]), float(eps),
                                                C.c_void_p(side.cuda_stream), C.byref(handle)), "st_graph_create")
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = handle
        return handle

    def graph_step(self, x=None, knobs=None, y=None):
        """One optimisation step = one hipGraphLaunch on the current stream.  x / knobs / y (optional) are copied into the
        graph's input buffers first; with all three None the buffers are used as they are (device-resident data)."""
        if x is not None:
            self.gx.copy_(x); self.gy.copy_(y)
            if self.dims.K: self.gk.copy_(knobs)
        self.step_count += 1; self.generation += 1
        self._call("st_graph_launch", self.graph, self._stream())
        return self.scalars

    def graph_destroy(self):
        if getattr(self, "graph", None) is not None:
            self._call("st_graph_destroy", self.graph); self.graph = None

    def dp_train_step(self, x, knobs, y, lr, betas=(0.9, 0.999), eps=1e-8, force_exchange=False, split_last=False, pack16=False):
        """The data-parallel step driven from C (st_dp_train_step): this rank's shard, both gradient buckets all-reduced on the
        library's RCCL communicator under the backward, clip after the reduction, replicated Adam.  split_last: the last exchange by basis
        (only 2.1 MB exposed); pack16: that exchange on bfloat16 values (the *_all arithmetic modes only)."""
        d, x, knobs, y = self._prep(x, knobs, y)
        self.step_count += 1; self.generation += 1; self.lr = float(lr)
        self._call("st_dp_train_step", self.dp, C.byref(d), _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.m),
                   _lib.ptr(self.v), _lib.ptr(self._stage_buf()), _lib.ptr(x), _lib.ptr(knobs), _lib.ptr(y), _lib.ptr(self.ws),
                   _lib.ptr(self.scalars), float(lr), float(betas[0]), float(betas[1]), float(eps),
                   int(self.step_count), (1 if force_exchange else 0) | (2 if split_last else 0) | (4 if pack16 else 0), self._stream())
        return self.scalars

    def clip_adam(self, lr, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8):
        """L1 clip + Adam on the current self.grads (data parallel: call after the all-reduce with
        grad_scale = 1/world; the norm is recomputed on the reduced gradient, identically on all ranks)."""
        self.step_count += 1; self.lr = float(lr)
        d = self._pending[0] if self._pending is not None else self._dims(self.dims.B)     # the STEP's dims: its partial-sum counts carve the workspace
        self._call("st_dp_clip_adam", C.byref(d), _lib.ptr(self.params), _lib.ptr(self.grads),
                   _lib.ptr(self.m), _lib.ptr(self.v), _lib.ptr(self.ws), _lib.ptr(self.scalars),
                   float(grad_scale), float(lr), float(betas[0]), float(betas[1]), float(eps),
                   int(self.step_count), self._stream())
        return self.scalars

    def loss(self):
        """Host copy of the last loss (device->host sync; the reference does this every 10 iterations)."""
        return float(self.scalars[0].item())

    def overflow_steps(self, reset=True):
        """Steps skipped so far because a gradient overflowed under the loss scale (scalars[5]; device->host sync).  The
        caller's loss-scale policy (train.train halves the scale, like Apex's dynamic scaler) acts on it."""
        n = int(self.scalars[5].item())
        if reset and n:
            self.scalars[5] = 0.0
        return n

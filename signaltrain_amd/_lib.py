"""ctypes binding of libsignaltrain_hip.so (C ABI declared in include/signaltrain_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  PyTorch is used only for device memory and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ST_LIB_PATH") or os.path.join(_HERE, "libsignaltrain_hip.so")     # ST_LIB_PATH: diagnostics (ablation builds, tools/ae_ablate.sh)


class st_dims(C.Structure):
    """Mirror of `struct st_dims` (include/signaltrain_hip.h)."""
    _fields_ = [(n, C.c_int) for n in ("B", "L", "N", "H", "T", "OT", "F", "K", "y")] + \
               [("prec", C.c_int), ("loss_scale", C.c_float), ("clip_all", C.c_int)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}

    def with_arith(self, prec=None, loss_scale=None, clip_all=None):
        d = st_dims(); C.memmove(C.byref(d), C.byref(self), C.sizeof(st_dims))
        if prec is not None: d.prec = int(prec)
        if loss_scale is not None: d.loss_scale = float(loss_scale)
        if clip_all is not None: d.clip_all = int(bool(clip_all))
        return d

    def with_batch(self, B):
        d = st_dims(); C.memmove(C.byref(d), C.byref(self), C.sizeof(st_dims)); d.B = int(B); return d


_p = C.c_void_p
_f = C.c_float
_i = C.c_int
_D = C.POINTER(st_dims)

# name -> (restype, argtypes)   -- must list every symbol declared in include/signaltrain_hip.h
SIGNATURES = {
    "st_kp": (_i, [_i]),
    "st_last_error": (C.c_char_p, []),
    "st_version": (_i, []),
    "st_set_tuning": (_i, [_i]),
    "st_get_tuning": (_i, [C.POINTER(C.c_int), _i]),
    "st_tuning_defaults": (_i, [C.POINTER(C.c_int), _i]),
    "st_reset_tuning": (_i, []),
    "st_effective_prec": (_i, [_D]),
    "st_set_debug": (_i, [_i]),
    "st_debug_read_stage_cycles": (_i, [C.POINTER(C.c_uint64)]),
    "st_profile_enable": (_i, [_i]),
    "st_profile_report": (_i, [C.c_char_p, _i]),
    "st_geometry": (_i, [C.c_double, C.c_double, _i, _i, _i, _D]),
    "st_param_offsets": (C.c_int64, [_D, C.POINTER(C.c_int64)]),
    "st_workspace_bytes": (C.c_size_t, [_D]),
    "st_workspace_bytes_max": (C.c_size_t, [_D]),
    "st_analysis_fwd": (_i, [_D, _p, _p, _p, _f, _p, _p, _p, _p, _p]),
    "st_ae_fwd": (_i, [_D, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "st_ae_fwd_ws_floats": (C.c_size_t, [_D]),
    "st_workspace_offsets": (_i, [_D, C.POINTER(C.c_int64)]),
    "st_ae_acts_floats": (C.c_size_t, [_D]),
    "st_ae_acts": (_i, [_D, _p, _p, _p, _i, _p, _p]),
    "st_compressor_4c": (_i, [_p, _p, C.c_float, C.c_int, C.c_int, C.c_int, _p, _p]),
    "st_synth_comp4c_scratch_floats": (C.c_size_t, [_i, _i]),
    "st_synth_comp4c": (_i, [C.c_uint, C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float),
                             C.c_int, C.c_int, _p, _p, _p, _p, _p, _p]),
    "st_fe_frames": (_i, [C.c_int] * 4),
    "st_fe_ws_floats": (C.c_size_t, [C.c_int] * 6),
    "st_fe_analysis_fwd": (_i, [_p, C.c_int, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p]),
    "st_fe_synthesis_fwd": (_i, [_p, C.c_int, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "st_model_input_grad_ws_floats": (C.c_size_t, [_D]),
    "st_model_input_grad": (_i, [_D, _p, _p, _p, _p, _p]),
    "st_fe_analysis_bwd": (_i, [_p, C.c_int, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p, _p, _p]),
    "st_fe_synthesis_bwd": (_i, [_p, C.c_int, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p, _p]),
    "st_ae_fwd_partials": (_i, [_D]),
    "st_synth_slabs": (_i, [_D]),
    "st_synth_frame_slabs": (_i, [_D]),
    "st_nt128_worklist": (_i, [_D, _i, _i, C.POINTER(C.c_uint), _i, C.POINTER(C.c_int)]),
    "st_fm_div_exact": (_i, [_i, _i]),
    "st_synth_fold": (_i, [_D, _p, _p, _p, _p]),
    "st_synthesis_frames": (_i, [_D, _p, _p, _p, _p]),
    "st_ola_loss": (_i, [_D, _p, _p, _p, _p, _p, _p, _p]),
    "st_ola_loss_partials": (_i, [_D]),
    "st_synthesis_dgrad": (_i, [_D, _p, _p, _p, _p]),
    "st_synthesis_wgrad": (_i, [_D, _p, _p, _p, _p, _p, _p, _p]),
    "st_ae_bwd": (_i, [_D, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p]),
    "st_ae_bwd_ws_floats": (C.c_size_t, [_D]),
    "st_ae_kept_activation_bytes": (C.c_size_t, [_D]),
    "st_polar_bwd": (_i, [_D, _p, _p, _p, _p, _p, _p, _p]),
    "st_analysis_wgrad": (_i, [_D, _p, _p, _f, _p, _p, _p, _p, _p]),
    "st_wgrad_ws_floats": (C.c_size_t, [_D]),
    "st_norm_partials": (_i, [_D]),
    "st_finalize_scalars": (_i, [_D, _p, _p, _p, _p, _f, _p, _p]),
    "st_clip_adam": (_i, [_p, _p, _p, _p, C.c_int64, C.c_int64, _p, _f, _f, _f, _f, _f, _i, _p]),
    "st_model_fwd": (_i, [_D, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "st_model_bwd": (_i, [_D, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "st_model_knob_grad": (_i, [_D, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "st_loss_backward": (_i, [_D, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "st_loss_backward_p1": (_i, [_D, _p, _p, _p, _p, _p, _p, _p]),
    "st_loss_backward_p2": (_i, [_D, _p, _p, _p, _p, _p]),
    "st_loss_backward_p2_staged": (_i, [_D, _p, _p, _p, _p, _p, _p]),
    "st_unstage_analysis": (_i, [_D, _p, _p, _p]),
    "st_loss_backward_stage": (_i, [_D, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "st_train_step": (_i, [_D, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _f, _f, _i, _p]),
    "st_dp_clip_adam": (_i, [_D, _p, _p, _p, _p, _p, _p, _f, _f, _f, _f, _f, _i, _p]),
    "st_graph_create": (_i, [_D, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _f, _f, _f, _p, C.POINTER(_p)]),
    "st_graph_launch": (_i, [_p, _p]),
    "st_graph_destroy": (_i, [_p]),
    "st_dp_unique_id": (_i, [_p]),
    "st_dp_init": (_i, [_p, _i, _i, C.POINTER(_p)]),
    "st_dp_destroy": (_i, [_p]),
    "st_dp_rank": (_i, [_p]),
    "st_dp_world": (_i, [_p]),
    "st_dp_rccl_version": (_i, [_p]),
    "st_dp_allreduce": (_i, [_p, _p, C.c_int64, _p]),
    "st_dp_broadcast": (_i, [_p, _p, C.c_int64, _i, _p]),
    "st_dp_sync": (_i, [_p, _p]),
    "st_dp_train_step": (_i, [_p, _D, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _f, _f, _i, _i, _p]),
}

# st_dims.prec levels (include/signaltrain_hip.h ST_PREC_*) by the names the Python surface uses
PREC = {"f32": 0, "bf16": 1, "bf16_all": 2, "f16": 3, "f16_all": 4, "f32x3": 5}

_lib = None


def load():
    """Load the HIP library; raises RuntimeError (never falls back) if it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"signaltrain_amd: {LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C signaltrain_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().st_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"signaltrain_hip {what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device (or host) pointer of a contiguous float32 torch tensor; None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "signaltrain_amd: tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


def on_arg_device(fn):
    """Decorator for the static forward / backward of the autograd Functions: run with the device of the first tensor argument
    current, so that torch.cuda.current_stream() inside is THAT device's stream and the library (which launches on the current
    device) works for modules living on cuda:1 while cuda:0 is current."""
    import functools
    import torch

    @functools.wraps(fn)
    def wrapped(ctx, first, *rest):
        if torch.is_tensor(first) and first.is_cuda:
            with torch.cuda.device(first.device):
                return fn(ctx, first, *rest)
        return fn(ctx, first, *rest)
    return wrapped


def geometry(scale_factor=1, shrink_factor=4, num_knobs=4, batch=1, scale_scheme="lean"):
    d = st_dims()
    check(load().st_geometry(float(scale_factor), float(shrink_factor), 0 if scale_scheme == "lean" else 1,
                             int(num_knobs), int(batch), C.byref(d)), "st_geometry")
    return d


def param_offsets(d):
    offs = (C.c_int64 * 40)()
    total = load().st_param_offsets(C.byref(d), offs)
    if total < 0:
        check(-1, "st_param_offsets")
    return list(offs), int(total)

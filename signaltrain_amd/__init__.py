"""signaltrain_amd -- MI355X-native SignalTrain training hot path (hand-written HIP behind a C ABI).

Mirrors the reference's Python surface for the hot path (signaltrain.nn_proc.st_model,
loss_functions.calc_loss, train.train); everything runs in libsignaltrain_hip.so on gfx950.
There is no CPU fallback: importing is cheap, but any compute call without the built library
and a ROCm device raises.
"""
__version__ = "0.1.0"
from . import _lib                      # noqa: F401
from .engine import StepEngine, ParamLayout, param_names   # noqa: F401

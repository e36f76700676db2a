"""Import the reference (/root/reference/signaltrain) in-process, read-only -- BUILD CONTAINER ONLY.

Shared by the golden-capture tools.  Applies the monkeypatches of SURVEY.md 8c (scipy window names, has_cudnn,
numba / librosa stubs) and returns the reference's modules.  Nothing from the reference is copied into the repo and
nothing here runs on the GPU box.
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

REF = "/root/reference/signaltrain"


def import_reference():
    import scipy.signal
    import scipy.signal.windows
    import torch
    scipy.signal.hamming = scipy.signal.windows.hamming
    scipy.signal.cosine = scipy.signal.windows.cosine
    torch.has_cudnn = False                       # True on this ROCm build even without a GPU: initialize() would call .cuda()
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")

        def _jit(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        nb.jit = _jit
        sys.modules["numba"] = nb
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import nn_proc, loss_functions, learningrate, misc          # noqa: E401  (the reference's modules)
    import audio as ref_audio
    return types.SimpleNamespace(nn_proc=nn_proc, loss_functions=loss_functions, learningrate=learningrate,
                                 misc=misc, audio=ref_audio)

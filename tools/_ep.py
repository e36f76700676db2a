import sys, time; sys.path.insert(0, '.')
import torch
from tests import gpu_checks as G
from signaltrain_amd.engine import StepEngine
from signaltrain_amd import _lib
lib = _lib.load()
for B in (2, 256):
    geo, X, Y, KN, P = G.make_case(B, 3, K=4)
    d = G.dims_of(geo, B, 4)
    eng = StepEngine(d, G.DEV); eng.load_state_dict(P)
    x, k, y = (torch.from_numpy(a).to(G.DEV) for a in (X, KN, Y))
    def run(n=60):
        for _ in range(10): eng.loss_backward(x, k, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): eng.loss_backward(x, k, y)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    lib.st_set_debug(0); a = run(); lib.st_set_debug(512); b = run(); lib.st_set_debug(0)
    print(f"B={B}: loss_backward {a:.1f} us, without the ae_bwd flush+copy {b:.1f} us -> epilogue {a-b:.1f} us")

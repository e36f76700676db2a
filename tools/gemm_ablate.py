#!/usr/bin/env python3
"""[needs a library built with -DST_GEMM_ABLATE: `make -C signaltrain_amd/csrc OUT=/path/libst_ablate.so EXTRA=-DST_GEMM_ABLATE`, then
ST_LIB_PATH=/path/libst_ablate.so python this_tool; with the default build the switches are compiled out and every line shows the same time]
Timing-only ablation of the analysis GEMM (results invalid under the debug switches)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from signaltrain_amd import _lib
lib = _lib.load()
B = 256
d = _lib.geometry(1, 4, 4, B)
dev = "cuda:0"
x = torch.randn(B, d.L, device=dev); Wr = torch.randn(d.N, d.N, device=dev) * 0.03; Wi = torch.randn(d.N, d.N, device=dev) * 0.03
re = torch.empty(B, d.T, d.F, device=dev); im = torch.empty_like(re); mag = torch.empty_like(re); phs = torch.empty_like(re)
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    lib.st_analysis_fwd(C.byref(d), _lib.ptr(x), _lib.ptr(Wr), _lib.ptr(Wi), 0.5, _lib.ptr(re), _lib.ptr(im), _lib.ptr(mag), _lib.ptr(phs), S())
def t(n=20):
    for _ in range(3): run()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for bk in (32, 16):
    lib.st_set_tuning(bk)
    for dbg, name in ((0, "full"), (1, "no loads/stores in loop"), (2, "no barriers"), (3, "no loads, no barriers (MFMA + LDS reads)"), (4, "no MFMA (data movement only)"), (7, "empty loop")):
        lib.st_set_debug(dbg)
        print(f"bk={bk} dbg={dbg} {name:44s} {t():8.1f} us (incl. ~7us dead-frame zeroing)")
lib.st_set_debug(0)

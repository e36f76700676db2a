#!/usr/bin/env python3
"""Verbose GPU diagnostic: per-stage parity report + per-kernel timings.  Run on the GPU box:
    python tools/gpu_diag.py [--perf-batch 256]
Writes gpurun_out/diag.txt (parity) and gpurun_out/perf.txt (timings)."""
import argparse, ctypes as C, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--perf-batch", type=int, default=256)
ap.add_argument("--skip-parity", action="store_true")
args = ap.parse_args()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", "diag.txt"), "w")


def out(s=""):
    print(s, flush=True); log.write(s + "\n"); log.flush()


from tests import gpu_checks as G
from signaltrain_amd import _lib
from signaltrain_amd.engine import StepEngine

out(f"device: {torch.cuda.get_device_name(0)}  CUs={torch.cuda.get_device_properties(0).multi_processor_count}")
nbad = 0
if not args.skip_parity:
    for (fn, kw) in ((G.run_all, dict(B=3, seed=0)), (G.run_all, dict(B=5, seed=2, K=3)), (G.run_fused, dict(B=3, seed=1)),
                     (G.run_fused, dict(B=7, seed=4, K=2, steps=2))):
        out(f"==== {fn.__name__} {kw}")
        try:
            nbad += G.report(fn(**kw), out)
        except Exception:
            out("EXCEPTION:\n" + traceback.format_exc()); nbad += 1
    out(f"TOTAL BAD: {nbad}")

# ------------------------------------------------------------------------------------ timings
B = args.perf_batch
geo, X, Y, KN, P = G.make_case(8, 3)
rng = np.random.default_rng(0)
reps = (B + 7) // 8
X = np.tile(X, (reps, 1))[:B] * rng.uniform(0.5, 1.0, (B, 1)).astype(np.float32)
Y = np.tile(Y, (reps, 1))[:B]; KN = np.tile(KN, (reps, 1))[:B]
d = G.dims_of(geo, B, 4)
eng = StepEngine(d, G.DEV); eng.load_state_dict(P)
x, kn, y = G.t(X), G.t(KN), G.t(Y)
perf = open(os.path.join(ROOT, "gpurun_out", "perf.txt"), "w")


def timeit(name, f, iters=20, warm=3, flops=None):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    s = f"{name:28s} {ms*1e3:10.1f} us" + (f"  {flops/ms/1e9:8.2f} TFLOP/s" if flops else "")
    print(s, flush=True); perf.write(s + "\n"); perf.flush()
    return ms


lib = eng.lib
from signaltrain_amd.engine import STFT_KEYS
V = eng.named; lay = eng.layout
ws = eng.ws
z = lambda *s: torch.zeros(*s, device=G.DEV)
T, OT, F, N, KP = d.T, d.OT, d.F, d.N, lib.st_kp(d.F)
re, im, mag, phs = z(B, T, F), z(B, T, F), z(B, T, F), z(B, T, F)
nsl = lib.st_synth_slabs(C.byref(d))
mag_hat, phs_hat, AA, dAA = z(B, OT, F), z(B, OT, F), z(B * OT, KP), z(nsl, B * OT, KP)
Sfold, frs = z(KP, N), z(lib.st_synth_frame_slabs(C.byref(d)), B * OT, N)
y_hat, dsyn = z(B, d.y), z(B, d.y)
regp, lp = z(lib.st_ae_fwd_partials(C.byref(d))), z(lib.st_ola_loss_partials(C.byref(d)))
wsg = z(lib.st_wgrad_ws_floats(C.byref(d))); aews = z(lib.st_ae_bwd_ws_floats(C.byref(d)))
gS = [z(N, N), z(N, N)]; gW = [z(N, N), z(N, N)]; npart = lib.st_norm_partials(C.byref(d)); na, ns = z(npart), z(npart)
dmag, dphs, dG = z(B, T, F), z(B, T, F), z(B * T, KP)
ae_m = eng.params[lay.offsets[4]:lay.offsets[22]]; ae_p = eng.params[lay.offsets[22]:]
PG = lay.offsets[22] - lay.offsets[4]
g_m, g_p = z(PG), z(PG)
p = _lib.ptr; D = C.byref(d); S = G.stream
MAC = 2.0
print(f"---- per-kernel timings at B={B}")
timeit("analysis_fwd(+polar)", lambda: lib.st_analysis_fwd(D, p(x), p(V[STFT_KEYS[0]]), p(V[STFT_KEYS[1]]), 0.5, p(re), p(im), p(mag), p(phs), S()),
       flops=MAC * B * T * 2 * F * N)
timeit("ae_fwd", lambda: lib.st_ae_fwd(D, p(mag), p(phs), p(kn), p(ae_m), p(ae_p), p(mag_hat), p(phs_hat), p(AA), p(regp), None, S()),
       flops=MAC * B * F * 2 * 8128)
timeit("synth_fold", lambda: lib.st_synth_fold(D, p(V[STFT_KEYS[2]]), p(V[STFT_KEYS[3]]), p(Sfold), S()))
timeit("synthesis_frames", lambda: lib.st_synthesis_frames(D, p(AA), p(Sfold), p(frs), S()), flops=MAC * B * OT * 2 * F * N)
timeit("ola_loss", lambda: lib.st_ola_loss(D, p(frs), p(x), p(y), p(y_hat), p(dsyn), p(lp), S()))
timeit("synthesis_dgrad", lambda: lib.st_synthesis_dgrad(D, p(dsyn), p(Sfold), p(dAA), S()), flops=MAC * B * OT * 2 * F * N)
timeit("synthesis_wgrad(+reduce)", lambda: lib.st_synthesis_wgrad(D, p(AA), p(dsyn), p(wsg), p(gS[0]), p(gS[1]), p(ns), S()), flops=MAC * B * OT * 2 * F * N)
timeit("ae_bwd(+reduce)", lambda: lib.st_ae_bwd(D, p(mag), p(phs), p(kn), p(ae_m), p(ae_p), p(mag_hat), p(phs_hat), p(dAA), None, 1e-9, p(dmag), p(dphs), p(aews), p(g_m), p(g_p), S()),
       flops=3 * MAC * B * F * 2 * 8128)
timeit("polar_bwd", lambda: lib.st_polar_bwd(D, p(re), p(im), p(dmag), p(dphs), None, p(dG), S()))
timeit("analysis_wgrad(+reduce)", lambda: lib.st_analysis_wgrad(D, p(dG), p(x), 0.5, p(wsg), p(gW[0]), p(gW[1]), p(na), S()), flops=MAC * B * T * 2 * F * N)
timeit("finalize_scalars", lambda: lib.st_finalize_scalars(D, p(lp), p(regp), p(na), p(ns), 1.0, p(eng.scalars), S()))
timeit("clip_adam", lambda: lib.st_clip_adam(p(eng.params), p(eng.grads), p(eng.m), p(eng.v), lay.total, lay.n_stft, p(eng.scalars), 1.0, 1e-6, 0.9, 0.999, 1e-8, 1, S()))
eng.load_state_dict(P); eng.m.zero_(); eng.v.zero_()
step_flops = 211.8e6 * B
ms = timeit("FULL forward (st_model_fwd)", lambda: eng.forward(x, kn), flops=88.1e6 * B)
ms = timeit("FULL train_step", lambda: eng.train_step(x, kn, y, 1e-6), flops=step_flops)
s = f"train step: {ms*1e3:.1f} us  -> {B/ms*1e3:.0f} windows/s = {25*B/ms*1e3/1e6:.3f} M frames/s ; {step_flops/ms/1e9:.2f} TFLOP/s algorithmic = {step_flops/ms/1e9/157.3*100:.1f}% of fp32 MFMA peak"
print(s); perf.write(s + "\n")
out(s)
perf.close(); log.close()
sys.exit(1 if nbad else 0)

"""GPU parity tests proper: every C-ABI entry point against the oracle (see tests/gpu_checks.py)."""
import os
import pytest

pytestmark = pytest.mark.gpu


def _assert_ok(res):
    bad = [r for r in res if not r["ok"]]
    assert not bad, "\n".join(f"{r['name']}: err={r['err']:.3e} scale={r['scale']:.3e} tol={r['tol']}" for r in bad)


@pytest.mark.parametrize("B,seed,K", [(3, 0, 4), (5, 2, 3), (1, 9, 4), (2, 4, 4), (6, 6, 1), (7, 7, 4), (2, 3, 16), (2, 3, 7)])
def test_per_op_parity(B, seed, K):
    from tests import gpu_checks as G
    _assert_ok(G.run_all(B=B, seed=seed, K=K))


@pytest.mark.parametrize("B,seed,K", [(1, 3, 4), (2, 4, 3), (2, 6, 16)])
def test_per_op_parity_scale8(B, seed, K):
    """BASELINE configs[4] geometry (65536-sample window, T=174, OT=46): the autoencoders take the wide
    feature-major GEMM path (st_ae_wide.h); everything else is the same kernels at larger sizes."""
    from tests import gpu_checks as G
    _assert_ok(G.run_all(B=B, seed=seed, K=K, scale=8))


@pytest.mark.parametrize("B,seed,K,steps", [(3, 1, 4, 3), (7, 4, 2, 2)])
def test_fused_step_parity(B, seed, K, steps):
    from tests import gpu_checks as G
    _assert_ok(G.run_fused(B=B, seed=seed, K=K, steps=steps))


@pytest.mark.parametrize("mode", ["f32", "f32x3", "bf16_all", "f16_all"])
@pytest.mark.parametrize("B,seed", [(64, 11), (130, 12)])
def test_trimmed_tiles_against_the_oracle(mode, B, seed):
    """Round 5: batches at which the structural-zero skipping of st_gemm_tn.h / st_gemm16.h is ACTIVE, against the oracle (the randomized sweep and the small
    fixed cases stay below 19 windows, where one 128-row tile holds every frame and nothing is skipped).  B = 64: tile rows of two frames each (frames 1-2: taps
    [256, 1024); frames 6-7: [0, 768)); B = 130: the first tile row is frame 1 alone (taps [640, 1024): 3 of 8 tile columns / 384 of 1024 reduction taps), the
    following ones straddle two frames.  Fused forward, loss, all 40 gradient tensors, clip norm and one optimizer step; misses of the fixed fp32 tolerance on
    ill-conditioned tensors are graded against the measured spread (tests/gpu_spread.py), never by name."""
    from tests import gpu_checks as G
    from tests import gpu_spread as S
    kw = dict(B=B, seed=seed, K=4)
    if mode == "f32":
        res = G.run_fused(steps=1, **kw)
    elif mode == "f32x3":
        with G.split_mode():
            res = G.run_fused(steps=1, **kw)
    else:
        half = "bf16" if mode.startswith("bf16") else "f16"
        with G.mixed_mode(2, half=half, tol_scale=(G.mixed_mode.FUSED_TOL if half == "bf16" else G.mixed_mode.FUSED_TOL_F16)[2]):
            res = G.run_fused(steps=1, **kw)
    still = S.grounded(res, kw) if mode in ("f32", "f32x3") else [r for r in res if not r["ok"]]
    assert not still, [(r["name"], r["rel"], r["tol"], r.get("spread")) for r in still]


@pytest.mark.parametrize("mode", ["f32", "f32x3"])
def test_branch_cut_window_is_graded_by_its_spread(mode):
    """Found by the round-5 sweep at batches where the trimmed tiles are active (tools/fuzz_parity.py 420 big): B = 64, shrink 8, K = 8, seed 643 in f32x3 --
    fwd.y_hat 2.4e-4 of the maximum against the 1e-4 of the north star.  Not the GEMMs: window 48 has a bin ON the branch cut of atan2 (re < 0, im ~ 0), phs jumps by
    2 pi under a 1e-6 perturbation of the input, the phase autoencoder is not 2 pi-periodic (nn_proc.py:310, :326), and that window's y_hat moves by 2.4e-4 in the float64
    ORACLE ITSELF (two of four perturbation draws).  The reference's forward is discontinuous there; a device that lands on the other side of the cut than the float64
    oracle is as right as the oracle.  Graded by tests/gpu_spread.py like every other miss: within 3 x the measured spread of that quantity, no names."""
    from tests import gpu_checks as G
    from tests import gpu_spread as S
    kw = dict(B=64, seed=643, K=8, scale=1, scheme="lean", shrink=8)
    if mode == "f32x3":
        with G.split_mode():
            res = G.run_fused(steps=1, **kw)
    else:
        res = G.run_fused(steps=1, **kw)
    still = S.grounded(res, kw)
    assert not still, [(r["name"], r["rel"], r["tol"], r.get("spread")) for r in still]


def test_legacy_large_fft_scheme():
    """nn_proc.py:374-376 (scale_scheme != 'lean'): ft and hop scale with the window -- scale 2: N=2048, H=768, F=1025,
    T=25, OT=9.  Same kernels, different GEMM sizes / spectral pitch."""
    from tests import gpu_checks as G
    _assert_ok(G.run_all(B=2, seed=3, K=4, scale=2, scheme="legacy"))
    _assert_ok(G.run_fused(B=3, seed=5, K=3, steps=2, scale=2, scheme="legacy"))


def test_legacy_large_fft_scheme_scale8_fp16():
    """BASELINE configs[4] read as "65536-sample window, large front-end FFT" = the legacy scheme at scale 8 in fp16 mixed precision
    (loss scale, clip over all parameters): two fused steps against the oracle with the same roundings."""
    from tests import gpu_checks as G
    # tolerance 1e-2: two windows x seven live frames -- every synthesis-side gradient element is a sum of 14 fp16 products and the phase
    # autoencoder's gradients come out of a cancellation (-dA_re sin + dA_im cos), so ONE d syn element that rounds the other way (its fp32
    # input differs by 1e-7 between two correct summation orders of the 8224-deep frames GEMM) moves them by ~1 %: tools/diag_g16b.py shows
    # the 16-bit operand pipeline and gemm_half_kernel bit-identical at scale 1 and 1.4 % apart here, on identical operands
    with G.mixed_mode(2, half="f16", tol_scale=100.0):
        _assert_ok(G.run_fused(B=2, seed=5, K=4, steps=2, scale=8, scheme="legacy"))


def test_legacy_large_fft_scheme_scale8():
    """SURVEY 8(f)-4 at the size it names: scale 8 of the legacy scheme = ft 8192, hop 3072, F 4097, T 25, OT 9 -- four
    8192 x 8192 bases (268 M parameters, 1.07 GB; the gradient / moment / slab buffers follow).  Same kernels, every per-op
    check and two fused steps against the oracle."""
    from tests import gpu_checks as G
    _assert_ok(G.run_all(B=1, seed=3, K=4, scale=8, scheme="legacy"))
    _assert_ok(G.run_fused(B=2, seed=5, K=3, steps=2, scale=8, scheme="legacy"))


@pytest.mark.parametrize("shrink,seed", [(2, 5), (1, 6), (8, 5)])
def test_other_shrink_factors(shrink, seed):
    """st_model(shrink_factor=...) (nn_proc.py:358-380): shrink 2 -> OT = 14 (fused autoencoder kernels), shrink 1 -> OT = 25
    = T (output as long as the input; wide autoencoder path with a narrow T), shrink 8 -> OT = 6."""
    from tests import gpu_checks as G
    _assert_ok(G.run_all(B=2, seed=3, K=4, shrink=shrink))
    _assert_ok(G.run_fused(B=3, seed=seed, K=3, steps=2, shrink=shrink))


def test_fused_step_parity_scale8():
    from tests import gpu_checks as G
    _assert_ok(G.run_fused(B=2, seed=5, K=4, steps=2, scale=8))


def test_ae_bwd_repeatable():
    """Run-to-run determinism of the dominant kernel (and a regression check for the timing-dependent MFMA
    result hazard found in round 1): three back-to-back launches with other work in between give identical bits."""
    import ctypes as C
    import numpy as np, torch
    from tests import gpu_checks as G
    from signaltrain_amd import _lib
    from signaltrain_amd.engine import StepEngine
    B, K = 5, 4
    geo, X, Y, KN, P = G.make_case(B, 11, K=K)
    d = G.dims_of(geo, B, K)
    eng = StepEngine(d, G.DEV); eng.load_state_dict(P)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)
    outs = []
    for rep in range(3):
        eng.loss_backward(x, kn, y)
        torch.cuda.synchronize()
        outs.append(eng.grads.clone())
        junk = torch.randn(2048, 2048, device=G.DEV); (junk @ junk).sum().item()
    names = [k for k, v in eng.layout.views(outs[0]).items()
             if not (torch.equal(v, eng.layout.views(outs[1])[k]) and torch.equal(v, eng.layout.views(outs[2])[k]))]
    assert not names, names


@pytest.mark.parametrize("dtype", ["f32", "bf16_all", "f16_all"])
@pytest.mark.parametrize("scale,B", [(1, 256), (8, 64), (1, 104), (1, 192), (1, 2048), (8, 512)])
def test_full_size_batch_properties(dtype, scale, B):
    """The per-GPU workloads of BASELINE configs[1..4] at FULL size -- 256 windows of 8192 samples, 64 windows of 65536 -- in the three
    arithmetic modes the configs name: properties that do not need the (slow) oracle.  Windows are independent, the loss is a mean over
    the batch and the L1 term a mean over B*OT*F, so
      forward(B) == concat(forward(halves)),   grads(B) == (grads(half 1) + grads(half 2)) / 2,
    and two runs give identical bits.  Exercises the full-size tiling / split-K / reduction paths (128 x 128 weight-gradient tiles + Nyquist
    partials, the 16-bit operand pipeline, the wide autoencoder path).  16-bit modes: every product is exact in fp32 on both sides, only
    fp32 sums re-associate -- and d loss / d y_hat carries 1 / (B y), a power of two between full and half batch, which commutes with the
    rounding for bf16: tolerance 2e-3.  Round 5: B = 104 and 192 at the short window -- 128-row tiles of the frame-major row order that hold parts of TWO frames
    (the union of their live tap ranges is computed) resp. one and a half tiles per frame, against halves (52, 96) whose work lists differ: the structural-zero
    skipping of st_gemm_tn.h / st_gemm16.h must not depend on how the frames fall on the tiles.  fp16: the polar backward SATURATES its output at +-65504 before the weight-gradient GEMM narrows it
    (1e7-sized atan2 sub-gradients on near-silent frames x loss scale 4096) and small values go subnormal -- a half batch's 2x
    larger gradients saturate / round where the full batch's do not, so the property holds to ~1 % only (measured 0.7 %): tolerance 2e-2.
    Round 6: eight times the bench sizes -- 2048 windows of 8192 samples (47 104 frame rows: several rounds of every tile grid, work lists at their largest) and 512 windows
    of 65536 samples (89 088 rows, 270 k autoencoder columns on the wide path) against their halves: the LARGE end of the size range, where nothing else in the suite goes."""
    import numpy as np, torch
    from tests import gpu_checks as G
    from signaltrain_amd.engine import StepEngine
    K = 4
    geo, X, Y, KN, P = G.make_case(8, 31, K=K, scale=scale)
    rng = np.random.default_rng(1)
    reps = B // 8
    X = (np.tile(X, (reps, 1)) * rng.uniform(0.4, 1.0, (B, 1))).astype(np.float32)
    Y = (np.tile(Y, (reps, 1)) * rng.uniform(0.4, 1.0, (B, 1))).astype(np.float32)
    KN = (rng.random((B, K)) - 0.5).astype(np.float32)
    x, y, kn = G.t(X), G.t(Y), G.t(KN)
    tf, tg = {"f32": (1e-6, 2e-5), "bf16_all": (1e-6, 2e-3), "f16_all": (1e-6, 2e-2)}[dtype]
    full = StepEngine(G.dims_of(geo, B, K), G.DEV, compute_dtype=dtype); full.load_state_dict(P)
    half = StepEngine(G.dims_of(geo, B // 2, K), G.DEV, compute_dtype=dtype); half.load_state_dict(P)
    yf, mf, hf = full.forward(x, kn)
    parts = [half.forward(x[i:i + B // 2], kn[i:i + B // 2]) for i in (0, B // 2)]
    for a, name, j in ((yf, "y_hat", 0), (mf, "mag", 1), (hf, "mag_hat", 2)):
        ref = torch.cat([p[j] for p in parts])
        assert (a - ref).abs().max().item() <= tf * ref.abs().max().item(), name
    full.loss_backward(x, kn, y); torch.cuda.synchronize(); g_full = full.grads.clone(); l_full = float(full.scalars[0])
    full.loss_backward(x, kn, y); torch.cuda.synchronize()
    assert torch.equal(g_full, full.grads)                          # bit-repeatable at full size
    assert bool(torch.isfinite(g_full).all())
    gs, ls = [], []
    for i in (0, B // 2):
        half.loss_backward(x[i:i + B // 2], kn[i:i + B // 2], y[i:i + B // 2]); torch.cuda.synchronize()
        gs.append(half.grads.clone()); ls.append(float(half.scalars[0]))
    g_ref = 0.5 * (gs[0] + gs[1])
    assert abs(l_full - 0.5 * (ls[0] + ls[1])) <= 1e-5 * abs(l_full)
    for name, v in full.layout.views(g_full).items():
        r = full.layout.views(g_ref)[name]
        assert (v - r).abs().max().item() <= tg * max(r.abs().max().item(), 1e-12), (name, (v - r).abs().max().item(), r.abs().max().item())


@pytest.mark.parametrize("B,seed,K", [(3, 0, 4), (5, 2, 3)])
def test_bf16_mode_per_op(B, seed, K):
    """st_dims.prec = ST_PREC_BF16: bf16 operands / fp32 accumulation in the STFT GEMMs (BASELINE configs[2], [3]) against the
    oracle with the SAME operands rounded to bfloat16 (oracle.GEMM_ROUND): per-op agreement stays at the 1e-6 level
    because both sides round identical inputs."""
    from tests import gpu_checks as G
    with G.bf16_mode():
        _assert_ok(G.run_all(B=B, seed=seed, K=K))


def test_bf16_mode_fused_step_and_differs_from_fp32():
    import torch
    from tests import gpu_checks as G
    from signaltrain_amd.engine import StepEngine
    with G.bf16_mode(1, tol_scale=G.bf16_mode.FUSED_TOL[1]):
        _assert_ok(G.run_fused(B=3, seed=1, K=4, steps=3))
    geo, X, Y, KN, P = G.make_case(3, 0, K=4)
    d = G.dims_of(geo, 3, 4)
    e32 = StepEngine(d, G.DEV); e32.load_state_dict(P)
    e16 = StepEngine(d, G.DEV, compute_dtype="bf16"); e16.load_state_dict(P)
    y32 = e32.forward(G.t(X), G.t(KN))[0]; y16 = e16.forward(G.t(X), G.t(KN))[0]
    y32b = e32.forward(G.t(X), G.t(KN))[0]
    rel = float((y32 - y16).abs().max() / y32.abs().max())
    assert 1e-5 < rel < 2e-2, rel                                  # really bf16 arithmetic, and sane
    assert torch.equal(y32, y32b)                                  # the fp32 engine is unaffected by the other engine's mode


def test_bf16_mode_scale8():
    """Geometry of BASELINE configs[4] (65536-sample window) with bf16 GEMM operands (the fp16 arithmetic that configuration
    names is test_f16_*_scale8 below)."""
    from tests import gpu_checks as G
    with G.bf16_mode():
        _assert_ok(G.run_all(B=1, seed=3, K=4, scale=8))
    with G.bf16_mode(1, tol_scale=G.bf16_mode.FUSED_TOL[1]):
        _assert_ok(G.run_fused(B=2, seed=5, K=4, steps=2, scale=8))


@pytest.mark.parametrize("B,seed,K", [(3, 0, 4), (2, 5, 7), (1, 9, 16)])
def test_bf16_all_mode_per_op(B, seed, K):
    """st_dims.prec = ST_PREC_BF16_ALL: bf16 operands also in the nine Linear layers of both autoencoders (forward, data gradient, weight
    gradient: the BF instantiations of st_ae.h, one v_mfma_f32_16x16x16_bf16 per tile) against the oracle with the same
    operands rounded to bfloat16 (oracle.AE_ROUND)."""
    from tests import gpu_checks as G
    with G.bf16_mode(2):
        _assert_ok(G.run_all(B=B, seed=seed, K=K))


def test_bf16_all_mode_fused_step_differs_from_bf16_gemm_mode():
    import torch
    from tests import gpu_checks as G
    from signaltrain_amd.engine import StepEngine
    B, K = 3, 4
    with G.bf16_mode(2, tol_scale=G.bf16_mode.FUSED_TOL[2]):
        _assert_ok(G.run_fused(B=B, seed=1, K=K, steps=2))
    geo, X, Y, KN, P = G.make_case(B, 1, K=K)
    d = G.dims_of(geo, B, K)
    e1 = StepEngine(d, G.DEV, compute_dtype="bf16"); e1.load_state_dict(P)
    e2 = StepEngine(d, G.DEV, compute_dtype="bf16_all"); e2.load_state_dict(P)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)
    e1.loss_backward(x, kn, y); e2.loss_backward(x, kn, y)
    torch.cuda.synchronize()
    o = e1.layout.offsets
    rel = (e1.grads[o[4]:] - e2.grads[o[4]:]).abs().max().item() / e1.grads[o[4]:].abs().max().item()
    assert 1e-5 < rel < 5e-2, rel                                    # the autoencoder gradients really went through bf16 products


def test_bf16_all_mode_scale8():
    """Level-2 precision at the 65536-sample geometry (BASELINE configs[4]): the wide autoencoder path runs its layer-1 / layer-9
    GEMMs on the bf16 kernel and the fused inner layers in their BF instantiation (needs an even batch: K = R operands must be
    a multiple of the 32-deep bf16 k-tile, otherwise the wide path stays fp32).  Per-op against the oracle with the same
    roundings; the fused step at the level-2 noise floor (2e-2, tools/bf16_noise_floor.py) -- the
    bias gradients of layers 1 / 9 come out of the GEMM with rounded dA where the oracle sums unrounded values."""
    from tests import gpu_checks as G
    with G.bf16_mode(2):
        _assert_ok(G.run_all(B=2, seed=3, K=4, scale=8))
    with G.bf16_mode(2, tol_scale=G.bf16_mode.FUSED_TOL[2]):
        _assert_ok(G.run_fused(B=2, seed=1, K=4, steps=2, scale=8))


# ------------------------------------------------------------------------------------------------ fp16 (BASELINE configs[4])
@pytest.mark.parametrize("level,B,seed,K,scale", [(1, 3, 0, 4, 1), (2, 3, 0, 4, 1), (2, 2, 5, 7, 1), (1, 1, 3, 4, 8), (2, 2, 3, 4, 8)])
def test_f16_mode_per_op(level, B, seed, K, scale):
    """st_dims.prec = ST_PREC_F16 / ST_PREC_F16_ALL with loss scale 4096: float16 operands (saturating conversion), fp32
    accumulation -- every per-op entry point against the oracle with the same operands rounded to IEEE half
    (oracle.fp16_round) and the same loss scale on the gradient operands; scale 8 = the 65536-sample window of configs[4]."""
    from tests import gpu_checks as G
    with G.mixed_mode(level, half="f16"):
        _assert_ok(G.run_all(B=B, seed=seed, K=K, scale=scale))


@pytest.mark.parametrize("level,scale,B", [(1, 1, 3), (2, 1, 3), (2, 8, 2)])
def test_f16_fused_step_with_loss_scale(level, scale, B):
    """The mixed-precision train step as the reference runs it under Apex (train.py:133-136): loss scaled by S = 4096 before
    the backward, gradients unscaled inside the optimizer kernel, L1 clip over ALL parameters -- against the oracle doing the
    same; no step may be skipped (overflow counter stays 0)."""
    import torch
    from tests import gpu_checks as G
    with G.mixed_mode(level, half="f16", tol_scale=G.mixed_mode.FUSED_TOL_F16[level] * (2.0 if scale == 8 else 1.0)):
        _assert_ok(G.run_fused(B=B, seed=1, K=4, steps=2, scale=scale))


def test_f16_overflow_skips_the_step_and_counts_it():
    """A loss scale large enough to overflow the fp16 gradient operands (inf -> non-finite L1 norm) must leave parameters and moments untouched and bump
    the overflow counter (scalars[5]) -- the signal Apex's dynamic loss scaler acts on; with a sane scale the step goes through
    and the result does not depend on the scale beyond rounding."""
    import torch
    from tests import gpu_checks as G
    from signaltrain_amd.engine import StepEngine
    geo, X, Y, KN, P = G.make_case(3, 2, K=4)
    d = G.dims_of(geo, 3, 4)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)
    eng = StepEngine(d, G.DEV, compute_dtype="f16_all", loss_scale=3.0e38); eng.load_state_dict(P)
    before = eng.params.clone()
    eng.train_step(x, kn, y, 1e-3); torch.cuda.synchronize()
    assert eng.overflow_steps(reset=False) == 1
    assert torch.equal(eng.params, before) and float(eng.m.abs().max()) == 0.0 and float(eng.v.abs().max()) == 0.0
    assert eng.overflow_steps() == 1 and eng.overflow_steps() == 0
    outs = []
    for S in (1024.0, 8192.0):
        e = StepEngine(d, G.DEV, compute_dtype="f16_all", loss_scale=S); e.load_state_dict(P)
        e.train_step(x, kn, y, 1e-3); torch.cuda.synchronize()
        assert e.overflow_steps() == 0 and torch.isfinite(e.params).all()
        outs.append((e.params.clone(), e.grads.clone()))
    assert (outs[0][0] - before).abs().max().item() > 1e-5                    # a real update happened
    # the unscaled, clipped gradient does not depend on the scale beyond fp16 rounding of the scaled operands (the first Adam
    # step itself is lr * sign(g): a noise-level element that changes sign moves its parameter by 2 lr, so parameters are compared loosely)
    g0, g1 = outs[0][1], outs[1][1]
    assert (g0 - g1).abs().max().item() <= 3e-3 * g0.abs().max().item()
    assert ((outs[0][0] - outs[1][0]).abs() > 2e-4).float().mean().item() < 2e-2


def test_engines_of_different_precision_coexist():
    """Precision is carried per call (st_dims.prec), not process-wide: interleaving a bf16_all, an f16_all and an f32 engine
    leaves the f32 engine's results bit-identical to running alone."""
    import torch
    from tests import gpu_checks as G
    from signaltrain_amd.engine import StepEngine
    geo, X, Y, KN, P = G.make_case(3, 4, K=4)
    d = G.dims_of(geo, 3, 4)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)
    alone = StepEngine(d, G.DEV); alone.load_state_dict(P)
    for _ in range(2):
        alone.train_step(x, kn, y, 1e-3)
    e32 = StepEngine(d, G.DEV); e32.load_state_dict(P)
    eb = StepEngine(d, G.DEV, compute_dtype="bf16_all"); eb.load_state_dict(P)
    eh = StepEngine(d, G.DEV, compute_dtype="f16_all"); eh.load_state_dict(P)
    for _ in range(2):
        eb.train_step(x, kn, y, 1e-3); e32.train_step(x, kn, y, 1e-3); eh.train_step(x, kn, y, 1e-3)
    torch.cuda.synchronize()
    assert torch.equal(alone.params, e32.params) and torch.equal(alone.scalars[:5], e32.scalars[:5])
    assert not torch.equal(eb.params, e32.params) and not torch.equal(eh.params, e32.params) and not torch.equal(eh.params, eb.params)


def test_engine_on_a_non_current_device_or_stream():
    """The engine launches on ITS device's current stream whatever device is current: with one GPU, a side stream
    made current for the engine's device must carry the work (the default stream stays idle, results identical); with two GPUs
    an engine on cuda:1 works while cuda:0 is current."""
    import torch
    from tests import gpu_checks as G
    from signaltrain_amd.engine import StepEngine
    geo, X, Y, KN, P = G.make_case(2, 6, K=4)
    d = G.dims_of(geo, 2, 4)
    ref = StepEngine(d, G.DEV); ref.load_state_dict(P)
    ref.train_step(G.t(X), G.t(KN), G.t(Y), 1e-3); torch.cuda.synchronize()
    side = torch.cuda.Stream(device=G.DEV)
    e = StepEngine(d, G.DEV); e.load_state_dict(P)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        e.train_step(x, kn, y, 1e-3)
    side.synchronize()
    assert torch.equal(e.params, ref.params)
    if torch.cuda.device_count() >= 2:
        e1 = StepEngine(d, "cuda:1"); e1.load_state_dict(P)
        assert torch.cuda.current_device() == 0
        e1.train_step(torch.from_numpy(X).to("cuda:1"), torch.from_numpy(KN).to("cuda:1"), torch.from_numpy(Y).to("cuda:1"), 1e-3)
        torch.cuda.synchronize("cuda:1")
        assert torch.equal(e1.params.cpu(), ref.params.cpu())


@pytest.mark.parametrize("dtype,codes,restore", [
    ("f32", (9000,), (9001,)),            # synthesis frames GEMM in the k-major form
    ("f32", (8001,), (8002,)),            # two-kernel autoencoder backward in fp32
    ("f32", (8200,), (8201,)),            # round 6: autoencoder backward that RECOMPUTES the activations instead of reading the ones the forward kept
    ("bf16", (8200,), (8201,)),           # ... with 16-bit STFT GEMMs around the fp32 autoencoders
    ("f32x3", (8200,), (8201,)),
    ("bf16_all", (8000,), (8002,)),       # single-kernel autoencoder backward with 16-bit Linear layers
    ("f16_all", (8000,), (8002,)),
    ("f32", (7001,), (7000,)),            # k-quad-major transposed staging of the weight-gradient GEMMs
    ("f32", (102,), (100,)),              # weight-gradient tile mode 2
    ("f32x3", (9201,), (9200,)),          # synthesis data gradient on the plane kernel
    ("f32x3", (9100,), (9103,)),          # 4-wave analysis plane tile
    ("f32x3", (9301,), (9300,)),          # weight gradients on the in-kernel three-plane split
    ("bf16", (9401,), (9400,)),           # one-plane (pre-converted bf16) bases on the plane kernel
    ("f32", (9950,), (9951,)),            # synthesis frames / data gradient on gemm_kernel<2, ...> instead of the 128 x 128 NT tiles
    ("f32", (9500,), (9501,)),            # weight gradients on gemm_kernel<3, ...> instead of the 128 x 128 TN tiles
    ("bf16_all", (9693,), (9690,)),       # 16-bit analysis forward + data gradient on the LDS-DMA kernel (producer / consumer waves)
    ("f16_all", (9693,), (9690,)),
    ("bf16_all", (9600,), (9601,)),       # converting GEMM (gemm_half_kernel) instead of the pre-rounded 16-bit operand pipeline
    ("bf16_all", (8100,), (8101,)),       # 16-bit autoencoder forward on 16-row groups (st_ae.h) instead of the 32-row kernel (st_ae32.h)
    ("f16_all", (8100,), (8101,)),
])
def test_alternative_code_paths_agree(dtype, codes, restore):
    """The variants kept behind st_set_tuning (measured slower, or experiments) compute the same thing as the default path: loss and
    all 40 gradient tensors of one batch (fp32-grade variants to 2e-5 of each tensor's largest element -- reassociation only --,
    the 16-bit ones to the rounding noise of their arithmetic)."""
    import numpy as np, torch
    from tests import gpu_checks as G
    from signaltrain_amd import _lib
    from signaltrain_amd.engine import StepEngine
    lib = _lib.load()
    B, K = 4, 4
    geo, X, Y, KN, P = G.make_case(B, 17, K=K)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)

    def run():
        d = G.dims_of(geo, B, K)
        eng = StepEngine(d, G.DEV, compute_dtype=dtype); eng.load_state_dict(P)
        eng.loss_backward(x, kn, y); torch.cuda.synchronize()
        return {k: v.clone() for k, v in eng.layout.views(eng.grads).items()}, float(eng.scalars[0])
    ref, l_ref = run()
    try:
        for c in codes: _lib.check(lib.st_set_tuning(c), "st_set_tuning")
        alt, l_alt = run()
    finally:
        for c in restore: _lib.check(lib.st_set_tuning(c), "st_set_tuning")
    tol = 2e-5 if dtype in ("f32", "f32x3") else 3e-2        # fp32-grade: reassociation only; 16-bit: a different summation order moves roundings
    assert abs(l_alt - l_ref) <= tol * abs(l_ref), (l_ref, l_alt)
    for k in ref:
        sc = ref[k].abs().max().item()
        assert (alt[k] - ref[k]).abs().max().item() <= tol * sc + 1e-12, (dtype, codes, k)


@pytest.mark.parametrize("B", [3, 64, 130])
def test_kept_activations_backward_is_the_recompute_backward_bit_for_bit(B):
    """Round 6: the fused fp32 step keeps the autoencoders' post-ELU activations in the forward kernel (what the reference's autograd keeps, nn_proc.py:77-126) and the
    backward kernel reads them instead of recomputing the forward chain.  Same values, same order of every sum: loss, all 40 gradient tensors and the parameters after
    two optimizer steps are IDENTICAL BITS to the recomputing backward (st_set_tuning(8200)), at a small batch, at a batch with several rounds per wave and at one whose
    last groups are ragged."""
    import numpy as np, torch
    from tests import gpu_checks as G
    from signaltrain_amd import _lib
    from signaltrain_amd.engine import StepEngine
    lib = _lib.load()
    K = 4
    geo, X, Y, KN, P = G.make_case(8, 23, K=K)
    rng = np.random.default_rng(11)
    reps = (B + 7) // 8
    X = (np.tile(X, (reps, 1))[:B] * rng.uniform(0.4, 1.0, (B, 1))).astype(np.float32)
    Y = (np.tile(Y, (reps, 1))[:B] * rng.uniform(0.4, 1.0, (B, 1))).astype(np.float32)
    KN = (rng.random((B, K)) - 0.5).astype(np.float32)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)

    def run():
        d = G.dims_of(geo, B, K)
        eng = StepEngine(d, G.DEV); eng.load_state_dict(P)
        eng.loss_backward(x, kn, y); torch.cuda.synchronize()
        g = {k: v.clone() for k, v in eng.layout.views(eng.grads).items()}; l = float(eng.scalars[0])
        eng.train_step(x, kn, y, 1e-3); eng.train_step(x, kn, y, 1e-3); torch.cuda.synchronize()
        return g, l, eng.params.clone()
    g_keep, l_keep, p_keep = run()
    try:
        _lib.check(lib.st_set_tuning(8200), "st_set_tuning")
        g_rec, l_rec, p_rec = run()
    finally:
        _lib.check(lib.st_set_tuning(8201), "st_set_tuning")
    assert l_keep == l_rec
    for k in g_keep:
        assert torch.equal(g_keep[k], g_rec[k]), k
    assert torch.equal(p_keep, p_rec)


@pytest.mark.parametrize("dtype,codes,restore", [
    ("f32", (9950,), (9951,)),            # small-tile gemm_kernel<2, ...> (window-major rows, every tap multiplied) instead of the work-list kernel
    ("f32", (9540,), (9543,)),            # weight gradients over ALL reduction rows in window-major order instead of the per-tile-column row ranges
    ("f32", (9500,), (9501,)),            # weight gradients on gemm_kernel<3, ...> instead of the 128 x 128-tile kernel
    ("bf16_all", (9560,), (9575,)),       # 16-bit synthesis GEMMs / weight gradients without the structural-zero skipping
    ("f16_all", (9560,), (9575,)),
])
def test_structural_zero_paths_agree_with_the_full_products(dtype, codes, restore):
    """Round 5: at a batch where the skipping is ACTIVE (B = 130: the first tile row is frame 1 alone, the others straddle two frames) the trimmed paths give what the
    untrimmed ones give -- the skipped products are exact zeros, so fp32 results differ by reassociation only (2e-5 of each tensor's largest element) and the 16-bit ones
    by the rounding noise a different summation order causes.  Loss and all 40 gradient tensors."""
    import numpy as np, torch
    from tests import gpu_checks as G
    from signaltrain_amd import _lib
    from signaltrain_amd.engine import StepEngine
    lib = _lib.load()
    B, K = 130, 4
    geo, X, Y, KN, P = G.make_case(8, 19, K=K)
    rng = np.random.default_rng(3)
    reps = (B + 7) // 8
    X = (np.tile(X, (reps, 1))[:B] * rng.uniform(0.4, 1.0, (B, 1))).astype(np.float32)
    Y = (np.tile(Y, (reps, 1))[:B] * rng.uniform(0.4, 1.0, (B, 1))).astype(np.float32)
    KN = (rng.random((B, K)) - 0.5).astype(np.float32)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)

    def run():
        d = G.dims_of(geo, B, K)
        eng = StepEngine(d, G.DEV, compute_dtype=dtype); eng.load_state_dict(P)
        eng.loss_backward(x, kn, y); torch.cuda.synchronize()
        return {k: v.clone() for k, v in eng.layout.views(eng.grads).items()}, float(eng.scalars[0])
    ref, l_ref = run()
    try:
        for c in codes: _lib.check(lib.st_set_tuning(c), "st_set_tuning")
        alt, l_alt = run()
    finally:
        for c in restore: _lib.check(lib.st_set_tuning(c), "st_set_tuning")
    tol = 2e-5 if dtype == "f32" else 3e-2
    assert abs(l_alt - l_ref) <= tol * abs(l_ref), (l_ref, l_alt)
    for k in ref:
        sc = ref[k].abs().max().item()
        assert (alt[k] - ref[k]).abs().max().item() <= tol * sc + 1e-12, (dtype, codes, k, (alt[k] - ref[k]).abs().max().item(), sc)


# ------------------------------------------------------------------------------------------------ the randomized sweep's 16-bit outliers, grounded
def _fuzz_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_ground", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_ground.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("idx", range(17))
def test_fuzz_outliers_grounded(idx):
    """VERDICT round 3 weak #1.  The six 16-bit configurations the randomized sweep flagged (profiles/r03_fuzz_parity.txt) and their even-batch / other-K
    neighbours (tools/fuzz_ground.py CASES; table: profiles/r04_fuzz_grounding.txt).  Five of the six were the SILENT fp32 fallback of the wide
    autoencoder path for odd batches (lean scale 2 and shrink 1 are wide geometries): the device ran fp32 autoencoder layers against an oracle rounding
    them to 16 bits.  The library now reports its effective arithmetic (st_effective_prec) and the checks' oracle follows it: per-op green at the per-op
    tolerance, fused within max(suite tolerance, 3 x the oracle's own spread for that configuration) -- profiles/r05_fuzz_self_noise.json, the rounding
    oracle against itself under eight 1e-6 perturbations.  The sixth (f16_all, 65536-sample window, K = 16) sits inside that spread.
    Round 5: the wide path takes 16-bit layers for ODD batches too, so all of these run the requested arithmetic (the self-noise table was recomputed with the layers
    rounded everywhere), and the one hard line of the round-5 sweep -- a single 65536-sample window in f16_all, seed 300 -- is case 13 (0.4-0.6 x its spread).
    Round 6: the three hard 16-bit lines of a sweep with a fresh seed (profiles/r06_fuzz_parity_seed4242.txt), all f16_all on the analysis-basis gradients, are cases 14-16
    (0.2-1.0 x their spread: profiles/r06_fuzz_grounding_16bit.txt)."""
    import json
    from tests import gpu_checks as G
    m = _fuzz_cases()
    mode, kw = m.CASES[idx]
    noise = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_fuzz_self_noise.json")))[m.tag(mode, kw)]
    half = "bf16" if mode.startswith("bf16") else "f16"
    with G.mixed_mode(2, half=half, tol_scale=(None if kw["scale"] != 8 else (40.0 if half == "bf16" else 20.0))):
        per = G.run_all(B=kw["B"], seed=kw["seed"], K=kw["K"], scale=kw["scale"], scheme=kw["scheme"], shrink=kw["shrink"])
    bad = [r for r in per if not r["ok"] and r["rel"] > noise.get(r["name"].replace("ae_bwd.g.", "grad."), 0.0)]
    assert not bad, [(r["name"], r["rel"], r["tol"]) for r in bad]
    ftol = (G.mixed_mode.FUSED_TOL if half == "bf16" else G.mixed_mode.FUSED_TOL_F16)[2]
    with G.mixed_mode(2, half=half, tol_scale=ftol):
        fused = G.run_fused(steps=1, **kw)
    # no tensor is exempt by name (round 5): every miss of the suite tolerance -- the analysis-basis gradients included -- has to sit within 3 x the oracle's own spread
    bad = [r for r in fused if not r["ok"] and r["rel"] > 3.0 * noise.get(r["name"], 0.0)]
    assert not bad, [(r["name"], r["rel"], r["tol"], noise.get(r["name"])) for r in bad]


def _f32_soft_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_ground_f32", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_ground_f32.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("idx", range(23))
def test_fp32_soft_lines_grounded(idx):
    """VERDICT round 4 weak #1 / next #1.  The 21 configurations of the round-4 randomized sweep (profiles/r04_fuzz_parity.txt) where the exact-fp32 or the
    f32x3 fused step missed the suite's 2e-4 on the analysis-basis gradients (up to 1.0e-3 of the tensor maximum) or 2e-5 on the parameters after one Adam
    step -- until round 4 filtered out by tensor NAME.  Now: a miss of the fixed tolerance passes only if the device's error is within 3 x the spread of that
    quantity for this configuration -- the oracle in float32 arithmetic against itself in float64 (the reference, PyTorch fp32, is on that side) and the
    float64 oracle under eight 1e-6 input perturbations (tests/gpu_spread.py; cached in profiles/r05_fuzz_f32_spread.json, table with the device columns in
    profiles/r05_fuzz_f32_grounding.txt).  The cause is the conditioning of d atan2(im, re) = (-im, re) / (re^2 + im^2) (nn_proc.py:309-310) at near-silent
    bins; golden G13 (tools/capture_golden_r5.py) shows the reference's own fp32 autograd moving by the same amount against float64.
    Case 21 (round 6, a sweep with a fresh seed): a single window at lean scale 2 whose spread is 1e-2 -- the device sits at 0.3 x that, which is 14 x the fixed tolerance and therefore over
    the cap; it passes as a LOCALIZED miss only (tests/gpu_spread.py LOCAL_ROWS: the elements over the tolerance lie in a handful of the tensor's 1024 rows).
    Case 22 (two more sweeps, 926 configurations, one window flagged): exact fp32, B = 5, seed 719 -- one row of the imaginary basis' gradient is 19 % of the tensor maximum off (spread
    32 %: a bin at the origin of atan2), localized like case 21, and it moves the published clip norm by 1 %: `step.l1norm` is graded by its spread as well since then."""
    from tests import gpu_checks as G
    from tests import gpu_spread as S
    m = _f32_soft_cases()
    mode, kw = m.CASES[idx]
    if mode == "f32x3":
        with G.split_mode():
            res = G.run_fused(steps=1, **kw)
    else:
        res = G.run_fused(steps=1, **kw)
    still = S.grounded(res, kw)
    assert not still, [(r["name"], r["rel"], r["tol"], r.get("spread"), r.get("ratio")) for r in still]


@pytest.mark.parametrize("scale,shrink,K,B", [(1, 4, 4, 3), (1, 4, 3, 2), (1, 2, 7, 2), (2, 4, 4, 3), (8, 4, 3, 2)])
def test_knob_gradient_against_oracle(scale, shrink, K, B):
    """st_model_knob_grad (d loss / d knobs; nn_proc.py:92-93 under autograd) against the oracle's d_knobs (pinned to the reference's autograd by golden
    G12) with the training loss's own upstream gradients: fused geometries, 3 / 7 knobs, an odd batch, and the wide autoencoder path (lean scale 2,
    the 65536-sample window).  fp32, the tolerance of the parameter gradients."""
    import numpy as np, torch
    from oracle import st_oracle as O
    from tests import gpu_checks as G
    geo, X, Y, KN, P = G.make_case(B=B, seed=31, scale=scale, shrink=shrink, K=K)
    _, _, c = O.model_loss_bwd(X, KN, Y, P, geo)
    F = geo["F"]
    w = O.freq_weights(F, np.float32)
    g_mh = (np.float32(O.L1_LAMBDA / 10) / np.float32(B * geo["OT"] * F) * np.sign(c["mag_hat"]) * w).astype(np.float32)      # the L1 term of calc_loss (loss_functions.py:36)
    d = G.dims_of(geo, B, K)
    eng = G.new_engine(d); eng.load_state_dict(P)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(G.DEV)
    before = eng.grads.clone()
    got = eng.knob_grad(dev(X), dev(KN), dev(c["dy"]), dev(g_mh)).cpu().numpy().astype(np.float64)
    ref = c["d_knobs"].astype(np.float64)
    assert got.shape == ref.shape == (B, K)
    assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max(), (got, ref)
    assert torch.equal(eng.grads, before)                 # the per-window passes write a scratch buffer, not the engine's gradients


@pytest.mark.parametrize("dtype,scale,shrink", [("f32", 2, 4), ("f16_all", 2, 4), ("bf16_all", 1, 1), ("f32", 8, 4)])
def test_wide_direct_input_equals_copy_kernel(dtype, scale, shrink):
    """Wide geometries (round 4): the analysis GEMM's polar epilogue writes mag / phs straight into the feature-major input of the wide autoencoder
    path and prep_kernel does the copy kernel's side jobs (pad columns, all-padding frames, knob rows, padded weight copies).  Same values, same
    arithmetic downstream: loss and all 40 gradient tensors are BITWISE those of the copy-kernel form (st_set_tuning(9970)), and the user-visible
    |STFT| of a forward call is still written."""
    import numpy as np, torch
    from tests import gpu_checks as G
    from signaltrain_amd import _lib
    from signaltrain_amd.engine import StepEngine
    lib = _lib.load()
    B, K = 2, 4
    geo, X, Y, KN, P = G.make_case(B, 23, K=K, scale=scale, shrink=shrink)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)

    def run():
        d = G.dims_of(geo, B, K)
        eng = StepEngine(d, G.DEV, compute_dtype=dtype); eng.load_state_dict(P)
        eng.loss_backward(x, kn, y); torch.cuda.synchronize()
        g = eng.grads.clone(); l = float(eng.scalars[0])
        yh, mag, mh = eng.forward(x, kn); torch.cuda.synchronize()
        return g, l, mag.clone()
    g1, l1, m1 = run()
    try:
        _lib.check(lib.st_set_tuning(9970), "st_set_tuning")
        g0, l0, m0 = run()
    finally:
        _lib.check(lib.st_set_tuning(9971), "st_set_tuning")
    assert l0 == l1 and torch.equal(g0, g1) and torch.equal(m0, m1) and float(m1.abs().max()) > 0


# Stated bounds of the 16-bit arithmetic modes against the UNROUNDED float64 oracle (the reference's fp32 model, pinned by the goldens), per tensor class:
# ~2.5 x the worst line of profiles/r05_16bit_vs_unrounded_oracle.txt (tools/loose_f32_check.py: five configurations incl. the 65536-sample window and shrink 2,
# two steps each).  "analysis grads" are the ill-conditioned tensors of tests/gpu_spread.py -- d atan2 amplifies the operand rounding of re / im by 1 / mag at
# near-silent bins, fp16's three extra mantissa bits do not help there -- hence one bound for both half types; "params" is absolute (Adam's first steps move a
# parameter by at most lr = 6.7e-5 each: a gradient element whose SIGN differs costs 2 lr per step).
LOOSE_BOUNDS = {
    "bf16": {"forward": 4e-2, "loss": 1.5e-2, "synthesis grads": 3e-2, "analysis grads": 0.15, "ae grads": 7e-2, "l1norm": 7e-2, "params": 7e-4},
    "f16": {"forward": 6e-3, "loss": 4e-3, "synthesis grads": 7e-3, "analysis grads": 0.15, "ae grads": 0.12, "l1norm": 3e-2, "params": 6e-4},
}


@pytest.mark.parametrize("mode,level,half", [("bf16", 1, "bf16"), ("bf16_all", 2, "bf16"), ("f16", 1, "f16"), ("f16_all", 2, "f16")])
@pytest.mark.parametrize("kw", [dict(B=3, seed=1, K=4), dict(B=8, seed=21, K=4), dict(B=2, seed=5, K=4, scale=8)], ids=["b3", "b8", "l65536"])
def test_16bit_modes_against_the_unrounded_oracle(mode, level, half, kw):
    """VERDICT round 4 missing #3 / weak #2, SURVEY.md section 5: every other 16-bit check compares the device with an oracle that rounds the same operands
    (gpu_checks.mixed_mode) -- that pins the kernels to the rounding oracle, not the rounding oracle to the reference.  Here the device runs the 16-bit mode and
    the oracle runs plain float64 (same loss scale and clip scope: those are the reference's Apex semantics, train.py:133-136, not roundings): forward outputs,
    loss, all 40 gradient tensors, the clip norm and the parameters after two steps within the stated per-class bounds above."""
    import importlib.util
    from tests import gpu_checks as G
    spec = importlib.util.spec_from_file_location("loose_f32_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "loose_f32_check.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    with G.mixed_mode(level, half=half, tol_scale=1e12, oracle_rounds=False):
        res = G.run_fused(steps=2, **kw)
    got = m.classes(res)
    bounds = LOOSE_BOUNDS[half]
    assert set(got) <= set(bounds), sorted(set(got) - set(bounds))
    over = {k: (v, bounds[k]) for k, v in got.items() if not (v[0] <= bounds[k])}
    assert not over, over


@pytest.mark.parametrize("B,nslab_scale", [(3, 1), (5, 8)])
def test_ola_loss_partials_do_not_depend_on_pointer_alignment(B, nslab_scale):
    """st_ola_loss picks the four-samples-per-thread kernel when every pointer is 16-byte aligned and the one-sample kernel otherwise.
    Both sum slot s = samples [256 s, 256 s + 256) of a window by the same tree, so the loss partials -- not only y_hat and d syn -- are bit-identical
    whichever one runs (the first four-wide version summed 1024 samples into one slot and zeroed three).  The same buffers are handed over once aligned and
    once from views that start 4 bytes into their allocation."""
    import ctypes as C
    import numpy as np
    import torch
    from signaltrain_amd import _lib
    from tests import gpu_checks as G
    lib = _lib.load()
    geo, X, Y, KN, P = G.make_case(B=B, seed=20 + B, scale=nslab_scale)
    d = G.dims_of(geo, B, 4)
    N, OT, ysz, L = geo["N"], geo["OT"], geo["y"], geo["L"]
    ns = lib.st_synth_frame_slabs(C.byref(d))
    nlp = lib.st_ola_loss_partials(C.byref(d)) // B      # slots per window (the entry returns the whole array's length)
    assert nlp == (ysz + 255) // 256
    rng = np.random.default_rng(7)
    frs_h = (0.05 * rng.standard_normal((ns, B * OT, N))).astype(np.float32)

    def run(off):      # off floats into each allocation: 0 -> every pointer 16-byte aligned, 1 -> none
        def buf(a):
            flat = torch.zeros(a.size + 8, dtype=torch.float32, device=G.DEV)
            v = flat[off:off + a.size]
            v.copy_(torch.from_numpy(a.reshape(-1)).to(G.DEV))
            return flat, v
        keep = []
        ptrs = []
        for a in (frs_h, X, Y, np.zeros((B, ysz), np.float32), np.zeros((B, ysz), np.float32), np.zeros((B * nlp,), np.float32)):
            flat, v = buf(a); keep.append((flat, v)); ptrs.append(C.c_void_p(v.data_ptr()))
            assert (v.data_ptr() % 16 == 0) == (off == 0)
        _lib.check(lib.st_ola_loss(C.byref(d), *ptrs, G.stream()), "ola")
        torch.cuda.synchronize()
        return [keep[i][1].cpu().numpy().copy() for i in (3, 4, 5)]

    ya, da, la = run(0)
    yb, db, lb = run(1)
    assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32)) and np.array_equal(da.view(np.uint32), db.view(np.uint32))
    assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), f"loss partials differ: max {np.abs(la - lb).max():.3e}"
    assert np.count_nonzero(la) == la.size      # every slot carries its own 256 samples (no zeroed slots)
    # ... and they are the log-cosh sums (float64 reference from the device's own y_hat)
    lc = np.zeros((B, nlp * 256)); lc[:, :ysz] = np.log(np.cosh(Y.astype(np.float64) - ya.reshape(B, ysz).astype(np.float64)))
    ref = lc.reshape(B, nlp, 256).sum(-1).reshape(-1)
    assert np.allclose(la, ref, rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("scale,shrink,B", [(1, 4, 3), (1, 2, 4), (8, 4, 2)])
@pytest.mark.parametrize("mode", ["f32", "f32x3", "bf16_all", "f16_all"])
def test_digital_silence_through_the_fused_step(mode, scale, shrink, B):
    """Edge case of the domain: DIGITAL SILENCE.  Window 0 is all zeros (input and target), window 1 starts with half a window of zeros -- frames that are exactly
    zero, then frames that straddle the onset.  At an exactly silent bin re = im = 0: mag = 0, phs = atan2(0, 1e-7) = 0 (nn_proc.py:309-310), d|.| = 0 by the
    sub-gradient the reference's autograd uses and d atan2 / d im = 1e7 (probed) -- finite, and multiplied by a frame of zeros in the analysis weight
    gradient, so a silent frame contributes exactly nothing; under fp16 that 1e7 x loss scale 4096 is what the polar backward saturates before the GEMM narrows it
    (inf x 0 would poison the basis gradient with NaN).  Forward, loss, all 40 gradient tensors, the clip norm and the parameters after one step against the oracle
    at the suite's own tolerances, in the four arithmetic families and at both windows of BASELINE.json."""
    import numpy as np
    from tests import gpu_checks as G
    geo, X, Y, KN, P = G.make_case(B, 77, K=4, scale=scale, shrink=shrink)
    X = X.copy(); Y = Y.copy()
    X[0] = 0.0; Y[0] = 0.0
    X[1, : X.shape[1] // 2] = 0.0

    def run():
        d = G.dims_of(geo, B, 4)
        restore = G.follow_effective_arithmetic(d)
        try:
            return G._run_fused(geo, X, Y, KN, P, d, B, 4, 1)
        finally:
            restore()
    if mode == "f32":
        res = run()
    elif mode == "f32x3":
        with G.split_mode(): res = run()
    elif mode == "bf16_all":
        with G.mixed_mode(2, half="bf16", tol_scale=G.mixed_mode.FUSED_TOL[2]): res = run()
    else:
        with G.mixed_mode(2, half="f16", tol_scale=G.mixed_mode.FUSED_TOL_F16[2] * (2.0 if scale == 8 else 1.0)): res = run()
    assert len(res) == 48
    assert all(np.isfinite(r["err"]) for r in res)
    bad = [r for r in res if not r["ok"]]
    assert not bad, [(r["name"], r["rel"], r["tol"]) for r in bad]


@pytest.mark.parametrize("scale,scheme,shrink", [(8, "lean", 4), (2, "lean", 4), (2, "legacy", 4), (1, "lean", 1)])
@pytest.mark.parametrize("mode", ["f32", "bf16_all", "f16_all"])
def test_model_without_knobs_on_the_wide_geometries(mode, scale, scheme, shrink):
    """num_knobs = 0 (golden G14 case 0 pins the oracle to the reference for it) where the autoencoders take the WIDE path (st_ae_wide.h: the 65536-sample window, lean and
    legacy scale 2, shrink 1): the knob rows of the layer-5 input are an empty job of prep_kernel there, the inner kernels mask their clamped knob load.  Fused step
    against the oracle at the mode's tolerances."""
    from tests import gpu_checks as G
    kw = dict(B=2, seed=41, K=0, steps=1, scale=scale, scheme=scheme, shrink=shrink)
    if mode == "f32":
        res = G.run_fused(**kw)
    else:
        half = "bf16" if mode == "bf16_all" else "f16"
        ftol = (G.mixed_mode.FUSED_TOL if half == "bf16" else G.mixed_mode.FUSED_TOL_F16)[2] * (2.0 if scale == 8 else 1.0)
        with G.mixed_mode(2, half=half, tol_scale=ftol):
            res = G.run_fused(**kw)
    bad = [r for r in res if not r["ok"]]
    assert not bad, [(r["name"], r["rel"], r["tol"]) for r in bad]
    if mode == "f32":              # ... and every per-op C entry (66 checks; the knob pointer of a knob-less model is the address of a resident zero: gpu_checks._run_all)
        per = G.run_all(B=2, seed=3, K=0, scale=scale, scheme=scheme, shrink=shrink)
        bad = [r for r in per if not r["ok"]]
        assert len(per) == 66 and not bad, [(r["name"], r["rel"], r["tol"]) for r in bad]

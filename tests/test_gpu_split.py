"""ST_PREC_F32X3 ("f32x3"): fp32 results from the bf16 matrix pipe -- operands as three bfloat16 planes, six partial products per
product, fp32 accumulation (include/signaltrain_hip.h, st_gemm_planes.h).  It is held to the UNMODIFIED fp32 oracle at the fp32
tolerances, and its error against float64 must not exceed that of the fp32 MFMA path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _assert_ok(res):
    bad = [r for r in res if not r["ok"]]
    assert not bad, "\n".join(f"{r['name']}: err={r['err']:.3e} scale={r['scale']:.3e} tol={r['tol']}" for r in bad)


@pytest.mark.parametrize("B,seed,K", [(3, 0, 4), (5, 2, 3), (2, 3, 7)])
def test_split_per_op_parity(B, seed, K):
    """Every per-op entry point (the in-kernel split of gemm_half_kernel PL = 3 for the three K-contiguous GEMMs)."""
    from tests import gpu_checks as G
    with G.split_mode():
        _assert_ok(G.run_all(B=B, seed=seed, K=K))


@pytest.mark.parametrize("B,seed,K,steps", [(3, 1, 4, 3), (7, 4, 2, 2)])
def test_split_fused_step_parity(B, seed, K, steps):
    """The fused step: pre-split bases (k-chunk-major planes written by wplanes_kernel) + activations split as they are staged."""
    from tests import gpu_checks as G
    with G.split_mode():
        _assert_ok(G.run_fused(B=B, seed=seed, K=K, steps=steps))


def test_split_scale8_and_legacy():
    from tests import gpu_checks as G
    with G.split_mode():
        _assert_ok(G.run_all(B=1, seed=3, K=4, scale=8))
        _assert_ok(G.run_fused(B=2, seed=5, K=4, steps=2, scale=8))
        _assert_ok(G.run_fused(B=3, seed=5, K=3, steps=2, scale=2, scheme="legacy"))


def test_split_is_fp32_grade():
    """Against the float64 oracle the split path is not less accurate than the fp32 MFMA path (forward outputs and gradients of a
    B = 6 batch through the fused entry points), and it is a different rounding, not the same bits."""
    import torch
    from tests import gpu_checks as G
    from oracle import st_oracle as O
    from signaltrain_amd.engine import StepEngine
    B, K = 6, 4
    geo, X, Y, KN, P = G.make_case(B, 21, K=K)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    loss, Gr, c = O.model_loss_bwd(X.astype(np.float64), KN.astype(np.float64), Y.astype(np.float64), P64, geo)
    y_ref, mag_ref, _ = O.model_fwd(X.astype(np.float64), KN.astype(np.float64), P64, geo)
    out = {}
    for mode in ("f32", "f32x3"):
        d = G.dims_of(geo, B, K)
        eng = StepEngine(d, G.DEV, compute_dtype=mode); eng.load_state_dict(P)
        y_hat, mag, mag_hat = eng.forward(G.t(X), G.t(KN))
        eng.loss_backward(G.t(X), G.t(KN), G.t(Y)); torch.cuda.synchronize()
        g = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in eng.layout.views(eng.grads).items()}
        out[mode] = dict(y=y_hat.cpu().numpy().astype(np.float64), mag=mag.cpu().numpy().astype(np.float64), g=g)
    rel = lambda a, r: float(np.abs(a - r).max() / max(np.abs(r).max(), 1e-300))
    e = {m: dict(y=rel(out[m]["y"], y_ref), mag=rel(out[m]["mag"], mag_ref),
                 g=max(rel(out[m]["g"][k], np.asarray(Gr[k], np.float64).reshape(out[m]["g"][k].shape)) for k in out[m]["g"])) for m in out}
    for q in ("y", "mag", "g"):
        assert e["f32x3"][q] <= 3.0 * e["f32"][q] + 5e-7, (q, e)      # same order (measured 0.6x .. 1.9x, tools/split_accuracy.py)
        assert e["f32x3"][q] < 2e-5, (q, e)
    assert not np.array_equal(out["f32"]["mag"], out["f32x3"]["mag"])      # really the other arithmetic


def test_split_trains_like_fp32():
    """End to end: 240 optimisation steps (B = 32, device-generated comp_4c windows, 1-cycle schedule) in f32 and in f32x3 from the same
    weights and data end at the same loss (training is chaotic: fp32 against itself from weights perturbed by 1e-6 differs by a few
    per cent at 1000 steps, tools/train_convergence.py / profiles/r02_train_convergence.txt)."""
    import torch
    from signaltrain_amd import _lib, nn_proc, audio, datasets, learningrate
    from signaltrain_amd.engine import StepEngine
    nn_proc._QUIET = True
    dev = torch.device("cuda:0"); B, STEPS = 32, 240
    torch.manual_seed(218); np.random.seed(218)
    sd = {k: v.detach().clone() for k, v in nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=4).state_dict().items()}
    ds = datasets.SynthAudioDataSet(8192, audio.Compressor_4c(), datapoints=STEPS * B, y_size=2048)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    x, y, kn = ds.batch_device(STEPS * B, dev, generator=gen)
    lrs, _ = learningrate.get_1cycle_schedule(lr_max=1e-3, n_data_points=STEPS * B, epochs=1, batch_size=B)
    d = _lib.geometry(1, 4, 4, B)
    tail = {}
    for mode in ("f32", "f32x3"):
        eng = StepEngine(d, dev, compute_dtype=mode); eng.load_state_dict(sd)
        losses = torch.zeros(STEPS, device=dev)
        for it in range(STEPS):
            sl = slice(it * B, (it + 1) * B)
            losses[it] = eng.train_step(x[sl], kn[sl], y[sl], float(lrs[max(it - 1, 0)]))[0]
        torch.cuda.synchronize()
        l = losses.cpu().numpy()
        assert np.isfinite(l).all() and l[-60:].mean() < 0.5 * l[:10].mean(), (mode, l[:10].mean(), l[-60:].mean())     # it trains
        tail[mode] = float(l[-60:].mean())
    assert abs(tail["f32x3"] / tail["f32"] - 1.0) < 0.15, tail


def test_split_known_limit_at_an_ill_conditioned_bin():
    """Where "fp32-grade" ends (round 6, found by the geometry-corner sweep: profiles/r06_fuzz_parity_geo_seed2026.txt).  L = 65536, shrink 2 (T = 174, OT = 89), B = 4, seed 243: the exact
    fp32 path is 4.5e-6 from the float64 oracle on every tensor; f32x3 misses the fixed 2e-4 on two -- the real analysis basis' gradient (9.5e-4) and the first layer of the phase
    autoencoder (2.3e-4) -- each in ONE row of the tensor, at 5.6 x that quantity's spread.  The six-term bfloat16 split carries ~2 x the STFT error of the fp32 MFMA path
    (profiles/r02_split_accuracy.txt: |STFT| 2.0e-6 vs 1.1e-6 against float64) and d atan2 amplifies it by 1 / mag at a near-silent bin.  Not a defect of a kernel (every other tensor of the
    step, every other row of these two, is inside the fp32 tolerance) and not hidden either: this test pins the SIZE of it -- localized (<= 2 rows), < 8 x the spread, < 5 x the tolerance
    -- so that the informational mode cannot drift further from the exact one without notice.  The headline and every parity claim are the exact-fp32 path's."""
    from tests import gpu_checks as G
    from tests import gpu_spread as S
    kw = dict(B=4, seed=243, K=2, steps=1, scale=8, scheme="lean", shrink=2)
    exact = G.run_fused(**kw)
    assert all(r["ok"] for r in exact) and max(r["rel"] / r["tol"] for r in exact) < 0.1          # the exact path: a tenth of every tolerance
    with G.split_mode():
        res = G.run_fused(**kw)
    miss = [r for r in res if not r["ok"]]
    sp = S.spread_of(kw)
    for r in miss:
        s = max(sp["f32"][r["name"]], sp["noise"][r["name"]])
        assert r["name"].startswith("grad.") and r.get("rows_over", 99) <= 2 and r["rel"] < 8.0 * s and r["rel"] < 5.0 * r["tol"], (r["name"], r["rel"], s, r.get("rows_over"))
    assert len(miss) <= 3

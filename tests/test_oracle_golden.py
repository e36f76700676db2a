"""CPU: the oracle (oracle/st_oracle.py) against the golden vectors captured from the imported
reference by tools/capture_golden.py.  This is what pins the oracle; it needs neither the
reference nor a GPU."""
import os
import numpy as np
import pytest
from oracle import st_oracle as O
from tests.golden_util import perturb_stft, projections, ae_keys, SAMPLE_ROWS, STFT_KEYS


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def close(a, b, rtol, what=""):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = max(float(np.max(np.abs(b))), 1e-30)
    err = float(np.max(np.abs(a - b)))
    assert err <= rtol * scale, f"{what}: max|d|={err:.3e} scale={scale:.3e}"


def golden_params(golden_dir, geo, fixture="g3_forward.npz", prefix="ae_", seed=7):
    g = load(golden_dir, fixture)
    P = O.init_params(geo, 4)
    for k in ae_keys():
        P[k] = g[prefix + k]
    perturb_stft(P, seed=seed)
    return P


def test_g1_geometry(golden_dir):
    tab = load(golden_dir, "g1_geometry.npz")["table"]
    assert len(tab) >= 20
    for s, sh, legacy, L, y, T, OT, N, H in tab:
        g = O.geometry(int(s), int(sh), "legacy" if legacy else "lean")
        assert (g["L"], g["y"], g["T"], g["OT"], g["N"], g["H"]) == (L, y, T, OT, N, H)
    g = O.geometry(1, 4)
    assert (g["L"], g["y"], g["T"], g["OT"]) == (8192, 2048, 25, 9)
    g = O.geometry(8, 4)
    assert (g["L"], g["y"], g["T"], g["OT"]) == (65536, 16256, 174, 46)


def test_g2_init_bases(golden_dir):
    g = load(golden_dir, "g2_init_bases.npz")
    P = O.init_params(O.geometry(1, 4), 4)
    np.testing.assert_allclose(O.hamming(1024), g["hamming"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(O.gla_window(1024, 384), g["gla"], rtol=0, atol=1e-14)
    for k in STFT_KEYS:
        w = P[k][:, 0, :]
        # <= 1 ulp f32 on sampled rows, and column sums as a whole-tensor fingerprint
        np.testing.assert_allclose(w[SAMPLE_ROWS], g["rows_" + k], rtol=0, atol=4e-9)
        np.testing.assert_allclose(w.astype(np.float64).sum(0), g["colsum_" + k], rtol=0, atol=2e-6)


def test_frame_indexing_bit_exact():
    """Integer contract: frame starts 384t-1024, OLA offsets 384t, crop [1024, 1024+y)."""
    geo = O.geometry(1, 4)
    st = O.frame_starts(geo["T"], geo["H"], geo["N"])
    assert st[0] == -1024 and st[1] == -640 and st[24] == 8192
    x = np.arange(1, 8193, dtype=np.float32)[None]
    fr = O.frames(x, 1024, 384, 25)
    assert np.all(fr[0, 0] == 0) and np.all(fr[0, 24] == 0)          # frames 0 and 24 are all padding
    assert fr[0, 3, 0] == x[0, 3 * 384 - 1024] and fr[0, 23, 383] == x[0, 8191] and fr[0, 23, 384] == 0
    frs = np.zeros((1, 9, 1024), np.float32); frs[0, 4, 10] = 1
    full = O.overlap_add(frs, 384)
    assert full.shape == (1, 4096) and full[0, 4 * 384 + 10] == 1 and full.sum() == 1


def test_g3_forward(golden_dir):
    g = load(golden_dir, "g3_forward.npz")
    geo = O.geometry(1, 4)
    P = golden_params(golden_dir, geo)
    y, mag, mag_hat, c = O.model_fwd(g["x"], g["knobs"], P, geo, return_all=True)
    close(y, g["y_hat"], 3e-6, "y_hat"); close(mag, g["mag"], 3e-6, "mag"); close(mag_hat, g["mag_hat"], 3e-6, "mag_hat")
    close(c["re"], g["re"], 3e-6, "re"); close(c["im"], g["im"], 3e-6, "im")
    close(c["Are"], g["an_real"], 5e-6, "an_real"); close(c["Aim"], g["an_imag"], 5e-6, "an_imag")
    close(c["syn"], g["x_fwdsyn"], 5e-6, "x_fwdsyn")
    # unfolded (literal flip/cat) synthesis == folded synthesis
    syn_lit = O.synthesis_fwd(c["Are"], c["Aim"], P[STFT_KEYS[2]], P[STFT_KEYS[3]], geo, folded=False)
    close(syn_lit, c["syn"], 5e-6, "fold")
    FB = g["act_bins"]
    for pref, hs, tol in (("m", c["hs_m"], 5e-6), ("p", c["hs_p"], 5e-5)):
        for j in range(9):
            o = hs[j + 1] if j < 4 else hs[j]
            if j == 3:
                o = o[:, :, :16]
            close(o[:, FB, :], g[f"{pref}_act{j}"], tol, f"{pref} act{j}")


def test_g4_backward(golden_dir):
    g3 = load(golden_dir, "g3_forward.npz"); g = load(golden_dir, "g4_backward.npz")
    geo = O.geometry(1, 4)
    P = golden_params(golden_dir, geo)
    X, KN, Y = (g3[k].astype(np.float64) for k in ("x", "knobs", "y"))
    loss, G, _ = O.model_loss_bwd(X, KN, Y, P, geo)
    close(loss, g["loss"], 3e-5, "loss")
    for k in ae_keys():
        close(G[k], g["g_" + k], 2e-5, k)
    PROJ = projections(seed=11)
    for k in STFT_KEYS:
        gk = G[k][:, 0, :]
        close(gk[SAMPLE_ROWS], g["rows_" + k], 2e-5, "rows " + k)
        close(gk[:, SAMPLE_ROWS], g["cols_" + k], 2e-5, "cols " + k)
        close(PROJ @ gk, g["proj_" + k], 2e-5, "proj " + k)
        close(np.abs(gk).sum(), g["l1_" + k], 1e-3, "l1 " + k)
    # structural facts of SURVEY.md a11: analysis rows >= 513 are exactly zero; synthesis Hermitian symmetry
    assert np.all(G[STFT_KEYS[0]][513:] == 0) and np.all(G[STFT_KEYS[1]][513:] == 0)
    k = np.arange(1, 512)
    np.testing.assert_array_equal(G[STFT_KEYS[2]][1024 - k, 0], G[STFT_KEYS[2]][k, 0])
    np.testing.assert_array_equal(G[STFT_KEYS[3]][1024 - k, 0], -G[STFT_KEYS[3]][k, 0])
    n, coef = O.clip_l1_stft({k: v.astype(np.float32) for k, v in G.items()})
    close(n, g["clip_norm"], 1e-3, "clip norm"); close(coef, g["clip_coef"], 1e-3, "clip coef")


def test_backward_finite_difference():
    """Independent of the reference: hand-derived backward vs central differences (float64, tiny B)."""
    geo = O.geometry(1, 4)
    rng = np.random.default_rng(0)
    P = O.init_params(geo, 4, rng)
    perturb_stft(P, seed=3)
    P = {k: v.astype(np.float64) for k, v in P.items()}
    X = rng.standard_normal((1, geo["L"])) * 0.3
    KN = rng.uniform(-.5, .5, (1, 4)); Y = rng.standard_normal((1, geo["y"])) * 0.3
    loss, G, _ = O.model_loss_bwd(X, KN, Y, P, geo)
    probes = [("mpaec.aenc.fnn_enc2.weight", (3, 5)), ("mpaec.phs_aenc.fnn_addknobs.weight", (2, 18)),
              ("mpaec.phs_aenc.fnn_dec.bias", (4,)), (STFT_KEYS[0], (37, 0, 500)), (STFT_KEYS[3], (100, 0, 411)),
              (STFT_KEYS[2], (1024 - 100, 0, 411))]
    for k, idx in probes:
        h = 1e-6 * max(1.0, abs(P[k][idx]))
        old = P[k][idx]
        P[k][idx] = old + h; lp = O.model_loss_bwd(X, KN, Y, P, geo)[0]
        P[k][idx] = old - h; lm = O.model_loss_bwd(X, KN, Y, P, geo)[0]
        P[k][idx] = old
        fd = (lp - lm) / (2 * h)
        assert abs(fd - G[k][idx]) <= 2e-4 * max(abs(fd), abs(G[k][idx])) + 1e-12, (k, idx, fd, G[k][idx])


def test_g5_adam_steps(golden_dir):
    g3 = load(golden_dir, "g3_forward.npz"); g = load(golden_dir, "g5_adam.npz")
    geo = O.geometry(1, 4)
    P = golden_params(golden_dir, geo)
    P = {k: P[k].copy() for k in O.param_order()}
    M = {k: np.zeros_like(v) for k, v in P.items()}; V = {k: np.zeros_like(v) for k, v in P.items()}
    lrs, _ = O.get_1cycle_schedule(lr_max=1e-3, n_data_points=200, epochs=1, batch_size=2)
    np.testing.assert_allclose(lrs[:4], g["lrs"], rtol=1e-14)
    PROJ = projections(seed=11)
    lr = lrs[0]
    for it in range(3):
        Xi = np.roll(g3["x"], 17 * it, axis=1).copy(); Yi = np.roll(g3["y"], 17 * it, axis=1).copy()
        assert lr == g[f"lr_used{it}"]        # lr used at iteration i is lr_sched[max(i-1,0)] (train.py:150)
        loss, _, _ = O.train_step(Xi, g3["knobs"], Yi, P, M, V, it + 1, lr, geo)
        lr = lrs[it]
        close(loss, g[f"loss{it}"], 3e-5, f"loss{it}")
        for k in ae_keys():
            np.testing.assert_allclose(P[k], g[f"s{it}_" + k], rtol=0, atol=6e-7, err_msg=k)
        for k in STFT_KEYS:
            np.testing.assert_allclose(P[k][SAMPLE_ROWS, 0, :], g[f"s{it}_rows_" + k], rtol=0, atol=2e-7)
            close(PROJ @ P[k][:, 0, :].astype(np.float64), g[f"s{it}_proj_" + k], 1e-5, "proj")


def test_g6_1cycle(golden_dir):
    g = load(golden_dir, "g6_1cycle.npz")
    lr, mom = O.get_1cycle_schedule(lr_max=1e-4, n_data_points=200000, epochs=100, batch_size=200)
    assert len(lr) == int(g["n"]) == 100000
    np.testing.assert_allclose(lr[g["idx"]], g["lr"], rtol=1e-15)
    np.testing.assert_allclose(mom[g["idx"]], g["mom"], rtol=1e-15)


def test_g7_dct_variant(golden_dir):
    g = load(golden_dir, "g7_dct.npz")
    W = O.dct_bases(1024, 2048)
    np.testing.assert_array_equal(W[SAMPLE_ROWS], g["basis_rows"])
    xft = O.dct_analysis_fwd(g["x"], W, g["bias"])
    assert xft.shape == (1, 9, 1024)
    close(xft[:, :, ::8], g["xft"], 3e-6, "dct analysis")
    wav = O.dct_synthesis_fwd(xft, W)
    assert wav.shape == (1, 1, 8192)
    close(wav, g["wav"], 5e-6, "dct synthesis")


def test_g8_scale8(golden_dir):
    g = load(golden_dir, "g8_scale8.npz")
    geo = O.geometry(8, 4)
    P = golden_params(golden_dir, geo, "g8_scale8.npz", "ae_", seed=9)
    y, mag, mag_hat = O.model_fwd(g["x"], g["knobs"], P, geo)
    assert y.shape == (1, 16256) and mag_hat.shape == (1, 46, 513)
    close(y, g["y_hat"], 5e-6, "y"); close(mag_hat[:, :, ::4], g["mag_hat"], 5e-6, "mag_hat")


def test_g9_compressor(golden_dir):
    g = load(golden_dir, "g9_compressor.npz")
    y = O.compressor_4controls(g["x"].copy(), *g["knobs"][:4], sr=g["knobs"][4])
    close(y, g["y"], 1e-6, "compressor_4controls")


def test_perfect_reconstruction_at_init():
    """Implicit invariant of the reference's init: analysis -> synthesis is identity."""
    geo = O.geometry(1, 1)                    # shrink 1: y = 8064 of the 8192 samples
    P = O.init_params(geo, 4)
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((1, geo["L"])) * 0.3).astype(np.float32)
    re, im = O.analysis_fwd(x, P[STFT_KEYS[0]], P[STFT_KEYS[1]], geo)
    syn = O.synthesis_fwd(re[:, geo["T"] - geo["OT"]:], im[:, geo["T"] - geo["OT"]:], P[STFT_KEYS[2]], P[STFT_KEYS[3]], geo)
    assert np.max(np.abs(syn - x[:, geo["L"] - geo["y"]:])) < 5e-6

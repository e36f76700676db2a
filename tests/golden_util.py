"""Helpers shared by tools/capture_golden.py and the tests that consume tests/golden/*.npz.

The 4 STFT tensors are 4 MiB each, so fixtures hold only sampled rows/columns, column sums and
random projections; the full "learned" tensors are regenerated as init bases (public formulas,
checked against the sampled rows) + a portable integer-hash perturbation (no RNG library state).
"""
import numpy as np

SAMPLE_ROWS = np.array([0, 1, 7, 256, 512, 513, 1023])
AE_LAYERS = ("fnn_enc", "fnn_enc2", "fnn_enc3", "fnn_enc4", "fnn_addknobs",
             "fnn_dec4", "fnn_dec3", "fnn_dec2", "fnn_dec")
STFT_KEYS = ("mpaec.dft_analysis.conv_analysis_real.weight",
             "mpaec.dft_analysis.conv_analysis_imag.weight",
             "mpaec.dft_synthesis.conv_synthesis_real.weight",
             "mpaec.dft_synthesis.conv_synthesis_imag.weight")


def ae_keys():
    return [f"mpaec.{ae}.{n}.{wb}" for ae in ("aenc", "phs_aenc") for n in AE_LAYERS for wb in ("weight", "bias")]


def hash_uniform(shape, seed):
    """Deterministic uniform[-0.5,0.5) field from pure uint64 arithmetic (portable)."""
    r = np.arange(shape[0], dtype=np.uint64)[:, None]
    c = np.arange(shape[1], dtype=np.uint64)[None, :]
    h = (r * np.uint64(73856093)) ^ (c * np.uint64(19349663)) ^ np.uint64(seed * 83492791 + 12345)
    h = (h * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    return (h.astype(np.float64) / 4294967296.0 - 0.5)


def perturb_stft(P, seed, amp=2e-3):
    """In place: STFT weights <- init + amp * hash noise, so the bases are 'learned' (not FFT-able)."""
    for j, k in enumerate(STFT_KEYS):
        w = P[k]
        n = hash_uniform(w[:, 0, :].shape, seed * 10 + j)
        P[k] = (w.astype(np.float64) + amp * n[:, None, :]).astype(np.float32)
    return P


def sample_rows(w):
    return w[SAMPLE_ROWS]


def projections(seed, n=4, size=1024):
    """n deterministic probe vectors (rows) used to fingerprint 1024x1024 tensors."""
    return hash_uniform((n, size), seed) * 2.0


def g14_case(ci):
    """The three edge cases of golden G14 (tools/capture_golden_r6.py), regenerated from portable generators: (geo, X, Y, KN, P, K).
    0: a model without knobs; 1: one knob; 2: digital silence -- window 0 all zeros (input and target), window 1 half a window of zeros, window 2 plain."""
    from tests import gpu_checks as G          # make_case only (numpy)
    if ci == 0:
        K = 0; geo, X, Y, KN, P = G.make_case(2, 5, K=K)
    elif ci == 1:
        K = 1; geo, X, Y, KN, P = G.make_case(2, 5, K=K)
    else:
        K = 4; geo, X, Y, KN, P = G.make_case(3, 77, K=K)
        X = X.copy(); Y = Y.copy()
        X[0] = 0.0; Y[0] = 0.0
        X[1, : X.shape[1] // 2] = 0.0
    return geo, X, Y, KN, P, K

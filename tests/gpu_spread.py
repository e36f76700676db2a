"""The spread of the checked QUANTITY itself, per configuration (CPU only; numpy + the oracle).

tests/gpu_checks.py holds every gradient tensor of a fused step to 2e-4 of the tensor maximum against the float64 oracle.  For the analysis-basis
gradients that tolerance is not always meaningful: d atan2(im, re) = (-im, re) / (re^2 + im^2) (nn_proc.py:309-310 under autograd) turns an absolute
error e in re / im -- fp32 GEMM rounding, ~1e-7 of the largest bin -- into a relative error e / mag of that bin's gradient, and near-silent bins
(mag ~ 1e-5 of the maximum) exist in comp_4c windows.  Rounds 1-4 filtered such misses BY TENSOR NAME; this module measures instead, for the
configuration at hand, how far the oracle moves when its OWN arithmetic changes:

    f32     the oracle in float32 arithmetic vs the oracle in float64, same fp32 inputs (the reference, PyTorch fp32, is on this side),
    noise   the float64 oracle under `npert` independent 1e-6 relative perturbations of inputs and parameters, max over draws,

both as max|difference| / max|tensor| per gradient tensor ("grad.<name>") and forward output ("fwd.y_hat", "fwd.mag_hat", "fwd.mag": the atan2 branch cut), as |difference| / |loss| for "step.loss", and as max absolute parameter
difference after one clip + Adam step ("train0.params").  gpu_checks.grounded() accepts a miss of the fixed tolerance only when the device's error
is within 3 x that spread -- for ANY tensor, no names.
"""
import numpy as np
from oracle import st_oracle as O


def _step_params(G, P, lr):
    """Parameters after one clip + Adam step from gradients G (train.py:136-147 ordering), float32 like the optimizer."""
    g = {k: (np.asarray(v, np.float64) / O.LOSS_SCALE).astype(np.float32) for k, v in G.items()}
    O.clip_l1_stft(g)
    Pq = {k: np.asarray(P[k], np.float32).copy() for k in O.param_order()}
    M = {k: np.zeros_like(v) for k, v in Pq.items()}; V = {k: np.zeros_like(v) for k, v in Pq.items()}
    O.adam_step(Pq, g, M, V, 1, lr)
    return Pq


def oracle_spread(B, seed, K=4, scale=1, scheme="lean", shrink=4, npert=8, steps=1):
    from tests import gpu_checks as G                     # make_case only (numpy); no GPU touched
    geo, X, Y, KN, P = G.make_case(B, seed, K=K, scale=scale, scheme=scheme, shrink=shrink)
    lr = O.get_1cycle_schedule(lr_max=1e-3, n_data_points=200, epochs=1, batch_size=2)[0][0]
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    X64, K64, Y64 = X.astype(np.float64), KN.astype(np.float64), Y.astype(np.float64)
    l0, G0, c0 = O.model_loss_bwd(X64, K64, Y64, P64, geo)
    p0 = _step_params(G0, P, lr)

    def diff(l1, G1, c1):
        d = {"grad." + k.replace("mpaec.", ""): float(np.abs(G0[k] - G1[k]).max() / max(np.abs(G0[k]).max(), 1e-30)) for k in G0}
        # the forward outputs have a spread too, rarely: atan2 is DISCONTINUOUS at its branch cut (re < 0, im ~ 0: phs jumps by 2 pi) and the phase autoencoder is
        # not 2 pi-periodic, so a bin on the cut moves that window's y_hat by ~1e-4 under a 1e-6 perturbation (nn_proc.py:310, :326; found by the round-5 sweep at B = 64)
        for nm, key in (("fwd.y_hat", "out"), ("step.y_hat", "out"), ("fwd.mag_hat", "mag_hat"), ("fwd.mag", "mag")):
            d[nm] = float(np.abs(np.asarray(c0[key], np.float64) - np.asarray(c1[key], np.float64)).max() / max(np.abs(c0[key]).max(), 1e-30))
        d["step.loss"] = abs(l0 - l1) / abs(l0)
        # the published clip norm (sum |g| over the four STFT tensors, nn_proc.py:299-302) inherits the spread of the analysis-basis gradients: one near-silent bin whose row moves by
        # 20 % of the tensor maximum moves the norm by 1 % (round 6: the sweep's seed-719 window; until then the norm was the one checked quantity without a spread)
        n0, n1 = (sum(float(np.abs(np.asarray(G[k], np.float64)).sum()) for k in O.param_order()[:4]) for G in (G0, G1))
        d["step.l1norm"] = abs(n0 - n1) / max(abs(n0), 1e-30)
        p1 = _step_params(G1, P, lr)
        d["train0.params"] = float(max(np.abs(p0[k].astype(np.float64) - p1[k]).max() for k in p0))
        return d
    l32, G32, c32 = O.model_loss_bwd(X, KN, Y, P, geo)
    out = {"f32": diff(float(l32), {k: v.astype(np.float64) for k, v in G32.items()}, c32), "noise": {}}
    for s in range(npert):
        rng = np.random.default_rng(1000 + s)
        Pp = {k: v * (1 + 1e-6 * rng.standard_normal(v.shape)) for k, v in P64.items()}
        Xp = X64 * (1 + 1e-6 * rng.standard_normal(X64.shape))
        l1, G1, c1 = O.model_loss_bwd(Xp, K64, Y64, Pp, geo)
        for k, v in diff(l1, G1, c1).items():
            out["noise"][k] = max(out["noise"].get(k, 0.0), v)
    return out


# ----------------------------------------------------------------------------- grading a run against the spread
import json, os
CACHE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_fuzz_f32_spread.json")


def case_tag(kw):
    return f"f32 B={kw['B']} K={kw.get('K', 4)} scale={kw.get('scale', 1)} {kw.get('scheme', 'lean')} shrink={kw.get('shrink', 4)} seed={kw['seed']}"


_MEM = {}      # spreads computed by THIS process for configurations the committed table does not hold


def spread_of(kw, npert=8, cache=CACHE, write=False):
    """oracle_spread(**kw), through the committed cache (fp32 arithmetic modes only: the rounding switches of the oracle are not part of the key).
    The committed file is READ-ONLY here (pytest must not rewrite a tracked file, and the unlocked read-modify-write raced under
    pytest-xdist): a configuration it does not hold is computed and kept in process memory.  Only tools/fuzz_ground_f32.py passes write=True."""
    plain = O.GEMM_ROUND is None and O.AE_ROUND is None and O.LOSS_SCALE == 1.0 and not O.CLIP_ALL
    tab = json.load(open(cache)) if (plain and os.path.isfile(cache)) else {}
    tg = case_tag(kw)
    if tg in tab:
        return tab[tg]
    if plain and tg in _MEM:
        return _MEM[tg]
    sp = oracle_spread(npert=npert, **{k: kw[k] for k in ("B", "seed", "K", "scale", "scheme", "shrink") if k in kw})
    if plain:
        _MEM[tg] = sp
        if write:
            tab[tg] = sp
            tmp = cache + f".tmp{os.getpid()}"
            json.dump(tab, open(tmp, "w"), indent=1, sort_keys=True)
            os.replace(tmp, cache)
    return sp


CAP_TOL = 10.0      # a grounded miss may never exceed this many times the check's own fixed tolerance, whatever the spread says ...
LOCAL_ROWS = 16     # ... unless it is LOCALIZED (round 6): at most this many rows (frequency bins) of a weight-gradient tensor hold an element over the fixed tolerance.  A single
                    # window at a long geometry can have a spread of 1e-2 on an analysis-basis gradient (profiles/r06_fuzz_grounding_f32.txt: f32x3, B = 1, lean scale 2, seed 499:
                    # device 2.9e-3 = 0.3 x spread = 14 x the tolerance); what the cap is there to stop -- a wrong tile / slab / k range hiding under a large spread -- covers
                    # 96 or 128 rows at least, so "within mult x spread AND in <= 16 rows" is a STRONGER statement about the kernels than the cap alone


def grounded(res, kw, mult=3.0, npert=8):
    """Re-grade the misses of gpu_checks.run_fused(**kw): a check that missed its fixed tolerance passes iff the device's error is within `mult` x the
    spread of that quantity for THIS configuration (max of f32-vs-f64 and self-noise).  Applies to every tensor the spread covers (all 40 gradient
    tensors, the loss, the parameters after the first step) -- no tensor is exempt by name, and checks the spread does not cover stay failures.
    The accepted error is capped at CAP_TOL x the check's fixed tolerance, except for a miss confined to <= LOCAL_ROWS rows of a weight-gradient tensor (see LOCAL_ROWS).  Returns the list of checks that remain failed; every re-graded check carries
    'spread' and 'ratio'."""
    bad = [r for r in res if not r["ok"]]
    if not bad:
        return []
    sp = spread_of(kw, npert=npert)
    still = []
    for r in bad:
        key = "train0.params" if r["name"].startswith("train0.params") else r["name"]
        if key not in sp["f32"]:
            still.append(r); continue
        s = max(sp["f32"][key], sp["noise"].get(key, 0.0))
        r["spread"] = s; r["ratio"] = r["rel"] / s if s > 0 else float("inf")
        # accepted bound = min(mult x spread, CAP_TOL x the fixed tolerance): at a configuration with a large self-noise (a bin on atan2's branch cut moves
        # y_hat by 2.4e-4, a near-silent bin moves an analysis-basis gradient by 1e-2) a genuinely wrong result of that size must not pass
        if r["rel"] <= min(mult * s, CAP_TOL * r["tol"]):
            r["ok"] = True; r["grounded"] = True
        elif r["rel"] <= mult * s and 0 < r.get("rows_over", 0) <= LOCAL_ROWS and r.get("rows", 0) >= 8 * LOCAL_ROWS:
            r["ok"] = True; r["grounded"] = True; r["localized"] = True        # over the cap, inside the spread, confined to a few bins
        else:
            still.append(r)
    return still

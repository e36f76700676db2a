"""Stage-by-stage parity checks of the HIP path against the oracle (needs a GPU).

Used by tests/test_gpu_parity.py (asserting) and tools/gpu_diag.py (verbose report).  Each check
calls the C ABI through ctypes on seeded inputs, computes the same quantity with oracle/st_oracle.py
in float64 *from the same fp32 inputs*, and returns (name, max_abs_err, scale, tolerance).
Tolerances are relative to max|reference| of the tensor (north_star: 1e-4 rel fp32).
"""
import ctypes as C
import numpy as np
import torch

from oracle import st_oracle as O
from signaltrain_amd import _lib
from signaltrain_amd.engine import StepEngine, ParamLayout, STFT_KEYS
from tests.golden_util import perturb_stft

DEV = "cuda:0"
TOL = 1e-4           # BASELINE.json north_star: <= 1e-4 relative, fp32


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def n(x):
    return x.detach().cpu().numpy().astype(np.float64)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_case(B=3, seed=0, scale=1, shrink=4, K=4, pert=7, scheme="lean"):
    """Seeded comp_4c-shaped inputs + 'learned' (perturbed) parameters, as fp32 numpy."""
    geo = O.geometry(scale, shrink, scheme)
    rng = np.random.default_rng(seed)
    X, Y, KN = O.synth_comp4c_batch(B, geo["L"], geo["y"], rng)
    if K != 4:
        KN = (rng.beta(0.8, 0.8, size=(B, K)) - 0.5).astype(np.float32)
    P = O.init_params(geo, K, np.random.default_rng(seed + 100))
    for k in P:                                   # non-zero biases so that bias paths are exercised
        if k.endswith(".bias"):
            P[k] = (0.05 * rng.standard_normal(P[k].shape)).astype(np.float32)
    perturb_stft(P, seed=pert)
    return geo, X, Y, KN, P


def dims_of(geo, B, K):
    """st_dims of the case; the arithmetic of the mode the checks currently run under (mixed_mode) travels in it."""
    d = _lib.st_dims()
    d.B, d.L, d.N, d.H, d.T, d.OT, d.F, d.K, d.y = B, geo["L"], geo["N"], geo["H"], geo["T"], geo["OT"], geo["F"], K, geo["y"]
    d.prec, d.loss_scale, d.clip_all = PREC_LEVEL, LOSS_SCALE, int(CLIP_ALL)
    return d


def follow_effective_arithmetic(d):
    """No silent arithmetic switch (st_effective_prec): where the geometry / batch cannot take 16-bit autoencoder layers (odd batch on the wide
    path) the library runs them in fp32 and SAYS so -- the oracle then rounds the GEMM operands only.  Returns a restore callable."""
    eff = int(_lib.load().st_effective_prec(C.byref(d)))
    saved = O.AE_ROUND
    if eff != d.prec and eff in (1, 3):
        O.AE_ROUND = None

    def restore():
        O.AE_ROUND = saved
    return restore


def to_kp(re, im, KP):
    """[R,F] pair -> padded [R,KP] (re at 0.., im at KP/2..)."""
    R, F = re.shape
    out = np.zeros((R, KP), re.dtype)
    out[:, :F] = re; out[:, KP // 2:KP // 2 + F] = im
    return out


def from_kp(a, F):
    KP = a.shape[1]
    return a[:, :F], a[:, KP // 2:KP // 2 + F]


PREC_LEVEL = 0           # st_dims.prec the checks run under (mixed_mode sets it); carried per call, nothing process-wide
LOSS_SCALE = 0.0         # st_dims.loss_scale (0 = none)
CLIP_ALL = False         # st_dims.clip_all
TOL_SCALE = 1.0          # multiplies every tolerance (mixed-precision runs: see mixed_mode)
ENGINE_DTYPE = "f32"     # compute_dtype of the StepEngines the checks create


class mixed_mode:
    """Context: the checks run with 16-bit operands in the HIP library (st_dims.prec) + the oracle's matching rounding.
    level 1: the STFT GEMMs; level 2: also the nine Linear layers of both autoencoders (the BF instantiations of st_ae.h /
    oracle.AE_ROUND).  half = "bf16" (BASELINE configs[2], [3]) or "f16" (configs[4]; runs with a loss scale and, like the
    reference's Apex branch train.py:136, the clip over all parameters).  Tolerances are widened (10x / 20x): both sides round
    the SAME quantities, but an operand that differs by 1e-6 between the two can land on the other side of a rounding boundary
    (bf16: 1 in ~4000 elements does, each then differs by 0.4 %; fp16 has 3 more mantissa bits: 0.05 %), and at level 2 those
    flips propagate through nine layers.  FUSED runs (device and oracle each consume their own intermediates) need more: the
    oracle against itself under a 1e-6 perturbation differs by up to 1.4e-2 at bf16 level 2 (tools/bf16_noise_floor.py) --
    FUSED_TOL below."""
    FUSED_TOL = {1: 30.0, 2: 200.0}          # tol_scale for run_fused under the two levels (3e-3 / 2e-2), bf16
    FUSED_TOL_F16 = {1: 10.0, 2: 40.0}       # fp16: 8x finer rounding

    def __init__(self, level=1, tol_scale=None, half="bf16", loss_scale=None, clip_all=None, oracle_rounds=True):
        assert half in ("bf16", "f16") and level in (1, 2)
        self.level, self.half = level, half
        # oracle_rounds = False: the device runs the 16-bit mode against the UNROUNDED fp32-grade oracle (same loss scale / clip scope: those are the
        # reference's Apex semantics, train.py:133-136, not roundings) -- the loose check SURVEY.md section 5 prescribes for the modes that have no
        # runnable reference here; tolerances then are the stated per-class bounds of tests/test_gpu_parity.py::test_16bit_modes_against_the_unrounded_oracle
        self.oracle_rounds = bool(oracle_rounds)
        self.tol_scale = tol_scale          # override, e.g. the 65536-sample geometry at level 2 (174-frame rows: more flips per sum)
        self.loss_scale = (4096.0 if half == "f16" else 0.0) if loss_scale is None else float(loss_scale)
        self.clip_all = (half == "f16") if clip_all is None else bool(clip_all)

    def __enter__(self):
        global TOL_SCALE, ENGINE_DTYPE, PREC_LEVEL, LOSS_SCALE, CLIP_ALL
        rnd = O.bf16_round if self.half == "bf16" else O.fp16_round
        PREC_LEVEL = {("bf16", 1): 1, ("bf16", 2): 2, ("f16", 1): 3, ("f16", 2): 4}[(self.half, self.level)]
        LOSS_SCALE, CLIP_ALL = self.loss_scale, self.clip_all
        O.GEMM_ROUND = rnd if self.oracle_rounds else None
        O.AE_ROUND = rnd if (self.level >= 2 and self.oracle_rounds) else None
        O.LOSS_SCALE = self.loss_scale if self.loss_scale > 0 else 1.0
        O.CLIP_ALL = self.clip_all
        base = (10.0 if self.level == 1 else 20.0) if self.half == "bf16" else (3.0 if self.level == 1 else 10.0)
        TOL_SCALE = self.tol_scale if self.tol_scale else base
        ENGINE_DTYPE = self.half + ("" if self.level == 1 else "_all")

    def __exit__(self, *a):
        global TOL_SCALE, ENGINE_DTYPE, PREC_LEVEL, LOSS_SCALE, CLIP_ALL
        PREC_LEVEL, LOSS_SCALE, CLIP_ALL = 0, 0.0, False
        O.GEMM_ROUND = None; O.AE_ROUND = None; O.LOSS_SCALE = 1.0; O.CLIP_ALL = False; TOL_SCALE = 1.0; ENGINE_DTYPE = "f32"


bf16_mode = mixed_mode       # round-1 name


class split_mode:
    """Context: the checks run with st_dims.prec = ST_PREC_F32X3 (fp32 operands as three bfloat16 planes, six partial products
    per product on the bf16 matrix pipe) against the UNMODIFIED fp32 oracle at the fp32 tolerances: the split is a different
    rounding of fp32 arithmetic, not a lower precision."""

    def __enter__(self):
        global ENGINE_DTYPE, PREC_LEVEL
        PREC_LEVEL, ENGINE_DTYPE = 5, "f32x3"

    def __exit__(self, *a):
        global ENGINE_DTYPE, PREC_LEVEL
        PREC_LEVEL, ENGINE_DTYPE = 0, "f32"


def new_engine(d, **kw):
    """A StepEngine with the arithmetic of the current mode."""
    return StepEngine(d, DEV, compute_dtype=ENGINE_DTYPE, loss_scale=(LOSS_SCALE if LOSS_SCALE > 0 else 1.0), clip_all=CLIP_ALL, **kw)


def err(name, got, ref, tol=TOL, scale=None):
    tol = tol * TOL_SCALE
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    sc = float(np.max(np.abs(ref))) if scale is None else float(scale)
    d = np.abs(got - ref)
    bad = ~np.isfinite(got)
    e = float(np.max(np.where(bad, np.inf, d))) if got.size else 0.0
    worst = np.unravel_index(int(np.argmax(np.where(bad, np.inf, d))), d.shape) if got.size else ()
    r = dict(name=name, err=e, scale=sc, rel=e / max(sc, 1e-30), tol=tol, ok=bool(e <= tol * max(sc, 1e-30)),
             worst=tuple(int(i) for i in worst), got=float(got[worst]) if got.size else 0.0,
             ref=float(ref[worst]) if got.size else 0.0)
    if got.ndim >= 2 and got.size and not r["ok"]:
        # WHERE a miss sits (round 6): how many leading-index slices ("rows": output channels of a weight tensor, i.e. frequency bins of the STFT bases) hold an
        # element over the tolerance, and the error of everything outside the worst few.  A conditioning miss (one near-silent bin under d atan2) lives in a
        # handful of rows; a kernel defect (a wrong tile, k range or slab) covers a tile's worth -- 96 / 128 rows or every row (tests/gpu_spread.py grounded()).
        per_row = np.where(bad, np.inf, d).reshape(d.shape[0], -1).max(axis=1)
        r["rows"] = int(d.shape[0]); r["rows_over"] = int((per_row > tol * max(sc, 1e-30)).sum())
    return r


def phase_err(name, got, ref, mag, tol=TOL):
    tol = tol * TOL_SCALE
    """Phase compared modulo 2*pi and weighted by mag/max(mag): atan2 is ill-conditioned where mag ~ 0
    and discontinuous at +-pi (SURVEY.md section 7 'hard parts')."""
    dphi = np.angle(np.exp(1j * (np.asarray(got, np.float64) - ref)))
    w = mag / max(float(mag.max()), 1e-30)
    e = np.abs(dphi) * w
    worst = np.unravel_index(int(np.argmax(e)), e.shape)
    return dict(name=name, err=float(e.max()), scale=1.0, rel=float(e.max()), tol=tol, ok=bool(e.max() <= tol),
                worst=tuple(int(i) for i in worst), got=float(got[worst]), ref=float(ref[worst]))


# --------------------------------------------------------------------------------------------- stages
def run_all(B=3, seed=0, K=4, verbose=False, scale=1, scheme="lean", shrink=4):
    """Run every per-op entry point on oracle-provided inputs; returns list of result dicts."""
    lib = _lib.load()
    geo, X, Y, KN, P = make_case(B, seed, K=K, scale=scale, scheme=scheme, shrink=shrink)
    d = dims_of(geo, B, K)
    restore_oracle = follow_effective_arithmetic(d)
    try:
        return _run_all(lib, geo, X, Y, KN, P, d, B, K, verbose)
    finally:
        restore_oracle()


def _run_all(lib, geo, X, Y, KN, P, d, B, K, verbose):
    KP = lib.st_kp(d.F); F, T, OT, N = d.F, d.T, d.OT, d.N
    res = []
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    loss, G, c = O.model_loss_bwd(X.astype(np.float64), KN.astype(np.float64), Y.astype(np.float64), P64, geo)

    lay = ParamLayout(d)
    flat = torch.zeros(lay.total, device=DEV)
    for k, v in lay.views(flat).items():
        v.copy_(t(P[k]).reshape(v.shape))
    V = lay.views(flat)
    Wr, Wi, Sr, Si = (V[k] for k in STFT_KEYS)
    ae_m = flat[lay.offsets[4]:lay.offsets[22]]; ae_p = flat[lay.offsets[22]:]
    PG = lay.offsets[22] - lay.offsets[4]
    x, kn, y = t(X), t(KN), t(Y)
    if K == 0:
        kn = None                            # a model without knobs: an empty tensor has no address; the ABI takes NULL for K == 0
    z = lambda *s: torch.zeros(*s, device=DEV)

    # 1. analysis + polar
    re, im, mag, phs = z(B, T, F), z(B, T, F), z(B, T, F), z(B, T, F)
    _lib.check(lib.st_analysis_fwd(C.byref(d), _lib.ptr(x), _lib.ptr(Wr), _lib.ptr(Wi), 0.5, _lib.ptr(re), _lib.ptr(im),
                                   _lib.ptr(mag), _lib.ptr(phs), stream()), "analysis_fwd")
    sc = float(np.abs(c["mag"]).max())
    res += [err("analysis.re", n(re), c["re"], scale=sc), err("analysis.im", n(im), c["im"], scale=sc),
            err("analysis.mag", n(mag), c["mag"]), phase_err("analysis.phs", n(phs), c["phs"], c["mag"])]
    # frame indexing is bit-exact: zero-padded frames give exact zeros
    dead = [tt for tt in range(T) if d.H * tt - d.N + d.N <= 0 or d.H * tt - d.N >= d.L]      # frames wholly inside the Conv1d padding
    assert 0 in dead
    res.append(err("analysis.zero_frames", n(re)[:, dead], np.zeros((B, len(dead), F)), scale=1.0, tol=0.0))

    # 2. autoencoders forward (oracle inputs)
    mag_o, phs_o = t(c["mag"]), t(c["phs"])
    mag_hat, phs_hat, AA = z(B, OT, F), z(B, OT, F), z(B * OT, KP)
    regp = z(lib.st_ae_fwd_partials(C.byref(d)))
    aefws = z(max(1, lib.st_ae_fwd_ws_floats(C.byref(d))))       # 0 floats needed unless the geometry is wide (T > 32 or OT > 16)
    _lib.check(lib.st_ae_fwd(C.byref(d), _lib.ptr(mag_o), _lib.ptr(phs_o), _lib.ptr(kn), _lib.ptr(ae_m), _lib.ptr(ae_p),
                             _lib.ptr(mag_hat), _lib.ptr(phs_hat), _lib.ptr(AA), _lib.ptr(regp), _lib.ptr(aefws), stream()), "ae_fwd")
    aa_re, aa_im = from_kp(n(AA), F)
    sa = float(max(np.abs(c["Are"]).max(), np.abs(c["Aim"]).max()))
    w = O.freq_weights(F, np.float64)
    res += [err("ae_fwd.mag_hat", n(mag_hat), c["mag_hat"]), err("ae_fwd.phs_hat", n(phs_hat), c["phs_hat"]),
            err("ae_fwd.an_real", aa_re.reshape(B, OT, F), c["Are"], scale=sa),
            err("ae_fwd.an_imag", aa_im.reshape(B, OT, F), c["Aim"], scale=sa),
            err("ae_fwd.pads", n(AA)[:, F:KP // 2], 0 * n(AA)[:, F:KP // 2], scale=1.0, tol=0.0),
            err("ae_fwd.reg_sum", n(regp).sum(), np.abs(c["mag_hat"] * w).sum())]

    # 3. fold + synthesis GEMM + OLA/loss
    Sfold = z(KP, N)
    _lib.check(lib.st_synth_fold(C.byref(d), _lib.ptr(Sr), _lib.ptr(Si), _lib.ptr(Sfold), stream()), "fold")
    fr_, fi_ = O.fold_synthesis(P64[STFT_KEYS[2]], P64[STFT_KEYS[3]], F)
    sf_re, sf_im = n(Sfold)[:F], n(Sfold)[KP // 2:KP // 2 + F]
    res += [err("fold.re", sf_re, fr_), err("fold.im", sf_im, fi_, scale=np.abs(fr_).max())]
    AAo = t(to_kp(c["Are"].reshape(-1, F), c["Aim"].reshape(-1, F), KP))
    nsl = lib.st_synth_slabs(C.byref(d))
    frs = z(lib.st_synth_frame_slabs(C.byref(d)), B * OT, N)
    _lib.check(lib.st_synthesis_frames(C.byref(d), _lib.ptr(AAo), _lib.ptr(Sfold), _lib.ptr(frs), stream()), "synth")
    frs_ref = O._r(c["Are"].reshape(-1, F)) @ O._r(fr_) + O._r(c["Aim"].reshape(-1, F)) @ O._r(fi_)     # O._r: identity unless bf16_mode
    # only what reaches the cropped output is computed (cls_fe_dft.py:113): frames t with 0 < H t and H t - N < y, and of the partly cropped frames
    # only the tile columns that hold surviving taps (round 5: N <= H t + n < N + y) -- compared on exactly the taps the overlap-add reads
    live = (geo["H"] * np.arange(OT)[:, None] + np.arange(N)[None, :] >= N) & (geo["H"] * np.arange(OT)[:, None] + np.arange(N)[None, :] < N + geo["y"])
    assert live.any(1)[1:-1].all()
    res.append(err("synthesis.frames", n(frs).sum(0).reshape(B, OT, N)[:, live], frs_ref.reshape(B, OT, N)[:, live]))
    y_hat, dsyn = z(B, d.y), z(B, d.y)
    lp = z(lib.st_ola_loss_partials(C.byref(d)))
    frs_o = z(lib.st_synth_frame_slabs(C.byref(d)), B * OT, N); frs_o[0] = t(frs_ref)
    _lib.check(lib.st_ola_loss(C.byref(d), _lib.ptr(frs_o), _lib.ptr(x), _lib.ptr(y), _lib.ptr(y_hat), _lib.ptr(dsyn),
                               _lib.ptr(lp), stream()), "ola")
    res += [err("ola.y_hat", n(y_hat), c["out"]), err("ola.dsyn", n(dsyn), 2 * c["dy"]),
            err("ola.logcosh", n(lp).sum() / (B * d.y), np.mean(O.logcosh(Y.astype(np.float64) - c["out"])))]

    # 4. synthesis dgrad / wgrad
    dsyn_o = t(2 * c["dy"])
    dAA = z(nsl, B * OT, KP)
    _lib.check(lib.st_synthesis_dgrad(C.byref(d), _lib.ptr(dsyn_o), _lib.ptr(Sfold), _lib.ptr(dAA), stream()), "dgrad")
    g_re, g_im = from_kp(n(dAA).sum(0), F)
    sd = float(max(np.abs(c["dAre"]).max(), np.abs(c["dAim"]).max()))
    res += [err("syn_dgrad.dAre", g_re.reshape(B, OT, F), c["dAre"], scale=sd),
            err("syn_dgrad.dAim", g_im.reshape(B, OT, F), c["dAim"], scale=sd)]
    wsg = z(lib.st_wgrad_ws_floats(C.byref(d)))
    gSr, gSi = z(N, N), z(N, N)
    npart = lib.st_norm_partials(C.byref(d))
    norm_s = z(npart)
    _lib.check(lib.st_synthesis_wgrad(C.byref(d), _lib.ptr(AAo), _lib.ptr(dsyn_o), _lib.ptr(wsg), _lib.ptr(gSr), _lib.ptr(gSi),
                                      _lib.ptr(norm_s), stream()), "syn_wgrad")
    ss = float(max(np.abs(G[STFT_KEYS[2]]).max(), np.abs(G[STFT_KEYS[3]]).max()))
    res += [err("syn_wgrad.gSr", n(gSr), G[STFT_KEYS[2]][:, 0], scale=ss), err("syn_wgrad.gSi", n(gSi), G[STFT_KEYS[3]][:, 0], scale=ss),
            err("syn_wgrad.l1", n(norm_s).sum(), np.abs(G[STFT_KEYS[2]]).sum() + np.abs(G[STFT_KEYS[3]]).sum(), tol=1e-3)]

    # 5. autoencoders backward (oracle inputs)
    dAAo = z(nsl, B * OT, KP); dAAo[0] = t(to_kp(c["dAre"].reshape(-1, F), c["dAim"].reshape(-1, F), KP))
    mh_o, ph_o = t(c["mag_hat"]), t(c["phs_hat"])
    dmag, dphs = z(B, T, F), z(B, T, F)
    aews = z(lib.st_ae_bwd_ws_floats(C.byref(d)))
    g_m, g_p = z(PG), z(lay.total - lay.offsets[22])
    reg_coef = O.LOSS_SCALE * (2e-5 / 10) / (B * OT * F)            # the L1 term's gradient carries the loss scale like every other gradient
    _lib.check(lib.st_ae_bwd(C.byref(d), _lib.ptr(mag_o), _lib.ptr(phs_o), _lib.ptr(kn), _lib.ptr(ae_m), _lib.ptr(ae_p),
                             _lib.ptr(mh_o), _lib.ptr(ph_o), _lib.ptr(dAAo), None, reg_coef, _lib.ptr(dmag), _lib.ptr(dphs),
                             _lib.ptr(aews), _lib.ptr(g_m), _lib.ptr(g_p), stream()), "ae_bwd")
    res += [err("ae_bwd.dmag", n(dmag), c["dmag"]), err("ae_bwd.dphs", n(dphs), c["dphs"])]
    for ai, (pref, gbuf) in enumerate((("mpaec.aenc", g_m), ("mpaec.phs_aenc", g_p))):
        gb = n(gbuf)
        for li, nm in enumerate(O.AE_LAYERS):
            for j, wb in enumerate(("weight", "bias")):
                k = f"{pref}.{nm}.{wb}"
                off = lay.offsets[4 + 2 * li + j] - lay.offsets[4]
                ref = G[k]
                res.append(err("ae_bwd.g." + k.replace("mpaec.", ""), gb[off:off + ref.size].reshape(ref.shape), ref))

    # 6. polar backward + analysis wgrad
    dG = z(B * T, KP)
    keep = [t(c["re"]), t(c["im"]), t(c["dmag"]), t(c["dphs"])]        # hold references until the launch is queued
    _lib.check(lib.st_polar_bwd(C.byref(d), _lib.ptr(keep[0]), _lib.ptr(keep[1]), _lib.ptr(keep[2]),
                                _lib.ptr(keep[3]), None, _lib.ptr(dG), stream()), "polar_bwd")
    # zero-padded frames carry d atan2 = 1e7 * dphs: compare on the live frames, and the
    # degenerate frames separately against the same formula
    live = np.ones(T, bool); live[[0, T - 1]] = False          # noqa: F841 (analysis frames 0 and T-1 are all padding)
    dre_g, dim_g = from_kp(n(dG), F)
    dre_g, dim_g = dre_g.reshape(B, T, F), dim_g.reshape(B, T, F)
    sl = float(max(np.abs(c["dre"][:, live]).max(), np.abs(c["dim"][:, live]).max()))
    res += [err("polar_bwd.dre", dre_g[:, live], c["dre"][:, live], scale=sl), err("polar_bwd.dim", dim_g[:, live], c["dim"][:, live], scale=sl),
            err("polar_bwd.dim(zero frames)", dim_g[:, ~live], c["dim"][:, ~live], tol=1e-3)]
    dGo = t(to_kp(c["dre"].reshape(-1, F), c["dim"].reshape(-1, F), KP))
    gWr, gWi = z(N, N), z(N, N)
    norm_a = z(npart)
    _lib.check(lib.st_analysis_wgrad(C.byref(d), _lib.ptr(dGo), _lib.ptr(x), 0.5, _lib.ptr(wsg), _lib.ptr(gWr), _lib.ptr(gWi),
                                     _lib.ptr(norm_a), stream()), "an_wgrad")
    sw = float(max(np.abs(G[STFT_KEYS[0]]).max(), np.abs(G[STFT_KEYS[1]]).max()))
    res += [err("an_wgrad.gWr", n(gWr), G[STFT_KEYS[0]][:, 0], scale=sw), err("an_wgrad.gWi", n(gWi), G[STFT_KEYS[1]][:, 0], scale=sw),
            err("an_wgrad.rows>=F", n(gWr)[F:], 0 * n(gWr)[F:], scale=1.0, tol=0.0)]
    torch.cuda.synchronize()
    return res


def run_fused(B=3, seed=1, K=4, steps=3, scale=1, scheme="lean", shrink=4, oracle_dtype="f64"):
    """Fused entry points: st_model_fwd, st_loss_backward, st_train_step x steps vs the oracle (float64 from the same fp32 inputs; oracle_dtype="f32":
    the oracle's forward / backward in float32 arithmetic instead -- the reference's own precision, tools/fuzz_ground_f32.py)."""
    geo, X, Y, KN, P = make_case(B, seed, K=K, scale=scale, scheme=scheme, shrink=shrink)
    d = dims_of(geo, B, K)
    restore_oracle = follow_effective_arithmetic(d)
    try:
        return _run_fused(geo, X, Y, KN, P, d, B, K, steps, oracle_dtype)
    finally:
        restore_oracle()


def _run_fused(geo, X, Y, KN, P, d, B, K, steps, oracle_dtype="f64"):
    eng = new_engine(d)
    eng.load_state_dict(P)
    res = []
    odt = np.float64 if oracle_dtype == "f64" else np.float32
    P64 = {k: v.astype(odt) for k, v in P.items()}
    loss, G, c = O.model_loss_bwd(X.astype(odt), KN.astype(odt), Y.astype(odt), P64, geo)
    y_hat, mag, mag_hat = eng.forward(t(X), t(KN))
    res += [err("fwd.y_hat", n(y_hat), c["out"]), err("fwd.mag", n(mag), c["mag"]), err("fwd.mag_hat", n(mag_hat), c["mag_hat"])]
    outs = eng.loss_backward(t(X), t(KN), t(Y), want_outputs=True)
    sc = n(eng.scalars)
    res += [err("step.y_hat", n(outs[0]), c["out"]), err("step.loss", sc[0], loss, tol=1e-4)]
    g = eng.layout.views(eng.grads)
    ss = {k: float(max(np.abs(G[a]).max(), np.abs(G[b]).max())) for k, (a, b) in
          {"an": STFT_KEYS[:2], "sy": STFT_KEYS[2:]}.items()}
    for k in eng.layout.names:
        scale = ss["an"] if k in STFT_KEYS[:2] else ss["sy"] if k in STFT_KEYS[2:] else None
        res.append(err("grad." + k.replace("mpaec.", ""), n(g[k]), G[k], tol=2e-4, scale=scale))
    l1 = sum(np.abs(G[k]).sum() for k in STFT_KEYS) / O.LOSS_SCALE       # the published norm is that of the unscaled gradient
    res.append(err("step.l1norm", sc[3], l1, tol=1e-3))
    # training steps (train.py:131-151 ordering) vs oracle in float32 arithmetic
    eng2 = new_engine(d); eng2.load_state_dict(P)
    Pq = {k: P[k].copy() for k in O.param_order()}
    Mq = {k: np.zeros_like(v) for k, v in Pq.items()}; Vq = {k: np.zeros_like(v) for k, v in Pq.items()}
    lrs, _ = O.get_1cycle_schedule(lr_max=1e-3, n_data_points=200, epochs=1, batch_size=2)
    lr = lrs[0]
    for it in range(steps):
        Xi = np.roll(X, 17 * it, axis=1).copy(); Yi = np.roll(Y, 17 * it, axis=1).copy()
        eng2.train_step(t(Xi), t(KN), t(Yi), lr)
        lo, no, co = O.train_step(Xi, KN, Yi, Pq, Mq, Vq, it + 1, lr, geo)
        lr = lrs[it]
        res.append(err(f"train{it}.loss", n(eng2.scalars)[0], lo, tol=1e-4))
        worst = max((err(k, n(v), Pq[k], scale=1.0, tol=2e-5) for k, v in eng2.named.items()), key=lambda r: r["err"])
        worst["name"] = f"train{it}.params(worst:{worst['name'].replace('mpaec.', '')})"
        res.append(worst)
    torch.cuda.synchronize()
    return res


def report(res, out=print):
    bad = 0
    for r in res:
        flag = "ok " if r["ok"] else "BAD"
        bad += not r["ok"]
        out(f"{flag} {r['name']:44s} err={r['err']:.3e} scale={r['scale']:.3e} rel={r['rel']:.2e} tol={r['tol']:.0e}"
            + ("" if r["ok"] else f"  worst@{r['worst']} got={r['got']:.6g} ref={r['ref']:.6g}"))
    return bad
